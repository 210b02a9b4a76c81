set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_hip_full_configs.py -q -m gpu -k "overlapped or deterministic" 2>&1 | tail -3 > $O/exp12.txt
python3 - >> $O/exp12.txt 2>&1 <<'PY'
import importlib, os, sys, time
import numpy as np, torch
sys.path[:0]=[".", "input-inference-for-control_amd"]
pkg=importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import make_env_model
for B in (1024, 8192, 16384, 65536):
    T=100; rng=np.random.default_rng(0)
    x0=np.array([np.pi,0.0])+1e-2*rng.normal(size=(B,2))
    for ov in (True, False):
        eng=pkg.BatchedI2c(make_env_model("PendulumKnownActReg"),T,None,np.diag([1.0]),None,300.0,1.0,np.zeros((B,T,1)),0.5*np.eye(1),np.array([0.0,0.0]),np.diag([1e-3,1e-3]),x0=x0,keep_zpost=False,overlap_propagation=ov)
        eng.use_expert_controller=False; eng._propagate=True; eng.propagate()
        eng.learn(5); torch.cuda.synchronize(); t0=time.perf_counter(); eng.learn(40); torch.cuda.synchronize()
        print(f"covariance control B={B} one call, overlap={ov}: {(time.perf_counter()-t0)/40*1e3:.3f} ms per iteration, fails {len(eng.failures())}")
    t0=time.perf_counter()
    for _ in range(40): eng.learn_msgs()
    torch.cuda.synchronize()
    print(f"covariance control B={B} stepwise learn_msgs: {(time.perf_counter()-t0)/40*1e3:.3f} ms per iteration")
PY
