import importlib, os, sys, time
import numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"input-inference-for-control_amd")]
pkg=importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import make_env_model
B=int(sys.argv[1]); T=100
mu_u=np.zeros((B,T,1)); rng=np.random.default_rng(0)
x0=np.array([np.pi,0.0])+1e-2*rng.normal(size=(B,2))
eng=pkg.BatchedI2c(make_env_model("PendulumKnownActReg"),T,None,np.diag([1.0]),None,300.0,1.0,mu_u,0.5*np.eye(1),np.array([0.0,0.0]),np.diag([1e-3,1e-3]),x0=x0,keep_zpost=False)
eng.use_expert_controller=False; eng._propagate=True
eng.propagate()
for _ in range(3): eng.learn_msgs()
torch.cuda.synchronize(); t0=time.perf_counter()
n=30
for _ in range(n): eng.learn_msgs()
torch.cuda.synchronize(); t1=time.perf_counter()
ev=[torch.cuda.Event(enable_timing=True) for _ in range(5)]
ev[0].record(); eng.forward_sweep(); ev[1].record(); eng.backward_sweep(); ev[2].record(); eng.propagate(); ev[3].record(); eng.maximize(); ev[4].record(); torch.cuda.synchronize()
print(f"B={B}: {1e3*(t1-t0)/n:.3f} ms/iteration wall; kernels fwd {ev[0].elapsed_time(ev[1]):.3f} bwd {ev[1].elapsed_time(ev[2]):.3f} prop {ev[2].elapsed_time(ev[3]):.3f} maximize {ev[3].elapsed_time(ev[4]):.3f} ms; fails {len(eng.failures())}")
