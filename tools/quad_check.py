"""Quick GPU check of the quad forward kernel (group_lanes = 64 on the d <= 8 models): goldens through the C ABI on cuda:0,
then forward / backward timings next to the default family. python tools/quad_check.py [golden ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "input-inference-for-control_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import parity  # noqa: E402

names = sys.argv[1:] or ["em_pendulum_T200", "em_dcp_T60", "em_cartpole_T100", "em_quadrotor_T20", "em_linear_T60", "em_dcp_nondiag_T30",
                         "em_pendulum_T30_tau7", "em_dcp_T300_run20"]
bad = 0
for name in names:
    t = time.time()
    try:
        eng = parity.check_against_golden(name, None, "cuda", 1e-6, 1e-5, group_lanes=164 if "quad12" in name else 64)
        print(name, "OK", eng.forward_family, eng.backward_family, f"{time.time() - t:.1f}s", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(name, "FAILED", repr(e)[:300], flush=True)
sys.exit(1 if bad else 0)
