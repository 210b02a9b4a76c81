"""Pendulum T=200 EM-iteration time across batch sizes:  python tools/sweep_batch.py [B ...]"""
import json
import subprocess
import sys

for B in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072]:
    out = subprocess.run([sys.executable, "bench.py", "--batch", str(B), "--steps", "30", "--warmup", "3", "--no-cpu-baseline",
                          "--no-saturated", "--no-extra"], capture_output=True, text=True).stdout.strip().split("\n")
    out = [l for l in out if l.startswith("{")][-1]
    d = json.loads(out)
    k = d["kernel_ms"]
    print(f"B={B:7d}  {d['ms_per_step']:7.4f} ms/iter  {d['value']:.3e} msg/s  fwd {k['forward_sweep']:.4f} bwd {k['backward_sweep']:.4f} "
          f"| whole-iteration {d['roofline'].get('whole_iteration_GBps', 0):7.1f} GB/s")
