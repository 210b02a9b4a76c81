"""The `extra.reference_shapes` legs of bench.py on their own (B = 1 through the drop-in facade; BASELINE.md section 2 shapes):
    python tools/bench_reference_shapes.py [--profile]   (--profile: cProfile of 200 facade iterations, pendulum T = 100)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package's drop-in `i2c` on sys.path)

pkg = importlib.import_module(bench.PKG)
print(json.dumps(bench.reference_shape_legs(pkg, "cuda"), indent=1))
if "--profile" in sys.argv:
    import cProfile
    import pstats

    import numpy as np
    from i2c.exp_types import CubatureQuadrature
    from i2c.i2c import I2cGraph
    from i2c.known_models import make_env_model

    Qp = np.diag([1.0, 100.0, 1.0])
    g = I2cGraph(make_env_model("PendulumKnown"), 100, Qp, np.diag([2.0]), Qp, 100.0, 0.0, 1e-2 * np.random.randn(100, 1), 2.0 * np.eye(1), None, None,
                 CubatureQuadrature(1, 0, 0), device="cuda")
    for _ in range(5):
        g.learn_msgs()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        g.learn_msgs()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
