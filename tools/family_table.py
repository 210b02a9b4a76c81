"""What runs by default: model x batch window -> forward family / backward family + schedule / propagation / filter, asked of the
library's own resolvers (i2c_kernel_family, i2c_backward_schedule -- they read the scalar fields of an I2cProblem only, so this
runs without a GPU) and written as a markdown table between the `family-table` markers of DESIGN.md, so that the document cannot
drift from the code (tests/test_abi.py::test_design_family_table_is_current regenerates and compares).
    python tools/family_table.py            print the table
    python tools/family_table.py --write    rewrite the block in DESIGN.md"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "input-inference-for-control_amd")]
pkg = importlib.import_module("input-inference-for-control_amd")
N = pkg._native

BEGIN, END = "<!-- family-table:begin (tools/family_table.py --write) -->", "<!-- family-table:end -->"
HORIZON = {"PendulumKnown": 200, "PendulumKnownActReg": 100, "CartpoleKnown": 500, "DoubleCartpoleKnown": 300, "LinearKnown": 60,
           "LinearKnownMinimumEnergy": 60, "PlanarQuadrotor": 50, "Quadrotor12": 50}
SCHED = {N.BWD_TWO_PASS: "two-pass", N.BWD_FUSED: "fused", N.BWD_CHUNKED: "chunked"}
B_MAX = 1 << 18
SEEDS = [1, 64, 65, 256, 257, 512, 768, 769, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 8193, 12287, 12288, 16384, 20479, 20480, 32768, 65536, 131072, B_MAX]


def shape(model_id, B, T, inference=N.INF_CUBATURE, post_layout=0):
    p = N.I2cProblem()
    p.abi_version, p.model_id, p.B, p.T, p.backward_mode, p.inference, p.dtype = N.ABI_VERSION, model_id, B, T, N.BWD_AUTO, inference, 0
    p.group_lanes, p.post_layout, p.gh_degree = 0, post_layout, 3
    p.quad_alpha, p.quad_beta, p.quad_kappa = 1.0, 0.0, 0.0
    return p


def resolve(lib, model_id, B, T, layout):
    p = shape(model_id, B, T, post_layout=layout)
    fam = lambda sweep: N.FAMILY_NAMES.get(lib.i2c_kernel_family(ctypes.byref(p), sweep), "refused")  # noqa: E731
    sched = SCHED.get(lib.i2c_backward_schedule(ctypes.byref(p)), "refused")
    if sched == "chunked" and fam(N.SWEEP_CHUNK_PASSES) == "quad":
        sched += " (quad passes)"
    elif sched == "chunked" and fam(N.SWEEP_CHUNK_STITCH) == "quad":
        sched += " (quad stitch)"
    return fam(N.SWEEP_FORWARD), f"{fam(N.SWEEP_BACKWARD)}, {sched}", fam(N.SWEEP_PROPAGATE), fam(N.SWEEP_FILTER)


def windows(lib, model_id, T, layout):
    """[(lo, hi, tuple)]: maximal batch ranges with one answer (bisection between seed points that differ)."""
    f = lambda B: resolve(lib, model_id, B, T, layout)  # noqa: E731
    pts = {B: f(B) for B in SEEDS}

    def bisect(lo, hi):  # f(lo) != f(hi): the first B in (lo, hi] with f(B) != f(lo)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if mid not in pts:
                pts[mid] = f(mid)
            if pts[mid] == pts[lo]:
                lo = mid
            else:
                hi = mid
        return hi

    changed = True
    while changed:
        changed = False
        ks = sorted(pts)
        for a, b in zip(ks, ks[1:]):
            if pts[a] != pts[b] and b - a > 1:
                bisect(a, b)
                changed = True
    out, ks = [], sorted(pts)
    start = ks[0]
    for a, b in zip(ks, ks[1:]):
        if pts[a] != pts[b]:
            out.append((start, a, pts[a]))
            start = b
    out.append((start, ks[-1], pts[ks[-1]]))
    return out


def table(lib=None):
    lib = lib or pkg.load_library()
    rows = ["| model (T of its config) | trajectories per GPU | forward sweep | backward sweep, schedule | propagation | filter step |", "|---|---|---|---|---|---|"]
    groups = {}  # models whose answers coincide at every batch size share their rows
    for name, mid in N.MODEL_IDS.items():
        d = lib.query(mid)
        T, layout = HORIZON[name], 1 if d.wave else 0
        key = (d.nx + d.nu, tuple(windows(lib, mid, T, layout)))
        groups.setdefault(key, []).append(f"{name} ({T})")
    for (dim, wins), names in groups.items():
        for k, (lo, hi, (ff, bb, pp, kk)) in enumerate(wins):
            rng = f"{lo} … {hi}" if hi < B_MAX else f"≥ {lo}"
            rows.append(f"| {', '.join(names) + f': d = {dim}' if k == 0 else '〃'} | {rng} | {ff} | {bb} | {pp} | {kk} |")
    return "\n".join(rows)


def block(lib=None):
    return (BEGIN + "\n" + table(lib) + "\n\n(`lam = 0`, fp64, nothing asked for; from `i2c_kernel_family` / `i2c_backward_schedule`; quad passes / quad stitch: "
            "compose + stitch / the stitch pass alone in the quad form; requests override it: INTEGRATION.md section 1b)\n" + END)


if __name__ == "__main__":
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        i, j = s.index(BEGIN), s.index(END) + len(END)
        open(path, "w").write(s[:i] + block() + s[j:])
        print("DESIGN.md: family table rewritten")
    else:
        print(block())
