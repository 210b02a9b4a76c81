"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as
MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc_traffic.json.

    python tools/pmc_summary.py <tag> <B> <T> <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of
a coalesced streaming read -> doubled here. Calibration for OUR access pattern (8 B per lane,
buffer_load_dwordx2): the forward sweep reads 11 rows x 8 B = 88 B per cell by construction and
FETCH_SIZE reports 44.3 B per cell; WRITE_SIZE reports the 160 B per cell written exactly.
Units: the counters are in KiB.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "i2c::" in r["Kernel_Name"]:
            rest = r["Kernel_Name"].split("i2c::")[1]
            name = rest.split("<")[0]
            if name in ("k_group", "k_wave"):  # one kernel template per family: the sweep (KIND) is its first argument
                kinds = {"0": "forward", "1": "backward", "2": "propagate", "3": "ckf"} if name == "k_group" else \
                    {"0": "forward", "1": "backward", "2": "scan", "3": "cell", "4": "forward"}  # 4: forward, pivot blocks through LDS
                name += "_" + kinds.get(rest.split("<")[1].split(",")[0].strip(), "x")
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in acc.items()}  # steady-state half


def main():
    tag, B, T, dfetch, dwrite = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    fs, ws = per_kernel(dfetch), per_kernel(dwrite)
    cells = B * T
    out = {"B": B, "T": T, "dtype": "f64", "fetch_correction": 2.0, "kernels": {}}
    for k in fs:
        rd, wr = 2.0 * fs[k] * 1024, ws.get(k, 0.0) * 1024
        out["kernels"][k] = {
            "FETCH_SIZE_KiB_raw": fs[k], "WRITE_SIZE_KiB": ws.get(k, 0.0),
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_cell": (rd + wr) / cells,
        }
    path = f"profiles/{tag}_pmc_traffic.json"
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    for k, v in out["kernels"].items():
        print(f"  {k:12s} {v['hbm_bytes_per_cell']:7.1f} B/cell  ({v['hbm_bytes_per_launch'] / 1e6:9.1f} MB per launch)")


if __name__ == "__main__":
    main()
