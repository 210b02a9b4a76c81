"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <output dir>"""
import csv
import glob
import sys

fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=lambda f: -sum(1 for l in open(f) if "i2c::" in l))
for r in csv.DictReader(open(fs[0])):
    n = r["Name"]
    short = n.split("i2c::")[1][:70] if "i2c::" in n else n[:70]
    print(f"{short:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
