"""Summarise the `placement <block> <live> <HW_ID> <XCC_ID>` lines of a -DI2C_QUAD_PLACEMENT build (one line per wave and launch of
k_quad_forward): per launch, how many live waves share a SIMD.   python tools/placement_summary.py < log"""
import collections
import sys

launches, cur, seen = [], [], set()
for line in sys.stdin:
    p = line.split()
    if len(p) != 5 or p[0] != "placement":
        continue
    blk, live, hw, xcc = int(p[1]), int(p[2]), int(p[3]), int(p[4])
    if blk in seen:
        launches.append(cur)
        cur, seen = [], set()
    seen.add(blk)
    cur.append((live, hw, xcc & 0xF))
launches.append(cur)
for i, L in enumerate(launches):
    simd = collections.Counter()
    cu = collections.Counter()
    for live, hw, xcc in L:
        if not live:
            continue
        key_cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        simd[key_cu + ((hw >> 4) & 3,)] += 1
        cu[key_cu] += 1
    hist = collections.Counter(simd.values())
    print(f"launch {i}: {sum(simd.values())} live waves on {len(simd)} SIMDs of {len(cu)} CUs; waves per occupied SIMD: {dict(sorted(hist.items()))}; "
          f"waves per occupied CU: {dict(sorted(collections.Counter(cu.values()).items()))}")
