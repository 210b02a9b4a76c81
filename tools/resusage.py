"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
K_SCR, K_OCC = r'ScratchSize \[bytes/lane\]', r'Occupancy \[waves/SIMD\]'
for b in blocks:
    name = b.split('\n')[0].strip()
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('i2c::', '').replace('void ', '')
    dem = re.sub(r'\(.*', '', dem)[:60]
    print("%-62s VGPR=%4d AGPR=%3d SGPR=%3d scratch=%5d occ=%d" % (dem, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(K_SCR), g(K_OCC)))
