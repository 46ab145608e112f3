#!/usr/bin/env python3
"""Per-kernel resource summary from hipcc's -Rpass-analysis remarks (csrc/*.remarks): VGPRs, spills, scratch, occupancy.
Usage: tools/kres.py [substring ...]"""
import glob, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alignnet-3d_amd", "csrc")
want = sys.argv[1:]
rows = {}
for f in glob.glob(os.path.join(root, "*.remarks")):
    cur = None
    for line in open(f, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
names = list(rows)
try:
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for n, d in zip(names, dem):
    d = d.replace("alignnet::", "").split("(")[0]
    if want and not any(w in d for w in want):
        continue
    r = rows[n]
    print("%-70s vgpr %3d agpr %3d vspill %3d sspill %3d scratch %4d occ %d" % (d[:70], r.get("vgpr", -1), r.get("agpr", 0), r.get("vspill", 0), r.get("sspill", 0), r.get("scratch", 0), r.get("occ", 0)))
