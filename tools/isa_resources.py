#!/usr/bin/env python3
"""Condense the `-Rpass-analysis=kernel-resource-usage` remarks of a full library build
(make -C radiativetransfer.jl_amd/csrc EXTRA=-Rpass-analysis=kernel-resource-usage 2> build.log) into one line per
kernel image: VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, static LDS, waves per SIMD.
usage: tools/isa_resources.py build.log > profiles/rNN_isa_resources.txt"""
import re, subprocess, sys

rows, cur = [], None
for ln in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", ln)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = sorted({r["name"] for r in rows})
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines())) if names else {}
seen = set()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spillV':>6} {'scratch':>7} {'LDS':>7} {'waves':>5}  kernel")
for r in rows:
    d = dem.get(r["name"], r["name"]).replace("(mom::LayerArgs)", "").replace("(mom4::LayerArgs)", "").replace("(momf::LayerArgs)", "")
    if d in seen or "k_" not in d:
        continue
    seen.add(d)
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('SGPRs','?'):>5} {r.get('VGPRs Spill','?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('LDS Size [bytes/block]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>5}  {d[:110]}")
