#!/usr/bin/env python3
"""Turn the rocprofv3 output merged under gpurun_out/<round>_* into the committed summaries in profiles/:
  profiles/<round>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python bench.py`
  profiles/<round>_pmc_summary.json        per-kernel FETCH_SIZE / WRITE_SIZE / MFMA-busy means
  profiles/traffic.json                    HBM bytes per k_layer launch (read by bench.py -> roofline.traffic)
Corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports 1/2 of the bytes of a wide coalesced read, so the read side is doubled (our reads are 8 B/lane, an
access width the guide marks as uncalibrated: the doubled figure is an upper estimate, the raw one a lower)."""
import csv, glob, json, shutil, sys, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "profiles"; out.mkdir(exist_ok=True)
import os
st = sorted(glob.glob(str(ROOT / "gpurun_out" / f"{rnd}_stats" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
if st: shutil.copy(st[-1], out / f"{rnd}_kernel_stats.csv")
summ = {}
for tag in ("fetch", "write", "mfma"):
    import os
    files = sorted(glob.glob(str(ROOT / "gpurun_out" / f"{rnd}_{tag}" / "*" / "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:  # newest collection only
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            for cn, v in d.items():
                summ.setdefault(k, {})[cn] = {"launches": len(v), "mean": sum(v) / len(v)}
(out / f"{rnd}_pmc_summary.json").write_text(json.dumps(summ, indent=1))
kl = next((v for k, v in summ.items() if k.startswith("mom::k_layer<true") or k.startswith("k_layer<true")), {})
if "FETCH_SIZE" in kl and "WRITE_SIZE" in kl:
    rd_raw, wr = kl["FETCH_SIZE"]["mean"] * 1024, kl["WRITE_SIZE"]["mean"] * 1024
    (out / "traffic.json").write_text(json.dumps({
        "round": rnd, "k_layer_hbm_bytes_per_launch": 2 * rd_raw + wr, "fetch_bytes_raw": rd_raw,
        "fetch_bytes_corrected_x2": 2 * rd_raw, "write_bytes": wr,
        "note": "mean over the 40 launches of mom::k_layer<true, 3, 15> in one bench step (S=10000, moments 1-2); FETCH_SIZE doubled per MI355X_MICROARCH.md"}, indent=1))
for f in ("bench.json", "bench_under_rocprof.json"):
    p = ROOT / "gpurun_out" / f"{rnd}_{f}"
    if p.exists(): shutil.copy(p, out / f"{rnd}_{f}")
print(json.dumps({k: v for k, v in summ.items() if "k_layer" in k or "k_surface" in k}, indent=1)[:2000])
