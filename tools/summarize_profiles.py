#!/usr/bin/env python3
"""Turn the rocprofv3 output merged under gpurun_out/<round>_<workload>_* (tools/collect_profiles.sh) into the
committed summaries in profiles/:
  profiles/<round>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --workload ...`
  profiles/<round>_pmc_summary.json              per workload and kernel: FETCH_SIZE / WRITE_SIZE / MFMA-busy means
  profiles/traffic.json                          per workload: HBM bytes per launch of the dominant kernel and its
                                                 MFMA-busy share of SIMD cycles (read by bench.py -> roofline)
Corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
1/2 of the bytes of a wide coalesced read, so the read side is doubled (our composite reads are 8 B/lane, an access
width the guide marks as uncalibrated: the doubled figure is an upper estimate, the raw one a lower).  MFMA-busy share =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections, csv, glob, json, os, shutil, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
try:
    commit = subprocess.check_output(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "unknown"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
DOMINANT = {"C2": "mom::k_layer<true, 3, 15>", "C4": "mom::k_layer<false, 3, 0>", "C1": "momsm::k_sweep<4, true>",
            "C5": "momr::k_dbl_pair1<false, 0>"}


def newest(pattern):
    files = sorted(glob.glob(str(ROOT / "gpurun_out" / pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


summ, traffic = {}, {}
for wl in ("C2", "C4", "C1", "C5", "voigt", "rrs_nt"):
    st = newest(f"{rnd}_{wl}_stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, out / f"{rnd}_{wl}_kernel_stats.csv")
    per = {}
    for tag in ("fetch", "write", "mfma"):
        f = newest(f"{rnd}_{wl}_{tag}/**/*counter_collection.csv")
        if not f:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            for cn, v in d.items():   # v: one value per launch, in dispatch order
                per.setdefault(k, {})[cn] = {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v), "values": v if len(v) <= 64 else None}
    if per:
        summ[wl] = per
    dom = DOMINANT.get(wl)
    if dom and dom not in per:  # later template parameters (k_layer's MT flag) follow the ones named here
        dom = next((k for k in per if k.startswith(dom[:-1] + ",")), dom)
    kl = per.get(dom, {})
    if kl:
        t = {"kernel": dom, "round": rnd, "source": f"tools/collect_profiles.sh {rnd} @ {commit}"}
        # One kernel NAME can cover dissimilar launches (C4: the m = 0 sub-problem at N0 = 128 and the full problem at
        # N = 256 are both k_layer<false, 3, 0>): "per launch" is the LARGEST launch of the step (the full problem, the one
        # roofline.achieved is computed for), taken at the same position of the dispatch order in every counter pass;
        # the step total is reported next to it.
        def pick(cn):
            e = kl.get(cn)
            if not e:
                return None, None
            v = e.get("values")
            if not v:
                return e["mean"], e["sum"]
            ref = kl.get("FETCH_SIZE", e).get("values") or v
            if max(ref) <= 2 * min(ref):     # similar launches: the LAST one (after the warm-up step of the counter pass)
                return v[-1], e["sum"]
            idx = max(range(len(ref)), key=lambda i: ref[i]) if len(ref) == len(v) else max(range(len(v)), key=lambda i: v[i])
            return v[idx], sum(v)
        fmax, fsum = pick("FETCH_SIZE")
        wmax, wsum = pick("WRITE_SIZE")
        if fmax is not None and wmax is not None:
            rd_raw, wr = fmax * 1024, wmax * 1024
            t.update(hbm_bytes_per_launch=2 * rd_raw + wr, fetch_bytes_raw=rd_raw, fetch_bytes_corrected_x2=2 * rd_raw, write_bytes=wr,
                     hbm_bytes_all_launches_of_the_pass=2 * fsum * 1024 + wsum * 1024, launches_in_the_pass=kl["FETCH_SIZE"]["launches"])
        bmax, _ = pick("SQ_VALU_MFMA_BUSY_CYCLES")
        gmax, _ = pick("GRBM_GUI_ACTIVE")
        if bmax is not None and gmax:
            t["mfma_busy_frac"] = bmax / (gmax / 8 * 1024)
        t["note"] = ("the LARGEST launch of the dominant kernel where its launches differ by more than 2 x, else the last one of the "
                     "counter pass (one warm-up + one profiled step; more than 64 launches: their mean); FETCH_SIZE "
                     "doubled per MI355X_MICROARCH.md; layer-sweep mode: ONE launch of the kernel covers all layers of the step")
        af = ROOT / "gpurun_out" / f"{rnd}_{wl}_args.txt"     # the bench arguments of the counter passes (collect_profiles.sh)
        if af.exists():
            args = af.read_text().split()
            t["bench_args"] = " ".join(args)
            if "--points" in args:
                t["points"] = int(args[args.index("--points") + 1])
        traffic[wl] = t
(out / f"{rnd}_pmc_summary.json").write_text(json.dumps(summ, indent=1))
(out / "traffic.json").write_text(json.dumps(traffic, indent=1))
for f in glob.glob(str(ROOT / "gpurun_out" / f"{rnd}_*bench*.json")):
    shutil.copy(f, out / Path(f).name)
print(json.dumps(traffic, indent=1))
