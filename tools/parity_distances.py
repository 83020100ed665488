"""Summarise a PARITY_LOG file (tests/helpers.py::_log_parity; one JSON line per GPU-vs-oracle comparison of a `pytest -m gpu`
run): the largest measured distance per tolerance class and the comparisons closest to their bar.
usage: PARITY_LOG=$PWD/gpurun_out/x/parity.jsonl python -m pytest tests -m gpu ; python tools/parity_distances.py gpurun_out/x/parity.jsonl"""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1])]
print(f"{len(rows)} comparisons, {sum(r['n'] for r in rows)} numbers compared")
by = collections.defaultdict(list)
for r in rows:
    by[r["rtol"]].append(r)
print("bar (rtol of |I|)  comparisons  largest measured  fraction of bar  where")
for k in sorted(by):
    mx = max(by[k], key=lambda r: r["measured"])
    print(f"{k:10.3e} {len(by[k]):6d}   {mx['measured']:10.3e}   {mx['measured'] / k:6.3f}   {mx['test'].split(' ')[0]} [{mx['what']}]")
print("\nten comparisons closest to their bar:")
for r in sorted(rows, key=lambda r: -r["measured"] / r["rtol"])[:10]:
    print(f"  {r['measured'] / r['rtol']:6.3f} of {r['rtol']:.2e}: {r['test'].split(' ')[0]} [{r['what']}] n={r['n']}")
