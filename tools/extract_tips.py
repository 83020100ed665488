#!/opt/conda/bin/python3.9
"""Extracts the TIPS-2017 partition-sum rows and the isotopologue weights of a few molecules from the reference's
NetCDF-4 tables (src/Absorption/constants/{TIPS_2017,iso_info}.nc) into a small .npz of plain arrays.

Run in the build container (the main Python has no HDF5 reader):
    /opt/conda/bin/python3.9 tools/extract_tips.py
The output, radiativetransfer.jl_amd/data/tips_2017_subset.npz, is DATA (HITRAN's TIPS-2017 tables, Gamache et al.
2017, exactly as the reference ships them: Float32, -1 = unfilled); the code that uses it (absorption.qoft,
oracle/absref.py) is written here.  Index convention of the reference (constants/TIPS_2017.jl:8-27,
mol_weights.jl:10-26): Julia reads the NetCDF variables as [molecule, isotope, index]; h5py sees the reversed
order [index, isotope, molecule]."""
import sys
from pathlib import Path

import h5py
import numpy as np

REF = Path("/root/reference/src/Absorption/constants")
OUT = Path(__file__).resolve().parents[1] / "radiativetransfer.jl_amd" / "data" / "tips_2017_subset.npz"
MOLECULES = None  # None: every HITRAN molecule id the reference's two tables both cover (round 4; rounds 1-3 shipped ids 1-7 only)


def main():
    out = {}
    with h5py.File(REF / "TIPS_2017.nc", "r") as f:
        T, Q = f["TIPS_2017_T"][...], f["TIPS_2017_Q"][...]  # [index, isotope, molecule], float32
    with h5py.File(REF / "iso_info.nc", "r") as f:
        W = f["mol_weight"][...]  # [isotope, molecule], float32
        A = f["abundance"][...]
    assert T.dtype == np.float32 and W.dtype == np.float32
    keys = []
    global MOLECULES
    if MOLECULES is None:
        MOLECULES = list(range(1, min(T.shape[2], W.shape[1]) + 1))
    for M in MOLECULES:
        for I in range(1, T.shape[1] + 1):
            t, q = T[:, I - 1, M - 1], Q[:, I - 1, M - 1]
            # get_TT / get_TQ: the entries before the first -1 (TIPS_2017.jl:15-27)
            nt = int(np.argmax(t == -1)) if np.any(t == -1) else t.size
            nq = int(np.argmax(q == -1)) if np.any(q == -1) else q.size
            if nt == 0 or nq == 0:
                continue
            out[f"T_{M}_{I}"] = t[:nt].copy()
            out[f"Q_{M}_{I}"] = q[:nq].copy()
            keys.append((M, I))
    out["mol_weight"] = W[:, [m - 1 for m in MOLECULES]].T.copy()  # [len(MOLECULES), 12], -1 = no such pair
    out["abundance"] = A[:, [m - 1 for m in MOLECULES]].T.copy()
    out["molecules"] = np.array(MOLECULES, dtype=np.int32)
    out["pairs"] = np.array(keys, dtype=np.int32)
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {len(keys)} (mol, iso) pairs, {OUT.stat().st_size} bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
