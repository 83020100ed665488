#!/usr/bin/env python3
"""Extract the NUMERIC known-answer tables held by the reference's own tests
(test/benchmarks/natraj_trues.jl, test/benchmarks/6SV1_R_trues.jl, used by
test/test_CoreRT.jl:3-83) into tests/golden/reference_tables.json.

Runs only in the build container (reads /root/reference); the JSON it writes is data
(expected outputs), committed so the GPU box / CI never need the reference tree.
"""
import json, re, sys
from pathlib import Path

REF = Path("/root/reference/test/benchmarks")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "reference_tables.json"

def numbers(s):
    return [float(x) for x in re.findall(r"[-+]?\d*\.\d+(?:[eE][-+]?\d+)?|[-+]?\d+(?:[eE][-+]?\d+)?", s)]

def natraj():
    txt = (REF / "natraj_trues.jl").read_text()
    out = {}
    for name in ("I_trues", "Q_trues", "U_trues"):
        mt = re.search(name + r"\s*=\s*\[(.*?)\]", txt, re.S)
        rows = [numbers(r) for r in mt.group(1).split(";")]
        rows = [r for r in rows if r]
        assert len(rows) == 16 and all(len(r) == 7 for r in rows), (name, len(rows))
        out[name] = rows  # [16 mu][7 azimuth]
    return out

def sixsv():
    txt = (REF / "6SV1_R_trues.jl").read_text()
    txt = re.sub(r"#.*", "", txt)
    vals = numbers(txt.split("=", 1)[1])
    assert len(vals) == 6 * 3 * 3 * 16, len(vals)
    it = iter(vals)
    return [[[[next(it) for _ in range(16)] for _ in range(3)] for _ in range(3)] for _ in range(6)]

if __name__ == "__main__":
    data = {
        "source": "reference test/benchmarks/natraj_trues.jl and 6SV1_R_trues.jl (numeric tables only)",
        "natraj": natraj(),
        "natraj_mu": [0.02, 0.06, 0.10, 0.16, 0.20, 0.28, 0.32, 0.40, 0.52, 0.64, 0.72, 0.84, 0.92, 0.96, 0.98, 1.00],
        "natraj_phi": [0.0, 30.0, 60.0, 90.0, 120.0, 150.0, 180.0],
        "natraj_tau": 0.5, "natraj_mu0": 0.2,
        "sixsv_R": sixsv(),  # [case 6][sza 3][az 3][vza 16]
        "sixsv_cases": [
            {"az": [180, 90, 0], "sza": [23.0739, 53.1301, 78.4630], "lambda_nm": 530, "tau": 0.1, "rho": 0.0},
            {"az": [180, 90, 0], "sza": [0.0001, 36.8699, 66.4218], "lambda_nm": 530, "tau": 0.1, "rho": 0.25},
            {"az": [180, 90, 0], "sza": [0.0001, 36.8699, 66.4218], "lambda_nm": 440, "tau": 0.25, "rho": 0.0},
            {"az": [180, 90, 0], "sza": [23.0739, 53.1301, 78.4630], "lambda_nm": 440, "tau": 0.25, "rho": 0.25},
            {"az": [180, 90, 0], "sza": [23.0739, 53.1301, 78.4630], "lambda_nm": 360, "tau": 0.50, "rho": 0.0},
            {"az": [180, 90, 0], "sza": [0.0001, 36.8699, 66.4218], "lambda_nm": 360, "tau": 0.50, "rho": 0.25},
        ],
        "sixsv_vza": [0.0, 11.4783, 16.2602, 23.0739, 32.8599, 43.9455, 50.2082, 58.6677, 66.4218, 71.3371,
                      73.7398, 78.463, 80.7931, 84.2608, 86.5602, 88.854],
        "yaml": {"quadrature_type": "RadauQuad", "polarization_type": "Stokes_IQUV", "max_m": 3, "l_trunc": 20,
                 "depol": 0.0, "Nz": 1},
    }
    OUT.write_text(json.dumps(data))
    print("wrote", OUT, OUT.stat().st_size, "bytes")
