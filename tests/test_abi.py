"""The C-ABI shared library: loads, exports every symbol include/momcore.h declares, and fails
loudly (status code + message, no crash, no fallback) when no GPU is present."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    txt = (ROOT / "include" / "momcore.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mom_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_and_binding_agree(rtamd):
    names = declared_symbols()
    assert len(names) >= 20
    assert set(names) == set(rtamd._lib.SIGNATURES), set(names) ^ set(rtamd._lib.SIGNATURES)


def test_library_exports_every_symbol(rtamd):
    lib = rtamd.load()  # raises MomError if libmomcore.so is missing: no CPU fallback exists
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_oracle_in_product():
    """The product path must not import, link, load or call anything under oracle/."""
    pat = re.compile(r"import\s+oracle|from\s+oracle|momref|libmomref|oracle/")
    for f in (ROOT / "radiativetransfer.jl_amd").rglob("*"):
        if f.is_file() and (f.suffix in (".py", ".hip", ".hpp", ".h") or f.name == "Makefile"):
            assert not pat.search(f.read_text()), f


def test_fails_loudly_without_gpu(rtamd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rtamd.MomError) as e:
        rtamd.Handle(12, 3, 4, 3)
    assert e.value.code == rtamd._lib.MOM_EHIP and "HIP device" in str(e.value)
    with pytest.raises(rtamd.MomError):
        rtamd.voigt_xsec(np.ones(1), np.ones(1), np.ones(1), np.ones(1), [1], [1], np.ones(4))
    m = rtamd.scenes.scene_C1(S=4)
    with pytest.raises(rtamd.MomError):
        rtamd.rt_run(m)


def test_argument_validation(rtamd):
    lib = rtamd.load()
    h = C.c_void_p()
    assert lib.mom_create(C.byref(h), 0, 0, 1, 1, 1, 0) == rtamd._lib.MOM_EINVAL
    assert lib.mom_create(C.byref(h), 0, 10, 3, 1, 1, 0) == rtamd._lib.MOM_EINVAL  # N % nStokes != 0
    assert lib.mom_create(C.byref(h), 0, 12, 3, 1, 1, 2) == rtamd._lib.MOM_EINVAL  # dtype: 0 = Float64, 1 = Float32
    assert b"dtype" in lib.mom_last_global_error()
    assert lib.mom_sync(None) == rtamd._lib.MOM_EINVAL and lib.mom_destroy(None) == 0
    g = np.ones(4)
    assert lib.mom_voigt_xsec(0, 1, None, None, None, None, None, None, 4, rtamd._lib.dp(g), rtamd._lib.dp(g)) == -1


@pytest.mark.parametrize("flags", [[], ["-DMOM_WAVES=4", "-DMOM_TJ=3", "-DMOM_NO_STRAIGHT", "-DMOM_NS=mom4"]])
def test_strip_images_fit_the_cu_lds(tmp_path, flags):
    """strip_lds_bytes(N, ns) <= 160 KB for every strip image size and 1..4 Stokes components per stream, in the
    8-wave and the 4-wave build (tools/lds_budget_check.hip: host code only, compiled by hipcc, needs no GPU)."""
    import subprocess
    exe = tmp_path / "lds_budget_check"
    csrc = ROOT / "radiativetransfer.jl_amd" / "csrc"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", f"-I{csrc}",
                           *flags, str(ROOT / "tools" / "lds_budget_check.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "OVER" not in out.stdout, out.stdout
    assert out.stdout.count(" ok") == 24
    if not flags:   # the 8-wave N = 60 image with two components per stream: no tables (they would need 171 KB)
        assert "N=60 ns=2 image=149392 tables=0" in out.stdout
