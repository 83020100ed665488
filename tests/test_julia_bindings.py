"""The Julia side of the drop-in boundary (integration/MomCore.jl, generated; integration/MomCoreRT.jl, hand-written)
checked WITHOUT Julia: the generated file is re-parsed with this test's own regexes and compared, symbol by symbol and
argument by argument, with (1) include/momcore.h, parsed here independently of the generator, and (2) the hand-written
ctypes table radiativetransfer.jl_amd/_lib.py::SIGNATURES -- the binding every GPU parity test runs through.  The host
layer may only call generated wrappers, with the header's arity.  INTEGRATION.md's tagged Julia blocks must be verbatim
excerpts.  Reference seams: src/Architectures.jl:20-55, src/CoreRT/rt_run.jl:19-21,41-230."""
import ctypes as C
import importlib.util
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
JL = ROOT / "integration" / "MomCore.jl"
HOST = ROOT / "integration" / "MomCoreRT.jl"
DOC = ROOT / "INTEGRATION.md"

C_TO_JL = {"int": "Cint", "double": "Cdouble", "size_t": "Csize_t", "mom_t**": "Ref{Ptr{Cvoid}}", "mom_t*": "Ptr{Cvoid}",
           "const mom_t*": "Ptr{Cvoid}", "const double*": "Ptr{Cdouble}", "double*": "Ptr{Cdouble}", "const int*": "Ptr{Cint}",
           "int*": "Ptr{Cint}", "const void*": "Ptr{Cvoid}", "void*": "Ptr{Cvoid}", "unsigned long long*": "Ptr{Culonglong}",
           "const char*": "Cstring"}


def header_prototypes():
    """name -> (julia return type, [julia arg types]) from include/momcore.h (this test's own parser)"""
    txt = (ROOT / "include" / "momcore.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for stmt in txt.split(";"):
        m = re.search(r"^\s*([\w\s\*]+?)\s*\b(mom_\w+)\s*\((.*)\)\s*$", " ".join(stmt.split()))
        if not m or "typedef" in stmt:
            continue
        ret, name, args = m.groups()

        def ctype(decl, has_name=True):
            decl = decl.strip()
            if has_name:
                decl = re.sub(r"\w+$", "", decl)              # drop the parameter name
            return re.sub(r"\s*\*\s*", "*", " ".join(decl.split())).replace("* *", "**").strip()

        argt = [] if args.strip() in ("", "void") else [C_TO_JL[ctype(a)] for a in args.split(",")]
        out[name] = (C_TO_JL[ctype(ret, has_name=False)], argt)
    return out


def split_top(s):
    """split on commas that are outside (), [], {}"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_bindings():
    """name -> (ret, [arg types], n call arguments, wrapper parameter count) parsed from integration/MomCore.jl"""
    txt = JL.read_text()
    out = {}
    for m in re.finditer(r"^(mom_\w+)\(([^)]*)\) =\n    ccall\(\(:(mom_\w+), libmomcore\), ([\w\{\}]+), \((.*?)\)((?:, [\wₐ-ₜ]+)*)\)$", txt, flags=re.M):
        fn, params, sym, ret, tup, callargs = m.groups()
        assert fn == sym, (fn, sym)
        types = [t for t in split_top(tup) if t]
        nparams = len([p for p in params.split(",") if p.strip()])
        ncall = len([a for a in callargs.split(",") if a.strip()])
        assert sym not in out, f"{sym} bound twice"
        out[sym] = (ret, types, ncall, nparams)
    return out


def test_generated_file_is_current():
    spec = importlib.util.spec_from_file_location("gen_julia_bindings", ROOT / "tools" / "gen_julia_bindings.py")
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    protos, enums = g.parse_header()
    assert JL.read_text() == g.emit(protos, enums, g.reference_citations()), "integration/MomCore.jl is stale: run tools/gen_julia_bindings.py"
    assert DOC.read_text() == g.sync_doc(DOC.read_text(), HOST.read_text()), "INTEGRATION.md excerpts are stale"


def test_every_symbol_bound_with_the_headers_types():
    hdr, jl = header_prototypes(), julia_bindings()
    assert len(hdr) >= 65
    assert set(hdr) == set(jl), set(hdr) ^ set(jl)
    for name, (ret, argt) in hdr.items():
        jret, jtypes, ncall, nparams = jl[name]
        assert jret == ret, (name, jret, ret)
        assert jtypes == argt, (name, jtypes, argt)
        assert ncall == nparams == len(argt), (name, ncall, nparams, len(argt))
    # the SYMBOLS tuple lists the same set
    syms = re.search(r"^const SYMBOLS = \((.*)\)$", JL.read_text(), flags=re.M).group(1)
    assert {s.strip().lstrip(":") for s in syms.split(",")} == set(hdr)


def test_julia_types_agree_with_the_ctypes_binding(rtamd):
    """the ctypes table is hand-written and exercised on the GPU; it must describe the same ABI as the generated Julia"""
    L = rtamd._lib
    ct = {C.c_int: "Cint", C.c_double: "Cdouble", C.c_size_t: "Csize_t", L.c_dp: "Ptr{Cdouble}", L.c_ip: "Ptr{Cint}",
          C.c_void_p: "Ptr{Cvoid}", C.POINTER(C.c_void_p): "Ref{Ptr{Cvoid}}", C.c_char_p: "Cstring",
          C.POINTER(C.c_ulonglong): "Ptr{Culonglong}", C.POINTER(C.c_ubyte): "Ptr{Cvoid}"}
    jl = julia_bindings()
    assert set(jl) == set(L.SIGNATURES)
    for name, (res, args) in L.SIGNATURES.items():
        jret, jtypes, _, _ = jl[name]
        assert ct[res] == jret, (name, res, jret)
        assert [ct[a] for a in args] == jtypes, (name, [ct[a] for a in args], jtypes)


def test_enumerators_match_the_header():
    txt = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "momcore.h").read_text(), flags=re.S)
    want = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", txt, flags=re.S):
        for item in body.split(","):
            if "=" in item:
                k, v = item.split("=")
                want[k.strip()] = int(v.strip(), 0)
    got = {k: int(v) for k, v in re.findall(r"^const (MOM_\w+) = Cint\((-?\d+)\)$", JL.read_text(), flags=re.M)}
    assert got == want and len(got) >= 40


def host_calls(txt):
    """(symbol, bang, n top-level arguments) of every MomCore.mom_*( ... ) call"""
    out = []
    for m in re.finditer(r"MomCore\.(mom_\w+)(!?)\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        out.append((m.group(1), m.group(2), len(split_top(txt[m.end():i - 1]))))
    return out


def test_host_layer_calls_only_generated_wrappers_with_the_right_arity():
    hdr = header_prototypes()
    jl_txt = JL.read_text()
    calls = host_calls(HOST.read_text())
    assert len(calls) >= 30
    for sym, bang, n in calls:
        assert sym in hdr, f"MomCoreRT.jl calls {sym}, which include/momcore.h does not declare"
        assert n == len(hdr[sym][1]), f"MomCoreRT.jl calls {sym}{bang} with {n} arguments, the header declares {len(hdr[sym][1])}"
        if bang:
            assert re.search(rf"^{sym}!\(", jl_txt, flags=re.M), f"no throwing wrapper {sym}! is generated"
    # the entry points of the three rt_run flavours, the operator level, the collective and the absorption path are all used
    used = {c[0] for c in calls}
    for need in ("mom_create", "mom_set_streams", "mom_scene_set", "mom_scene_set_surface", "mom_rt_run", "mom_get_RT", "mom_get_hdr",
                 "mom_rrs_set", "mom_rrs_set_shard", "mom_scene_set_rrs", "mom_rt_run_rrs", "mom_get_RT_rrs", "mom_rt_run_multisensor",
                 "mom_elemental", "mom_doubling", "mom_interaction", "mom_surface_lambertian", "mom_postprocess", "mom_batch_inv",
                 "mom_batched_mul", "mom_comm_unique_id", "mom_comm_init", "mom_allgather_RT", "mom_absorption_set_lines",
                 "mom_voigt_tau_abs_profile", "mom_voigt_xsec", "mom_destroy"):
        assert need in used, need
    # the seam: the architecture type and the three methods
    host = HOST.read_text()
    assert "struct MI355X <: AbstractArchitecture" in host
    assert re.search(r"function rt_run\(RS_type::noRS, model::vSmartMOM_Model, iBand, arch::MI355X\)", host)
    assert re.search(r"function rt_run\(RS_type::RRS, model::vSmartMOM_Model, iBand, arch::MI355X", host)
    assert re.search(r"function rt_run_test_ms\(RS_type::noRS, sensor_levels::Vector\{Int64\}, model::vSmartMOM_Model, iBand, arch::MI355X\)", host)


def test_integration_md_blocks_are_excerpts():
    doc, host = DOC.read_text(), HOST.read_text()
    blocks = re.findall(r"```julia\n# excerpt: integration/MomCoreRT\.jl \[(\w+)\]\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 8
    for name, body in blocks:
        assert body.strip(), name
        assert f"# >>> {name}\n{body}# <<< {name}\n" in host, f"INTEGRATION.md block [{name}] is not a verbatim excerpt"
    # no hand-written raw ccall is left in the document: every binding shown comes from the checked files
    assert not re.search(r"ccall\(\(:mom_", doc)
