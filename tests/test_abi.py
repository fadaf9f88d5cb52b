"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/ppo_hip.h
declares with the declared arity, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "ppo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"int32_t\s+(ppo_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        protos[name] = n
    return protos


def test_header_matches_binding_table(ppo):
    protos = _header_prototypes()
    sig = ppo._lib.SIGNATURES
    assert set(protos) == set(sig), (set(protos) ^ set(sig))
    for name, n in protos.items():
        assert len(sig[name]) == n, name


def test_library_exports_every_symbol(ppo):
    L = ppo._lib.lib()
    for name in _header_prototypes():
        assert hasattr(L, name), name
    assert L.ppo_version() >= 100


def test_no_cpu_fallback(ppo):
    if ppo.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ppo.PPOError, match="no CPU fallback"):
        ppo.compute_returns(np.ones(4, np.float32), np.zeros(4, np.uint8), 1.0)
    with pytest.raises(ppo.PPOError):
        ppo.HipVecEnv(num_envs=2)
    with pytest.raises(ppo.PPOError):
        ppo.HipPolicy(72, 128, 2, 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower().replace("the oracle", "").replace("cpu oracle", ""), (dp, f)


# ---------------------------------------------------------------- Julia shim: every ccall against the header
def _split_top(s):
    """split on commas that are not nested in () or {}"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({":
            depth += 1
        elif ch in ")}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _header_signatures():
    """name -> list of normalised C parameter types"""
    src = open(os.path.join(ROOT, "include", "ppo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    handles = set(re.findall(r"typedef\s+struct\s+\w+\*\s*(\w+);", src))
    sigs = {}
    for m in re.finditer(r"int32_t\s+(ppo_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        params = []
        if args not in ("", "void"):
            for a in _split_top(args):
                a = a.replace("const ", "").strip()
                ptr = a.count("*")
                base = a.replace("*", " ").split()[0]
                if base in handles:
                    params.append("handle" + "*" * ptr)
                elif base == "ppo_allreduce_fn":
                    params.append("fnptr")
                else:
                    params.append(base + "*" * ptr)
        sigs[name] = params
    return sigs


_JL = {"Int32": "int32_t", "Int64": "int64_t", "UInt64": "uint64_t", "UInt32": "uint32_t", "UInt8": "uint8_t", "Int8": "int8_t",
       "Float32": "float", "Float64": "double"}


def _julia_matches(jl, c):
    jl = jl.strip()
    m = re.fullmatch(r"(Ptr|Ref)\{(.+)\}", jl)
    if not m:
        return _JL.get(jl) == c
    inner = m.group(2)
    if inner == "Cvoid":                                   # opaque handle, void*, or any pointer passed as C_NULL-able
        return c in ("handle", "void*", "fnptr") or c.endswith("*")
    if inner == "Ptr{Cvoid}":
        return c in ("handle*", "void**")
    if inner == "UInt8" and c == "char*":
        return True
    return _JL.get(inner, "?") + "*" == c


def test_julia_shim_ccalls_match_the_header():
    """julia/ProximalPolicyOptimizationHIP.jl cannot run here (no julia): parse every
    `ccall((:sym, LIB), Int32, (types...), args...)` and check symbol, arity, argument types and argument COUNT against
    include/ppo_hip.h."""
    sigs = _header_signatures()
    src = open(os.path.join(ROOT, "julia", "ProximalPolicyOptimizationHIP.jl")).read()
    assert _check_julia_ccalls(src, sigs) >= 25
    # the checker itself: wrong arity, wrong scalar type, wrong pointer element type, missing value are all caught
    for bad in ("ccall((:ppo_env_reset, LIB), Int32, (Ptr{Cvoid}, Int32), env.h, 1)",
                "ccall((:ppo_adam_set_lr, LIB), Int32, (Ptr{Cvoid}, Float32), o.h, 1f0)",
                "ccall((:ppo_env_get_reward, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), env.h, r)",
                "ccall((:ppo_env_get_reward, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), env.h)",
                "ccall((:ppo_no_such_symbol, LIB), Int32, (Ptr{Cvoid},), env.h)"):
        with pytest.raises(AssertionError):
            _check_julia_ccalls(bad, sigs)
    # the defects VERDICT r1 listed stay fixed
    assert "import Flux" in src and "rand(UInt64)" not in src and "Val{:hip}" not in src and "* 128" not in src


def _check_julia_ccalls(src, sigs):
    src = re.sub(r"#[^\n]*", "", src)
    calls = 0
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*(\w+),\s*\(", src):
        sym, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1                              # find the matching ')' of the type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        types = _split_top(src[m.end():i - 1])
        j, depth = i, 1                                    # ... and of the ccall itself: the remaining items are the values
        while depth:
            depth += {"(": 1, ")": -1, "[": 1, "]": -1}.get(src[j], 0)
            j += 1
        values = _split_top(src[i:j - 1].lstrip(", \n"))
        assert sym in sigs, "ccall of a symbol the header does not declare: %s" % sym
        assert ret == "Int32", sym
        assert len(types) == len(sigs[sym]), "%s: %d Julia argument types, %d C parameters" % (sym, len(types), len(sigs[sym]))
        assert len(values) == len(types), "%s: %d values for %d argument types" % (sym, len(values), len(types))
        for k, (jt, ct) in enumerate(zip(types, sigs[sym])):
            assert _julia_matches(jt, ct), "%s argument %d: Julia %s vs C %s" % (sym, k + 1, jt, ct)
        calls += 1
    return calls
