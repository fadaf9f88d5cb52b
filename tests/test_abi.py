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
