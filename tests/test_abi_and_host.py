"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/roborugby_amd.h
declares (no compute calls -- there is no GPU here), and the host logic around it behaves like the reference's
gym surface (constants, spaces, error behaviour)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "roborugby_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from roborugby_amd import _lib
    from roborugby_amd.build import build_hip_library
    build_hip_library()  # hipcc cross-compiles gfx950 without a GPU
    lib = C.CDLL(_lib.LIB_PATH)
    decl = _declared_symbols()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/roborugby_amd.h but not exported"
    assert set(decl) == set(_lib.SYMBOLS), (set(decl) ^ set(_lib.SYMBOLS))
    lib.rr_abi_version.restype = C.c_int
    assert lib.rr_abi_version() == 4


def test_config_struct_matches_header_layout():
    from roborugby_amd import _lib
    # 2+4 int32, 2 double, 7 int32 (+4 pad), 2 uint64, 2 uint32 with natural alignment = 96 bytes
    assert C.sizeof(_lib.RRConfig) == 96 and _lib.RRConfig.step_budget_clocks.offset == 88
    assert _lib.RRConfig.arena_w.offset == 24 and _lib.RRConfig.reset_on_fault.offset == 56
    assert _lib.RRConfig.seed.offset == 72


def test_presets_reproduce_reference_constants(golden_dir):
    from roborugby_amd.config import PRESETS
    for name in ("T", "G"):
        c = np.load(f"{golden_dir}/kat_{name}.npz")["consts"]
        p = PRESETS[name]
        assert (p.arena_w, p.arena_h, p.game_len_steps) == (c[0], c[1], c[2])
        assert p.points_ball_travel_mult == c[3] and p.points_robot_travel_mult == c[4]
        assert (p.nr_happy, p.nr_grumpy, p.nb_pos, p.nb_neg) == tuple(int(v) for v in c[5:9])
    # algorithmic bytes per env-step
    assert PRESETS["T"].algorithmic_bytes_per_step(1) == 149
    assert PRESETS["G"].algorithmic_bytes_per_step(4) == 601


def test_spaces_and_make_fail_loudly_without_gpu():
    import torch
    import roborugby_amd as rr
    from roborugby_amd.spaces import Box, Discrete
    b = Box(-600, 600, (11,), np.float32)
    assert b.shape == (11,) and b.high[0] == 600 and b.contains(np.zeros(11, np.float32))
    d = Discrete(8)
    assert d.n == 8 and d.contains(7) and not d.contains(8)
    with pytest.raises(KeyError):
        rr.make("RoboRugby-v0")  # not constructible in the reference either
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            rr.make("RoboRugbySimpleDuel-v3", num_envs=4)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under roborugby_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "roborugby_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "rr_oracle" not in txt and "librr_oracle" not in txt, f
                assert "emu_lib" not in txt and "librr_emu" not in txt, f
