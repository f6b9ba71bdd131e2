"""Oracle primitives vs known-answer tables captured from the reference (tests/golden/kat_*.npz).

Bar: bit-exact (the oracle repeats the reference's fp64 operation order and libm calls)."""
import ctypes as C
import json

import numpy as np
import pytest

import oracle_lib as ol


def _load(golden_dir, preset):
    return np.load(f"{golden_dir}/kat_{preset}.npz")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_line_intersection_and_within(golden_dir, preset):
    k = _load(golden_dir, preset)
    L = ol.lib()
    for seg, want, win in zip(k["li_in"], k["li_out"], k["li_within"]):
        seg = np.ascontiguousarray(seg)
        out = np.zeros(2)
        w3 = np.zeros(3, np.uint8)
        L.rro_kat_line_intersection(_dp(seg), _dp(out), w3.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert np.array_equal(out, want), (seg, out, want)
        assert np.array_equal(w3, win)


@pytest.mark.parametrize("preset", ["T"])
def test_distance_and_angle(golden_dir, preset):
    k = _load(golden_dir, preset)
    L = ol.lib()
    for p, d, a in zip(k["pt_in"], k["pt_dist"], k["pt_angle"]):
        p = np.ascontiguousarray(p)
        out = np.zeros(2)
        L.rro_kat_dist_angle(_dp(p), _dp(out))
        assert out[0] == d and out[1] == a, (p, out, d, a)


@pytest.mark.parametrize("preset", ["T"])
def test_floatrect_rotation_corners_copy(golden_dir, preset):
    k = _load(golden_dir, preset)
    L = ol.lib()
    for i, (inp, want, cp) in enumerate(zip(k["fr_in"], k["fr_out"], k["fr_copy"])):
        inp = np.ascontiguousarray(inp)
        out, c7 = np.zeros(15), np.zeros(7)
        L.rro_kat_floatrect(_dp(inp), _dp(out), _dp(c7))
        assert np.array_equal(out, want), (i, inp, out - want)
        assert np.array_equal(c7, cp), (i, inp, c7 - cp)


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_two_way_lidar(golden_dir, preset):
    k = _load(golden_dir, preset)
    L = ol.lib()
    W, H = k["consts"][0], k["consts"][1]
    for inp, want in zip(k["lidar_in"], k["lidar_out"]):
        inp = np.ascontiguousarray(inp)
        out = np.zeros(2)
        L.rro_kat_lidar(W, H, _dp(inp), _dp(out))
        assert np.array_equal(out, want), (inp, out, want)


@pytest.mark.parametrize("preset", ["T"])
def test_collision_predicates(golden_dir, preset):
    k = _load(golden_dir, preset)
    L = ol.lib()
    for inp, want in zip(k["brc_in"], k["brc_out"]):
        assert L.rro_kat_ball_robot(_dp(np.ascontiguousarray(inp))) == int(want), inp
    assert 0.15 < k["brc_out"].mean() < 0.85  # the table exercises both answers
    for inp, want in zip(k["rrc_in"], k["rrc_out"]):
        assert L.rro_kat_robots(_dp(np.ascontiguousarray(inp))) == int(want), inp
    assert 0.15 < k["rrc_out"].mean() < 0.85


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_constants(golden_dir, preset):
    k = _load(golden_dir, preset)
    c = k["consts"]
    cfg = ol.PRESETS[preset]
    assert (cfg["W"], cfg["H"], cfg["game_len"]) == (c[0], c[1], c[2])
    assert (cfg["nr_h"], cfg["nr_g"], cfg["nb_p"], cfg["nb_n"]) == tuple(int(v) for v in c[5:9])
    assert c[9] == 12 and c[10] == 16.0
    meta = json.loads(str(k["meta"]))
    assert meta["preset"] == preset
