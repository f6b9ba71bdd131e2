"""Unit tests of the kernel source's own arithmetic primitives (compiled by the host emulation harness): the one-rounding
`x % 360` that replaces libm's fmod must be BIT-identical to Python's float %, and the fma-based sincos must stay within
1.2e-16 absolute of libm on the angle range the simulator uses."""
import ctypes as C
import math
import struct

import numpy as np

import emu_lib as el


def _bits(x):
    return struct.pack("<d", x)


def test_py_mod_360_is_bit_identical_to_python():
    f = el.lib().emu_py_mod360
    rng = np.random.RandomState(0)
    vals = list(rng.uniform(0, 2000, 200000)) + list(rng.uniform(0, 1e6, 50000)) + list(-rng.uniform(0, 2000, 50000))
    for k in range(0, 8):  # the neighbourhoods of the multiples of 360, both signs: where a rounded quotient goes wrong
        for sgn in (1.0, -1.0):
            x = sgn * 360.0 * k
            vals += [x, np.nextafter(x, 1e9), np.nextafter(x, -1e9)] + list(x + rng.uniform(-1e-12, 1e-12, 2000))
    # the values the rotation bookkeeping really produces: multiples of 0.6 / 1.2 accumulated in floating point, + 720
    r = 0.0
    for i in range(20000):
        r = (r + (0.6 if i % 3 else 1.2) + 720) % 360
        vals += [r + 720, r + 45 + 720, r + 180]
    bad = 0
    for a in vals:
        a = float(a)
        if _bits(f(a)) != _bits(a % 360.0):
            bad += 1
    assert bad == 0, bad


def test_sincos_accuracy():
    L = el.lib()
    s, c = C.c_double(), C.c_double()
    rng = np.random.RandomState(1)
    worst = 0.0
    for deg in list(rng.uniform(-90, 810, 100000)) + [0.0, 90.0, 180.0, 270.0, 360.0, 45.0, 0.6, 1.2]:
        x = math.radians(float(deg))
        L.emu_sincos(x, C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(x)), abs(c.value - math.cos(x)))
    assert worst < 1.2e-16, worst
