"""Provenance of the committed fixtures: when the reference tree is present (the build container), the generator is run again --
one fresh process, its own invocation line -- into a scratch directory and must reproduce the committed file array for array.
`mix_T` is the part VERDICT r2 found irreproducible (the constructor's placement came from Python's unseeded global random);
`kat_T` is the cheapest one; `thrust_D` pins the third shape's continuous entry (its `mix_D` regenerates the same way: 40 s, left to `gen_golden.py D mix`), `thrust_X` / `reset_X` the shape outside the library's
built list -- the reference with its four entity-count constants patched, nothing else (round 4).  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle", "refgen"))
from load_reference import reference_available  # noqa: E402  (reads no reference code: only checks that the tree exists)

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("preset,part,name", [("T", "mix", "mix_T.npz"), ("T", "kat", "kat_T.npz"), ("D", "thrust", "thrust_D.npz"),
                                               ("X", "thrust", "thrust_X.npz"), ("X", "reset", "reset_X.npz")])
def test_generator_reproduces_the_committed_fixture(tmp_path, golden_dir, preset, part, name):
    env = dict(os.environ, RR_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(REPO, "oracle", "refgen", "gen_golden.py"), preset, part], env=env,
                          stdout=subprocess.DEVNULL, timeout=840)
    new, old = np.load(tmp_path / name), np.load(os.path.join(golden_dir, name))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if k == "meta":
            continue
        assert np.array_equal(new[k], old[k], equal_nan=(new[k].dtype.kind == "f")), k
