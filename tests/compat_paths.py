"""sys.path set-up for the tests that import the `robo_rugby` compat shim (compat/) the way a caller of the reference
would: `import gym, robo_rugby` -- with compat/gym_minimal standing in when no real gym is installed."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_compat(game_mode):
    os.environ["ROBO_RUGBY_GAME_MODE"] = "1" if game_mode else "0"
    for name in [m for m in sys.modules if m == "robo_rugby" or m.startswith("robo_rugby.")]:
        del sys.modules[name]  # constants are read at import (like the reference's module-level GAME_MODE)
    compat = os.path.join(REPO, "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    if "gym" not in sys.modules and importlib.util.find_spec("gym") is None:
        sys.path.insert(0, os.path.join(compat, "gym_minimal"))
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_registry", None) is not None:
        sys.modules["gym"]._registry.clear()  # the stand-in's registry: a re-import registers again
