"""roborugby_amd -- MI355X-native batched lockstep simulator of RoboRugby's env.step() hot path.

Product = the HIP library (csrc/, built in-tree as libroborugby_amd.so) behind the C-ABI in
include/roborugby_amd.h, plus this thin Python mirror of the reference's gym.Env surface."""
from .config import PRESETS, Preset  # noqa: F401

__all__ = ["PRESETS", "Preset", "BatchedRoboRugbyEnv", "RoboRugbyEnv", "make", "Direction", "DebugInfo", "ShardedPipeline"]


def __getattr__(name):  # torch / the HIP library are only needed once an env is actually used
    if name in ("BatchedRoboRugbyEnv", "RoboRugbyEnv", "make", "Direction", "DebugInfo"):
        from . import env
        return getattr(env, name)
    if name == "ShardedPipeline":
        from .pipeline import ShardedPipeline
        return ShardedPipeline
    raise AttributeError(name)
