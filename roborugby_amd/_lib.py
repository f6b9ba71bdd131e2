"""ctypes binding of the C-ABI in include/roborugby_amd.h.

There is NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises."""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime this process loads FIRST so that the
# library below (linked against the same soname) shares it -- device pointers and streams handed across the
# C-ABI are only meaningful inside one runtime instance.
import torch  # noqa: F401  (import order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
# RR_LIB_PATH lets tools/ A/B-test alternative builds of the same ABI (kernel tuning); default = the in-tree build
LIB_PATH = os.environ.get("RR_LIB_PATH", os.path.join(HERE, "libroborugby_amd.so"))
LIB_PATH_EXACT = os.path.join(HERE, "libroborugby_amd_exact.so")  # the exact-trig parity build (build.py: -DRR_EXACT_TRIG=1)


class RRConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("num_envs", C.c_int32),
        ("nr_happy", C.c_int32), ("nr_grumpy", C.c_int32), ("nb_pos", C.c_int32), ("nb_neg", C.c_int32),
        ("arena_w", C.c_double), ("arena_h", C.c_double),
        ("game_len_steps", C.c_int32), ("game_mode", C.c_int32), ("time_limit", C.c_int32), ("auto_reset", C.c_int32),
        ("reset_on_fault", C.c_int32),
        ("dtype", C.c_int32), ("device", C.c_int32),
        ("seed", C.c_uint64), ("arena_offset", C.c_uint64),
        ("step_budget_clocks", C.c_uint32), ("reserved_", C.c_uint32),
    ]


class RRDqnArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("batch", C.c_int32),
        ("eval_params", C.c_void_p * 6), ("target_params", C.c_void_p * 6),
        ("state_memory", C.c_void_p), ("new_state_memory", C.c_void_p), ("action_memory", C.c_void_p),
        ("reward_memory", C.c_void_p), ("terminal_memory", C.c_void_p), ("batch_index", C.c_void_p),
        ("gamma", C.c_float), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("loss_out", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/roborugby_amd.h declares
_vp = C.c_void_p
SYMBOLS = {
    "rr_abi_version": (C.c_int, []),
    "rr_last_error": (C.c_char_p, []),
    "rr_exact_trig": (C.c_int, []),
    "rr_create": (C.c_int, [C.POINTER(RRConfig), C.POINTER(_vp)]),
    "rr_destroy": (C.c_int, [_vp]),
    "rr_reset": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "rr_step": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_step_f64": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_set_step_budget": (C.c_int, [_vp, C.c_uint32]),
    "rr_step_thrust": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_step_thrust_f64": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_rollout": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_observe": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "rr_observe_f64": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "rr_set_reward_program": (C.c_int, [_vp, _vp, C.c_int32]),
    "rr_observe_kind": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp]),
    "rr_observe_kind_f64": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp]),
    "rr_track_prior_step": (C.c_int, [_vp, C.c_int32, _vp]),
    "rr_set_goal_scoring": (C.c_int, [_vp, C.c_int32, _vp]),
    "rr_goal_scores": (C.c_int, [_vp, _vp, _vp]),
    "rr_set_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_set_poses": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rr_reset_to_poses": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_get_scratch_rect": (C.c_int, [_vp, _vp, _vp]),
    "rr_set_scratch_rect": (C.c_int, [_vp, _vp, _vp]),
    "rr_get_episode_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rr_set_episode_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rr_episode_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_policy_chase": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_float, C.c_uint64, _vp, C.c_int32, _vp]),
    "rr_dqn_create": (C.c_int, [C.c_int32, C.POINTER(_vp)]),
    "rr_dqn_destroy": (C.c_int, [_vp]),
    "rr_dqn_last_error": (C.c_char_p, []),
    "rr_dqn_update": (C.c_int, [_vp, C.POINTER(RRDqnArgs), _vp]),
    "rr_dqn_grads": (C.c_int, [_vp, C.POINTER(RRDqnArgs), _vp, _vp]),
    "rr_dqn_param_count": (C.c_int32, []),
    "rr_dqn_store": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rr_dqn_act": (C.c_int, [_vp, C.POINTER(_vp * 6), _vp, C.c_int32, C.c_float, C.c_uint64, C.c_uint32, _vp, _vp, _vp]),
    "rr_dqn_adam_state": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int64), C.c_int32, _vp]),
    "rr_probe_hbm_copy": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "rr_state_bytes_per_env": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rr_lanes_per_env": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
}

_lib = None
_lib_exact = None


def load(exact=False):
    """Loads libroborugby_amd.so (built by roborugby_amd.build / __graft_entry__.build); exact=True: the exact-trig parity build
    libroborugby_amd_exact.so (same ABI, same sources, -DRR_EXACT_TRIG=1)."""
    global _lib, _lib_exact
    if exact:
        if _lib_exact is None:
            _lib_exact = _load_path(LIB_PATH_EXACT, True)
        return _lib_exact
    if _lib is None:
        _lib = _load_path(LIB_PATH, False)
    return _lib


_shape_libs = {}


def load_shape(counts, exact=False):
    """The one-shape library of entity counts outside the built list (build.build_shape_library), compiled on demand when it is missing
    or older than its sources -- hipcc has to be there for that; the same ABI, no fallback of any kind."""
    from . import build as _build
    key = (tuple(int(c) for c in counts), bool(exact))
    _build.shape_lanes(*key[0])  # ValueError for counts the contact masks cannot hold: nothing to compile
    if key not in _shape_libs:
        path = _build.shape_lib_path(*key[0], exact=exact)
        if _build.shape_is_stale(*key[0], exact=exact) and (_build.sources_present() or not os.path.exists(path)):
            try:
                _build.build_shape_library(*key[0], verbose=True, exact=exact)
            except Exception as exc:
                raise ImportError(f"{path}: no library for {key[0][0]}+{key[0][1]} robots / {key[0][2]}+{key[0][3]} balls and it could not be "
                                  f"built ({exc}): `python -m roborugby_amd.build --shape={','.join(map(str, key[0]))}` needs hipcc") from exc
        _shape_libs[key] = _bind(C.CDLL(path), exact, False, path)
    return _shape_libs[key]


def _bind(lib, exact, overridden, path):
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    assert lib.rr_exact_trig() == (1 if exact else 0) or overridden, path
    return lib


def _load_path(path, exact):
    from . import build as _build
    overridden = not exact and "RR_LIB_PATH" in os.environ
    stale = not overridden and os.path.exists(path) and _build.sources_present() and _build.is_stale(exact)
    if not os.path.exists(path) or stale:
        # not a fallback: the only thing ever loaded is the HIP library, (re)built here if the in-tree .so is absent or
        # older than its sources (an edited rr_sim.hpp must never be tested against yesterday's kernels)
        try:
            if overridden:
                raise RuntimeError("RR_LIB_PATH points to a missing file")
            _build.build_hip_library(verbose=True, exact=exact)
        except Exception as exc:
            if stale:
                raise ImportError(f"{path} is older than its sources and could not be rebuilt ({exc})") from exc
            raise ImportError(
                f"{path} is missing and could not be built ({exc}): run `python -m roborugby_amd.build`. "
                "roborugby_amd has no CPU fallback.") from exc
    return _bind(C.CDLL(path), exact, overridden, path)


class RRError(RuntimeError):
    pass


def check(rc, what, lib=None, err_fn="rr_last_error"):
    """Raises RRError with the message of the library that returned `rc` (`lib`; the env passes its own -- the default and the
    parity library keep separate thread-local strings).  `err_fn`: the entry point that holds the message -- the rr_dqn_* calls
    write theirs to rr_dqn_last_error."""
    if rc != 0:
        libs = [lib] if lib is not None else [x for x in (_lib, _lib_exact) if x is not None] or [load()]
        msg = b"; ".join(m for m in (getattr(x, err_fn)() for x in libs) if m)
        raise RRError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def check_dqn(rc, what, lib):
    check(rc, what, lib, "rr_dqn_last_error")
