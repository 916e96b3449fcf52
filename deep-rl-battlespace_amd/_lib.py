"""ctypes binding of include/battlespace_hip.h.  No fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library is the in-tree build.  BSX_LIB_PATH points the binding at ANOTHER build of the same sources (profiling
# variants under csrc/variants/, built by tools/): diagnostic builds never overwrite the product file, and one whose
# bsx_build_flags() is non-zero (results are not the reference's) is refused unless BSX_ALLOW_DIAG=1 is set as well.
LIB_PATH = os.environ.get("BSX_LIB_PATH") or os.path.join(_HERE, "csrc", "libbattlespace_hip.so")

ABI_VERSION = 14
BULLET_SLOTS = 12
MAX_N = 16
MAX_T = 65535
ACTOR_F32, ACTOR_BF16X3, ACTOR_BF16X6 = 0, 1, 2
F_AUTO_RESET = 1
F_EMPTY_CALL = 2
F_WIDE_OFFSETS = 4
F_ONE_WAVE = 8
ACT_I32, ACT_LOGITS_F32 = 0, 1
ACT_F32, ACT_F64, ACT_F32X4 = 0, 1, 2
WINNER_NAMES = ("none", "red", "blue", "tie")

c_void_p, c_int, c_int64, c_uint32, c_uint64, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                          ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t)


class BsxRewards(ctypes.Structure):
    _fields_ = [("hit_base_reward", ctypes.c_double), ("hit_plane_reward", ctypes.c_double),
                ("miss_punishment", ctypes.c_double), ("die_punishment", ctypes.c_double),
                ("lose_punishment", ctypes.c_double)]


class BsxActorNoise(ctypes.Structure):
    _fields_ = [("gaussian_std", ctypes.c_float), ("ou_scale", ctypes.c_float), ("ou_theta", ctypes.c_float),
                ("ou_sigma", ctypes.c_float), ("ou_mu", ctypes.c_float), ("ou_state", ctypes.c_void_p),
                ("env_done", ctypes.c_void_p), ("z_inject", ctypes.c_void_p), ("ou_keep", ctypes.c_int),
                ("sample_mode", ctypes.c_int), ("temperature", ctypes.c_float), ("logp", ctypes.c_void_p), ("u_inject", ctypes.c_void_p),
                ("value_weights", ctypes.c_void_p), ("value", ctypes.c_void_p)]


EXPORT_FIELDS = ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner",
                 "bl_live", "bl_x", "bl_y", "bl_dir", "counters")


class BsxExport(ctypes.Structure):
    _fields_ = [(f, c_void_p) for f in EXPORT_FIELDS]


# name -> (restype, argtypes): every symbol include/battlespace_hip.h declares
SIGNATURES = {
    "bsx_abi_version": (c_int, []),
    "bsx_build_flags": (c_int, []),
    "bsx_state_bytes": (c_int, [c_int64, c_int, ctypes.POINTER(c_size_t)]),
    "bsx_state_init": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "bsx_state_release": (c_int, [c_void_p]),
    "bsx_reset": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_uint64, c_uint64, c_int64, c_void_p, c_void_p]),
    "bsx_step_discrete": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_uint64, c_int64, c_void_p]),
    "bsx_step_continuous": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_uint64, c_int64, c_void_p]),
    "bsx_step_discrete_range": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_uint64, c_int64, c_void_p]),
    "bsx_step_continuous_range": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_uint64, c_int64, c_void_p]),
    "bsx_step_many_discrete": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_int, c_uint64, c_int64, c_void_p]),
    "bsx_step_many_continuous": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, c_int, c_uint64, c_int64, c_void_p]),
    "bsx_rollout_discrete": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, ctypes.POINTER(BsxActorNoise), c_uint64,
                                     c_uint64, c_void_p, c_uint64, c_int64, c_void_p]),
    "bsx_rollout_continuous": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, ctypes.POINTER(BsxRewards), c_uint32, ctypes.POINTER(BsxActorNoise), c_uint64,
                                       c_uint64, c_void_p, c_uint64, c_int64, c_void_p]),
    "bsx_observe": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "bsx_export_state": (c_int, [c_void_p, c_int64, c_int, ctypes.POINTER(BsxExport), c_void_p]),
    "bsx_tie_tick": (c_int, [c_int]),
    "bsx_stream_synchronize": (c_int, [c_void_p]),
    "bsx_host_device_pointer": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "bsx_selftest_atan2": (c_int, [c_int, c_void_p, c_void_p]),
    "bsx_instinct_discrete": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "bsx_instinct_continuous": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_uint64, c_uint64, c_void_p, c_void_p]),
    "bsx_actor_blob_floats": (c_int, [c_int, ctypes.POINTER(c_int)]),
    "bsx_actor_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, ctypes.POINTER(BsxActorNoise), c_uint64, c_uint64,
                                  c_void_p, c_int64, c_void_p]),
}

_lib = None


def load():
    """dlopen the in-tree HIP library and type its entry points.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built.  Run "
            f"`python deep-rl-battlespace_amd/build.py` (needs hipcc; cross-compiles for gfx950 without a GPU). "
            f"There is no CPU fallback for the step() path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.bsx_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {v}, binding expects {ABI_VERSION}; rebuild the extension")
    flags = lib.bsx_build_flags()
    if flags and os.environ.get("BSX_ALLOW_DIAG") != "1":
        raise ImportError(f"{LIB_PATH} is a diagnostic build (bsx_build_flags() = {flags:#x}): its results are not the "
                          f"reference's.  Rebuild the product library, or set BSX_ALLOW_DIAG=1 for a timing-only run.")
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError(f"{what}: invalid argument (BSX_E_ARG)")
    if rc == -2:
        raise ValueError(f"{what}: misaligned pointer (BSX_E_ALIGN)")
    if rc == -3:
        raise ValueError(f"{what}: this state block was advanced by the other action family (BSX_E_FAMILY): a block is discrete or "
                         f"continuous from its first step until every game is reset")
    raise RuntimeError(f"{what}: HIP error {rc}")
