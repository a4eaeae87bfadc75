"""
ctypes binding of include/rsf_abi.h.

The product loads exactly one library through this module: csrc/librsf_hip.so (hand-written
gfx950 kernels).  There is no CPU fallback: if the library is missing, cannot be loaded, or no
GPU is visible, `load()` raises.  `bind()` only attaches prototypes to an already opened
library handle, so the test-suite can type the oracle library with the same declarations.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# RSF_HIP_LIB: alternative build of the same HIP library (kernel A/B experiments); never a CPU library
LIB_PATH = os.environ.get("RSF_HIP_LIB") or os.path.join(HERE, "csrc", "librsf_hip.so")

ABI_VERSION = 1
OK = 0
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_RADIATION_DAMPING = 1
FLAG_FP32_SOLVE = 2
FLAG_DOP853 = 4
ADAPT_NONE, ADAPT_REFERENCE_DICT, ADAPT_AM = 0, 1, 2
ADAPT_MODES = {"none": ADAPT_NONE, "reference_dict": ADAPT_REFERENCE_DICT, "am": ADAPT_AM}
MAX_PARAMS = 3
ERR_NOT_POSDEF = -6
# rsf_mcmc_counters: index = RSF_CNT_* of include/rsf_abi.h
COUNTERS = ("accepted", "evaluated", "nonfinite", "out_of_bounds", "early_rejected", "wave_solves", "wave_skips",
            "steps_tight", "steps_narrow", "steps_wide", "steps_full", "steps_redone", "lane_steps")


class RsfError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rsf error {code}: {message}")
        self.code = code


class Config(ctypes.Structure):
    _fields_ = [
        ("size", c_uint32),
        ("version", c_uint32),
        ("device", c_int32),
        ("mem_space", c_int32),
        ("stream", c_void_p),
        ("block_threads", c_uint32),
        ("cpu_threads", c_uint32),
    ]


class Model(ctypes.Structure):
    _fields_ = [
        ("size", c_uint32),
        ("flags", c_uint32),
        ("nsteps", c_int32),
        ("substeps", c_int32),
        ("t_start", c_double),
        ("t_final", c_double),
        ("mu_ref", c_double),
        ("V_ref", c_double),
        ("k1", c_double),
        ("mu_t_zero", c_double),
        ("a", c_double),
        ("b", c_double),
    ]


class McmcConfig(ctypes.Structure):
    _fields_ = [
        ("size", c_uint32),
        ("n_params", c_int32),
        ("n_chains", c_int64),
        ("chain_offset", c_int64),
        ("seed", c_uint64),
        ("n0", c_double),
        ("prior_len", c_int32),
        ("adapt_mode", c_int32),
        ("adapt_interval", c_int32),
        ("n_groups", c_int32),
        ("fd_rel_step", c_double),
        ("lo", c_double * MAX_PARAMS),
        ("hi", c_double * MAX_PARAMS),
    ]


_P = c_void_p  # array arguments travel as raw addresses (host ndarray or device tensor)

PROTOTYPES = {
    "rsf_version": (c_int, []),
    "rsf_backend": (c_char_p, []),
    "rsf_build_id": (c_char_p, []),
    "rsf_last_error": (c_char_p, []),
    "rsf_device_count": (c_int, []),
    "rsf_create": (c_int, [POINTER(Config), POINTER(c_void_p)]),
    "rsf_destroy": (c_int, [c_void_p]),
    "rsf_sync": (c_int, [c_void_p]),
    "rsf_set_model": (c_int, [c_void_p, POINTER(Model)]),
    "rsf_model_nout": (c_int, [c_void_p, POINTER(c_int32)]),
    "rsf_forward_batch": (c_int, [c_void_p, c_int64, _P, _P, _P, _P, _P, _P]),
    "rsf_mcmc_init": (c_int, [c_void_p, POINTER(McmcConfig), _P, _P]),
    "rsf_mcmc_get_state": (c_int, [c_void_p, _P, _P, _P, _P]),
    "rsf_mcmc_set_state": (c_int, [c_void_p, _P, _P, _P, _P]),
    "rsf_mcmc_run": (c_int, [c_void_p, c_int64, _P, _P, _P]),
    "rsf_mcmc_replay": (c_int, [c_void_p, c_int64, _P, _P, _P, _P, _P, _P]),
    "rsf_mcmc_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "rsf_mcmc_counters": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "rsf_mcmc_init_state": (c_int, [c_void_p, POINTER(McmcConfig), _P, _P, _P, _P]),
    "rsf_mcmc_propose": (c_int, [c_void_p, _P, _P, _P]),
    "rsf_mcmc_replay_ssq": (c_int, [c_void_p, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "rsf_pool_summary": (c_int, [c_void_p, c_int64, _P, c_int64, POINTER(c_double)]),
    "rsf_pool_kde": (c_int, [c_void_p, c_int64, _P, c_int64, c_int32, _P, c_double, _P]),
    "rsf_pool_histogram": (c_int, [c_void_p, c_int64, _P, c_int64, c_int32, c_double, c_double, _P]),
    "rsf_comm_unique_id": (c_int, [POINTER(c_uint8)]),
    "rsf_comm_init": (c_int, [c_void_p, c_int32, c_int32, POINTER(c_uint8)]),
    "rsf_comm_destroy": (c_int, [c_void_p]),
    "rsf_pool_allgather": (c_int, [c_void_p, _P, c_int64, _P]),
    "rsf_pool_allreduce_sum": (c_int, [c_void_p, _P, c_int64]),
    "rsf_comm_init_all": (c_int, [POINTER(c_void_p), c_int32]),
    "rsf_pool_allgather_all": (c_int, [POINTER(c_void_p), c_int32, POINTER(c_void_p), c_int64, POINTER(c_void_p)]),
    "rsf_pool_allreduce_sum_all": (c_int, [POINTER(c_void_p), c_int32, POINTER(c_void_p), c_int64]),
    "rsf_philox4x32_10": (c_int, [POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "rsf_mcmc_adapt": (c_int, [c_int32, c_int32, _P, c_int32, c_int32, _P]),
    "rsf_mcmc_draws": (c_int, [c_uint64, c_int64, c_int64, c_int32, c_double, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
}


def bind(lib):
    """Attach the rsf_abi.h prototypes to an opened CDLL; raises if a symbol is missing."""
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError names the missing symbol
        fn.restype, fn.argtypes = restype, argtypes
    return lib


_lib = None


def hip_runtime_path():
    """The ONE HIP runtime this process uses.  librsf_hip.so is linked without a runtime dependency
    because two HIP/HSA runtimes in a process cannot both own the GPU: PyTorch's ROCm wheel bundles
    its own libamdhip64.so (no SONAME, so the loader does not unify it with /opt/rocm's), and the
    engine shares device memory and streams with torch.  Order: $RSF_HIP_RUNTIME, torch's bundled
    runtime, then $ROCM_PATH or /opt/rocm."""
    env = os.environ.get("RSF_HIP_RUNTIME")
    if env:
        return env
    try:
        import torch

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    except ImportError:
        pass
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "libamdhip64.so")):
            return os.path.join(root, "lib", "libamdhip64.so")
    raise RsfError(-2, "no HIP runtime (libamdhip64.so) found; set RSF_HIP_RUNTIME")


def load():
    """The product library (HIP).  Fails loudly; never substitutes a CPU implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsfError(-2, f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        ctypes.CDLL(hip_runtime_path(), mode=ctypes.RTLD_GLOBAL)  # resolves the library's hip* symbols
        lib = bind(ctypes.CDLL(LIB_PATH))
        if lib.rsf_version() != ABI_VERSION:
            raise RsfError(-1, f"ABI version mismatch: library {lib.rsf_version()}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(lib, status):
    if status != OK:
        raise RsfError(status, lib.rsf_last_error().decode("utf-8", "replace"))


def require_device(lib):
    n = lib.rsf_device_count()
    if n <= 0:
        raise RsfError(-2, "no HIP device visible (rsf_device_count() = %d): %s; there is no CPU fallback"
                       % (n, lib.rsf_last_error().decode("utf-8", "replace")))
    return n
