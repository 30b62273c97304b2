"""ctypes binding of include/ferromic_hip.h (libferromic_hip.so).

This is the only door from Python to the device layer.  It fails loudly when the shared library
is missing or when no GPU is present: there is no CPU fallback anywhere in ``ferromic_amd``.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMH_LIB_PATH") or os.path.join(_HERE, "lib", "libferromic_hip.so")

FMH_OK = 0
FMH_ERR_INVALID = 1
FMH_ERR_HIP = 2
FMH_ERR_NO_DEVICE = 3
FMH_ERR_UNSUPPORTED = 4

FORMULA_SPARSE = 0
FORMULA_DENSE = 1
FORMULA_SUMMARY = 2

MAX_GROUPS = 8
MAX_GROUPS_MANY = 256
MAX_PAIRS = 28
HUDSON_PACK_F64 = 10
HUDSON_PACK_U64 = 10
COMM_ID_BYTES = 128

WC_STATES = (
    "calculable",
    "components_yield_indeterminate_ratio",
    "no_inter_population_variance",
    "insufficient_data_for_estimation",
)


class FerromicHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libferromic_hip status {status}: {message}")
        self.status = status


class NoDeviceError(FerromicHipError):
    pass


class PopTotals(C.Structure):
    _fields_ = [
        ("haplotype_capacity", C.c_uint64),
        ("segregating_sites", C.c_uint64),
        ("uncallable_sites", C.c_uint64),
        ("pi_sum", C.c_double),
    ]


class HudsonTotals(C.Structure):
    _fields_ = [
        ("numerator_sum", C.c_double),
        ("denominator_sum", C.c_double),
        ("pi1_sum", C.c_double),
        ("pi2_sum", C.c_double),
        ("dxy_sum_all", C.c_double),
        ("dxy_uncallable_sites", C.c_uint64),
        ("site_num_sum", C.c_double),
        ("site_den_sum", C.c_double),
        ("sites_with_components", C.c_uint64),
        ("site_dxy_sum", C.c_double),
        ("site_dxy_skipped", C.c_uint64),
        ("pop", PopTotals * 2),
    ]


class HudsonSites(C.Structure):
    _fields_ = [
        ("d_fst", C.c_void_p),
        ("d_dxy", C.c_void_p),
        ("d_pi1", C.c_void_p),
        ("d_pi2", C.c_void_p),
        ("d_num", C.c_void_p),
        ("d_den", C.c_void_p),
        ("d_alt", C.c_void_p),
        ("d_called", C.c_void_p),
    ]


class PairDiversitySites(C.Structure):
    _fields_ = [("d_pi", C.c_void_p), ("d_theta", C.c_void_p)]


class WcTotals(C.Structure):
    _fields_ = [
        ("sum_a", C.c_double * (1 + MAX_PAIRS)),
        ("sum_b", C.c_double * (1 + MAX_PAIRS)),
        ("informative_sites", C.c_uint64 * (1 + MAX_PAIRS)),
        ("sites_attempted", C.c_uint64),
    ]


# name -> (restype, argtypes); every symbol include/ferromic_hip.h declares
_vp, _sz, _i, _u8, _u32, _u64, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_uint8, C.c_uint32, C.c_uint64, C.c_double
_P = C.POINTER
SYMBOLS = {
    "fmh_last_error": (C.c_char_p, []),
    "fmh_abi_version": (_i, []),
    "fmh_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "fmh_get_option": (_i, [C.c_char_p, _P(C.c_longlong)]),
    "fmh_device_count": (_i, [_P(_i)]),
    "fmh_device_info": (_i, [_i, C.c_char_p, _sz, _P(_i), _P(_u64)]),
    "fmh_device_alloc": (_i, [_i, _sz, _P(_vp)]),
    "fmh_device_free": (_i, [_i, _vp]),
    "fmh_device_release_scratch": (_i, [_i]),
    "fmh_copy_to_host": (_i, [_i, _vp, _vp, _sz, _vp]),
    "fmh_copy_to_device": (_i, [_i, _vp, _vp, _sz, _vp]),
    "fmh_device_zero": (_i, [_i, _vp, _sz, _vp]),
    "fmh_stream_synchronize": (_i, [_i, _vp]),
    "fmh_matrix_create": (_i, [_vp, _vp, _sz, _sz, _sz, _u8, _i, _P(_vp)]),
    "fmh_matrix_create_packed": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _u8, _i, _P(_vp)]),
    "fmh_matrix_alloc": (_i, [_sz, _sz, _sz, _i, _u8, _i, _P(_vp)]),
    "fmh_matrix_wrap": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _sz, _u8, _i, _P(_vp)]),
    "fmh_matrix_destroy": (_i, [_vp]),
    "fmh_matrix_info": (_i, [_vp, _P(_sz), _P(_sz), _P(_sz), _P(_sz), _P(_sz), _P(_i), _P(_u8), _P(_i)]),
    "fmh_matrix_device_ptrs": (_i, [_vp, _P(_vp), _P(_vp)]),
    "fmh_matrix_download": (_i, [_vp, _vp, _vp]),
    "fmh_matrix_scan_max_allele": (_i, [_vp, _P(_u8), _vp]),
    "fmh_matrix_pack": (_i, [_vp, _i]),
    "fmh_matrix_generate": (_i, [_vp, _u64, _u64, _vp, _vp, _i, _u32, _vp]),
    "fmh_groups_create": (_i, [_vp, _vp, _i, _P(_vp)]),
    "fmh_groups_destroy": (_i, [_vp]),
    "fmh_groups_sizes": (_i, [_vp, _P(_i), _P(_u64)]),
    "fmh_population_summaries": (_i, [_vp, _vp, _sz, _sz, _i, _vp, _vp, _P(PopTotals), _vp]),
    "fmh_hudson_sweep": (_i, [_vp, _vp, _sz, _sz, _i, _P(HudsonSites), _P(HudsonTotals), _vp]),
    "fmh_hudson_from_counts": (_i, [_i, _vp, _vp, _u64, _vp, _vp, _u64, _sz, _i, _i, _P(HudsonSites), _P(HudsonTotals), _vp]),
    "fmh_diversity_sites": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _P(PopTotals), _vp]),
    "fmh_pair_region_sweep": (_i, [_vp, _vp, _sz, _sz, _i, _i, _P(PairDiversitySites), _P(HudsonSites), _P(HudsonTotals), _vp]),
    "fmh_wc_sweep": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _P(WcTotals), _vp]),
    "fmh_wc_sweep_many": (_i, [_vp, _vp, _i, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fmh_pairwise_differences": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "fmh_hudson_totals_pack": (_i, [_P(HudsonTotals), _P(_d), _P(_u64)]),
    "fmh_hudson_totals_unpack": (_i, [_P(HudsonTotals), _P(_d), _P(_u64)]),
    "fmh_pop_totals_pack": (_i, [_P(PopTotals), _i, _P(_d), _P(_u64)]),
    "fmh_pop_totals_unpack": (_i, [_P(PopTotals), _i, _P(_d), _P(_u64)]),
    "fmh_wc_totals_pack": (_i, [_P(WcTotals), _i, _P(_d), _P(_u64)]),
    "fmh_wc_totals_unpack": (_i, [_P(WcTotals), _i, _P(_d), _P(_u64)]),
    "fmh_comm_get_unique_id": (_i, [_vp]),
    "fmh_comm_init_rank": (_i, [_vp, _i, _i, _i, _P(_vp)]),
    "fmh_comm_init_all": (_i, [_P(_i), _i, _P(_vp)]),
    "fmh_comm_init_local": (_i, [_i, _P(_vp)]),
    "fmh_comm_destroy": (_i, [_vp]),
    "fmh_comm_info": (_i, [_vp, _P(_i), _P(_i), _P(_i), _P(_i)]),
    "fmh_comm_describe": (_i, [_vp, C.c_char_p, _sz]),
    "fmh_comm_abort": (_i, [_vp]),
    "fmh_allreduce_totals": (_i, [_vp, _P(_d), _sz, _P(_u64), _sz]),
    "fmh_allreduce_totals_begin": (_i, [_vp, _P(_d), _sz, _P(_u64), _sz]),
    "fmh_allreduce_totals_end": (_i, [_vp, _P(_d), _P(_u64)]),
    "fmh_hudson_sweep_sharded_begin": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _P(HudsonSites), _vp]),
    "fmh_hudson_sweep_sharded_end": (_i, [_vp, _P(HudsonTotals)]),
    "fmh_hudson_sweep_sharded": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _P(HudsonSites), _P(HudsonTotals), _vp]),
    "fmh_wc_sweep_sharded_begin": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "fmh_wc_sweep_sharded_end": (_i, [_vp, _P(WcTotals)]),
    "fmh_wc_sweep_sharded": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _P(WcTotals), _vp]),
    "fmh_population_summaries_sharded_begin": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _vp, _vp, _vp]),
    "fmh_population_summaries_sharded_end": (_i, [_vp, _P(PopTotals)]),
    "fmh_population_summaries_sharded": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _vp, _vp, _P(PopTotals), _vp]),
    "fmh_pair_region_sweep_sharded_begin": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _i, _P(PairDiversitySites), _P(HudsonSites), _vp]),
    "fmh_pair_region_sweep_sharded": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _i, _P(PairDiversitySites), _P(HudsonSites), _P(HudsonTotals), _vp]),
    "fmh_timing_read_reduce": (_i, [_P(_d), _P(_u64)]),
    "fmh_timing_reset_reduce": (_i, []),
    "fmh_timing_enable": (_i, [_i]),
    "fmh_timing_reset": (_i, []),
    "fmh_timing_read": (_i, [_P(_d), _P(_u64)]),
    "fmh_timing_read_minmax": (_i, [_P(_d), _P(_d)]),
}

_lib: Optional[C.CDLL] = None


def _share_hip_runtime_with_torch() -> None:
    """One process must hold ONE HIP runtime.  A PyTorch-ROCm wheel bundles its own libamdhip64.so; if
    libferromic_hip.so pulled in /opt/rocm's copy first, a later `import torch` would load a second
    runtime and see no GPUs.  When a torch wheel with a bundled runtime is installed (found without
    importing torch), map that copy first so both sides bind to it."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch already loaded its runtime; the dynamic linker reuses it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for base in spec.submodule_search_locations:
        cand = os.path.join(base, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load() -> C.CDLL:
    """Load libferromic_hip.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ferromic_amd/csrc`). ferromic_amd has no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    if status == FMH_OK:
        return
    msg = load().fmh_last_error().decode("utf-8", "replace")
    if status == FMH_ERR_NO_DEVICE:
        raise NoDeviceError(status, msg)
    raise FerromicHipError(status, msg)


def set_option(key: str, value=None) -> None:
    """fmh_set_option: a per-process switch of the library (keys = the FMH_* names of include/ferromic_hip.h); None = default.
    The library reads the FMH_* environment once, at first use - later changes go through here, not through os.environ."""
    check(load().fmh_set_option(key.encode(), None if value is None else str(value).encode()))


def get_option(key: str) -> int:
    v = C.c_longlong()
    check(load().fmh_get_option(key.encode(), C.byref(v)))
    return int(v.value)


class options:
    """`with options(FMH_LAYOUT="bytes", FMH_DEFER_TILES=1): ...` - set for the block, previous values restored after it."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def device_count() -> int:
    n = C.c_int(0)
    check(load().fmh_device_count(C.byref(n)))
    return n.value
