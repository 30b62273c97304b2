"""ctypes wrapper of oracle/liboracle_dense.so (the C half of the CPU oracle).

TEST INFRASTRUCTURE ONLY — see the header of oracle/dense_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FERROMIC_ORACLE_LIB") or os.path.join(_HERE, "liboracle_dense.so")  # the override: the sanitizer build (oracle/Makefile asan)


class PopTotals(C.Structure):
    _fields_ = [("haplotype_capacity", C.c_uint64), ("segregating_sites", C.c_uint64),
                ("uncallable_sites", C.c_uint64), ("pi_sum", C.c_double)]


class HudsonTotals(C.Structure):
    _fields_ = [("numerator_sum", C.c_double), ("denominator_sum", C.c_double), ("pi1_sum", C.c_double),
                ("pi2_sum", C.c_double), ("dxy_sum_all", C.c_double), ("dxy_uncallable_sites", C.c_uint64),
                ("site_num_sum", C.c_double), ("site_den_sum", C.c_double), ("sites_with_components", C.c_uint64)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", _HERE], check=True)
        lib = C.CDLL(LIB_PATH)
        vp, sz, u64, u32, i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int
        lib.fo_generate.argtypes = [vp, vp, sz, sz, u64, u64, vp, vp, u32, i]
        lib.fo_generate.restype = None
        lib.fo_hudson_sweep_threaded.argtypes = [vp, vp, sz, sz, vp, sz, vp, sz] + [vp] * 10 + [
            C.POINTER(PopTotals), C.POINTER(HudsonTotals), i]
        lib.fo_hudson_sweep_threaded.restype = None
        lib.fo_wc_sites_threaded.argtypes = [vp, vp, sz, sz, vp, i, vp, vp, vp, vp, vp, vp, i]
        lib.fo_wc_sites_threaded.restype = None
        _lib = lib
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def generate(variants: int, columns: int, seed: int, first_site: int, thresholds24: np.ndarray,
             pop_of_column: np.ndarray, missing_threshold24: int = 0, nthreads: int = 1):
    """Host-layout matrix (and missing words) of the counter-based synthetic cohort."""
    thr = np.ascontiguousarray(thresholds24, dtype=np.uint32)
    poc = np.ascontiguousarray(pop_of_column, dtype=np.uint8)
    assert thr.shape[1] == variants and poc.size == columns
    data = np.empty(variants * columns, dtype=np.uint8)
    words = np.zeros((variants * columns + 63) // 64, dtype=np.uint64) if missing_threshold24 else None
    load().fo_generate(_p(data), _p(words), variants, columns, seed, first_site, _p(thr), _p(poc),
                       missing_threshold24, nthreads)
    return data, words


@dataclass
class SweepOut:
    alt: np.ndarray      # [2][S] u32
    called: np.ndarray   # [2][S]
    fst: Optional[np.ndarray]
    dxy: Optional[np.ndarray]
    pi1: Optional[np.ndarray]
    pi2: Optional[np.ndarray]
    num: Optional[np.ndarray]
    den: Optional[np.ndarray]
    pop: list
    totals: dict


def hudson_sweep(data: np.ndarray, missing_words: Optional[np.ndarray], variants: int, stride: int,
                 offsets1: np.ndarray, offsets2: np.ndarray, nthreads: int = 1, want_sites: bool = True) -> SweepOut:
    """fo_hudson_sweep_threaded: summaries x2 + Hudson totals + per-site records (biallelic dense)."""
    o1 = np.ascontiguousarray(offsets1, dtype=np.uint64)
    o2 = np.ascontiguousarray(offsets2, dtype=np.uint64)
    alt = np.empty((2, variants), dtype=np.uint32)
    called = np.empty((2, variants), dtype=np.uint32)
    tracks = [np.empty(variants, dtype=np.float64) if want_sites else None for _ in range(6)]
    pop = (PopTotals * 2)()
    tot = HudsonTotals()
    load().fo_hudson_sweep_threaded(_p(data), _p(missing_words), variants, stride, _p(o1), o1.size, _p(o2), o2.size,
                                    _p(alt[0]), _p(called[0]), _p(alt[1]), _p(called[1]),
                                    *[_p(t) for t in tracks], pop, C.byref(tot), nthreads)
    return SweepOut(alt, called, *tracks,
                    pop=[{k: getattr(pop[i], k) for k, _ in PopTotals._fields_} for i in range(2)],
                    totals={k: getattr(tot, k) for k, _ in HudsonTotals._fields_})


@dataclass
class WcOut:
    a: np.ndarray       # [slots][S]; slot 0 = overall, then pairs (0,1), (0,2), ...
    b: np.ndarray
    state: np.ndarray   # 0 calculable, 1 indeterminate, 2 no variance, 3 insufficient
    sum_a: np.ndarray
    sum_b: np.ndarray
    informative: np.ndarray


def wc_sites(data: np.ndarray, missing_words: Optional[np.ndarray], variants: int, stride: int, group_of_column: np.ndarray,
             n_groups: int, nthreads: int = 1) -> WcOut:
    """fo_wc_sites_threaded: Weir & Cockerham per site + regional sums (stats.rs:1814-2032, 2145-2374) on a dense matrix."""
    goc = np.ascontiguousarray(group_of_column, dtype=np.uint8)
    assert goc.size == stride and 2 <= n_groups <= 16
    slots = 1 + n_groups * (n_groups - 1) // 2
    a = np.empty((slots, variants), dtype=np.float64)
    b = np.empty((slots, variants), dtype=np.float64)
    st = np.empty((slots, variants), dtype=np.uint8)
    sa, sb, inf = np.zeros(slots), np.zeros(slots), np.zeros(slots, dtype=np.uint64)
    load().fo_wc_sites_threaded(_p(data), _p(missing_words), variants, stride, _p(goc), n_groups, _p(a), _p(b), _p(st), _p(sa), _p(sb),
                                _p(inf), nthreads)
    return WcOut(a, b, st, sa, sb, inf)
