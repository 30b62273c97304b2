"""Thin object layer over the C-ABI: device matrices, memberships and sweeps returning numpy.

Everything here runs on the GPU through libferromic_hip.so.  Host work is limited to building
masks and reshaping outputs.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi
from ._abi import FORMULA_DENSE, FORMULA_SPARSE, FORMULA_SUMMARY, WC_STATES  # noqa: F401


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A hipMalloc'd buffer freed with the object."""

    def __init__(self, device: int, nbytes: int):
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _abi.check(_abi.load().fmh_device_alloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def to_numpy(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if count:
            _abi.check(_abi.load().fmh_copy_to_host(self.device, _ptr(out), self.ptr, out.nbytes, None))
        return out

    @classmethod
    def from_numpy(cls, device: int, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(device, max(a.nbytes, 1))
        if a.nbytes:
            _abi.check(_abi.load().fmh_copy_to_device(device, buf.ptr, _ptr(a), a.nbytes, None))
        return buf

    def free(self):
        if getattr(self, "ptr", None):
            _abi.load().fmh_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceMatrix:
    """fmh_matrix handle == DenseGenotypeMatrix (stats.rs:250-331) resident in HBM."""

    def __init__(self, handle: int):
        self._h = handle
        lib = _abi.load()
        v, s, p, pitch, bp = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        hm, ma, dev = C.c_int(), C.c_uint8(), C.c_int()
        _abi.check(lib.fmh_matrix_info(handle, C.byref(v), C.byref(s), C.byref(p), C.byref(pitch), C.byref(bp),
                                       C.byref(hm), C.byref(ma), C.byref(dev)))
        self.variants, self.samples, self.ploidy = v.value, s.value, p.value
        self.pitch, self.bits_pitch = pitch.value, bp.value
        self.has_missing, self.max_allele, self.device = bool(hm.value), ma.value, dev.value
        self.columns = self.samples * self.ploidy

    @classmethod
    def from_host(cls, data: np.ndarray, missing_words: Optional[np.ndarray], variants: int, samples: int,
                  ploidy: int, max_allele: int, device: int = 0) -> "DeviceMatrix":
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        if data.size != variants * samples * ploidy:
            raise ValueError("dense genotype matrix requires variants * samples * ploidy entries")
        if missing_words is not None:
            missing_words = np.ascontiguousarray(missing_words, dtype=np.uint64)
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_matrix_create(_ptr(data), _ptr(missing_words), variants, samples, ploidy,
                                                 int(max_allele), device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_host_planes(cls, planes: Sequence[np.ndarray], called: Optional[np.ndarray], variants: int, samples: int, ploidy: int,
                         max_allele: int, device: int = 0) -> "DeviceMatrix":
        """fmh_matrix_create_packed: host bit planes [variants][pitch] uint8 (column c = bit c & 7 of byte c >> 3), plane k = bit k of
        the allele value; `called` the same shape with 1 bits for called entries, or None."""
        planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        pitch = planes[0].shape[1] if variants else 0
        ptrs = [_ptr(p) for p in planes] + [None] * (3 - len(planes))
        if called is not None:
            called = np.ascontiguousarray(called, dtype=np.uint8)
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_matrix_create_packed(ptrs[0], ptrs[1], ptrs[2], _ptr(called), pitch, variants, samples, ploidy,
                                                        int(max_allele), device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def alloc(cls, variants: int, samples: int, ploidy: int, with_missing: bool, max_allele: int = 1,
              device: int = 0) -> "DeviceMatrix":
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_matrix_alloc(variants, samples, ploidy, int(with_missing), max_allele, device,
                                                C.byref(h)))
        return cls(h.value)

    @classmethod
    def wrap(cls, d_data: int, pitch: int, d_bits: Optional[int], bits_pitch: int, variants: int, samples: int,
             ploidy: int, max_allele: int, device: int = 0) -> "DeviceMatrix":
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_matrix_wrap(d_data, pitch, d_bits, bits_pitch, variants, samples, ploidy,
                                               max_allele, device, C.byref(h)))
        return cls(h.value)

    def generate(self, seed: int, first_global_site: int, thresholds24: np.ndarray, pop_of_column: np.ndarray,
                 missing_threshold24: int = 0) -> None:
        thr = np.ascontiguousarray(thresholds24, dtype=np.uint32)
        poc = np.ascontiguousarray(pop_of_column, dtype=np.uint8)
        if thr.ndim != 2 or thr.shape[1] != self.variants or poc.size != self.columns:
            raise ValueError("thresholds must be [n_pops][variants] and pop_of_column [columns]")
        _abi.check(_abi.load().fmh_matrix_generate(self._h, seed, first_global_site, _ptr(thr), _ptr(poc),
                                                   thr.shape[0], missing_threshold24, None))
        self.max_allele = max(self.max_allele, 1)

    def pack(self, release_bytes: bool = False) -> None:
        """fmh_matrix_pack: build the bit-packed image the sweeps prefer (alleles 0..3); optionally drop the u8 rows."""
        _abi.check(_abi.load().fmh_matrix_pack(self._h, 1 if release_bytes else 0))

    def download(self):
        data = np.empty(self.variants * self.columns, dtype=np.uint8)
        words = None
        if self.has_missing:
            words = np.zeros((self.variants * self.columns + 63) // 64, dtype=np.uint64)
        _abi.check(_abi.load().fmh_matrix_download(self._h, _ptr(data), _ptr(words)))
        return data, words

    def scan_max_allele(self) -> int:
        out = C.c_uint8()
        _abi.check(_abi.load().fmh_matrix_scan_max_allele(self._h, C.byref(out), None))
        return out.value

    def close(self):
        if getattr(self, "_h", None):
            _abi.load().fmh_matrix_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Groups:
    """fmh_groups handle: P column memberships as 0/1 masks."""

    def __init__(self, matrix: DeviceMatrix, masks: np.ndarray):
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        if masks.ndim != 2 or masks.shape[1] != matrix.columns:
            raise ValueError("masks must be [n_groups][samples*ploidy]")
        self.matrix = matrix
        self.n_groups = masks.shape[0]
        self.sizes = [int(x) for x in masks.astype(bool).sum(axis=1)]
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_groups_create(matrix._h, _ptr(masks), self.n_groups, C.byref(h)))
        self._h = h.value

    @staticmethod
    def mask_from_haplotypes(matrix: DeviceMatrix, haplotypes: Sequence) -> np.ndarray:
        """DenseMembership::build (stats.rs:1252-1284) / HapMembership::build (1212-1238) as a column
        mask: duplicates collapse, out-of-range samples are skipped, Right is skipped if ploidy <= 1."""
        mask = np.zeros(matrix.columns, dtype=np.uint8)
        for sample_idx, side in haplotypes:
            if sample_idx >= matrix.samples:
                continue
            if side == 0:
                mask[sample_idx * matrix.ploidy] = 1
            else:
                if matrix.ploidy <= 1:
                    continue
                mask[sample_idx * matrix.ploidy + 1] = 1
        return mask

    @classmethod
    def from_haplotype_lists(cls, matrix: DeviceMatrix, lists: Sequence[Sequence]) -> "Groups":
        return cls(matrix, np.stack([cls.mask_from_haplotypes(matrix, hl) for hl in lists]))

    def close(self):
        if getattr(self, "_h", None):
            _abi.load().fmh_groups_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _pop_totals(t: _abi.PopTotals) -> Dict[str, float]:
    return dict(haplotype_capacity=int(t.haplotype_capacity), segregating_sites=int(t.segregating_sites),
                uncallable_sites=int(t.uncallable_sites), pi_sum=float(t.pi_sum))


@dataclass
class SummaryResult:
    totals: List[Dict[str, float]]
    alt: Optional[np.ndarray]  # [P][rows] u32
    called: Optional[np.ndarray]


def population_summaries(m: DeviceMatrix, g: Groups, formula: int = FORMULA_SUMMARY, row_begin: int = 0,
                         row_count: Optional[int] = None, want_sites: bool = True) -> SummaryResult:
    rows = m.variants - row_begin if row_count is None else row_count
    P = g.n_groups
    d_alt = DeviceBuffer(m.device, 4 * P * rows) if want_sites else None
    d_called = DeviceBuffer(m.device, 4 * P * rows) if want_sites else None
    totals = (_abi.PopTotals * P)()
    _abi.check(_abi.load().fmh_population_summaries(m._h, g._h, row_begin, rows, formula,
                                                    d_alt.ptr if d_alt else None, d_called.ptr if d_called else None,
                                                    totals, None))
    alt = d_alt.to_numpy(np.uint32, P * rows).reshape(P, rows) if want_sites else None
    called = d_called.to_numpy(np.uint32, P * rows).reshape(P, rows) if want_sites else None
    return SummaryResult([_pop_totals(totals[p]) for p in range(P)], alt, called)


@dataclass
class HudsonResult:
    totals: Dict[str, float]
    pop: List[Dict[str, float]]
    sites: Optional[Dict[str, np.ndarray]]


def hudson_totals_dict(t: _abi.HudsonTotals) -> Dict[str, float]:
    return {k: (float(getattr(t, k)) if isinstance(getattr(t, k), float) else int(getattr(t, k)))
            for k, _ in _abi.HudsonTotals._fields_ if k != "pop"}


def hudson_sweep(m: DeviceMatrix, g: Groups, formula: int, row_begin: int = 0, row_count: Optional[int] = None,
                 want_sites: bool = True) -> HudsonResult:
    rows = m.variants - row_begin if row_count is None else row_count
    bufs = {}
    sites = None
    if want_sites:
        for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            bufs[name] = DeviceBuffer(m.device, 8 * rows)
        bufs["alt"] = DeviceBuffer(m.device, 4 * 2 * rows)
        bufs["called"] = DeviceBuffer(m.device, 4 * 2 * rows)
        sites = _abi.HudsonSites(*(bufs[n].ptr for n in ("fst", "dxy", "pi1", "pi2", "num", "den", "alt", "called")))
    totals = _abi.HudsonTotals()
    _abi.check(_abi.load().fmh_hudson_sweep(m._h, g._h, row_begin, rows, formula,
                                            C.byref(sites) if sites is not None else None, C.byref(totals), None))
    out_sites = None
    if want_sites:
        out_sites = {n: bufs[n].to_numpy(np.float64, rows) for n in ("fst", "dxy", "pi1", "pi2", "num", "den")}
        out_sites["alt"] = bufs["alt"].to_numpy(np.uint32, 2 * rows).reshape(2, rows)
        out_sites["called"] = bufs["called"].to_numpy(np.uint32, 2 * rows).reshape(2, rows)
    return HudsonResult(hudson_totals_dict(totals), [_pop_totals(totals.pop[0]), _pop_totals(totals.pop[1])], out_sites)


def hudson_from_counts(device: int, called1: "DeviceBuffer", alt1: "DeviceBuffer", capacity1: int, called2: "DeviceBuffer", alt2: "DeviceBuffer",
                       capacity2: int, rows: int, formula: int, any_missing: bool = False, want_sites: bool = True) -> HudsonResult:
    """fmh_hudson_from_counts: the Hudson pair of two populations from their per-site count tables (u32 device arrays of `rows` entries)."""
    bufs = {}
    sites = None
    if want_sites:
        for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            bufs[name] = DeviceBuffer(device, 8 * rows)
        bufs["alt"] = DeviceBuffer(device, 4 * 2 * rows)
        bufs["called"] = DeviceBuffer(device, 4 * 2 * rows)
        sites = _abi.HudsonSites(*(bufs[n].ptr for n in ("fst", "dxy", "pi1", "pi2", "num", "den", "alt", "called")))
    totals = _abi.HudsonTotals()
    _abi.check(_abi.load().fmh_hudson_from_counts(device, called1.ptr, alt1.ptr, capacity1, called2.ptr, alt2.ptr, capacity2, rows, formula,
                                                  1 if any_missing else 0, C.byref(sites) if sites is not None else None, C.byref(totals), None))
    out_sites = None
    if want_sites:
        out_sites = {n: bufs[n].to_numpy(np.float64, rows) for n in ("fst", "dxy", "pi1", "pi2", "num", "den")}
        out_sites["alt"] = bufs["alt"].to_numpy(np.uint32, 2 * rows).reshape(2, rows)
        out_sites["called"] = bufs["called"].to_numpy(np.uint32, 2 * rows).reshape(2, rows)
    return HudsonResult(hudson_totals_dict(totals), [_pop_totals(totals.pop[0]), _pop_totals(totals.pop[1])], out_sites)


@dataclass
class DiversityResult:
    totals: Dict[str, float]
    pi: np.ndarray
    theta: np.ndarray
    called: np.ndarray
    distinct: np.ndarray


def diversity_sites(m: DeviceMatrix, g: Groups, row_begin: int = 0, row_count: Optional[int] = None) -> DiversityResult:
    rows = m.variants - row_begin if row_count is None else row_count
    d_pi, d_th = DeviceBuffer(m.device, 8 * rows), DeviceBuffer(m.device, 8 * rows)
    d_ca, d_di = DeviceBuffer(m.device, 4 * rows), DeviceBuffer(m.device, 4 * rows)
    totals = _abi.PopTotals()
    _abi.check(_abi.load().fmh_diversity_sites(m._h, g._h, row_begin, rows, d_pi.ptr, d_th.ptr, d_ca.ptr, d_di.ptr,
                                               C.byref(totals), None))
    return DiversityResult(_pop_totals(totals), d_pi.to_numpy(np.float64, rows), d_th.to_numpy(np.float64, rows),
                           d_ca.to_numpy(np.uint32, rows), d_di.to_numpy(np.uint32, rows))


@dataclass
class WcResult:
    sum_a: np.ndarray  # [1+npairs]
    sum_b: np.ndarray
    informative_sites: np.ndarray
    sites_attempted: int
    a: np.ndarray  # [1+npairs][rows]
    b: np.ndarray
    state: np.ndarray
    group_called: np.ndarray  # [P][rows]


def wc_sweep(m: DeviceMatrix, g: Groups, row_begin: int = 0, row_count: Optional[int] = None) -> WcResult:
    rows = m.variants - row_begin if row_count is None else row_count
    P = g.n_groups
    nw = 1 + P * (P - 1) // 2
    d_a, d_b = DeviceBuffer(m.device, 8 * nw * rows), DeviceBuffer(m.device, 8 * nw * rows)
    d_s, d_n = DeviceBuffer(m.device, nw * rows), DeviceBuffer(m.device, 4 * P * rows)
    totals = _abi.WcTotals()
    _abi.check(_abi.load().fmh_wc_sweep(m._h, g._h, row_begin, rows, d_a.ptr, d_b.ptr, d_s.ptr, d_n.ptr,
                                        C.byref(totals), None))
    return WcResult(np.array(totals.sum_a[:nw]), np.array(totals.sum_b[:nw]),
                    np.array(totals.informative_sites[:nw], dtype=np.uint64), int(totals.sites_attempted),
                    d_a.to_numpy(np.float64, nw * rows).reshape(nw, rows),
                    d_b.to_numpy(np.float64, nw * rows).reshape(nw, rows),
                    d_s.to_numpy(np.uint8, nw * rows).reshape(nw, rows),
                    d_n.to_numpy(np.uint32, P * rows).reshape(P, rows))


def wc_sweep_many(m: DeviceMatrix, masks: np.ndarray, row_begin: int = 0, row_count: Optional[int] = None, sites: bool = True) -> WcResult:
    """fmh_wc_sweep_many: W&C for any number of groups (counting in batches of 8, arithmetic from count tables).
    sites=False: no per-site track is asked for - the regional sums come straight from the count tables (a, b, state, group_called = None)."""
    rows = m.variants - row_begin if row_count is None else row_count
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    G = int(masks.shape[0])
    nw = 1 + G * (G - 1) // 2
    if not sites:
        sum_a, sum_b = np.zeros(nw, dtype=np.float64), np.zeros(nw, dtype=np.float64)
        inf = np.zeros(nw, dtype=np.uint64)
        _abi.check(_abi.load().fmh_wc_sweep_many(m._h, _ptr(masks), G, row_begin, rows, None, None, None, None, _ptr(sum_a), _ptr(sum_b), _ptr(inf), None))
        return WcResult(sum_a, sum_b, inf, int(rows), None, None, None, None)
    d_a, d_b = DeviceBuffer(m.device, 8 * nw * rows), DeviceBuffer(m.device, 8 * nw * rows)
    d_s, d_n = DeviceBuffer(m.device, nw * rows), DeviceBuffer(m.device, 4 * G * rows)
    sum_a, sum_b = np.zeros(nw, dtype=np.float64), np.zeros(nw, dtype=np.float64)
    inf = np.zeros(nw, dtype=np.uint64)
    _abi.check(_abi.load().fmh_wc_sweep_many(m._h, _ptr(masks), G, row_begin, rows, d_a.ptr, d_b.ptr, d_s.ptr, d_n.ptr,
                                             _ptr(sum_a), _ptr(sum_b), _ptr(inf), None))
    return WcResult(sum_a, sum_b, inf, int(rows),
                    d_a.to_numpy(np.float64, nw * rows).reshape(nw, rows),
                    d_b.to_numpy(np.float64, nw * rows).reshape(nw, rows),
                    d_s.to_numpy(np.uint8, nw * rows).reshape(nw, rows),
                    d_n.to_numpy(np.uint32, G * rows).reshape(G, rows))


def pairwise_differences(m: DeviceMatrix, n_samples: int):
    """fmh_pairwise_differences -> (diff, both) as [n, n] uint64 arrays (upper triangle filled)."""
    n = int(n_samples)
    d_diff, d_both = DeviceBuffer(m.device, 8 * n * n), DeviceBuffer(m.device, 8 * n * n)
    lib = _abi.load()
    if n:
        _abi.check(lib.fmh_device_zero(m.device, d_diff.ptr, 8 * n * n, None))
        _abi.check(lib.fmh_device_zero(m.device, d_both.ptr, 8 * n * n, None))
    _abi.check(lib.fmh_pairwise_differences(m._h, n, d_diff.ptr, d_both.ptr, None))
    return d_diff.to_numpy(np.uint64, n * n).reshape(n, n), d_both.to_numpy(np.uint64, n * n).reshape(n, n)
