"""Host-side mirror of the reference's Python module ``ferromic`` (PyO3, src/lib.rs:2227-2270).

Same names, keyword names, defaults, attribute names and error behaviour as the reference for the
per-site diversity / FST path; every statistic is computed on the GPU through libferromic_hip.so
(ferromic_amd.device).  The host only coerces inputs, chooses the code path exactly as
src/stats.rs does (summary / dense / sparse), builds column masks and turns device tracks into the
reference's result objects.  PCA entry points are outside this path and raise NotImplementedError.

There is no CPU fallback: without the HIP library or a GPU these functions raise.
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import device as dev

__version__ = "0.1.0"

FST_EPSILON = 1e-12  # stats.rs:26
LEFT, RIGHT = 0, 1
_I64_MAX = (1 << 63) - 1
_I64_MIN = -(1 << 63)


# --------------------------------------------------------------------------------------------------
# input coercion (lib.rs:825-1080, 1301-1367)
# --------------------------------------------------------------------------------------------------


def _extract_u8(obj) -> int:
    if isinstance(obj, bool):
        obj = int(obj)
    if not isinstance(obj, (int, np.integer)):
        raise TypeError("expected an integer allele")
    v = int(obj)
    if v < 0 or v > 255:
        raise OverflowError("out of range integral type conversion attempted")
    return v


def _regular_genotypes(genotypes_obj) -> Optional[np.ndarray]:
    """Fast path of parse_genotypes for the common shape - every sample a same-length sequence of in-range ints,
    no None: one numpy conversion instead of a Python loop per allele.  Anything else returns None and takes the
    literal path (which also raises the literal errors)."""
    if isinstance(genotypes_obj, np.ndarray):
        arr = genotypes_obj
    elif isinstance(genotypes_obj, (list, tuple)) and genotypes_obj and isinstance(genotypes_obj[0], (list, tuple, np.ndarray)):
        try:
            arr = np.array(genotypes_obj)
        except ValueError:  # ragged
            return None
    else:
        return None
    if arr.ndim != 2 or arr.dtype.kind not in "iu" or arr.shape[1] == 0:
        return None
    if arr.size and (int(arr.min()) < 0 or int(arr.max()) > 255):
        return None
    return arr.astype(np.uint8, copy=False)


def _parse_genotypes(genotypes_obj):
    """parse_genotypes, lib.rs:1301-1332: None | int (haploid) | iterable of ints per sample.
    Returns a list of Optional[List[int]], or a uint8 array [samples][ploidy] when every sample is regular."""
    fast = _regular_genotypes(genotypes_obj)
    if fast is not None:
        return fast
    out: List[Optional[List[int]]] = []
    for entry in genotypes_obj:
        if entry is None:
            out.append(None)
            continue
        if isinstance(entry, (int, np.integer)) and not isinstance(entry, bool):
            try:
                out.append([_extract_u8(entry)])
                continue
            except OverflowError:
                pass
        try:
            it = iter(entry)
        except TypeError:
            raise ValueError("genotypes must be sequences of allele integers or None") from None
        out.append([_extract_u8(a) for a in it])
    return out


def _field(obj, names: Sequence[str]):
    """extract_optional_field, lib.rs:1369-1379: item access first, then attribute."""
    for name in names:
        try:
            return obj[name]
        except Exception:
            pass
        if hasattr(obj, name):
            return getattr(obj, name)
    return None


def _mapping_field(mapping: dict, names: Sequence[str]):
    for name in names:
        if name in mapping:
            return mapping[name]
    raise ValueError("mapping missing required field: " + " / ".join(names))


def _parse_variant(obj) -> Tuple[int, List[Optional[List[int]]]]:
    """VariantInput::extract, lib.rs:834-873."""
    if isinstance(obj, tuple):
        if len(obj) != 2:
            raise ValueError("variant tuples must have length 2: (position, genotypes)")
        return int(obj[0]), _parse_genotypes(obj[1])
    if isinstance(obj, dict):
        position = int(_mapping_field(obj, ["position", "pos", "site"]))
        return position, _parse_genotypes(_mapping_field(obj, ["genotypes", "calls"]))
    position = _field(obj, ["position", "pos", "site"])
    if position is None:
        raise ValueError("variant is missing a position")
    genotypes = _field(obj, ["genotypes", "calls"])
    if genotypes is None:
        raise ValueError("variant is missing genotypes")
    return int(position), _parse_genotypes(genotypes)


def _parse_side(obj) -> int:
    """parse_side, lib.rs:1334-1367."""
    if isinstance(obj, (int, np.integer)) and not isinstance(obj, bool):
        if int(obj) == 0:
            return LEFT
        if int(obj) == 1:
            return RIGHT
        raise ValueError("haplotype side must be 0 or 1")
    if isinstance(obj, str):
        lower = obj.lower()
        if lower in ("l", "left", "0"):
            return LEFT
        if lower in ("r", "right", "1"):
            return RIGHT
        raise ValueError("haplotype side must be one of 0, 1, 'L', 'R', 'left', 'right'")
    raise ValueError("haplotype side must be 0/1 or a left/right string")


def _parse_haplotype(obj) -> Tuple[int, int]:
    """HaplotypeInput::extract, lib.rs:887-923."""
    if isinstance(obj, (tuple, list)):
        if len(obj) < 2:
            raise ValueError("haplotypes must contain (sample_index, side)")
        idx = int(obj[0])
        if idx < 0:
            raise OverflowError("can't convert negative int to unsigned")
        return idx, _parse_side(obj[1])
    index_obj = _field(obj, ["sample_index", "sample", "index"])
    if index_obj is None:
        raise ValueError("haplotype missing sample index")
    side_obj = _field(obj, ["side", "haplotype", "haplotype_side"])
    if side_obj is None:
        raise ValueError("haplotype missing side")
    return int(index_obj), _parse_side(side_obj)


def _parse_population_id(obj):
    """PopulationIdInput::extract, lib.rs:928-965 -> ('group', u8) | ('named', str)."""
    if isinstance(obj, dict):
        if "haplotype_group" in obj:
            return ("group", _extract_u8(obj["haplotype_group"]))
        if "named" in obj:
            return ("named", str(obj["named"]))
        raise ValueError("population id dictionaries must provide 'haplotype_group' or 'named'")
    if isinstance(obj, (int, np.integer)) and not isinstance(obj, bool):
        v = int(obj)
        if 0 <= v <= 255:
            return ("group", v)
        if v > 255:
            raise ValueError("haplotype_group ids must be <= 255")
    if isinstance(obj, str):
        return ("named", obj)
    raise ValueError("could not interpret population id; pass an int, string, or mapping")


def _vcf_error(kind: str, msg: str) -> ValueError:
    """vcf_error_to_pyerr, lib.rs:1551: ValueError(f"VCF error: {err:?}")."""
    return ValueError(f'VCF error: {kind}("{msg}")')


# --------------------------------------------------------------------------------------------------
# variant store: the sparse model (process.rs:431-536) kept as arrays
# --------------------------------------------------------------------------------------------------


class _Store:
    """All variants of a population in the reference's SPARSE semantics: entry (site, sample, k) is
    called iff the sample's genotype is Some and has more than k alleles (CompressedGenotypes::get,
    process.rs:479-496: a leading 0xFF byte means None, a later 0xFF truncates)."""

    def __init__(self, positions: np.ndarray, data: Optional[np.ndarray], called: Optional[np.ndarray],
                 num_samples: np.ndarray, shape: Optional[Tuple[int, int, int]] = None, lazy=None):
        self.positions = positions            # int64 [S]
        self._data = data                     # u8 [S, N, P]
        self._called = called                 # bool [S, N, P]
        self._lazy = lazy                     # () -> (data, called): from_numpy defers its full-array passes
        self.shape = tuple(data.shape) if data is not None else shape
        self.num_samples = num_samples        # int64 [S]: genotypes.len() of every variant
        self._device: Dict[int, dev.DeviceMatrix] = {}
        self.twin: Optional["_Dense"] = None  # dense matrix whose device image equals this store's (from_numpy, nothing missing)
        self._root: Optional["_Store"] = None  # contiguous row view of another store: its rows live in the root's device matrix
        self._row0 = 0

    def _materialise(self):
        if self._data is None:
            self._data, self._called = self._lazy()
            self._lazy = None

    @property
    def data(self) -> np.ndarray:
        self._materialise()
        return self._data

    @property
    def called(self) -> np.ndarray:
        self._materialise()
        return self._called

    @property
    def count(self) -> int:
        return int(self.positions.shape[0])

    @property
    def first_sample_count(self) -> int:
        return int(self.num_samples[0]) if self.count else 0

    @classmethod
    def from_python(cls, variants: Iterable) -> "_Store":
        parsed = [_parse_variant(v) for v in variants]
        S = len(parsed)
        N = max((len(g) for _, g in parsed), default=0)
        P = 1
        for _, g in parsed:
            if isinstance(g, np.ndarray):
                P = max(P, g.shape[1])
                continue
            for gt in g:
                if gt is not None:
                    P = max(P, len(gt))
        data = np.zeros((S, max(N, 1), P), dtype=np.uint8)
        called = np.zeros((S, max(N, 1), P), dtype=bool)
        for s, (_, g) in enumerate(parsed):
            if isinstance(g, np.ndarray):  # regular block: every sample has g.shape[1] alleles
                n, k = g.shape
                ok = g != 0xFF
                if not ok.all():  # 0xFF is the missing sentinel: leading -> None, later -> truncation
                    ok = np.logical_and.accumulate(ok, axis=1)
                    g = np.where(ok, g, 0)
                data[s, :n, :k] = g
                called[s, :n, :k] = ok
                continue
            for i, gt in enumerate(g):
                if gt is None:
                    continue
                # CompressedGenotypes::new + get: the stride of THIS variant is its own max ploidy
                for k, allele in enumerate(gt):
                    if allele == 0xFF:
                        break  # 0xFF is the missing sentinel: leading -> None, later -> truncation
                    data[s, i, k] = allele
                    called[s, i, k] = True
        positions = np.array([p for p, _ in parsed], dtype=np.int64).reshape(S)
        num_samples = np.array([len(g) for _, g in parsed], dtype=np.int64).reshape(S)
        return cls(positions, data, called, num_samples)

    @classmethod
    def from_numpy(cls, g: np.ndarray, neg: Optional[np.ndarray], positions: np.ndarray) -> "_Store":
        """The sparse half of convert_numeric_array (lib.rs:1165-1206): a sample with ANY missing
        allele is None as a whole."""
        S, N, P = g.shape

        def build():
            if N == 0:
                return np.zeros((S, 1, max(P, 1)), dtype=np.uint8), np.zeros((S, 1, max(P, 1)), dtype=bool)
            sample_ok = np.ones((S, N), dtype=bool) if neg is None else ~neg.any(axis=2)
            # 0xFF sentinel semantics of CompressedGenotypes (allele 255 is indistinguishable from missing)
            not_ff = g != 0xFF
            prefix_ok = np.logical_and.accumulate(not_ff, axis=2)
            called = prefix_ok & sample_ok[:, :, None]
            return np.where(called, g, 0).astype(np.uint8), called

        # the reference builds this sparse copy eagerly (lib.rs:1165-1206); the statistics only touch it on the
        # paths that leave the dense matrix (ploidy != 2, incompatible variant sets), so it is built on first use
        shape = (S, 1, max(P, 1)) if N == 0 else (S, N, P)
        return cls(np.asarray(positions, dtype=np.int64), None, None, np.full(S, N, dtype=np.int64), shape=shape, lazy=build)

    def subset(self, idx: np.ndarray) -> "_Store":
        """Rows `idx` (ascending, as np.nonzero gives them).  A contiguous run - every region query over sorted
        positions - becomes a VIEW: no host copy, and its sweeps run over a row range of the root's resident
        device matrix instead of uploading the rows again."""
        if idx.size and int(idx[-1]) - int(idx[0]) + 1 == idx.size:
            a, b = int(idx[0]), int(idx[-1]) + 1
            view = _Store(self.positions[a:b], None, None, self.num_samples[a:b], shape=(b - a,) + tuple(self.shape[1:]),
                          lazy=lambda: (self.data[a:b], self.called[a:b]))
            view._root = self._root if self._root is not None else self
            view._row0 = self._row0 + a
            return view
        return _Store(self.positions[idx], self.data[idx], self.called[idx], self.num_samples[idx])

    def device_rows(self, device: int = 0) -> Tuple[dev.DeviceMatrix, int, int]:
        """(device matrix, first row, row count) holding this store's variants."""
        if self._root is not None:
            return self._root.device_matrix(device), self._row0, self.count
        return self.device_matrix(device), 0, self.count

    def device_matrix(self, device: int = 0) -> dev.DeviceMatrix:
        if self.twin is not None:
            return self.twin.device_matrix(device)  # same bytes, same (absent) missing mask: one copy in HBM
        if device not in self._device:
            S, N, P = self.data.shape
            flat_called = self.called.reshape(-1)
            words = None
            if not flat_called.all():
                bits = np.packbits(~flat_called, bitorder="little")
                pad = (-bits.size) % 8
                words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8").copy()
            max_allele = int(self.data.max()) if self.data.size else 0
            self._device[device] = dev.DeviceMatrix.from_host(self.data.reshape(-1), words, S, N, P, max_allele, device)
        return self._device[device]

    def mask_for(self, haplotypes: Sequence[Tuple[int, int]], sample_count: Optional[int]) -> np.ndarray:
        """HapMembership::build (stats.rs:1212-1238) as a column mask; `sample_count` is the
        function-specific bound the reference passes (None = no bound beyond the data)."""
        _, N, P = self.shape
        mask = np.zeros(N * P, dtype=np.uint8)
        limit = N if sample_count is None else min(sample_count, N)
        for sample_idx, side in haplotypes:
            if sample_idx >= limit:
                continue
            if side >= P:
                continue
            mask[sample_idx * P + side] = 1
        return mask


class _Dense:
    """DenseGenotypeMatrix built by from_numpy (lib.rs:1208-1224): per-allele missing bits."""

    def __init__(self, g: np.ndarray, neg: Optional[np.ndarray]):
        self.g = g
        self.neg = neg
        self.variant_count, self.sample_count, self.ploidy = g.shape
        self.max_allele = int(g.max()) if g.size else 0
        self._device: Dict[int, dev.DeviceMatrix] = {}

    def device_matrix(self, device: int = 0) -> dev.DeviceMatrix:
        if device not in self._device:
            words = None
            if self.neg is not None and self.neg.any():
                bits = np.packbits(self.neg.reshape(-1), bitorder="little")
                pad = (-bits.size) % 8
                words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8").copy()
            self._device[device] = dev.DeviceMatrix.from_host(self.g.reshape(-1), words, self.variant_count,
                                                              self.sample_count, self.ploidy, self.max_allele, device)
        return self._device[device]

    def mask_for(self, haplotypes: Sequence[Tuple[int, int]]) -> np.ndarray:
        """DenseMembership::build, stats.rs:1252-1284."""
        mask = np.zeros(self.sample_count * self.ploidy, dtype=np.uint8)
        for sample_idx, side in haplotypes:
            if sample_idx >= self.sample_count:
                continue
            if side == LEFT:
                mask[sample_idx * self.ploidy] = 1
            else:
                if self.ploidy <= 1:
                    continue
                mask[sample_idx * self.ploidy + 1] = 1
        return mask


def _convert_numeric_array(genotypes, positions) -> Tuple[_Store, Optional[_Dense]]:
    """build_variants_from_numpy + convert_numeric_array, lib.rs:1082-1227."""
    if not isinstance(genotypes, np.ndarray) or genotypes.ndim != 3 or genotypes.dtype not in (
            np.dtype(np.uint8), np.dtype(np.int8), np.dtype(np.uint16), np.dtype(np.int16)):
        raise ValueError("genotypes must be a numpy.ndarray with dtype uint8/int8/uint16/int16 and shape "
                         "(variants, samples, ploidy)")
    S = genotypes.shape[0]
    pos = _extract_positions(positions, S)
    neg = None
    if genotypes.dtype.kind == "i":
        neg = genotypes < 0
        if not neg.any():
            neg = None
    if genotypes.dtype.itemsize > 1:
        too_big = genotypes > 255
        if too_big.any():
            raise ValueError("allele values must be <= 255")
    g = np.where(genotypes < 0, 0, genotypes).astype(np.uint8) if genotypes.dtype.kind == "i" else genotypes.astype(np.uint8)
    g = np.ascontiguousarray(g)
    store = _Store.from_numpy(g, neg, pos)
    dense = _Dense(g, neg) if genotypes.shape[2] == 2 else None
    if dense is not None and neg is None and dense.max_allele != 0xFF and genotypes.shape[1] > 0:
        store.twin = dense  # sparse semantics == dense semantics: no None sample, no 0xFF sentinel
    return store, dense


def _extract_positions(positions_obj, expected_len: int) -> np.ndarray:
    """extract_positions, lib.rs:1229-1299."""
    if isinstance(positions_obj, np.ndarray):
        if positions_obj.ndim != 1 or positions_obj.dtype not in (
                np.dtype(np.int64), np.dtype(np.int32), np.dtype(np.uint32), np.dtype(np.uint64)):
            raise ValueError("positions must be a sequence of integers (NumPy array with dtype "
                             "int64/int32/uint32/uint64 or an iterable of ints)")
        if positions_obj.dtype == np.uint64 and positions_obj.size and int(positions_obj.max()) > _I64_MAX:
            raise ValueError("positions must fit into signed 64-bit integers")
        arr = positions_obj.astype(np.int64)
    else:
        try:
            arr = np.array([int(x) for x in positions_obj], dtype=np.int64)
        except Exception:
            raise ValueError("positions must be a sequence of integers (NumPy array with dtype "
                             "int64/int32/uint32/uint64 or an iterable of ints)") from None
    if arr.shape[0] != expected_len:
        raise ValueError(f"positions length {arr.shape[0]} does not match variant dimension {expected_len}")
    return arr


# --------------------------------------------------------------------------------------------------
# result classes (lib.rs:75-545)
# --------------------------------------------------------------------------------------------------


def _optional_float_display(value: Optional[float]) -> str:  # lib.rs:817-823
    if value is None:
        return "None"
    if math.isfinite(value):
        return f"{value:.6f}"
    return "NaN" if math.isnan(value) else ("inf" if value > 0 else "-inf")


def _rust_debug_opt(v) -> str:
    if v is None:
        return "None"
    if isinstance(v, float):
        s = repr(v)
        return f"Some({s})"
    return f"Some({v})"


_set = object.__setattr__  # result objects are immutable for callers (PyO3 `#[pyo3(get)]`); constructors write through this


class _ReadOnly:
    __slots__ = ()

    def __setattr__(self, key, value):
        raise AttributeError(f"attribute '{key}' of '{type(self).__name__}' objects is not writable")

    def __delattr__(self, key):
        raise AttributeError(f"attribute '{key}' of '{type(self).__name__}' objects is not writable")


class FstEstimate(_ReadOnly):
    """ferromic.FstEstimate, lib.rs:76-165."""

    __slots__ = ("state", "value", "sum_a", "sum_b", "sites")

    def __init__(self, state, value, sum_a, sum_b, sites):
        _set(self, "state", state); _set(self, "value", value); _set(self, "sum_a", sum_a); _set(self, "sum_b", sum_b); _set(self, "sites", sites)

    def components(self):
        return (self.value, self.sum_a, self.sum_b, self.sites)

    def __repr__(self):
        if self.value is not None:
            return (f"FstEstimate(state='{self.state}', value={self.value:.6f}, sum_a={_rust_debug_opt(self.sum_a)}, "
                    f"sum_b={_rust_debug_opt(self.sum_b)}, sites={_rust_debug_opt(self.sites)})")
        return (f"FstEstimate(state='{self.state}', value=None, sum_a={_rust_debug_opt(self.sum_a)}, "
                f"sum_b={_rust_debug_opt(self.sum_b)}, sites={_rust_debug_opt(self.sites)})")


class PairwiseDifference(_ReadOnly):
    __slots__ = ("sample_i", "sample_j", "differences", "comparable_sites")

    def __init__(self, sample_i, sample_j, differences, comparable_sites):
        _set(self, "sample_i", sample_i); _set(self, "sample_j", sample_j); _set(self, "differences", differences)
        _set(self, "comparable_sites", comparable_sites)

    def __repr__(self):
        return (f"PairwiseDifference(sample_i={self.sample_i}, sample_j={self.sample_j}, differences={self.differences}, "
                f"comparable_sites={self.comparable_sites} [genomic bases])")


class ChromosomePcaResult:  # lib.rs:195-257 — PCA is outside the hot path
    def __init__(self, *a, **k):
        raise NotImplementedError("PCA is outside the per-site diversity/FST path implemented by ferromic_amd")


class DiversitySite(_ReadOnly):
    __slots__ = ("position", "pi", "watterson_theta")

    def __init__(self, position, pi, watterson_theta):
        _set(self, "position", position); _set(self, "pi", pi); _set(self, "watterson_theta", watterson_theta)

    def __repr__(self):
        return f"DiversitySite(position={self.position}, pi={self.pi:.6f}, watterson_theta={self.watterson_theta:.6f})"


class HudsonDxyResult(_ReadOnly):
    __slots__ = ("d_xy")

    def __init__(self, d_xy):
        _set(self, "d_xy", d_xy)

    def __repr__(self):
        return "HudsonDxyResult(d_xy=None)" if self.d_xy is None else f"HudsonDxyResult(d_xy={self.d_xy:.6f})"


class HudsonFstSite(_ReadOnly):
    __slots__ = ("position", "fst", "d_xy", "pi_pop1", "pi_pop2", "n1_called", "n2_called", "numerator_component",
                 "denominator_component")

    def __init__(self, position, fst, d_xy, pi_pop1, pi_pop2, n1_called, n2_called, num, den):
        _set(self, "position", position); _set(self, "fst", fst); _set(self, "d_xy", d_xy); _set(self, "pi_pop1", pi_pop1)
        _set(self, "pi_pop2", pi_pop2); _set(self, "n1_called", n1_called); _set(self, "n2_called", n2_called)
        _set(self, "numerator_component", num); _set(self, "denominator_component", den)

    def __repr__(self):
        return (f"HudsonFstSite(position={self.position}, fst={_optional_float_display(self.fst)}, "
                f"d_xy={_optional_float_display(self.d_xy)}, pi_pop1={_optional_float_display(self.pi_pop1)}, "
                f"pi_pop2={_optional_float_display(self.pi_pop2)}, n1_called={self.n1_called}, n2_called={self.n2_called})")


class HudsonFstResult(_ReadOnly):
    __slots__ = ("fst", "d_xy", "pi_pop1", "pi_pop2", "pi_xy_avg", "population1_label", "population1_haplotype_group",
                 "population2_label", "population2_haplotype_group")

    def __init__(self, fst, d_xy, pi_pop1, pi_pop2, pi_xy_avg, id1, id2):
        _set(self, "fst", fst); _set(self, "d_xy", d_xy); _set(self, "pi_pop1", pi_pop1); _set(self, "pi_pop2", pi_pop2)
        _set(self, "pi_xy_avg", pi_xy_avg)
        l1, g1 = _population_label(id1)
        l2, g2 = _population_label(id2)
        _set(self, "population1_label", l1); _set(self, "population1_haplotype_group", g1)
        _set(self, "population2_label", l2); _set(self, "population2_haplotype_group", g2)

    def __repr__(self):
        return (f"HudsonFstResult(fst={_optional_float_display(self.fst)}, d_xy={_optional_float_display(self.d_xy)}, "
                f"pi_pop1={_optional_float_display(self.pi_pop1)}, pi_pop2={_optional_float_display(self.pi_pop2)}, "
                f"pi_xy_avg={_optional_float_display(self.pi_xy_avg)}, pop1={self.population1_label or '<unknown>'}, "
                f"pop2={self.population2_label or '<unknown>'})")


class WcFstSite(_ReadOnly):
    __slots__ = ("position", "overall_fst", "pairwise_fst", "variance_components_a", "variance_components_b",
                 "population_sizes", "pairwise_variance_components")

    def __init__(self, position, overall_fst, pairwise_fst, a, b, population_sizes, pairwise_variance_components):
        _set(self, "position", position); _set(self, "overall_fst", overall_fst); _set(self, "pairwise_fst", pairwise_fst)
        _set(self, "variance_components_a", a); _set(self, "variance_components_b", b)
        _set(self, "population_sizes", population_sizes); _set(self, "pairwise_variance_components", pairwise_variance_components)

    def variance_components(self):
        return (self.variance_components_a, self.variance_components_b)

    def __repr__(self):
        return f"WcFstSite(position={self.position}, overall_fst={self.overall_fst!r})"


class WcFstResult(_ReadOnly):
    __slots__ = ("overall_fst", "pairwise_fst", "pairwise_variance_components", "site_fst", "fst_type")

    def __init__(self, overall_fst, pairwise_fst, pairwise_variance_components, site_fst, fst_type):
        _set(self, "overall_fst", overall_fst); _set(self, "pairwise_fst", pairwise_fst)
        _set(self, "pairwise_variance_components", pairwise_variance_components); _set(self, "site_fst", site_fst)
        _set(self, "fst_type", fst_type)

    def __repr__(self):
        return f"WcFstResult(overall_fst={self.overall_fst!r})"


def _population_label(pid):  # lib.rs:1393-1400
    if pid is None:
        return None, None
    kind, val = pid
    if kind == "group":
        return f"haplotype_group_{val}", val
    return val, None


# --------------------------------------------------------------------------------------------------
# Population (lib.rs:547-814)
# --------------------------------------------------------------------------------------------------


def _sat_sub(a: int, b: int) -> int:
    return max(_I64_MIN, min(_I64_MAX, a - b))


class _Summary:
    """DensePopulationSummary scalars (stats.rs:1311-1317) of one population, kept with the counts
    resident on the device side of the sweep that produced them."""

    def __init__(self, totals: dict):
        self.haplotype_capacity = totals["haplotype_capacity"]
        self.segregating_sites = totals["segregating_sites"]
        self.uncallable_sites = totals["uncallable_sites"]
        self.pi_sum = totals["pi_sum"]


class Population:
    """ferromic.Population, lib.rs:547-728."""

    def __init__(self, id, variants, haplotypes, sequence_length, sample_names=None):
        if sequence_length <= 0:
            raise ValueError("sequence_length must be a positive integer")
        self._init(_parse_population_id(id), _Store.from_python(variants), [_parse_haplotype(h) for h in haplotypes],
                   list(sample_names) if sample_names is not None else [], int(sequence_length), None)

    def _init(self, pid, store, haplotypes, sample_names, sequence_length, dense):
        self._id = pid
        self._store: _Store = store
        self._haplotypes: List[Tuple[int, int]] = haplotypes
        self._sample_names: List[str] = sample_names
        self._sequence_length = sequence_length
        self._dense: Optional[_Dense] = dense
        self._summary: Optional[_Summary] = None  # OnceLock, lib.rs:738

    @classmethod
    def _make(cls, pid, store, haplotypes, sample_names, sequence_length, dense) -> "Population":
        self = cls.__new__(cls)
        self._init(pid, store, haplotypes, sample_names, sequence_length, dense)
        return self

    @staticmethod
    def from_numpy(id, genotypes, positions, haplotypes, sequence_length, sample_names=None):
        if sequence_length <= 0:
            raise ValueError("sequence_length must be a positive integer")
        pid = _parse_population_id(id)
        haps = [_parse_haplotype(h) for h in haplotypes]
        store, dense = _convert_numeric_array(genotypes, positions)
        return Population._make(pid, store, haps, list(sample_names) if sample_names is not None else [],
                                int(sequence_length), dense)

    def with_haplotypes(self, id, haplotypes):
        # clone_with_haplotypes, lib.rs:761-775: variants and dense matrix are shared, summary is not
        return Population._make(_parse_population_id(id), self._store, [_parse_haplotype(h) for h in haplotypes],
                                self._sample_names, self._sequence_length, self._dense)

    # ---- as_population_context, lib.rs:777-799 ----
    def _dense_summary(self) -> Optional[_Summary]:
        if self._dense is None or self._dense.max_allele > 1:
            return None
        if self._summary is None:
            dm = self._dense.device_matrix()
            g = dev.Groups(dm, self._dense.mask_for(self._haplotypes)[None, :])
            res = dev.population_summaries(dm, g, dev.FORMULA_SUMMARY, want_sites=False)
            self._summary = _Summary(res.totals[0])
        return self._summary

    def segregating_sites(self) -> int:
        return _count_segregating_sites_for_population(self)

    def nucleotide_diversity(self) -> float:
        return _calculate_pi_for_population(self)

    @property
    def id(self):
        return self._id[1]

    @property
    def haplotype_group(self):
        return self._id[1] if self._id[0] == "group" else None

    @property
    def label(self):
        return self._id[1] if self._id[0] == "named" else None

    @property
    def sequence_length(self):
        return self._sequence_length

    @property
    def variant_count(self):
        return self._store.count

    @property
    def sample_names(self):
        return list(self._sample_names)

    @property
    def haplotypes(self):
        return [(idx, side) for idx, side in self._haplotypes]

    def __repr__(self):
        label = f"haplotype_group {self._id[1]}" if self._id[0] == "group" else f"named '{self._id[1]}'"
        return (f"Population({label}, haplotypes={len(self._haplotypes)}, variants={self._store.count}, "
                f"sequence_length={self._sequence_length})")


def _coerce_population(obj) -> Population:
    """PopulationInput::extract, lib.rs:978-1080."""
    if isinstance(obj, Population):
        return obj
    if isinstance(obj, dict):
        pid = _parse_population_id(_mapping_field(obj, ["id", "population_id", "name"]))
        if "variants" not in obj:
            raise ValueError("population requires 'variants'")
        if "haplotypes" not in obj:
            raise ValueError("population requires 'haplotypes'")
        L = int(_mapping_field(obj, ["sequence_length", "length", "L"]))
        if L <= 0:
            raise ValueError("sequence_length must be a positive integer")
        names = list(obj["sample_names"]) if "sample_names" in obj else []
        return Population._make(pid, _Store.from_python(obj["variants"]), [_parse_haplotype(h) for h in obj["haplotypes"]],
                                names, L, None)
    pid = _field(obj, ["id", "population_id", "name"])
    if pid is None:
        raise ValueError("population-like object missing 'id'")
    variants = _field(obj, ["variants"])
    if variants is None:
        raise ValueError("population-like object missing 'variants'")
    haplotypes = _field(obj, ["haplotypes"])
    if haplotypes is None:
        raise ValueError("population-like object missing 'haplotypes'")
    L = _field(obj, ["sequence_length", "length", "L"])
    if L is None:
        raise ValueError("population-like object missing 'sequence_length'")
    if int(L) <= 0:
        raise ValueError("sequence_length must be a positive integer")
    names = _field(obj, ["sample_names", "samples"])
    return Population._make(_parse_population_id(pid), _Store.from_python(variants), [_parse_haplotype(h) for h in haplotypes],
                            list(names) if names is not None else [], int(L), None)


# --------------------------------------------------------------------------------------------------
# statistics: path selection mirrors src/stats.rs, arithmetic runs on the GPU
# --------------------------------------------------------------------------------------------------


def _sparse_pop_sweep(store: _Store, masks: np.ndarray, rows: Optional[np.ndarray] = None):
    """Population-summary sweep over the sparse model (FORMULA_SPARSE)."""
    if rows is not None:
        store = store.subset(rows)
    if store.count == 0:
        return [dict(haplotype_capacity=int(m.sum()), segregating_sites=0, uncallable_sites=0, pi_sum=0.0) for m in masks]
    dm, r0, rc = store.device_rows()
    g = dev.Groups(dm, masks)
    return dev.population_summaries(dm, g, dev.FORMULA_SPARSE, r0, rc, want_sites=False).totals


def _count_segregating_sites_for_population(pop: Population) -> int:
    """count_segregating_sites_for_population, stats.rs:3831-3851."""
    summary = pop._dense_summary()
    if summary is not None:
        return int(summary.segregating_sites)
    d = pop._dense
    if d is not None and d.ploidy == 2:
        mask = d.mask_for(pop._haplotypes)
        if int(mask.sum()) <= 1:
            return 0
        if d.variant_count == 0:
            return 0
        dm = d.device_matrix()
        res = dev.population_summaries(dm, dev.Groups(dm, mask[None, :]), dev.FORMULA_DENSE, want_sites=False)
        return int(res.totals[0]["segregating_sites"])
    # count_segregating_sites_for_haplotypes (3858-3889): no sample-count bound, no dedup needed
    mask = pop._store.mask_for(pop._haplotypes, None)
    return int(_sparse_pop_sweep(pop._store, mask[None, :])[0]["segregating_sites"])


def _pi_guards(n_haplotypes: int, seq_length: int) -> Optional[float]:
    if n_haplotypes <= 1:
        return math.nan
    if seq_length < 0:
        return 0.0
    if seq_length == 0:
        return math.inf
    return None


def _calculate_pi_sparse(store: _Store, haplotypes, seq_length: int) -> float:
    """calculate_pi, stats.rs:4317-4432."""
    early = _pi_guards(len(haplotypes), seq_length)
    if early is not None:
        return early
    hap_sample_count = max((s + 1 for s, _ in haplotypes), default=0)
    sample_count = max(store.first_sample_count, hap_sample_count)
    mask = store.mask_for(haplotypes, sample_count)
    # HapMembership.total counts members inside sample_count even beyond the stored columns
    total = len({(s, side) for s, side in haplotypes if s < sample_count})
    if total <= 1:
        return math.nan
    t = _sparse_pop_sweep(store, mask[None, :])[0]
    effective = _sat_sub(seq_length, t["uncallable_sites"])
    if effective == 0:
        return math.nan
    return t["pi_sum"] / float(effective)


def _calculate_pi_for_population(pop: Population) -> float:
    """calculate_pi_for_population, stats.rs:4599-4614."""
    L = pop._sequence_length
    summary = pop._dense_summary()
    if summary is not None:  # calculate_pi_from_summary, stats.rs:1476-1542
        return _pi_from_summary(summary, L, None)
    d = pop._dense
    if d is not None and d.ploidy == 2:  # calculate_pi_dense, stats.rs:4534-4597
        mask = d.mask_for(pop._haplotypes)
        early = _pi_guards(int(mask.sum()), L)
        if early is not None:
            return early
        if d.variant_count == 0:
            return 0.0 / float(L)
        dm = d.device_matrix()
        t = dev.population_summaries(dm, dev.Groups(dm, mask[None, :]), dev.FORMULA_DENSE, want_sites=False).totals[0]
        effective = _sat_sub(L, t["uncallable_sites"])
        if effective == 0:
            return math.nan
        return t["pi_sum"] / float(effective)
    return _calculate_pi_sparse(pop._store, pop._haplotypes, L)


def _pi_from_summary(summary: _Summary, seq_length: int, precomputed: Optional[float]) -> float:
    early = _pi_guards(summary.haplotype_capacity, seq_length)
    if early is not None:
        return early
    effective = _sat_sub(seq_length, summary.uncallable_sites)
    if effective == 0:
        return math.nan
    sum_pi = summary.pi_sum if precomputed is None else precomputed
    return sum_pi / float(effective)


def _variants_compatible(a: _Store, b: _Store) -> bool:  # stats.rs:3399-3401
    return a.count == b.count and bool(np.array_equal(a.positions, b.positions))


def _nan_to_none(x: float) -> Optional[float]:
    return None if x != x else float(x)


def _joint_sparse_matrix(p1: Population, p2: Population, rows: Optional[np.ndarray]):
    """Both populations' membership over ONE device matrix.  The reference reads pop1's genotypes
    through pop1.variants and pop2's through pop2.variants (stats.rs:2974-2975 receives one
    variant because both slices are the same object in every supported call; when they are
    different objects with equal positions, pop1's variants are used for both, 3050-3055)."""
    store = p1._store if rows is None else p1._store.subset(rows)
    m1 = store.mask_for(p1._haplotypes, len(p1._sample_names))
    m2 = store.mask_for(p2._haplotypes, len(p2._sample_names))
    return store, np.stack([m1, m2])


def _hudson_sites_sparse(p1: Population, p2: Population, rows: Optional[np.ndarray]):
    store, masks = _joint_sparse_matrix(p1, p2, rows)
    if store.count == 0:
        return store, None
    dm, r0, rc = store.device_rows()
    return store, dev.hudson_sweep(dm, dev.Groups(dm, masks), dev.FORMULA_SPARSE, r0, rc)


def _opt_list(a: np.ndarray) -> list:
    """f64 track -> Python floats with NaN -> None (Option<f64>)."""
    out = a.tolist()
    nan = np.flatnonzero(a != a)
    for i in nan.tolist():
        out[i] = None
    return out


def _sites_to_py(store: _Store, res) -> List[HudsonFstSite]:
    if res is None:
        return []
    s = res.sites
    n = store.count
    return list(map(HudsonFstSite, (store.positions[:n] + 1).tolist(), _opt_list(s["fst"][:n]), _opt_list(s["dxy"][:n]),
                    _opt_list(s["pi1"][:n]), _opt_list(s["pi2"][:n]), s["called"][0][:n].tolist(), s["called"][1][:n].tolist(),
                    _opt_list(s["num"][:n]), _opt_list(s["den"][:n])))


def _calculate_d_xy_hudson(p1: Population, p2: Population) -> Optional[float]:
    """calculate_d_xy_hudson, stats.rs:2403-2524."""
    if p1._sequence_length <= 0:
        raise _vcf_error("InvalidRegion", "Sequence length must be positive for Dxy calculation")
    if p1._sequence_length != p2._sequence_length:
        raise _vcf_error("Parse", "Sequence length mismatch in Dxy calculation")
    if not _variants_compatible(p1._store, p2._store):
        raise _vcf_error("Parse", "Variant slices differ in positions/length for Dxy calculation")
    if not p1._haplotypes or not p2._haplotypes:
        return None
    L = p1._sequence_length
    s1, s2 = p1._dense_summary(), p2._dense_summary()
    if s1 is not None and s2 is not None:  # dxy_from_summaries, 1637-1662
        totals = _summary_pair_totals(p1, p2)
        effective = _sat_sub(L, totals["dxy_uncallable_sites"])
        return totals["dxy_sum_all"] / float(effective) if effective > 0 else None
    if p1._dense is not None and p1._dense is p2._dense and p1._dense.ploidy == 2:  # calculate_dxy_dense, 2526-2611
        d = p1._dense
        ma, mb = d.mask_for(p1._haplotypes), d.mask_for(p2._haplotypes)
        if int(ma.sum()) == 0 or int(mb.sum()) == 0:
            return None
        if d.variant_count == 0:
            return 0.0 / float(L)
        dm = d.device_matrix()
        t = dev.hudson_sweep(dm, dev.Groups(dm, np.stack([ma, mb])), dev.FORMULA_DENSE, want_sites=False).totals
        effective = _sat_sub(L, t["site_dxy_skipped"])
        return t["site_dxy_sum"] / float(effective) if effective > 0 else None
    _, res = _hudson_sites_sparse(p1, p2, None)  # sparse fold, 2476-2496
    sum_dxy, skipped = (0.0, 0) if res is None else (res.totals["site_dxy_sum"], res.totals["site_dxy_skipped"])
    effective = _sat_sub(L, skipped)
    return sum_dxy / float(effective) if effective > 0 else None


def _summary_pair_totals(p1: Population, p2: Population) -> dict:
    """aggregate_hudson_components_from_summaries (stats.rs:1554-1623) in one fused sweep.  Both
    summaries come from dense matrices; when they are the same matrix the sweep reads it once,
    otherwise the first min(S1, S2) sites of the two matrices are laid side by side."""
    d1, d2 = p1._dense, p2._dense
    if d1 is d2:
        dm = d1.device_matrix()
        masks = np.stack([d1.mask_for(p1._haplotypes), d2.mask_for(p2._haplotypes)])
    else:
        S = min(d1.variant_count, d2.variant_count)
        g = np.concatenate([d1.g[:S].reshape(S, -1), d2.g[:S].reshape(S, -1)], axis=1)
        neg = None
        if d1.neg is not None or d2.neg is not None:
            n1 = d1.neg[:S].reshape(S, -1) if d1.neg is not None else np.zeros((S, d1.sample_count * 2), bool)
            n2 = d2.neg[:S].reshape(S, -1) if d2.neg is not None else np.zeros((S, d2.sample_count * 2), bool)
            neg = np.concatenate([n1, n2], axis=1).reshape(S, -1, 2)
        joint = _Dense(np.ascontiguousarray(g).reshape(S, -1, 2), neg)
        dm = joint.device_matrix()
        m1 = np.concatenate([d1.mask_for(p1._haplotypes), np.zeros(d2.sample_count * 2, np.uint8)])
        m2 = np.concatenate([np.zeros(d1.sample_count * 2, np.uint8), d2.mask_for(p2._haplotypes)])
        masks = np.stack([m1, m2])
    if dm.variants == 0:
        return dict(numerator_sum=0.0, denominator_sum=0.0, pi1_sum=0.0, pi2_sum=0.0, dxy_sum_all=0.0, dxy_uncallable_sites=0)
    return dev.hudson_sweep(dm, dev.Groups(dm, masks), dev.FORMULA_SUMMARY, want_sites=False).totals


def _hudson_core(p1: Population, p2: Population, region: Optional[Tuple[int, int]]):
    """calculate_hudson_fst_for_pair_core, stats.rs:3435-3599."""
    if p1._sequence_length <= 0:
        raise _vcf_error("InvalidRegion", "Sequence length must be positive for Hudson FST calculation.")
    if p1._sequence_length != p2._sequence_length:
        raise _vcf_error("Parse", "Sequence length mismatch between population contexts for Hudson FST calculation.")
    if not _variants_compatible(p1._store, p2._store):
        raise _vcf_error("Parse", "Variant slices differ in positions/length.")
    L = p1._sequence_length
    s1, s2 = p1._dense_summary(), p2._dense_summary()
    summary_totals = None
    dense_shared = p1._dense if (p1._dense is not None and p1._dense is p2._dense and p1._dense.ploidy == 2) else None
    sites: List[HudsonFstSite] = []
    if region is not None:
        rows = _region_rows(p1._store, region)
        store, res = _hudson_sites_sparse(p1, p2, rows)
        sites = _sites_to_py(store, res)
        num_sum, den_sum = (0.0, 0.0) if res is None else (res.totals["site_num_sum"], res.totals["site_den_sum"])
    elif s1 is not None and s2 is not None:
        summary_totals = _summary_pair_totals(p1, p2)
        num_sum, den_sum = summary_totals["numerator_sum"], summary_totals["denominator_sum"]
    elif dense_shared is not None:
        if p1._store.count == 0:
            num_sum, den_sum = 0.0, 0.0
        else:
            dm = dense_shared.device_matrix()
            masks = np.stack([dense_shared.mask_for(p1._haplotypes), dense_shared.mask_for(p2._haplotypes)])
            res = dev.hudson_sweep(dm, dev.Groups(dm, masks), dev.FORMULA_DENSE)
            num_sum, den_sum = res.totals["site_num_sum"], res.totals["site_den_sum"]
            sites = _sites_to_py(p1._store, res)
    elif p1._store.count == 0:
        num_sum, den_sum = 0.0, 0.0
    else:
        store, res = _hudson_sites_sparse(p1, p2, None)
        sites = _sites_to_py(store, res)
        num_sum, den_sum = res.totals["site_num_sum"], res.totals["site_den_sum"]
    fst = num_sum / den_sum if den_sum > FST_EPSILON else None
    if summary_totals is not None:
        pi1_raw = _pi_from_summary(s1, L, summary_totals["pi1_sum"])
        pi2_raw = _pi_from_summary(s2, L, summary_totals["pi2_sum"])
        if not p1._haplotypes or not p2._haplotypes:
            dxy = None
        else:
            effective = _sat_sub(L, summary_totals["dxy_uncallable_sites"])
            dxy = summary_totals["dxy_sum_all"] / float(effective) if effective > 0 else None
    else:
        pi1_raw = _calculate_pi_for_population(p1)
        pi2_raw = _calculate_pi_for_population(p2)
        dxy = _calculate_d_xy_hudson(p1, p2)
    pi1 = pi1_raw if math.isfinite(pi1_raw) else None
    pi2 = pi2_raw if math.isfinite(pi2_raw) else None
    avg = 0.5 * (pi1 + pi2) if (pi1 is not None and pi2 is not None) else None
    return HudsonFstResult(fst, dxy, pi1, pi2, avg, p1._id, p2._id), sites


def _build_region(region) -> Tuple[int, int]:  # lib.rs:1402-1410
    start, end = int(region[0]), int(region[1])
    if end < start:
        raise ValueError("region end must be greater than or equal to region start")
    return start, end


def _region_rows(store: _Store, region: Tuple[int, int]) -> np.ndarray:
    start, end = region
    return np.nonzero((store.positions >= start) & (store.positions <= end))[0]


def _region_len(region: Tuple[int, int]) -> int:  # QueryRegion::len, process.rs:573-584
    start, end = region
    if start > end:
        return 0
    a = max(start, 0)
    b = a if end < a else max(end + 1, a)
    return b - a if b > a else 0


# ---- module-level functions (lib.rs:1555-1778, 2190-2225) ----


def segregating_sites(variants) -> int:
    """count_segregating_sites, stats.rs:3808-3829: every called allele of every sample."""
    store = _Store.from_python(variants)
    if store.count == 0:
        return 0
    mask = np.ones((1, store.shape[1] * store.shape[2]), dtype=np.uint8)
    return int(_sparse_pop_sweep(store, mask)[0]["segregating_sites"])


def nucleotide_diversity(variants, haplotypes, sequence_length) -> float:
    if sequence_length <= 0:
        raise ValueError("sequence_length must be a positive integer")
    return _calculate_pi_sparse(_Store.from_python(variants), [_parse_haplotype(h) for h in haplotypes], int(sequence_length))


def _harmonic(n: int) -> float:  # stats.rs:4234-4240
    s = 0.0
    for k in range(1, n + 1):
        s += 1.0 / float(k)
    return s


def watterson_theta(segregating_sites, sample_count, sequence_length) -> float:
    """watterson_theta_py (lib.rs:1593-1613) -> calculate_watterson_theta (stats.rs:4243-4307).
    Scalar host arithmetic: nothing here touches genotype data."""
    if segregating_sites < 0 or sample_count < 0:
        raise OverflowError("can't convert negative int to unsigned")
    if sample_count <= 1:
        raise ValueError("sample_count must be greater than 1 for Watterson's theta")
    if sequence_length <= 0:
        raise ValueError("sequence_length must be a positive integer")
    h = _harmonic(sample_count - 1)
    if h > 0.0:
        return float(segregating_sites) / h / float(sequence_length)
    return math.nan if segregating_sites == 0 else math.inf


def pairwise_differences(variants, sample_count, sequence_length) -> List[PairwiseDifference]:
    """pairwise_differences_py (lib.rs:1620-1636) -> calculate_pairwise_differences (stats.rs:4106-4231).
    The all-vs-all allele comparison of every sample pair is a Gram product over sites on the GPU
    (fmh_pairwise_differences); ploidy detection and the comparable-site arithmetic are host scalars."""
    if sequence_length <= 0:
        raise ValueError("sequence_length must be a positive integer")
    if sample_count < 0:
        raise OverflowError("can't convert negative int to unsigned")
    store = _Store.from_python(variants)
    n = int(sample_count)
    S = store.count
    N = store.shape[1] if S else 0
    # haplotype_counts: length of the first Some genotype of each sample (stats.rs:4124-4137)
    hap_counts = [0] * n
    if S:
        some = store.called[:, :, 0]                       # [S, N]
        lens = store.called.sum(axis=2)                    # prefix property -> genotype length
        for idx in range(min(n, N)):
            rows = np.nonzero(some[:, idx] & (idx < store.num_samples))[0]
            if rows.size:
                hap_counts[idx] = int(lens[rows[0], idx])
    m = min(n, N)
    diff = both = None
    if S and m >= 2:
        diff, both = dev.pairwise_differences(store.device_matrix(), m)
    out = []
    L = int(sequence_length)
    for i in range(n):
        for j in range(i + 1, n):
            hi, hj = hap_counts[i], hap_counts[j]
            if hi == 0 or hj == 0:
                out.append(PairwiseDifference(i, j, 0, 0))
                continue
            product = hi * hj
            d = int(diff[i, j])
            missing_sites = S - int(both[i, j])
            out.append(PairwiseDifference(i, j, d, max(L * product - missing_sites * product, 0)))
    return out


def per_site_diversity(variants, haplotypes, region=None) -> List[DiversitySite]:
    """per_site_diversity_py (lib.rs:1644-1665) -> calculate_per_site_diversity (stats.rs:4628-4806)."""
    store = _Store.from_python(variants)
    haps = [_parse_haplotype(h) for h in haplotypes]
    if len(haps) < 2:
        raise ValueError("at least two haplotypes are required for diversity calculations")
    if region is not None:
        reg = _build_region(region)
    else:
        if store.count == 0:
            raise ValueError("region must be provided when no variants are supplied")
        reg = (int(store.positions.min()), int(store.positions.max()))
    if _region_len(reg) <= 0:
        return []
    rows = _region_rows(store, reg)
    if rows.size == 0:
        return []
    mask = store.mask_for(haps, store.first_sample_count)  # membership built from the FIRST variant's sample count
    sub = store.subset(rows)
    dm, r0, rc = sub.device_rows()
    res = dev.diversity_sites(dm, dev.Groups(dm, mask[None, :]), r0, rc)
    return list(map(DiversitySite, (sub.positions + 1).tolist(), res.pi.tolist(), res.theta.tolist()))


def hudson_dxy(population1, population2) -> HudsonDxyResult:
    return HudsonDxyResult(_calculate_d_xy_hudson(_coerce_population(population1), _coerce_population(population2)))


def hudson_fst(population1, population2) -> HudsonFstResult:
    return _hudson_core(_coerce_population(population1), _coerce_population(population2), None)[0]


def hudson_fst_sites(population1, population2, region) -> List[HudsonFstSite]:
    """calculate_hudson_fst_per_site, stats.rs:3021-3058: incompatible variants -> []."""
    reg = _build_region(region)
    p1, p2 = _coerce_population(population1), _coerce_population(population2)
    if not _variants_compatible(p1._store, p2._store):
        return []
    store, res = _hudson_sites_sparse(p1, p2, _region_rows(p1._store, reg))
    return _sites_to_py(store, res)


def hudson_fst_with_sites(population1, population2, region):
    reg = _build_region(region)
    return _hudson_core(_coerce_population(population1), _coerce_population(population2), reg)


def _normalize_sample_name_for_lookup(name: str) -> str:  # process.rs:1192-1196
    if name.endswith("_L") or name.endswith("_R"):
        return name[:-2]
    return name


def _map_sample_names_to_indices(sample_names: Sequence[str]) -> Dict[str, int]:  # process.rs:1198-1241
    exact: Dict[str, int] = {}
    alias: Dict[str, Optional[int]] = {}
    for i, name in enumerate(sample_names):
        exact[name] = i
        suffix = name.rsplit("_", 1)[-1]
        if suffix != name:
            if suffix not in alias:
                alias[suffix] = i
            elif alias[suffix] != i:
                alias[suffix] = None
    for a, target in alias.items():
        if target is not None and a not in exact:
            exact[a] = target
    return exact


def _extract_sample_group_map(obj) -> Dict[str, Tuple[int, int]]:  # lib.rs:1485-1505
    if not isinstance(obj, dict):
        raise ValueError("sample_to_group must be a dict mapping sample -> (left, right)")
    out = {}
    for key, value in obj.items():
        try:
            left, right = value[0], value[1]
        except Exception:
            raise ValueError("group tuples must contain two entries") from None
        out[str(key)] = (_extract_u8(left), _extract_u8(right))
    return out


def _classify(a: float, b: float, sites: int) -> FstEstimate:
    """fst_estimate_from_components (stats.rs:1781-1812) / regional ladder (2234-2270)."""
    denominator = a + b

    def div(x, y):
        try:
            return x / y
        except ZeroDivisionError:
            if x != x or x == 0.0:
                return math.nan
            return math.copysign(math.inf, x) * math.copysign(1.0, y)

    if denominator > FST_EPSILON:
        return FstEstimate("calculable", div(a, denominator), a, b, sites)
    if denominator < -FST_EPSILON:
        return FstEstimate("components_yield_indeterminate_ratio", None, a, b, sites)
    if abs(a) > FST_EPSILON:
        return FstEstimate("calculable", div(a, denominator), a, b, sites)
    return FstEstimate("no_inter_population_variance", None, a, b, sites)


def _insufficient(sites: int) -> FstEstimate:
    return FstEstimate("insufficient_data_for_estimation", None, 0.0, 0.0, sites)


def wc_fst(variants, sample_names, sample_to_group, region) -> WcFstResult:
    """wc_fst_py (lib.rs:1750-1770) -> calculate_fst_wc_haplotype_groups (stats.rs:675-806)."""
    sample_names = [str(s) for s in sample_names]
    if not sample_names:
        raise ValueError("sample_names must contain at least one sample")
    store = _Store.from_python(variants)
    group_map = _extract_sample_group_map(sample_to_group)
    reg = _build_region(region)
    # map_samples_to_haplotype_groups (1036-1052) + SubpopulationMembership::from_map (1104-1150)
    idx_of = _map_sample_names_to_indices(sample_names)
    hap_to_group: Dict[Tuple[int, int], str] = {}
    for name, (lg, rg) in group_map.items():
        vcf_idx = idx_of.get(_normalize_sample_name_for_lookup(name))
        if vcf_idx is not None:
            hap_to_group[(vcf_idx, LEFT)] = str(lg)
            hap_to_group[(vcf_idx, RIGHT)] = str(rg)
    labels = sorted(set(hap_to_group.values()))
    G = len(labels)
    keys = [f"{labels[i]}_vs_{labels[j]}" for i in range(G) for j in range(i + 1, G)]
    rows = _region_rows(store, reg)
    if rows.size == 0:
        return WcFstResult(_insufficient(0), {}, {}, [], "haplotype_groups")  # stats.rs:2152-2159
    sub = store.subset(rows)
    _, N, P = sub.shape
    masks = np.zeros((max(G, 1), N * P), dtype=np.uint8)
    label_idx = {lab: i for i, lab in enumerate(labels)}
    for (sample_idx, side), lab in hap_to_group.items():
        if sample_idx >= len(sample_names) or sample_idx >= N or side >= P:
            continue
        masks[label_idx[lab], sample_idx * P + side] = 1
    dm, r0, rc = sub.device_rows()
    if G < 2:
        # fewer than two groups: every site with any called allele is NoInterPopulationVariance (0, 0),
        # sites without any call are InsufficientData (stats.rs:1925-1930, 1987-2003).  One summary
        # sweep over all columns supplies "any call"; the single group's sizes come with it.
        allm = np.ones((1, N * P), dtype=np.uint8)
        gm = np.concatenate([allm, masks[:1]]) if G == 1 else allm
        res = dev.population_summaries(dm, dev.Groups(dm, gm), dev.FORMULA_SPARSE, r0, rc)
        site_objs = []
        for i in range(sub.count):
            if res.called[0][i] == 0:
                site_objs.append(WcFstSite(int(sub.positions[i]) + 1, _insufficient(1), {}, 0.0, 0.0, {}, {}))
            else:
                sizes = {labels[0]: int(res.called[1][i])} if (G == 1 and res.called[1][i] > 0) else {}
                site_objs.append(WcFstSite(int(sub.positions[i]) + 1, _classify(0.0, 0.0, 1), {}, 0.0, 0.0, sizes, {}))
        n_inf = sum(1 for s in site_objs if s.overall_fst.state != "insufficient_data_for_estimation")
        overall = _insufficient(len(site_objs)) if n_inf == 0 else _classify(0.0, 0.0, n_inf)
        return WcFstResult(overall, {}, {}, site_objs, "haplotype_groups")
    # up to 8 groups: the fused sweep; more: counting sweeps in batches of 8 + the counts kernel (same per-site bits)
    w = dev.wc_sweep(dm, dev.Groups(dm, masks), r0, rc) if G <= dev._abi.MAX_GROUPS else dev.wc_sweep_many(dm, masks, r0, rc)
    states = dev.WC_STATES
    site_objs = []
    for i in range(sub.count):
        if w.state[0][i] == 3:
            site_objs.append(WcFstSite(int(sub.positions[i]) + 1, _insufficient(1), {}, 0.0, 0.0, {}, {}))
            continue
        overall = _classify(float(w.a[0][i]), float(w.b[0][i]), 1)
        pw, pwc = {}, {}
        for k, key in enumerate(keys, start=1):
            if w.state[k][i] == 3:
                pw[key] = _insufficient(1)
                pwc[key] = (0.0, 0.0)
            else:
                pw[key] = _classify(float(w.a[k][i]), float(w.b[k][i]), 1)
                pwc[key] = (float(w.a[k][i]), float(w.b[k][i]))
            assert states[w.state[k][i]] == pw[key].state
        sizes = {labels[gidx]: int(w.group_called[gidx][i]) for gidx in range(G) if w.group_called[gidx][i] > 0}
        site_objs.append(WcFstSite(int(sub.positions[i]) + 1, overall, pw, float(w.a[0][i]), float(w.b[0][i]), sizes, pwc))
    # calculate_overall_fst_wc, stats.rs:2145-2374 (sums come from the device sweep)
    if int(w.informative_sites[0]) == 0:
        overall = _insufficient(sub.count)
    else:
        overall = _classify(float(w.sum_a[0]), float(w.sum_b[0]), int(w.informative_sites[0]))
    pairwise, agg = {}, {}
    any_site_with_pairs = any(s.overall_fst.state != "insufficient_data_for_estimation" for s in site_objs)
    if any_site_with_pairs:
        for k, key in enumerate(keys, start=1):
            n_inf = int(w.informative_sites[k])
            if n_inf > 0:
                pairwise[key] = _classify(float(w.sum_a[k]), float(w.sum_b[k]), n_inf)
                agg[key] = (float(w.sum_a[k]), float(w.sum_b[k]))
            else:
                attempted = sum(1 for s in site_objs if key in s.pairwise_variance_components or key in s.pairwise_fst)
                pairwise[key] = _insufficient(attempted)
                agg[key] = (0.0, 0.0)
    return WcFstResult(overall, pairwise, agg, site_objs, "haplotype_groups")


def wc_fst_components(estimate: FstEstimate):
    return estimate.components()


def _extract_interval_list(obj):  # lib.rs:1527-1549
    if obj is None:
        return None
    out = []
    for entry in obj:
        try:
            start, end = int(entry[0]), int(entry[1])
        except Exception:
            raise ValueError("intervals must be (start, end)") from None
        if end < start:
            raise ValueError("interval end must be greater than or equal to start")
        out.append((start, end))
    return out


def adjusted_sequence_length(start, end, allow=None, mask=None) -> int:
    """adjusted_sequence_length_py (lib.rs:2196-2215) -> calculate_adjusted_sequence_length
    (stats.rs:3644-3775).  Interval arithmetic only (host)."""
    if end < start:
        raise ValueError("end must be greater than or equal to start")
    allow = _extract_interval_list(allow)
    mask = _extract_interval_list(mask)
    # region as 0-based half-open (process.rs:193-206)
    a = max(start, 1)
    b = end if end >= a else a
    r0, r1 = a - 1, b
    allowed = []
    if allow is not None:
        for s, e in allow:
            lo, hi = max(r0, s), min(r1, e)
            if lo < hi:
                allowed.append((lo + 1, hi))
    else:
        allowed.append((start, end))
    if mask is not None:
        masks = [(s + 1, e) for s, e in mask]
        out = []
        for a_start, a_end in allowed:  # subtract_regions, stats.rs:3739-3775
            parts = [(a_start, a_end)]
            for m_start, m_end in masks:
                nxt = []
                for s, e in parts:
                    if m_end < s or m_start > e:
                        nxt.append((s, e))
                        continue
                    if m_start > s and m_start - 1 >= s:
                        nxt.append((s, m_start - 1))
                    if m_end < e and m_end + 1 <= e:
                        nxt.append((m_end + 1, e))
                parts = nxt
                if not parts:
                    break
            out.extend(parts)
        allowed = out
    total = 0
    for s, e in allowed:
        lo = max(s, 1)
        hi = e if e >= lo else lo
        total += max(hi - (lo - 1), 0)
    return total


def inversion_allele_frequency(sample_map) -> Optional[float]:
    """calculate_inversion_allele_frequency, stats.rs:3778-3805 (host scalar)."""
    m = _extract_sample_group_map(sample_map)
    ones = total = 0
    for h1, h2 in m.values():
        for allele in (h1, h2):
            if allele in (0, 1):
                total += 1
                ones += allele
    return float(ones) / float(total) if total > 0 else None


def _pca_unavailable(*_a, **_k):
    raise NotImplementedError("PCA (src/pca.rs) is outside the per-site diversity/FST path implemented by ferromic_amd")


chromosome_pca = chromosome_pca_to_file = per_chromosome_pca = global_pca = _pca_unavailable
