"""Region sharding across the GPUs of one node.

Sites are independent units: rank r owns one contiguous genomic slab and sweeps only that slab;
per-site tracks never leave the owning GPU.  The only exchange step is the sum of the regional
accumulators (20 scalars for a Hudson pair), done with one all-reduce over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm) — latency-bound, a few microseconds of payload.
Integer totals are exact; f64 totals are summed rank-by-rank by the collective, inside the
1e-9 contract.
"""

from __future__ import annotations

import ctypes as C
from typing import Tuple

from . import _abi


def slab_for_rank(total_sites: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous slab: r*S/G .. (r+1)*S/G."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return total_sites * rank // world, total_sites * (rank + 1) // world


_EXACT_IN_F64 = 1 << 53


def _pack(totals: _abi.HudsonTotals):
    """The 10 f64 + 10 u64 accumulators of fmh_hudson_totals_pack as ONE f64 vector: the counts (sites, at most a few
    10^9 per node) are far below 2^53, so they and their sums are exact in f64 and one collective carries everything."""
    lib = _abi.load()
    f64 = (C.c_double * _abi.HUDSON_PACK_F64)()
    u64 = (C.c_uint64 * _abi.HUDSON_PACK_U64)()
    _abi.check(lib.fmh_hudson_totals_pack(C.byref(totals), f64, u64))
    ints = [int(x) for x in u64]
    if any(x >= _EXACT_IN_F64 for x in ints):
        raise OverflowError("a count accumulator reached 2^53: reduce the integer totals separately")
    return list(f64) + [float(x) for x in ints]


def _unpack(values) -> _abi.HudsonTotals:
    lib = _abi.load()
    f2 = (C.c_double * _abi.HUDSON_PACK_F64)(*values[:_abi.HUDSON_PACK_F64])
    u2 = (C.c_uint64 * _abi.HUDSON_PACK_U64)(*[int(x) for x in values[_abi.HUDSON_PACK_F64:]])
    out = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_totals_unpack(C.byref(out), f2, u2))
    return out


def allreduce_hudson_totals(totals: _abi.HudsonTotals, dist, device) -> _abi.HudsonTotals:
    """Sum fmh_hudson_totals over all ranks with ONE collective.  `dist` is torch.distributed (nccl on GPUs, gloo in
    CPU tests)."""
    import torch

    t = torch.tensor(_pack(totals), dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return _unpack(t.cpu().tolist())


class HudsonTotalsPipeline:
    """One step late: the all-reduce of step k runs (on the collective's own stream) while the sweep of step k + 1
    streams the matrix, and is collected when step k + 1 hands in its totals.  `flush()` waits for the last one."""

    def __init__(self, dist, device):
        self.dist, self.device, self.pending, self.latest = dist, device, None, None

    def submit(self, totals: _abi.HudsonTotals) -> None:
        import torch

        t = torch.tensor(_pack(totals), dtype=torch.float64, device=self.device)
        work = self.dist.all_reduce(t, async_op=True)
        previous, self.pending = self.pending, (work, t)
        if previous is not None:
            self._collect(previous)

    def _collect(self, item) -> None:
        work, t = item
        work.wait()
        self.latest = _unpack(t.cpu().tolist())

    def flush(self) -> _abi.HudsonTotals:
        if self.pending is not None:
            self._collect(self.pending)
            self.pending = None
        return self.latest
