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


def allreduce_hudson_totals(totals: _abi.HudsonTotals, dist, device) -> _abi.HudsonTotals:
    """Sum fmh_hudson_totals over all ranks with ONE collective per dtype (f64 + i64 vectors packed by
    fmh_hudson_totals_pack).  `dist` is torch.distributed (nccl on GPUs, gloo in CPU tests)."""
    import torch

    lib = _abi.load()
    f64 = (C.c_double * _abi.HUDSON_PACK_F64)()
    u64 = (C.c_uint64 * _abi.HUDSON_PACK_U64)()
    _abi.check(lib.fmh_hudson_totals_pack(C.byref(totals), f64, u64))
    tf = torch.tensor(list(f64), dtype=torch.float64, device=device)
    tu = torch.tensor([int(x) for x in u64], dtype=torch.int64, device=device)
    dist.all_reduce(tf)
    dist.all_reduce(tu)
    f2 = (C.c_double * _abi.HUDSON_PACK_F64)(*tf.cpu().tolist())
    u2 = (C.c_uint64 * _abi.HUDSON_PACK_U64)(*tu.cpu().tolist())
    out = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_totals_unpack(C.byref(out), f2, u2))
    return out
