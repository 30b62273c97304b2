"""Region sharding across the GPUs of one node.

Sites are independent units: rank r owns one contiguous genomic slab and sweeps only that slab;
per-site tracks never leave the owning GPU.  The only exchange step is the sum of the regional
accumulators, a few hundred bytes - latency-bound.  Integer totals are exact; f64 totals are summed
in the collective's order, inside the 1e-9 contract.

Two carriers of that sum:
  * `Comm` - the product path: libferromic_hip.so's own communicator (fmh_comm_*), RCCL over xGMI
    issued by the library on its own stream, with the pipelined fmh_hudson_sweep_sharded_begin/_end.
  * `allreduce_*_totals(..., dist, device)` - the packed vectors through torch.distributed: the "any
    other transport" form of the same packing (gloo in the CPU tests and in the one-GPU rehearsal).
"""

from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

from . import _abi


class Comm:
    """fmh_comm handle.  One process per GPU: `from_torch_distributed` lets rank 0 create the RCCL id and ships its
    128 bytes through the process group the launcher set up (any backend; CPU tensors); one process: `single`."""

    def __init__(self, handle: int):
        self._h = handle
        lib = _abi.load()
        w, r, d, t = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _abi.check(lib.fmh_comm_info(handle, C.byref(w), C.byref(r), C.byref(d), C.byref(t)))
        self.world, self.rank, self.device, self.transport = w.value, r.value, d.value, ("rccl", "host", "local")[t.value]

    @classmethod
    def from_unique_id(cls, uid: bytes, world: int, rank: int, device: int) -> "Comm":
        if len(uid) != _abi.COMM_ID_BYTES:
            raise ValueError("an RCCL unique id is 128 bytes")
        buf = (C.c_char * _abi.COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_comm_init_rank(buf, world, rank, device, C.byref(h)))
        return cls(h.value)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * _abi.COMM_ID_BYTES)()
        _abi.check(_abi.load().fmh_comm_get_unique_id(buf))
        return bytes(buf)

    @classmethod
    def single(cls, device: int = 0) -> "Comm":
        """A one-rank communicator (the collective is then an identity, but runs through RCCL all the same)."""
        return cls.from_unique_id(cls.unique_id(), 1, 0, device)

    @classmethod
    def local(cls, device: int = 0) -> "Comm":
        """One rank without any transport (no RCCL): the pipelined begin / end sweeps of a single GPU."""
        h = C.c_void_p()
        _abi.check(_abi.load().fmh_comm_init_local(device, C.byref(h)))
        return cls(h.value)

    init_stuck = False  # a communicator creation that outlived its timeout is still blocked in a helper thread: leave through os._exit

    @classmethod
    def from_torch_distributed(cls, dist, device: int, timeout_s: float = 300.0) -> "Comm":
        """Rank 0 makes the RCCL id, the process group ships it, every rank joins.  ncclCommInitRank blocks until ALL ranks have joined: if
        one of them failed before it got there the others would wait for ever, so the join runs in a helper thread and is given up after
        `timeout_s` (the caller then falls back to another transport and says so)."""
        import threading

        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        out = {}

        def join():
            try:
                out["comm"] = cls.from_unique_id(box[0], world, rank, device)
            except BaseException as exc:  # noqa: BLE001 - re-raised on the caller's thread
                out["error"] = exc

        t = threading.Thread(target=join, daemon=True)
        t.start()
        t.join(timeout_s)
        if t.is_alive():
            Comm.init_stuck = True
            raise TimeoutError(f"ncclCommInitRank(rank {rank} of {world}) did not return within {timeout_s:.0f} s: a peer never joined")
        if "error" in out:
            raise out["error"]
        return out["comm"]

    @classmethod
    def init_all(cls, devices: Sequence[int]) -> List["Comm"]:
        """One process driving several GPUs (one thread each); a device listed twice selects the in-process host rendezvous."""
        n = len(devices)
        arr = (C.c_int * n)(*devices)
        out = (C.c_void_p * n)()
        _abi.check(_abi.load().fmh_comm_init_all(arr, n, out))
        return [cls(out[i]) for i in range(n)]

    def allreduce(self, f64: Sequence[float], u64: Sequence[int]):
        """Element-wise sum over the ranks of one f64 and one u64 vector (fmh_allreduce_totals)."""
        nf, nu = len(f64), len(u64)
        f = (C.c_double * max(nf, 1))(*f64)
        u = (C.c_uint64 * max(nu, 1))(*u64)
        _abi.check(_abi.load().fmh_allreduce_totals(self._h, f, nf, u, nu))
        return list(f)[:nf], [int(x) for x in u][:nu]

    def describe(self) -> dict:
        """fmh_comm_describe parsed: transport, world, rank, device, rccl_library, rccl_version, in_flight."""
        buf = C.create_string_buffer(1024)
        _abi.check(_abi.load().fmh_comm_describe(self._h, buf, len(buf)))
        out = {}
        for item in buf.value.decode("utf-8", "replace").split(" "):
            if "=" in item:
                k, v = item.split("=", 1)
                out[k] = int(v) if v.lstrip("-").isdigit() else v
            elif out:  # a path with blanks: glue it back on the previous value
                last = list(out)[-1]
                out[last] = f"{out[last]} {item}"
        return out

    def abort(self) -> None:
        _abi.check(_abi.load().fmh_comm_abort(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _abi.load().fmh_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


def _sum_vectors(f64, u64, dist, device):
    """One collective for both vectors: the counts ride as f64 (exact below 2^53, refused above)."""
    import torch

    ints = [int(x) for x in u64]
    if any(x >= _EXACT_IN_F64 for x in ints):
        raise OverflowError("a count accumulator reached 2^53: reduce the integer totals separately")
    t = torch.tensor(list(f64) + [float(x) for x in ints], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    v = t.cpu().tolist()
    return v[:len(f64)], [int(x) for x in v[len(f64):]]


def allreduce_wc_totals(totals: _abi.WcTotals, n_groups: int, dist, device) -> _abi.WcTotals:
    """W&C regional sums of every slot (calculate_overall_fst_wc, stats.rs:2145-2374) summed over the ranks."""
    lib = _abi.load()
    slots = 1 + n_groups * (n_groups - 1) // 2
    f = (C.c_double * (2 * slots))()
    u = (C.c_uint64 * (slots + 1))()
    _abi.check(lib.fmh_wc_totals_pack(C.byref(totals), n_groups, f, u))
    fs, us = _sum_vectors(list(f), list(u), dist, device)
    out = _abi.WcTotals()
    _abi.check(lib.fmh_wc_totals_unpack(C.byref(out), n_groups, (C.c_double * (2 * slots))(*fs), (C.c_uint64 * (slots + 1))(*us)))
    return out


def allreduce_pop_totals(totals: Sequence[_abi.PopTotals], dist, device) -> List[_abi.PopTotals]:
    """Per-population summaries (build_dense_population_summary's fold/reduce, stats.rs:1365-1461) summed over the ranks."""
    lib = _abi.load()
    n = len(totals)
    arr = (_abi.PopTotals * n)(*totals)
    f = (C.c_double * n)()
    u = (C.c_uint64 * (3 * n + 1))()
    _abi.check(lib.fmh_pop_totals_pack(arr, n, f, u))
    fs, us = _sum_vectors(list(f), list(u), dist, device)
    out = (_abi.PopTotals * n)()
    _abi.check(lib.fmh_pop_totals_unpack(out, n, (C.c_double * n)(*fs), (C.c_uint64 * (3 * n + 1))(*us)))
    return list(out)
