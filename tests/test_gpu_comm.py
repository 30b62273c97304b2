"""fmh_comm on the GPU box: the RCCL transport with a one-rank group (all a one-GPU box allows), the in-process host
transport with three "ranks" aliasing the device, and the sharded Hudson sweep against one sweep over the whole matrix."""

import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cohort(S, N, seed, missing=0.0):
    rng = np.random.default_rng(seed)
    data = (rng.random((S, 2 * N)) < rng.beta(0.8, 0.8, size=(S, 1))).astype(np.uint8)
    words = None
    if missing:
        miss = rng.random((S, 2 * N)) < missing
        words = np.packbits(miss.reshape(-1), bitorder="little")
        words = np.frombuffer(np.pad(words, (0, (-words.size) % 8)).tobytes(), dtype="<u8").copy()
    return data, words


def _totals_equal(a, b, rel):
    from ferromic_amd import _abi

    for k, _ in _abi.HudsonTotals._fields_:
        if k == "pop":
            continue
        x, y = getattr(a, k), getattr(b, k)
        if isinstance(x, int):
            assert x == y, (k, x, y)
        else:
            assert x == pytest.approx(y, rel=rel, abs=1e-300), (k, x, y)
    for p in range(2):
        for k in ("haplotype_capacity", "segregating_sites", "uncallable_sites"):
            assert getattr(a.pop[p], k) == getattr(b.pop[p], k), (p, k)
        assert a.pop[p].pi_sum == pytest.approx(b.pop[p].pi_sum, rel=rel)


def test_rccl_one_rank_allreduce_and_sharded_sweep():
    from ferromic_amd import _abi, device, sharding

    lib = _abi.load()
    comm = sharding.Comm.single(0)
    assert (comm.world, comm.rank, comm.transport) == (1, 0, "rccl")
    f, u = comm.allreduce([1.5, -2.25, 1e300], [7, (1 << 63) + 5])
    assert f == [1.5, -2.25, 1e300] and u == [7, (1 << 63) + 5]  # u64 travels as u64: nothing is squeezed through f64
    f, u = comm.allreduce([], [3])
    assert f == [] and u == [3]
    S, N = 70_001, 150
    data, words = _cohort(S, N, 5, missing=0.02)
    dm = device.DeviceMatrix.from_host(data, words, S, N, 2, 1)
    masks = np.zeros((2, 2 * N), np.uint8)
    masks[0, :N], masks[1, N:] = 1, 1
    g = device.Groups(dm, masks)
    plain = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, S, _abi.FORMULA_DENSE, None, C.byref(plain), None))
    got = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep_sharded(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_DENSE, None, C.byref(got), None))
    _totals_equal(got, plain, 0.0)  # same kernel, same grid, one rank: the very same bits
    # pipelined: two windows in flight, collected in order; a third begin without an end is refused
    half = S // 2
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 0, half, _abi.FORMULA_DENSE, None, None))
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, half, S - half, _abi.FORMULA_DENSE, None, None))
    assert lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_DENSE, None, None) == _abi.FMH_ERR_INVALID
    a, b = _abi.HudsonTotals(), _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(a)))
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(b)))
    assert lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(b)) == _abi.FMH_ERR_INVALID
    assert a.sites_with_components + b.sites_with_components == plain.sites_with_components
    assert a.pop[0].segregating_sites + b.pop[0].segregating_sites == plain.pop[0].segregating_sites
    assert a.numerator_sum + b.numerator_sum == pytest.approx(plain.numerator_sum, rel=1e-12)
    # a local communicator (no RCCL at all) runs the same pipelined calls
    loc = sharding.Comm.local(0)
    assert (loc.world, loc.transport) == (1, "local")
    l2 = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep_sharded(loc._h, dm._h, g._h, 0, S, _abi.FORMULA_DENSE, None, C.byref(l2), None))
    _totals_equal(l2, plain, 0.0)
    f2, u2 = loc.allreduce([0.5, 2.0], [9])
    assert f2 == [0.5, 2.0] and u2 == [9]
    loc.close()
    # an empty window is a valid collective participant
    _abi.check(lib.fmh_hudson_sweep_sharded(comm._h, dm._h, g._h, 10, 0, _abi.FORMULA_DENSE, None, C.byref(a), None))
    assert a.sites_with_components == 0 and a.numerator_sum == 0.0 and a.pop[1].haplotype_capacity == N
    comm.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_host_transport_slabs_equal_whole(ranks):
    """`ranks` threads, each with its own slab matrix on the one device (a device listed twice selects the in-process rendezvous):
    Hudson totals through the sharded sweep, W&C and summaries through pack -> fmh_allreduce_totals -> unpack."""
    from ferromic_amd import _abi, device, sharding

    lib = _abi.load()
    S, N, G = 50_003, 90, 3
    data, words = _cohort(S, N, 11 + ranks, missing=0.03)
    masks = np.zeros((G, 2 * N), np.uint8)
    for k in range(G):
        masks[k, 2 * (k * N // G):2 * ((k + 1) * N // G)] = 1
    whole = device.DeviceMatrix.from_host(data, words, S, N, 2, 1)
    g2, g3 = device.Groups(whole, masks[:2]), device.Groups(whole, masks)  # kept alive: the handles die with the objects
    ref_h = _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep(whole._h, g2._h, 0, S, _abi.FORMULA_SPARSE, None, C.byref(ref_h), None))
    ref_w = _abi.WcTotals()
    _abi.check(lib.fmh_wc_sweep(whole._h, g3._h, 0, S, None, None, None, None, C.byref(ref_w), None))
    ref_p = (_abi.PopTotals * G)()
    _abi.check(lib.fmh_population_summaries(whole._h, g3._h, 0, S, _abi.FORMULA_SPARSE, None, None, ref_p, None))

    comms = sharding.Comm.init_all([0] * ranks)
    assert all(c.transport == "host" and c.world == ranks for c in comms)
    miss_bits = None if words is None else np.unpackbits(words.view(np.uint8), bitorder="little")[:S * 2 * N].reshape(S, 2 * N)
    out, errors = [None] * ranks, []

    def work(r):
        try:
            b, e = sharding.slab_for_rank(S, r, ranks)
            w = None
            if miss_bits is not None:
                w = np.packbits(miss_bits[b:e].reshape(-1), bitorder="little")
                w = np.frombuffer(np.pad(w, (0, (-w.size) % 8)).tobytes(), dtype="<u8").copy()
            dm = device.DeviceMatrix.from_host(data[b:e], w, e - b, N, 2, 1)
            s2, s3 = device.Groups(dm, masks[:2]), device.Groups(dm, masks)
            h = _abi.HudsonTotals()
            _abi.check(lib.fmh_hudson_sweep_sharded(comms[r]._h, dm._h, s2._h, 0, e - b, _abi.FORMULA_SPARSE, None, C.byref(h), None))
            wc = _abi.WcTotals()
            _abi.check(lib.fmh_wc_sweep(dm._h, s3._h, 0, e - b, None, None, None, None, C.byref(wc), None))
            slots = 1 + G * (G - 1) // 2
            f, u = (C.c_double * (2 * slots))(), (C.c_uint64 * (slots + 1))()
            _abi.check(lib.fmh_wc_totals_pack(C.byref(wc), G, f, u))
            _abi.check(lib.fmh_allreduce_totals(comms[r]._h, f, 2 * slots, u, slots + 1))
            wc_all = _abi.WcTotals()
            _abi.check(lib.fmh_wc_totals_unpack(C.byref(wc_all), G, f, u))
            pt = (_abi.PopTotals * G)()
            _abi.check(lib.fmh_population_summaries(dm._h, s3._h, 0, e - b, _abi.FORMULA_SPARSE, None, None, pt, None))
            pf, pu = (C.c_double * G)(), (C.c_uint64 * (3 * G + 1))()
            _abi.check(lib.fmh_pop_totals_pack(pt, G, pf, pu))
            _abi.check(lib.fmh_allreduce_totals(comms[r]._h, pf, G, pu, 3 * G + 1))
            p_all = (_abi.PopTotals * G)()
            _abi.check(lib.fmh_pop_totals_unpack(p_all, G, pf, pu))
            out[r] = (h, wc_all, p_all)
        except Exception as exc:  # noqa: BLE001 - reported by the main thread
            errors.append((r, exc))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(ranks):
        h, wc_all, p_all = out[r]
        _totals_equal(h, ref_h, 1e-12)
        for k in range(1 + G * (G - 1) // 2):
            assert wc_all.informative_sites[k] == ref_w.informative_sites[k]
            assert wc_all.sum_a[k] == pytest.approx(ref_w.sum_a[k], rel=1e-11, abs=1e-13)
            assert wc_all.sum_b[k] == pytest.approx(ref_w.sum_b[k], rel=1e-11, abs=1e-13)
        assert wc_all.sites_attempted == S
        for p in range(G):
            assert (p_all[p].segregating_sites, p_all[p].uncallable_sites, p_all[p].haplotype_capacity) == \
                   (ref_p[p].segregating_sites, ref_p[p].uncallable_sites, ref_p[p].haplotype_capacity)
            assert p_all[p].pi_sum == pytest.approx(ref_p[p].pi_sum, rel=1e-12)
    # every rank left the rendezvous with the same bits (rank-order summation)
    for r in range(1, ranks):
        assert out[r][0].numerator_sum == out[0][0].numerator_sum and out[r][1].sum_a[0] == out[0][1].sum_a[0]
    for c in comms:
        c.close()
