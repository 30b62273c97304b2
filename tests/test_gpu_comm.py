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
    # pipelined: windows in flight up to FMH_SHARDED_IN_FLIGHT (4), collected in order; one more begin without an end is refused
    half = S // 2
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 0, half, _abi.FORMULA_DENSE, None, None))
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, half, S - half, _abi.FORMULA_DENSE, None, None))
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 0, 100, _abi.FORMULA_DENSE, None, None))
    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 100, 0, _abi.FORMULA_DENSE, None, None))
    assert lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_DENSE, None, None) == _abi.FMH_ERR_INVALID
    a, b, c3, c4 = _abi.HudsonTotals(), _abi.HudsonTotals(), _abi.HudsonTotals(), _abi.HudsonTotals()
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(a)))
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(b)))
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(c3)))
    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(c4)))
    assert c3.sites_with_components <= 100 and c4.sites_with_components == 0 and c4.numerator_sum == 0.0
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


def _wc_equal(a, b, slots, rel):
    assert a.sites_attempted == b.sites_attempted
    for k in range(slots):
        assert a.informative_sites[k] == b.informative_sites[k], k
        if rel == 0.0:
            assert a.sum_a[k] == b.sum_a[k] and a.sum_b[k] == b.sum_b[k], k
        else:
            assert a.sum_a[k] == pytest.approx(b.sum_a[k], rel=rel, abs=1e-13) and a.sum_b[k] == pytest.approx(b.sum_b[k], rel=rel, abs=1e-13), k


def test_rccl_collective_path_with_one_rank():
    """The device-side sharded W&C and population-summaries sweeps on a one-rank RCCL group equal the plain calls bit for bit (same
    kernel, same grid, a sum over one rank); the communicator describes itself; the reduce latency is measured; mixed kinds
    pipeline in FIFO order and an _end of the wrong kind is refused."""
    from ferromic_amd import _abi, device, sharding

    lib = _abi.load()
    comm = sharding.Comm.single(0)
    d = comm.describe()
    assert d["transport"] == "rccl" and d["world"] == 1 and d["rank"] == 0 and d["device"] == 0 and d["in_flight"] == 4
    assert "rccl" in str(d["rccl_library"]).lower() and str(d["rccl_version"]).isdigit(), d
    S, N = 40_001, 130
    for (G, max_allele, missing) in ((4, 1, 0.0), (3, 1, 0.02), (2, 3, 0.01), (6, 1, 0.0)):
        rng = np.random.default_rng(G * 100 + max_allele)
        data, words = _cohort(S, N, 17 + G, missing=missing)
        if max_allele > 1:
            data = (data * rng.integers(1, max_allele + 1, size=data.shape)).astype(np.uint8)
        dm = device.DeviceMatrix.from_host(data, words, S, N, 2, max_allele)
        masks = np.zeros((G, 2 * N), np.uint8)
        for k in range(G):
            masks[k, 2 * (k * N // G):2 * ((k + 1) * N // G)] = 1
        g = device.Groups(dm, masks)
        slots = 1 + G * (G - 1) // 2
        plain_w, got_w = _abi.WcTotals(), _abi.WcTotals()
        da, db = device.DeviceBuffer(0, 8 * slots * S), device.DeviceBuffer(0, 8 * slots * S)
        _abi.check(lib.fmh_wc_sweep(dm._h, g._h, 0, S, da.ptr, db.ptr, None, None, C.byref(plain_w), None))
        a_plain = da.to_numpy(np.float64, slots * S)
        _abi.check(lib.fmh_wc_sweep_sharded(comm._h, dm._h, g._h, 0, S, da.ptr, db.ptr, None, None, C.byref(got_w), None))
        _wc_equal(got_w, plain_w, slots, 0.0)
        assert np.array_equal(da.to_numpy(np.float64, slots * S), a_plain, equal_nan=True)
        plain_p, got_p = (_abi.PopTotals * G)(), (_abi.PopTotals * G)()
        _abi.check(lib.fmh_population_summaries(dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, None, None, plain_p, None))
        _abi.check(lib.fmh_population_summaries_sharded(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, None, None, got_p, None))
        for p in range(G):
            for k, _ in _abi.PopTotals._fields_:
                assert getattr(got_p[p], k) == getattr(plain_p[p], k), (G, p, k)
        # two kinds in flight: collected oldest first, by their own _end only
        half = S // 2
        _abi.check(lib.fmh_wc_sweep_sharded_begin(comm._h, dm._h, g._h, 0, half, None, None, None, None, None))
        _abi.check(lib.fmh_population_summaries_sharded_begin(comm._h, dm._h, g._h, half, S - half, _abi.FORMULA_SUMMARY, None, None, None))
        assert lib.fmh_population_summaries_sharded_end(comm._h, got_p) == _abi.FMH_ERR_INVALID and b"W&C" in lib.fmh_last_error()
        assert lib.fmh_hudson_sweep_sharded_end(comm._h, None) == _abi.FMH_ERR_INVALID
        w1 = _abi.WcTotals()
        _abi.check(lib.fmh_wc_sweep_sharded_end(comm._h, C.byref(w1)))
        _abi.check(lib.fmh_population_summaries_sharded_end(comm._h, got_p))
        w2 = _abi.WcTotals()
        _abi.check(lib.fmh_wc_sweep_sharded(comm._h, dm._h, g._h, half, S - half, None, None, None, None, C.byref(w2), None))
        assert w1.sites_attempted == half and w2.sites_attempted == S - half
        for k in range(slots):
            assert w1.informative_sites[k] + w2.informative_sites[k] == plain_w.informative_sites[k]
            assert w1.sum_a[k] + w2.sum_a[k] == pytest.approx(plain_w.sum_a[k], rel=1e-11, abs=1e-13)
        # an empty slab takes part with zeros
        _abi.check(lib.fmh_wc_sweep_sharded(comm._h, dm._h, g._h, 5, 0, None, None, None, None, C.byref(w2), None))
        assert w2.sites_attempted == 0 and w2.informative_sites[0] == 0 and w2.sum_a[0] == 0.0
    # the reduce is timed with events on the communicator's stream when sweeps are timed
    lib.fmh_timing_enable(1)
    lib.fmh_timing_reset()
    lib.fmh_timing_reset_reduce()
    for _ in range(5):
        _abi.check(lib.fmh_population_summaries_sharded(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, None, None, got_p, None))
    ms, n = C.c_double(), C.c_uint64()
    lib.fmh_timing_read_reduce(C.byref(ms), C.byref(n))
    lib.fmh_timing_enable(0)
    assert n.value == 5 and 0.0 < ms.value < 1000.0, (n.value, ms.value)
    # refusals happen before anything is enqueued
    assert lib.fmh_wc_sweep_sharded_begin(comm._h, dm._h, g._h, S, 1, None, None, None, None, None) == _abi.FMH_ERR_INVALID
    one = device.Groups(dm, masks[:1])
    assert lib.fmh_wc_sweep_sharded_begin(comm._h, dm._h, one._h, 0, S, None, None, None, None, None) == _abi.FMH_ERR_INVALID
    _abi.check(lib.fmh_population_summaries_sharded(comm._h, dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, None, None, got_p, None))  # still usable
    comm.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_host_transport_device_side_wc_and_summaries(ranks):
    """fmh_wc_sweep_sharded / fmh_population_summaries_sharded over `ranks` slab matrices (in-process transport) equal the whole
    matrix: integers exactly, f64 sums to 1e-11 (the rank-order sum differs from the one-grid order), for 2..4 groups (one fused
    kernel) and 5 groups (the blocking route inside _begin)."""
    from ferromic_amd import _abi, device, sharding

    lib = _abi.load()
    S, N = 30_011, 100
    data, words = _cohort(S, N, 23 + ranks, missing=0.02)
    miss_bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:S * 2 * N].reshape(S, 2 * N)
    whole = device.DeviceMatrix.from_host(data, words, S, N, 2, 1)
    comms = sharding.Comm.init_all([0] * ranks)
    for G in (2, 4, 5):
        masks = np.zeros((G, 2 * N), np.uint8)
        for k in range(G):
            masks[k, 2 * (k * N // G):2 * ((k + 1) * N // G)] = 1
        gw = device.Groups(whole, masks)
        slots = 1 + G * (G - 1) // 2
        ref_w, ref_p = _abi.WcTotals(), (_abi.PopTotals * G)()
        _abi.check(lib.fmh_wc_sweep(whole._h, gw._h, 0, S, None, None, None, None, C.byref(ref_w), None))
        _abi.check(lib.fmh_population_summaries(whole._h, gw._h, 0, S, _abi.FORMULA_SPARSE, None, None, ref_p, None))
        out, errors = [None] * ranks, []

        def work(r):
            try:
                b, e = sharding.slab_for_rank(S, r, ranks)
                w = np.packbits(miss_bits[b:e].reshape(-1), bitorder="little")
                w = np.frombuffer(np.pad(w, (0, (-w.size) % 8)).tobytes(), dtype="<u8").copy()
                dm = device.DeviceMatrix.from_host(data[b:e], w, e - b, N, 2, 1)
                g = device.Groups(dm, masks)
                wc, pt = _abi.WcTotals(), (_abi.PopTotals * G)()
                # both enqueued before either is collected: the two reduces travel back to back
                _abi.check(lib.fmh_wc_sweep_sharded_begin(comms[r]._h, dm._h, g._h, 0, e - b, None, None, None, None, None))
                _abi.check(lib.fmh_population_summaries_sharded_begin(comms[r]._h, dm._h, g._h, 0, e - b, _abi.FORMULA_SPARSE, None, None, None))
                _abi.check(lib.fmh_wc_sweep_sharded_end(comms[r]._h, C.byref(wc)))
                _abi.check(lib.fmh_population_summaries_sharded_end(comms[r]._h, pt))
                out[r] = (wc, pt)
            except Exception as exc:  # noqa: BLE001
                errors.append((r, exc))

        threads = [threading.Thread(target=work, args=(r,)) for r in range(ranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        for r in range(ranks):
            wc, pt = out[r]
            _wc_equal(wc, ref_w, slots, 1e-11)
            assert wc.sites_attempted == S
            for p in range(G):
                assert (pt[p].segregating_sites, pt[p].uncallable_sites, pt[p].haplotype_capacity) == \
                       (ref_p[p].segregating_sites, ref_p[p].uncallable_sites, ref_p[p].haplotype_capacity)
                assert pt[p].pi_sum == pytest.approx(ref_p[p].pi_sum, rel=1e-12)
        for r in range(1, ranks):  # rank-order summation: every rank leaves with the same bits
            assert out[r][0].sum_a[0] == out[0][0].sum_a[0] and out[r][1][0].pi_sum == out[0][1][0].pi_sum
    for c in comms:
        c.close()


def test_abort_wakes_the_peers_instead_of_hanging():
    """A rank that fails before its collective calls fmh_comm_abort: the peers blocked in the in-process rendezvous return an error
    (they used to wait for ever), and every later collective on the group is refused."""
    import time

    from ferromic_amd import _abi, sharding

    lib = _abi.load()
    comms = sharding.Comm.init_all([0, 0, 0])
    results = {}

    def waiter(r):
        f, u = (C.c_double * 2)(1.0, 2.0), (C.c_uint64 * 1)(3)
        results[r] = (lib.fmh_allreduce_totals(comms[r]._h, f, 2, u, 1), lib.fmh_last_error())

    threads = [threading.Thread(target=waiter, args=(r,)) for r in (0, 2)]
    for t in threads:
        t.start()
    time.sleep(0.3)  # both are inside the rendezvous, waiting for rank 1
    assert all(t.is_alive() for t in threads)
    comms[1].abort()  # rank 1 "failed" before its collective
    for t in threads:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in threads), "peers still blocked after fmh_comm_abort"
    for r in (0, 2):
        assert results[r][0] == _abi.FMH_ERR_INVALID and b"aborted by rank 1" in results[r][1], results[r]
    f, u = (C.c_double * 1)(1.0), (C.c_uint64 * 1)(1)
    assert lib.fmh_allreduce_totals(comms[0]._h, f, 1, u, 1) == _abi.FMH_ERR_INVALID
    for c in comms:
        c.close()
