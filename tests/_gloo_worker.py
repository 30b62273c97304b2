"""World-size-2 rehearsal of the region-sharded path on CPU (gloo): every rank computes the regional
accumulators of ITS slab (here with the oracle, since no GPU exists in this container), the ranks
combine them through ferromic_amd.sharding.allreduce_hudson_totals — the exact code bench.py runs
over RCCL — and rank 0 checks the result against the unsharded cohort."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from ferromic_amd import _abi, sharding
    from oracle import dense as D

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    S, Hcols, seed = 3001, 64, 4242
    rng = np.random.default_rng(seed)
    thr = (np.clip(rng.beta(0.8, 0.8, size=(1, S)) + np.array([[0.03], [-0.03]]), 0.001, 0.999) * (1 << 24)).astype(np.uint32)
    poc = (np.arange(Hcols) >= Hcols // 2).astype(np.uint8)
    off1 = np.nonzero(poc == 0)[0]
    off2 = np.nonzero(poc == 1)[0]

    def totals_of(begin, end):
        data, words = D.generate(end - begin, Hcols, seed, begin, np.ascontiguousarray(thr[:, begin:end]), poc,
                                 int(0.03 * (1 << 24)), 2)
        out = D.hudson_sweep(data, words, end - begin, Hcols, off1, off2, 2, want_sites=False)
        t = _abi.HudsonTotals()
        for k, v in out.totals.items():
            setattr(t, k, v)
        for p in range(2):
            for k, v in out.pop[p].items():
                setattr(t.pop[p], k, v)
        return t

    begin, end = sharding.slab_for_rank(S, rank, world)
    mine = totals_of(begin, end)
    merged = sharding.allreduce_hudson_totals(mine, dist, "cpu")
    # the pipelined form bench.py uses: step k's reduce is collected when step k + 1 submits, the last one by flush()
    pipe = sharding.HudsonTotalsPipeline(dist, "cpu")
    half = totals_of(begin, begin + (end - begin) // 2)
    pipe.submit(half)
    assert pipe.latest is None
    pipe.submit(mine)
    first = pipe.latest
    piped = pipe.flush()
    assert pipe.flush() is piped
    assert first.sites_with_components < piped.sites_with_components or S < 4
    for k, _ in _abi.HudsonTotals._fields_:
        if k != "pop":
            assert getattr(piped, k) == getattr(merged, k), k
    for p in range(2):
        assert piped.pop[p].segregating_sites == merged.pop[p].segregating_sites and piped.pop[p].pi_sum == merged.pop[p].pi_sum
    # W&C regional sums (every slot of 3 groups) and per-population summaries ride the same way: pack -> one sum -> unpack.
    # Each rank's slab totals are a deterministic function of the rank, so every rank can check the merged result.
    G, slots = 3, 4

    def wc_of(r):
        t = _abi.WcTotals()
        for k in range(slots):
            t.sum_a[k] = 0.125 * (r + 1) * (k + 1) + 1e-9 * r
            t.sum_b[k] = 3.5 * (r + 2) - k
            t.informative_sites[k] = 1000 * (r + 1) + k
        t.sites_attempted = 1200 * (r + 1)
        return t

    def pops_of(r):
        out = []
        for p in range(G):
            t = _abi.PopTotals()
            t.haplotype_capacity, t.segregating_sites, t.uncallable_sites, t.pi_sum = 20 + p, 300 * (r + 1) + p, 7 * r + p, 11.25 * (r + 1) + 0.5 * p
            out.append(t)
        return out

    wc = sharding.allreduce_wc_totals(wc_of(rank), G, dist, "cpu")
    pops = sharding.allreduce_pop_totals(pops_of(rank), dist, "cpu")
    for k in range(slots):
        assert wc.informative_sites[k] == sum(wc_of(r).informative_sites[k] for r in range(world))
        assert abs(wc.sum_a[k] - sum(wc_of(r).sum_a[k] for r in range(world))) <= 1e-12
        assert abs(wc.sum_b[k] - sum(wc_of(r).sum_b[k] for r in range(world))) <= 1e-12
    assert wc.sites_attempted == sum(1200 * (r + 1) for r in range(world)) and wc.sum_a[slots] == 0.0
    for p in range(G):
        assert pops[p].haplotype_capacity == 20 + p  # a per-rank constant, not a sum
        assert pops[p].segregating_sites == sum(300 * (r + 1) + p for r in range(world))
        assert pops[p].uncallable_sites == sum(7 * r + p for r in range(world))
        assert abs(pops[p].pi_sum - sum(11.25 * (r + 1) + 0.5 * p for r in range(world))) <= 1e-12
    if rank == 0:
        whole = totals_of(0, S)
        for k, _ in _abi.HudsonTotals._fields_:
            if k == "pop":
                continue
            a, b = getattr(merged, k), getattr(whole, k)
            if isinstance(a, int):
                assert a == b, (k, a, b)
            else:
                assert abs(a - b) <= 1e-9 * max(abs(b), 1e-12), (k, a, b)
        for p in range(2):
            assert merged.pop[p].segregating_sites == whole.pop[p].segregating_sites
            assert merged.pop[p].uncallable_sites == whole.pop[p].uncallable_sites
            assert merged.pop[p].haplotype_capacity == whole.pop[p].haplotype_capacity == Hcols // 2
            assert abs(merged.pop[p].pi_sum - whole.pop[p].pi_sum) <= 1e-9 * whole.pop[p].pi_sum
        covered = sum(sharding.slab_for_rank(S, r, world)[1] - sharding.slab_for_rank(S, r, world)[0] for r in range(world))
        assert covered == S
        print(f"GLOO_SHARDING_OK world={world} fst={merged.numerator_sum / merged.denominator_sum:.12f}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
