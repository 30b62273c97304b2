"""The N>1 path on CPU: two gloo ranks, region-sharded accumulators, one all-reduce."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_partition():
    from ferromic_amd import sharding

    for S, G in ((10, 3), (10_000_000, 8), (7, 8), (0, 2)):
        slabs = [sharding.slab_for_rank(S, r, G) for r in range(G)]
        assert slabs[0][0] == 0 and slabs[-1][1] == S
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
    with pytest.raises(ValueError):
        sharding.slab_for_rank(10, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_region_sharded_allreduce_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29511 + world), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"GLOO_SHARDING_OK world={world}" in out.stdout


def test_totals_ride_exactly_in_one_f64_vector():
    """One all-reduce carries the f64 sums and the u64 counts: counts below 2^53 survive the f64 round trip exactly, larger
    ones are refused rather than rounded."""
    from ferromic_amd import _abi, sharding

    t = _abi.HudsonTotals()
    t.numerator_sum, t.denominator_sum, t.pi1_sum = 0.1 + 0.2, 1e-300, 3.0e15 + 0.5
    t.dxy_uncallable_sites = (1 << 53) - 1
    t.sites_with_components = 10_000_000 * 8
    t.pop[0].segregating_sites = 123_456_789_012
    t.pop[1].uncallable_sites = 7
    v = sharding._pack(t)
    assert len(v) == _abi.HUDSON_PACK_F64 + _abi.HUDSON_PACK_U64 and all(isinstance(x, float) for x in v)
    back = sharding._unpack(v)
    assert back.numerator_sum == t.numerator_sum and back.denominator_sum == t.denominator_sum and back.pi1_sum == t.pi1_sum
    assert back.dxy_uncallable_sites == (1 << 53) - 1 and back.sites_with_components == 80_000_000
    assert back.pop[0].segregating_sites == 123_456_789_012 and back.pop[1].uncallable_sites == 7
    doubled = sharding._unpack([2 * x for x in v[:_abi.HUDSON_PACK_F64]] + [x + x for x in v[_abi.HUDSON_PACK_F64:]])  # what a 2-rank sum does
    assert doubled.sites_with_components == 160_000_000 and doubled.pop[0].segregating_sites == 246_913_578_024
    t.sites_with_components = 1 << 53
    with pytest.raises(OverflowError):
        sharding._pack(t)
