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
