"""bench.py's N > 1 path rehearsed on one GPU: two ranks (gloo, both on cuda:0) each own a slab of the cohort,
sweep it and sum the accumulators; the combined totals must equal one rank sweeping both slabs.  The library's own RCCL
communicator is exercised with a one-rank group (tests/test_gpu_comm.py covers its in-process multi-rank transport)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_equal_one_rank():
    """Two ranks on one GPU (accumulators summed through torch.distributed/gloo: RCCL cannot place two ranks on one device), weak and
    strong scaling, against one rank sweeping the same cohort."""
    S = 300_000
    common = ["--steps", "2", "--warmup", "1", "--haplotypes", "1000", "--no-cpu-baseline"]
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    rehearse = ["--transport", "torch", "--backend", "gloo", "--rehearse-on-one-device"]
    weak = run(launch + ["--master-port", "29541", "bench.py", "--gpus", "2", "--sites", str(S), "--scaling", "weak", "--no-secondary"] + rehearse + common)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["steps"] == 2 and weak["config"]["sites_per_gpu"] == S
    assert weak["value"] == pytest.approx(2 * S * 2 / (weak["ms_per_step"] * 2 / 1e3), rel=1e-6)  # whole-job sites / max-over-ranks time
    strong = run(launch + ["--master-port", "29542", "bench.py", "--gpus", "2", "--sites", str(2 * S)] + rehearse + common)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["config"]["sites_per_gpu"] == S and strong["config"]["total_sites"] == 2 * S
    assert strong["value"] == pytest.approx(2 * S * 2 / (strong["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert strong["secondary"]["scaling"] == "weak" and strong["secondary"]["sites_per_gpu"] == 2 * S and strong["secondary"]["total_sites"] == 4 * S
    # the N > 1 line carries its own evidence: who summed, what every rank swept and measured, and the all-rank totals against one GPU
    # sweeping the whole cohort (computed by rank 0 after the timed region)
    c = strong["comm"]
    assert c["world"] == 2 and "torch.distributed" in c["transport"] and c["ranks_reporting"] == 2 and c["rccl_library"] is None
    assert [r["rank"] for r in c["per_rank"]] == [0, 1] and [r["slab"] for r in c["per_rank"]] == [[0, S], [S, 2 * S]]
    for r in c["per_rank"]:
        assert r["sites"] == S and r["kernel_launches_timed"] >= 1 and 0 < r["kernel_ms_min"] <= r["kernel_ms_avg"] <= r["kernel_ms_max"]
        assert r["elapsed_ms_per_step"] > 0 and r["reduce_ms_avg"] is None  # torch transport: no library-side reduce to time
    assert c["kernel_ms_avg_min_over_ranks"] <= c["kernel_ms_avg_max_over_ranks"]
    pv = strong["parity_vs_n1"]
    assert pv["mode"] == "strong" and pv["recorded"] is None  # not the default cohort: no recorded constants
    assert pv["computed"]["ok"] is True and pv["computed"]["segregating_sites"] is True and pv["computed"]["sites_with_components"] is True
    assert pv["computed"]["hudson_fst_rel_err"] <= 1e-9 and pv["computed"]["pi_sum_rel_err"] <= 1e-9
    assert weak["parity_vs_n1"]["mode"] == "weak" and "computed" not in weak["parity_vs_n1"]
    # the same 2 x S sites generated and swept by one rank (same seed recipe: seed = total sites + samples)
    one = run([sys.executable, "bench.py", "--sites", str(2 * S)] + common)
    assert one["scaling"] == "strong" and "secondary" not in one
    for two in (weak, strong):
        assert one["config"]["seed"] == two["config"]["seed"]
        assert one["results"]["segregating_sites"] == two["results"]["segregating_sites"]
        for a, b in zip(one["results"]["pi_sum"], two["results"]["pi_sum"]):
            assert a == pytest.approx(b, rel=1e-9)
        assert one["results"]["hudson_fst"] == pytest.approx(two["results"]["hudson_fst"], rel=1e-9)


def test_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher): the script starts torch.distributed.run itself before touching the GPU."""
    d = run([sys.executable, "bench.py", "--gpus", "2", "--sites", "200000", "--haplotypes", "400", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             "--no-secondary", "--transport", "torch", "--backend", "gloo", "--rehearse-on-one-device"])
    assert d["n_gpus"] == 2 and d["config"]["sites_per_gpu"] == 100000 and d["scaling"] == "strong"


def test_rccl_collective_path_with_one_rank():
    """The sharded path through the library's own RCCL communicator (a one-rank group): sweep, device-side finalise, ncclAllReduce on the
    communicator's stream, pipelined one step deep - same totals as the run without it, and both layouts' roofline blocks present."""
    common = ["--steps", "4", "--warmup", "1", "--sites", "250000", "--haplotypes", "1000", "--no-cpu-baseline"]
    plain = run([sys.executable, "bench.py"] + common)
    coll = run([sys.executable, "bench.py", "--force-collective"] + common)
    assert "RCCL" in coll["config"]["parallelism"] and "no collective" in plain["config"]["parallelism"] and "pipelined" in plain["config"]["parallelism"]
    assert coll["results"] == plain["results"]
    # what summed: reported by the library itself (fmh_comm_describe), with the librccl file it bound and the measured reduce latency
    c = coll["comm"]
    assert c["transport"] == "rccl" and c["world"] == 1 and c["rank"] == 0 and "rccl" in str(c["rccl_library"]).lower() and str(c["rccl_version"]).isdigit()
    assert c["source"].startswith("fmh_comm_describe") and c["ranks_reporting"] == 1 and c["per_rank"][0]["slab"] == [0, 250000]
    assert c["per_rank"][0]["reduces_timed"] >= 1 and 0 < c["per_rank"][0]["reduce_ms_avg"] < 50 and c["reduce_ms_avg_max_over_ranks"] > 0
    assert coll["parity_vs_n1"]["computed"]["ok"] is True and coll["parity_vs_n1"]["computed"]["hudson_fst_rel_err"] == 0.0  # one rank: the same sweep
    assert "comm" not in plain and "parity_vs_n1" not in plain
    blocking = run([sys.executable, "bench.py", "--sync-steps"] + common)
    assert "blocking" in blocking["config"]["parallelism"] and blocking["results"] == plain["results"]
    assert coll["roofline"]["kernel_ms_avg"] > 0  # HIP events of the pipelined launches
    for d in (plain, coll):
        r = d["roofline"]
        assert r["algorithmic_bytes_per_site"] == 125 + 56 and r["u8_layout_bytes_per_site"] == 1000 + 56
        # u8 rows and bit planes: the same per-site values; the regional sums are taken over differently sized grids (1e-9 contract)
        assert r["u8_layout_measured"]["hudson_fst"] == pytest.approx(d["results"]["hudson_fst"], rel=1e-12)
        assert 0 < r["frac"] and 0 < r["u8_layout_measured"]["frac"]
    u8 = run([sys.executable, "bench.py", "--layout", "bytes"] + common)
    assert u8["results"]["segregating_sites"] == plain["results"]["segregating_sites"]
    assert u8["results"]["hudson_fst"] == pytest.approx(plain["results"]["hudson_fst"], rel=1e-12)
    for a, b in zip(u8["results"]["pi_sum"], plain["results"]["pi_sum"]):
        assert a == pytest.approx(b, rel=1e-12)
    assert u8["roofline"]["algorithmic_bytes_per_site"] == 1056 and u8["dtype"] == "u8"


def test_single_process_route():
    """`bench.py --single-process`: one process, one host thread per rank, fmh_comm_init_all - the route run_vcf --devices takes, without
    torch.distributed.run or gloo.  [0]: the library's RCCL communicator with one rank; [0, 0]: two ranks on the in-process host rendezvous
    (RCCL cannot place two ranks on one device).  Both carry the same evidence as the launcher route and agree with one plain rank."""
    S = 300_000
    common = ["--steps", "3", "--warmup", "1", "--haplotypes", "1000", "--no-cpu-baseline", "--sites", str(2 * S)]
    one = run([sys.executable, "bench.py"] + common)
    rccl = run([sys.executable, "bench.py", "--single-process", "--devices", "0"] + common)
    c = rccl["comm"]
    assert rccl["n_gpus"] == 1 and c["transport"] == "rccl" and c["world"] == 1 and "rccl" in str(c["rccl_library"]).lower() and c["source"].startswith("fmh_comm_describe")
    assert c["ranks_reporting"] == 1 and c["per_rank"][0]["slab"] == [0, 2 * S] and c["per_rank"][0]["reduces_timed"] >= 1 and c["per_rank"][0]["reduce_ms_avg"] > 0
    assert rccl["results"] == one["results"] and "launcher" in rccl["config"] and "transport_fallback" not in rccl["config"]
    assert rccl["parity_vs_n1"]["computed"]["ok"] is True
    host = run([sys.executable, "bench.py", "--single-process", "--gpus", "2", "--devices", "0,0"] + common)
    c = host["comm"]
    assert host["n_gpus"] == 2 and c["transport"] == "host" and c["world"] == 2 and c["ranks_reporting"] == 2
    assert [r["rank"] for r in c["per_rank"]] == [0, 1] and [r["slab"] for r in c["per_rank"]] == [[0, S], [S, 2 * S]]
    for r in c["per_rank"]:
        assert r["sites"] == S and r["kernel_launches_timed"] >= 1 and 0 < r["kernel_ms_min"] <= r["kernel_ms_avg"] <= r["kernel_ms_max"] and r["elapsed_ms_per_step"] > 0
        assert r["comm"]["world"] == 2 and r["comm"]["transport"] == "host"
    assert host["value"] == pytest.approx(2 * S * 3 / (host["ms_per_step"] * 3 / 1e3), rel=1e-6)
    assert host["results"]["segregating_sites"] == one["results"]["segregating_sites"]
    assert host["results"]["hudson_fst"] == pytest.approx(one["results"]["hudson_fst"], rel=1e-9)
    pv = host["parity_vs_n1"]
    assert pv["mode"] == "strong" and pv["computed"]["ok"] is True and pv["computed"]["hudson_fst_rel_err"] <= 1e-9 and pv["computed"]["pi_sum_rel_err"] <= 1e-9
    weak = run([sys.executable, "bench.py", "--single-process", "--devices", "0,0", "--scaling", "weak", "--steps", "2", "--warmup", "1", "--haplotypes", "400",
                "--sites", "100000", "--no-cpu-baseline"])
    assert weak["scaling"] == "weak" and weak["config"]["total_sites"] == 200000 and weak["config"]["sites_per_gpu"] == 100000


def test_bench_line_contract():
    """The one JSON line the driver parses: every field of the contract, with the roofline and cpu_baseline objects."""
    d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--sites", "200000", "--haplotypes", "400", "--cpu-sample-sites", "50000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "sites/s" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(200000 * 3 / (d["ms_per_step"] * 3 / 1e3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert "traffic" in r and "traffic_source" in r and r["algorithmic_bytes_per_site"] == 50 + 56
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] == c["cores_usable"] <= c["cores_visible"] and c["value"] > 0 and c["unit"] == "sites/s" and "sample" in c
    assert c["parity_vs_gpu"]["alt_counts_bit_exact"] is True and c["parity_vs_gpu"]["per_site_f64_max_rel_err"] <= 1e-9
