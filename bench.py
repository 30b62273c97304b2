#!/usr/bin/env python3
"""bench.py — variant-sites/sec of the fused "pi + Hudson FST" sweep (BASELINE.json metric).

A step = one pass of the hot path (per-site allele counts for both populations, per-site pi1/pi2/Dxy/num/den tracks,
regional accumulators summed over all ranks and back on the host) over the synthetic cohort already resident in HBM.
Workload = BASELINE config C4: 10 M sites x 5 000 haplotypes, 2 populations.

  N = 1   the whole cohort on one GPU; the same begin / end calls on a local one-rank communicator (no RCCL, nothing to exchange), i.e. the
          next sweep is enqueued while the previous one's totals travel to the host (--sync-steps: one blocking fmh_hudson_sweep per step).
  N > 1   region sharding: rank r owns one contiguous slab and only that slab; the regional accumulators are
          summed by RCCL inside libferromic_hip.so (fmh_hudson_sweep_sharded_begin/_end: finalise on the device, ncclAllReduce
          on the communicator's stream, pipelined one step deep so the reduce of step k overlaps the sweep of step k + 1).
          --scaling strong (default): the 10 M sites are SPLIT over the ranks (the metric's "10M sites, 1 -> 8 GPUs");
          --scaling weak: every rank owns --sites sites of an N x larger cohort.  The other mode is measured in the same run
          and reported under "secondary".

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 --master-port P bench.py --gpus N --steps K --warmup W
or simply `python bench.py --gpus N`: the script then starts that launcher itself, before touching the GPU.
Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
W_OUT_HUDSON = 56      # bytes of per-site results written: 2*(4+4) counts + 5*8 (pi1, pi2, dxy, num, den); SURVEY.md 8(d)


def synthetic_thresholds(sites: int, first_site: int, seed: int, sigma: float = 0.05) -> np.ndarray:
    """Per-site population frequencies following the reference's benchmark recipe
    (src/pybenches/test_population_statistics_benchmarks.py:113-158): base ~ Beta(0.8, 0.8),
    divergence ~ N(0, sigma), clip [0.001, 0.999]; rows 0 and 1 of the cohort forced informative.
    Returned as 24-bit integer thresholds [2][sites] for the counter-based generator."""
    # a pure function of the GLOBAL site index (blocks of 2^20 sites, one stream per block), so any sharding of the
    # cohort sees the same frequencies as one GPU holding all of it
    block = 1 << 20
    f1 = np.empty(sites, dtype=np.float64)
    f2 = np.empty(sites, dtype=np.float64)
    b = first_site // block
    while b * block < first_site + sites:
        rng = np.random.default_rng([seed, b])
        base = rng.beta(0.8, 0.8, size=block)
        div = rng.normal(0.0, sigma, size=block)
        lo, hi = max(first_site, b * block), min(first_site + sites, (b + 1) * block)
        f1[lo - first_site:hi - first_site] = np.clip(base + div, 0.001, 0.999)[lo - b * block:hi - b * block]
        f2[lo - first_site:hi - first_site] = np.clip(base - div, 0.001, 0.999)[lo - b * block:hi - b * block]
        b += 1
    thr = np.stack([f1, f2]) * float(1 << 24)
    thr = thr.astype(np.uint32)
    if first_site == 0 and sites > 0:
        thr[0, 0], thr[1, 0] = 0, 1 << 24  # pop1 all ref, pop2 all alt
    if first_site <= 1 < first_site + sites:
        thr[0, 1 - first_site], thr[1, 1 - first_site] = 1 << 23, 1 << 24
    return thr


def usable_cores() -> dict:
    """Host cores this process can actually run on: the affinity mask, further capped by the cgroup CPU quota
    (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  os.cpu_count() is the machine, not the share."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = int(fq.read()), int(fp.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    usable = affinity
    if quota is not None:
        usable = max(1, min(affinity, int(quota + 0.5)))
    return {"cores_visible": visible, "cores_affinity": affinity, "cgroup_cpu_quota": quota, "cores_usable": usable}


def cpu_baseline(args, sample: int, thr_full: np.ndarray, poc: np.ndarray, seed: int, gpu_check):
    """The reference algorithm restated in C (oracle/dense_oracle.c), site ranges over host threads,
    on a bounded sample of the same cohort (rows [0, sample) are bit-identical to the GPU's)."""
    from oracle import dense as D

    host = usable_cores()
    cores = args.cpu_threads if args.cpu_threads > 0 else host["cores_usable"]
    H = args.haplotypes
    data, _ = D.generate(sample, H, seed, 0, np.ascontiguousarray(thr_full[:, :sample]), poc, 0, cores)
    off1 = np.nonzero(poc == 0)[0].astype(np.uint64)
    off2 = np.nonzero(poc == 1)[0].astype(np.uint64)
    best = float("inf")
    out = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = D.hudson_sweep(data, None, sample, H, off1, off2, cores, want_sites=True)
        best = min(best, time.perf_counter() - t0)
    parity = gpu_check(sample, out)
    return {
        "value": sample / best,
        "unit": "sites/s",
        "cores": cores,  # threads the restatement ran on = the cores this process may use (affinity mask capped by the cgroup quota)
        "cores_visible": host["cores_visible"],
        "cores_usable": host["cores_usable"],
        "cgroup_cpu_quota": host["cgroup_cpu_quota"],
        "kind": "port",
        "sample": f"first {sample} sites x {H} haplotypes of the same cohort, best of 3 passes "
                  f"({best:.3f} s each), C restatement of stats.rs:1367-1470 + 1554-1623 + 3179-3278",
        "parity_vs_gpu": parity,
    }


def kernel_source_sha() -> str:
    """Identity of the sweep kernel sources: profiles/pmc_traffic.json records the one its counters were collected on."""
    h = hashlib.sha256()
    # the kernels, their launcher (grid, deferral depth) and abi.hip, where the batch depth, lanes per row and the deferral's LDS room - which
    # decide the access pattern - are chosen
    for name in ("sweep_kernels.hpp", "sweep_launch.inc", "abi.hip"):
        with open(os.path.join(ROOT, "ferromic_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_traffic(sites: int, haplotypes: int, layout: str):
    """HBM bytes per launch from the committed rocprofv3 --pmc collection (never measured in this run)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None, "none: profiles/pmc_traffic.json is absent"
    try:
        rec = json.load(open(tpath)).get(f"{sites}x{haplotypes}:{layout}")
    except (OSError, ValueError):
        rec = None
    if not rec:
        return None, "none: no recorded collection for this shape and layout"
    sha = rec.get("kernel_source_sha")
    where = f"static profiles/pmc_traffic.json ({rec.get('source', '?')}); separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run"
    if sha != kernel_source_sha():
        return None, f"stale: collected on kernel sources {sha}, this build is {kernel_source_sha()} ({where})"
    return rec["hbm_bytes_per_launch"], where + f", kernel sources {sha}"


# Recorded from N = 1 runs of the default cohort (10 M sites x 5 000 haplotypes, seed 10 002 500, profiles/r02/c4_bench.json and this round's):
# the integer totals are exact whatever the sharding, so any N must reproduce them; the f64 ratio within the 1e-9 contract.
N1_RECORDED = {
    (10_000_000, 5000, 10_002_500): {"segregating_sites": [9949061, 9949159], "sites_with_components": 10_000_000, "dxy_uncallable_sites": 0,
                                     "hudson_fst": 0.028471767808637662, "pi_sum": [3068822.9373544613, 3069429.427405202]},
}


def parity_vs_n1(args, first, totals, rank, world, local_rank, lib, masks, poc, N, H):
    """N > 1 strong scaling: the all-rank totals against one GPU sweeping the WHOLE cohort - (a) the recorded N = 1 constants when the run
    is the default cohort, (b) computed here, after the timed region, by rank 0 (the cohort is a pure function of the global site index, so
    rank 0 regenerates all of it: 10 M x 5 000 is 50 GB of u8 rows for the generator + 6.4 GB of planes, on a 288 GB card)."""
    from ferromic_amd import _abi, device

    out = {"mode": first["scaling"]}
    if first["scaling"] != "strong":
        out["note"] = "weak scaling sweeps an N x larger cohort: no one-GPU counterpart of the same cohort"
        return out
    got = {"segregating_sites": [int(totals.pop[0].segregating_sites), int(totals.pop[1].segregating_sites)],
           "sites_with_components": int(totals.sites_with_components), "dxy_uncallable_sites": int(totals.dxy_uncallable_sites),
           "hudson_fst": totals.numerator_sum / totals.denominator_sum if totals.denominator_sum > 1e-12 else None,
           "pi_sum": [totals.pop[0].pi_sum, totals.pop[1].pi_sum]}

    def compare(ref):
        res = {}
        for k in ("segregating_sites", "sites_with_components", "dxy_uncallable_sites"):
            if k in ref:
                res[k] = ref[k] == got[k]
        if ref.get("hudson_fst") is not None and got["hudson_fst"] is not None:
            res["hudson_fst_rel_err"] = abs(got["hudson_fst"] - ref["hudson_fst"]) / abs(ref["hudson_fst"])
        if "pi_sum" in ref:
            res["pi_sum_rel_err"] = max(abs(a - b) / abs(b) for a, b in zip(got["pi_sum"], ref["pi_sum"]))
        res["ok"] = all(v for k, v in res.items() if isinstance(v, bool)) and res.get("hudson_fst_rel_err", 0.0) <= 1e-9 and res.get("pi_sum_rel_err", 0.0) <= 1e-9
        return res

    rec = N1_RECORDED.get((first["total_sites"], H, first["seed"]))
    out["recorded"] = dict(compare(rec), source="bench.py N1_RECORDED (N = 1 runs of this cohort)") if rec else None
    if rank != 0 or args.no_verify_n1:
        return out
    total = first["total_sites"]
    if total * H > 120e9:
        out["computed"] = {"skipped": f"{total} x {H} u8 rows exceed the verification budget"}
        return out
    try:
        thr = synthetic_thresholds(total, 0, first["seed"])
        dm = device.DeviceMatrix.alloc(total, N, 2, with_missing=False, max_allele=1, device=local_rank)
        dm.generate(first["seed"], 0, thr, poc, 0)
        if args.layout == "packed":
            dm.pack(release_bytes=True)
        g = device.Groups(dm, masks)
        ref_t = _abi.HudsonTotals()
        _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, total, _abi.FORMULA_DENSE, None, C.byref(ref_t), None))
        ref = {"segregating_sites": [int(ref_t.pop[0].segregating_sites), int(ref_t.pop[1].segregating_sites)],
               "sites_with_components": int(ref_t.sites_with_components), "dxy_uncallable_sites": int(ref_t.dxy_uncallable_sites),
               "hudson_fst": ref_t.numerator_sum / ref_t.denominator_sum if ref_t.denominator_sum > 1e-12 else None,
               "pi_sum": [ref_t.pop[0].pi_sum, ref_t.pop[1].pi_sum]}
        out["computed"] = dict(compare(ref), source=f"rank 0 swept all {total} sites itself after the timed region (fmh_hudson_sweep, one GPU)", n1_values=ref)
        del g, dm
    except Exception as exc:  # noqa: BLE001 - the check must never cost the line
        out["computed"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--timing-sample", type=int, default=4,
                    help="bracket every n-th sweep of the timed region with HIP events (roofline.achieved comes from their average); the two event "
                         "records cost a few microseconds of stream time per timed launch, which is 2-3 %% of a 1.25 M-site step")
    ap.add_argument("--sites", type=int, default=10_000_000,
                    help="sites of the cohort: the WHOLE cohort with --scaling strong (split over the ranks), per GPU with --scaling weak")
    ap.add_argument("--haplotypes", type=int, default=5000)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: skip the measurement of the other scaling mode")
    ap.add_argument("--cpu-sample-sites", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = the cores this process may use)")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="who sums the accumulators over the ranks: libferromic_hip.so's own RCCL communicator (the product path) or "
                         "torch.distributed (with --backend gloo: the rehearsal of N > 1 on a one-GPU box)")
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend of the control plane (barriers, id exchange, max over ranks) and of --transport torch")
    ap.add_argument("--layout", choices=["packed", "bytes"], default="packed",
                    help="resident layout of the cohort: bit-packed planes (what fmh_matrix_create keeps for alleles 0..3) or the u8 rows")
    ap.add_argument("--u8-reference-steps", type=int, default=5,
                    help="N = 1 with --layout packed: before the u8 rows are released, time this many sweeps over them (the layout "
                         "north_star names) and report their HBM fraction next to the headline; 0 skips it")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the sharded path (communicator, device-side reduce, pipelining) even with one rank: measures its "
                         "software cost on a one-GPU box")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="sharded / pipelined steps the host keeps enqueued before it collects the oldest (at most FMH_SHARDED_IN_FLIGHT - 1 = 3): 1 = the "
                         "reduce of step k overlaps the sweep of step k + 1 only; 2 (default) also covers a reduce that has to wait for compute units "
                         "held by the next sweep's persistent grid, or that takes longer than one 0.17 ms sweep")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: run the pipelined steps on an explicit stream with FMH_GRAPH=1 - a repeated step is replayed from a captured hipGraph "
                         "(sweep + finalize + D2H as one launch); measures the fixed cost per step against the eager path")
    ap.add_argument("--explicit-stream", action="store_true", help="N = 1: the eager pipelined steps on the same explicit stream (the A side of --graph)")
    ap.add_argument("--no-verify-n1", action="store_true",
                    help="N > 1, strong scaling: skip rank 0's own sweep of the whole cohort after the timed region (parity_vs_n1.computed)")
    ap.add_argument("--sync-steps", action="store_true",
                    help="N = 1: one blocking fmh_hudson_sweep per step instead of the pipelined begin / end pair on a local communicator")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs driven by ONE process: N host threads, fmh_comm_init_all over --devices (default 0..N-1), the same pipelined "
                         "fmh_hudson_sweep_sharded_begin / _end loop and the same comm / per_rank / parity_vs_n1 evidence as the launcher route - the route "
                         "run_vcf --devices takes; needs neither torch.distributed.run nor gloo")
    ap.add_argument("--devices", default=None,
                    help="--single-process: comma-separated device indices, one rank each (default 0..gpus-1); a device listed twice selects the "
                         "library's in-process host rendezvous (the rehearsal of N > 1 on a one-GPU box)")
    ap.add_argument("--rehearse-on-one-device", action="store_true",
                    help="all ranks share cuda:0 (needs --transport torch --backend gloo): exercises the sharded code path on a one-GPU box")
    args = ap.parse_args()
    if not 1 <= args.pipeline_depth <= 3:  # FMH_SHARDED_IN_FLIGHT - 1: one more and _begin refuses mid-run
        ap.error("--pipeline-depth must be 1, 2 or 3 (the library keeps at most FMH_SHARDED_IN_FLIGHT = 4 sharded sweeps in flight per communicator)")
    return args


def main_single_process(args) -> int:
    """One process, one host thread per GPU, the library's own communicators (fmh_comm_init_all): what run_vcf --devices does, with the
    bench's step structure.  The threads exist before anything touches a GPU; if the RCCL communicators cannot be created the run goes
    through the library's in-process host rendezvous instead, says so (config.transport_fallback) and exits 3 after printing the line."""
    import threading

    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    world = len(devices)
    args.gpus = world
    H = args.haplotypes
    if H % 2:
        raise SystemExit("haplotypes must be even (diploid samples)")
    N = H // 2
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)
    masks = np.stack([(poc == 0), (poc == 1)]).astype(np.uint8)
    gate = threading.Barrier(world + 1)   # main <-> workers: communicators are ready (or the run is off)
    sync = threading.Barrier(world)       # the workers among themselves: both sides of the timed region
    shared = {"comms": None, "abort": False}
    reports = [None] * world
    errors = [None] * world

    def worker(r):
        try:
            gate.wait()
            if shared["abort"]:
                return
            from ferromic_amd import _abi, device, sharding

            lib = _abi.load()
            comm, dev = shared["comms"][r], devices[r]
            if args.scaling == "strong":
                total = args.sites
                begin, end = sharding.slab_for_rank(total, r, world)
            else:
                total = args.sites * world
                begin, end = r * args.sites, (r + 1) * args.sites
            S = end - begin
            seed = total + N
            thr = synthetic_thresholds(S, begin, seed)
            dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=False, max_allele=1, device=dev)
            dm.generate(seed, begin, thr, poc, 0)
            if args.layout == "packed":
                dm.pack(release_bytes=True)
            groups = device.Groups(dm, masks)
            bufs = {n: device.DeviceBuffer(dev, 8 * max(S, 1)) for n in ("dxy", "pi1", "pi2", "num", "den")}
            bufs["alt"] = device.DeviceBuffer(dev, 4 * 2 * max(S, 1))
            bufs["called"] = device.DeviceBuffer(dev, 4 * 2 * max(S, 1))
            sites = _abi.HudsonSites(None, bufs["dxy"].ptr, bufs["pi1"].ptr, bufs["pi2"].ptr, bufs["num"].ptr, bufs["den"].ptr, bufs["alt"].ptr, bufs["called"].ptr)
            totals = _abi.HudsonTotals()
            in_flight = 0

            def run_steps(k):
                nonlocal in_flight
                for _ in range(k):
                    _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), None))
                    in_flight += 1
                    if in_flight > args.pipeline_depth:
                        _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(totals)))
                        in_flight -= 1
                while in_flight > 0:
                    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(totals)))
                    in_flight -= 1
                _abi.check(lib.fmh_stream_synchronize(dev, None))

            run_steps(args.warmup)
            sync.wait()
            if r == 0:
                lib.fmh_timing_enable(max(1, args.timing_sample))
                lib.fmh_timing_reset()
                lib.fmh_timing_reset_reduce()
            sync.wait()
            t0 = time.perf_counter()
            run_steps(args.steps)
            mine_elapsed = time.perf_counter() - t0
            sync.wait()
            elapsed = time.perf_counter() - t0  # barrier + device synchronisation on both sides: the slowest rank's time
            reports[r] = {"rank": r, "device": dev, "slab": [begin, end], "sites": S, "elapsed_ms_per_step": mine_elapsed / args.steps * 1e3,
                          "elapsed": elapsed, "total": total, "seed": seed, "totals": totals, "comm": comm.describe(), "keep": (dm, groups, bufs)}
        except BaseException as exc:  # noqa: BLE001 - reported by the main thread; the peers are released through the communicators' abort
            errors[r] = exc
            shared["abort"] = True
            try:
                for c in shared["comms"] or []:
                    c.abort()
            except Exception:  # noqa: BLE001
                pass
            for b in (gate, sync):
                b.abort()

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    # ---- only now the first GPU call -----------------------------------------------------------------------------------------
    from ferromic_amd import _abi, sharding

    lib = _abi.load()
    transport_note = None
    try:
        shared["comms"] = sharding.Comm.init_all(devices)
    except Exception as exc:  # noqa: BLE001 - RCCL could not come up: the in-process rendezvous carries the sums, and the line says so
        transport_note = f"fmh_comm_init_all over RCCL failed ({type(exc).__name__}: {exc}); accumulators summed through the library's in-process host rendezvous"
        print("bench.py: " + transport_note, file=sys.stderr)
        try:
            with _abi.options(FMH_COMM_TRANSPORT="host"):
                shared["comms"] = sharding.Comm.init_all(devices)
        except Exception as exc2:  # noqa: BLE001
            shared["abort"] = True
            gate.wait()
            print(f"bench.py: no communicator could be created: {exc2}", file=sys.stderr)
            return 2
    gate.wait()
    for t in threads:
        t.join()
    failed = [(r, e) for r, e in enumerate(errors) if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if failed or any(rep is None for rep in reports):
        for r, e in failed:
            print(f"bench.py: rank {r} failed: {type(e).__name__}: {e}", file=sys.stderr)
        return 2
    kernel_ms, launches, kmin, kmax = C.c_double(), C.c_uint64(), C.c_double(), C.c_double()
    lib.fmh_timing_read(C.byref(kernel_ms), C.byref(launches))
    lib.fmh_timing_read_minmax(C.byref(kmin), C.byref(kmax))
    red_ms, red_n = C.c_double(), C.c_uint64()
    lib.fmh_timing_read_reduce(C.byref(red_ms), C.byref(red_n))
    lib.fmh_timing_enable(0)
    avg_kernel_s = kernel_ms.value / 1e3 / max(launches.value, 1)
    first = reports[0]
    elapsed = max(rep["elapsed"] for rep in reports)
    totals, total, S = first["totals"], first["total"], first["sites"]
    packed = args.layout == "packed"
    b_site = ((H + 7) // 8 if packed else H) + W_OUT_HUDSON
    achieved = b_site * S / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    comm_report = dict(first["comm"], source="fmh_comm_describe (libferromic_hip.so)")
    per_rank = []
    for rep in reports:
        # one process: the library's sweep / reduce timers are process-wide, so every rank carries the all-rank figures (equal slabs)
        per_rank.append({"rank": rep["rank"], "device": rep["device"], "slab": rep["slab"], "sites": rep["sites"], "kernel_ms_avg": avg_kernel_s * 1e3,
                         "kernel_ms_min": kmin.value, "kernel_ms_max": kmax.value, "kernel_launches_timed": int(launches.value),
                         "reduce_ms_avg": (red_ms.value / red_n.value) if red_n.value else None, "reduces_timed": int(red_n.value),
                         "elapsed_ms_per_step": rep["elapsed_ms_per_step"],
                         "comm": {k: rep["comm"].get(k) for k in ("transport", "world", "rank", "device")}})
    reduce_name = {"rccl": "RCCL (ncclAllReduce issued by libferromic_hip.so on each communicator's stream)",
                   "host": "the library's in-process host rendezvous", "local": "nothing (one rank)"}.get(str(comm_report.get("transport")), str(comm_report.get("transport")))
    result = {
        "metric": "variant-sites/sec (pi + Hudson FST)", "value": total * args.steps / elapsed, "unit": "sites/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u1" if packed else "u8", "data": "synthetic",
        "config": {"workload": f"C4 fused per-site pi + Hudson FST sweep: {total} sites x {H} haplotypes, {S} sites on each of {world} ranks, 2 populations, biallelic, "
                               "no missing data, matrix resident in HBM " + ("bit-packed" if packed else "as u8 rows"),
                   "total_sites": total, "sites_per_gpu": S, "haplotypes": H, "populations": 2, "seed": first["seed"], "layout": args.layout,
                   "devices": devices, "launcher": "none: one process, one host thread per rank, fmh_comm_init_all",
                   "parallelism": f"region-sharded x{world} from ONE process: one contiguous slab per rank, per-site tracks stay on the owning GPU, the 128 regional accumulators "
                                  f"summed by {reduce_name}, every host thread {args.pipeline_depth} steps ahead of its oldest uncollected reduce",
                   **({"transport_fallback": transport_note} if transport_note else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "none: not collected on the single-process route", "kernel_ms_avg": avg_kernel_s * 1e3,
                     "kernel_launches_timed": int(launches.value), "timing_sample": max(1, args.timing_sample), "kernel_sites_per_launch": S,
                     "algorithmic_bytes_per_site": b_site},
        "results": {"hudson_fst": totals.numerator_sum / totals.denominator_sum if totals.denominator_sum > 1e-12 else None,
                    "segregating_sites": [int(totals.pop[0].segregating_sites), int(totals.pop[1].segregating_sites)],
                    "sites_with_components": int(totals.sites_with_components), "dxy_uncallable_sites": int(totals.dxy_uncallable_sites),
                    "pi_sum": [totals.pop[0].pi_sum, totals.pop[1].pi_sum]},
        "comm": dict(comm_report, per_rank=per_rank, ranks_reporting=len(per_rank), kernel_ms_avg_min_over_ranks=avg_kernel_s * 1e3,
                     kernel_ms_avg_max_over_ranks=avg_kernel_s * 1e3, reduce_ms_avg_max_over_ranks=(red_ms.value / red_n.value) if red_n.value else None,
                     reduce_note="HIP events on the communicators' streams around ncclGroupStart .. ncclGroupEnd (process-wide average); null on the host rendezvous"),
    }
    for rep in reports[1:]:  # every rank must hold the same all-rank totals
        t = rep["totals"]
        assert int(t.pop[0].segregating_sites) == int(totals.pop[0].segregating_sites) and t.numerator_sum == totals.numerator_sum, "ranks disagree on the reduced totals"
    for rep in reports:
        rep["keep"] = None  # release the slabs before rank 0 sweeps the whole cohort
    result["parity_vs_n1"] = parity_vs_n1(args, {"scaling": args.scaling, "total_sites": total, "seed": first["seed"]}, totals, 0, world, devices[0], lib, masks, poc, N, H)
    print(json.dumps(result), flush=True)
    for c in shared["comms"]:
        c.close()
    return 3 if transport_note else 0


def main() -> int:
    args = parse_args()
    if args.single_process:
        return main_single_process(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: become the launcher (nothing has touched the GPU yet) and hand the job to N ranks
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        return subprocess.run(cmd, env=env).returncode

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_on_one_device:
        local_rank = 0
        if args.transport != "torch":
            raise SystemExit("--rehearse-on-one-device needs --transport torch (RCCL cannot place two ranks on one GPU)")
    args.gpus = world

    import torch

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback exists for the product path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (args.force_collective and args.transport == "torch"):
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from ferromic_amd import _abi, device, sharding

    lib = _abi.load()
    H = args.haplotypes
    if H % 2:
        raise SystemExit("haplotypes must be even (diploid samples)")
    N = H // 2
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)  # contiguous equal populations
    masks = np.stack([(poc == 0), (poc == 1)]).astype(np.uint8)
    sharded = world > 1 or args.force_collective
    comm = None
    transport_note = None
    if sharded and args.transport == "rccl":
        # the library's own communicator; creating it is a collective, so the ranks agree on the outcome before anyone uses it - if RCCL
        # cannot come up on this host the run falls back to torch.distributed for the reduce (slower, same numbers) and says so
        try:
            comm = sharding.Comm.from_torch_distributed(dist, local_rank) if world > 1 else sharding.Comm.single(local_rank)
            failure = None
        except Exception as exc:  # noqa: BLE001 - any failure takes the fallback
            comm, failure = None, f"{type(exc).__name__}: {exc}"
        if world > 1:
            ok = torch.tensor([0 if failure else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                    comm = None
                failure = failure or "another rank could not create its communicator"
        if comm is None:
            if dist is None:
                raise SystemExit(f"the RCCL communicator could not be created: {failure}")
            args.transport = "torch"
            transport_note = f"libferromic_hip's RCCL communicator could not be created ({failure}); accumulators summed through torch.distributed ({args.backend})"
            print("bench.py: " + transport_note, file=sys.stderr)

    if not sharded and not args.sync_steps:
        # one GPU, nothing to exchange: the same pipelined begin / end calls on a local one-rank communicator (no RCCL), so that N = 1 and
        # N > 1 run the same step structure - the next sweep is enqueued while the previous one's totals travel to the host
        comm = sharding.Comm.local(local_rank)

    user_stream = None
    if args.graph or args.explicit_stream:
        user_stream = torch.cuda.Stream(device=local_rank)  # a non-blocking stream of the caller's own (the NULL stream cannot be captured)
        if args.graph:
            _abi.set_option("FMH_GRAPH", 1)
    stream_ptr = C.c_void_p(user_stream.cuda_stream) if user_stream is not None else None

    def barrier():
        if dist is not None and world > 1:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if dist is None or world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_mode(scaling: str, primary: bool):
        """Builds this rank's slab for one scaling mode, times K steps, returns the measurements."""
        if scaling == "strong":
            total = args.sites
            begin, end = sharding.slab_for_rank(total, rank, world)
        else:
            total = args.sites * world
            begin, end = rank * args.sites, (rank + 1) * args.sites
        S = end - begin
        seed = total + N  # reference recipe: seed = variants + samples
        thr = synthetic_thresholds(S, begin, seed)
        dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=False, max_allele=1, device=local_rank)
        t0 = time.perf_counter()
        dm.generate(seed, begin, thr, poc, 0)
        gen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        if args.layout == "packed":
            dm.pack(release_bytes=False)  # the resident image fmh_matrix_create keeps: one bit per haplotype
        pack_s = time.perf_counter() - t0
        groups = device.Groups(dm, masks)
        # per-site tracks stay in HBM (56 B/site): counts + pi1, pi2, dxy, num, den
        bufs = {n: device.DeviceBuffer(local_rank, 8 * max(S, 1)) for n in ("dxy", "pi1", "pi2", "num", "den")}
        bufs["alt"] = device.DeviceBuffer(local_rank, 4 * 2 * max(S, 1))
        bufs["called"] = device.DeviceBuffer(local_rank, 4 * 2 * max(S, 1))
        sites = _abi.HudsonSites(None, bufs["dxy"].ptr, bufs["pi1"].ptr, bufs["pi2"].ptr, bufs["num"].ptr,
                                 bufs["den"].ptr, bufs["alt"].ptr, bufs["called"].ptr)
        state = {"totals": _abi.HudsonTotals(), "in_flight": 0}
        pipeline = sharding.HudsonTotalsPipeline(dist, "cuda" if args.backend == "nccl" else "cpu") if sharded and comm is None else None

        def sweep_local():
            local = _abi.HudsonTotals()
            _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(local), None))
            return local

        def step():
            if comm is not None:
                # enqueue this step's sweep + device-side reduce, then collect the previous step's region-wide totals
                _abi.check(lib.fmh_hudson_sweep_sharded_begin(comm._h, dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), stream_ptr))
                state["in_flight"] += 1
                if state["in_flight"] > args.pipeline_depth:
                    _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(state["totals"])))
                    state["in_flight"] -= 1
            elif pipeline is not None:
                pipeline.submit(sweep_local())
            else:
                state["totals"] = sweep_local()

        def fence():
            while comm is not None and state["in_flight"] > 0:
                _abi.check(lib.fmh_hudson_sweep_sharded_end(comm._h, C.byref(state["totals"])))
                state["in_flight"] -= 1
            if pipeline is not None:
                merged = pipeline.flush()
                if merged is not None:
                    state["totals"] = merged
            barrier()
            torch.cuda.synchronize()
            _abi.check(lib.fmh_stream_synchronize(local_rank, stream_ptr))

        # the same sweep over the u8 rows (one byte per haplotype, the layout the reference and north_star name), timed while the
        # matrix still holds them: its HBM fraction is reported next to the headline; then the rows are released
        u8_reference = None
        if args.layout == "packed":
            if primary and world == 1 and args.u8_reference_steps > 0 and S > 0:
                with _abi.options(FMH_LAYOUT="bytes"):  # fmh_set_option: sweeps take the u8 kernels while the rows exist
                    local = sweep_local()
                    lib.fmh_timing_enable(1)
                    lib.fmh_timing_reset()
                    for _ in range(args.u8_reference_steps):
                        local = sweep_local()
                    ms8, n8 = C.c_double(), C.c_uint64()
                    lib.fmh_timing_read(C.byref(ms8), C.byref(n8))
                    lib.fmh_timing_enable(0)
                k8 = ms8.value / 1e3 / max(n8.value, 1)
                u8_reference = {"kernel": "fmh::sweep_kernel<2, Summary|Hudson, no-missing, biallelic, u8>", "kernel_ms_avg": k8 * 1e3,
                                "algorithmic_bytes_per_site": H + W_OUT_HUDSON, "achieved": (H + W_OUT_HUDSON) * S / k8 / 1e9,
                                "frac": (H + W_OUT_HUDSON) * S / k8 / 1e9 / HBM_PEAK_GBS, "unit": "GB/s", "sites_per_s_kernel": S / k8,
                                "steps": args.u8_reference_steps,
                                "hudson_fst": local.numerator_sum / local.denominator_sum if local.denominator_sum > 1e-12 else None}
            dm.pack(release_bytes=True)  # from here on the matrix is what fmh_matrix_create leaves: planes only

        for _ in range(args.warmup):
            step()
        fence()
        lib.fmh_timing_enable(max(1, args.timing_sample))  # HIP events around every n-th sweep of the timed region
        lib.fmh_timing_reset()
        lib.fmh_timing_reset_reduce()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        kernel_ms, launches = C.c_double(), C.c_uint64()
        lib.fmh_timing_read(C.byref(kernel_ms), C.byref(launches))
        kmin, kmax = C.c_double(), C.c_double()
        lib.fmh_timing_read_minmax(C.byref(kmin), C.byref(kmax))
        red_ms, red_n = C.c_double(), C.c_uint64()
        lib.fmh_timing_read_reduce(C.byref(red_ms), C.byref(red_n))
        lib.fmh_timing_enable(0)
        local_elapsed = elapsed
        elapsed = max_over_ranks(elapsed)
        avg_kernel_s = (kernel_ms.value / 1e3) / max(launches.value, 1)
        # what THIS rank did, gathered on rank 0 below: its slab, its kernel times (HIP events on its launch stream), the device-side latency of
        # its grouped all-reduce (events on the communicator's stream), its own wall clock over the timed steps
        mine = {"rank": rank, "device": local_rank, "slab": [begin, end], "sites": S, "kernel_ms_avg": avg_kernel_s * 1e3, "kernel_ms_min": kmin.value,
                "kernel_ms_max": kmax.value, "kernel_launches_timed": int(launches.value),
                "reduce_ms_avg": (red_ms.value / red_n.value) if red_n.value else None, "reduces_timed": int(red_n.value),
                "elapsed_ms_per_step": local_elapsed / args.steps * 1e3}
        return {"scaling": scaling, "total_sites": total, "slab": (begin, end), "S": S, "seed": seed, "gen_s": gen_s, "pack_s": pack_s,
                "elapsed": elapsed, "avg_kernel_s": avg_kernel_s, "timed_launches": int(launches.value), "totals": state["totals"], "u8_reference": u8_reference,
                "bufs": bufs, "thr": thr, "dm": dm, "groups": groups, "mine": mine}

    first = run_mode(args.scaling, True)
    totals = first["totals"]
    S = first["S"]
    value = first["total_sites"] * args.steps / first["elapsed"]
    # ---- N > 1: the line carries its own evidence (who reduced, over what, what every rank did) --------------------------------
    comm_report = None
    per_rank = None
    if sharded:
        if comm is not None:
            comm_report = comm.describe()  # from the library: transport, world, rank, device, the librccl file it bound, ncclGetVersion
            comm_report["source"] = "fmh_comm_describe (libferromic_hip.so)"
        else:
            comm_report = {"transport": f"torch.distributed ({args.backend})", "world": world, "rank": rank, "device": local_rank, "rccl_library": None,
                           "source": "bench.py: the library's communicator is not in use" + (" (rehearsal on one device)" if args.rehearse_on_one_device else "")}
        mine = dict(first["mine"])
        mine["comm"] = {k: comm_report.get(k) for k in ("transport", "world", "rank", "device")}
        if dist is not None and world > 1:
            box = [None] * world
            dist.all_gather_object(box, mine)
            per_rank = box
        else:
            per_rank = [mine]
    secondary = None
    if world > 1 and not args.no_secondary:
        # release the primary cohort, then the other mode on the same ranks
        keep = {k: first[k] for k in ("scaling", "total_sites", "elapsed", "avg_kernel_s", "timed_launches", "S", "seed", "gen_s", "pack_s", "slab")}
        first.clear()
        first.update(keep)
        other = run_mode("weak" if args.scaling == "strong" else "strong", False)
        ot = other["totals"]
        secondary = {"scaling": other["scaling"], "value": other["total_sites"] * args.steps / other["elapsed"], "unit": "sites/s",
                     "ms_per_step": other["elapsed"] / args.steps * 1e3, "total_sites": other["total_sites"], "sites_per_gpu": other["S"],
                     "kernel_ms_avg": other["avg_kernel_s"] * 1e3,
                     "hudson_fst": ot.numerator_sum / ot.denominator_sum if ot.denominator_sum > 1e-12 else None}
        other.clear()

    # algorithmic bytes per site of the kernel that runs: the genotype row as it is resident in HBM (one bit per
    # haplotype when packed, one byte in the u8 layout) + the 56 B of per-site tracks it writes
    packed = args.layout == "packed"
    b_site_u8 = H + W_OUT_HUDSON
    b_site = ((H + 7) // 8 if packed else H) + W_OUT_HUDSON
    avg_kernel_s = first["avg_kernel_s"]
    achieved = b_site * S / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    traffic, traffic_source = recorded_traffic(S, H, args.layout)
    reduce_name = {"rccl": "RCCL (ncclAllReduce issued by libferromic_hip.so on its own stream)", "torch": f"torch.distributed ({args.backend})"}[args.transport]

    result = {
        "metric": "variant-sites/sec (pi + Hudson FST)",
        "value": value,
        "unit": "sites/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": first["elapsed"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": first["scaling"],
        "vs_baseline": None,
        "dtype": "u1" if packed else "u8",
        "data": "synthetic",
        "config": {
            "workload": f"C4 fused per-site pi + Hudson FST sweep: {first['total_sites']} sites x {H} haplotypes"
                        + (f", {S} sites on each of {world} GPUs" if world > 1 else "")
                        + ", 2 populations, biallelic, no missing data, matrix resident in HBM "
                        + ("bit-packed (1 bit per haplotype, the layout fmh_matrix_create keeps)" if packed else "as u8 rows"),
            "total_sites": first["total_sites"],
            "sites_per_gpu": S,
            "haplotypes": H,
            "populations": 2,
            "parallelism": (f"region-sharded x{world}: one contiguous slab per GPU, per-site tracks stay on the owning GPU, the 128 regional "
                            f"accumulators summed by {reduce_name}, the host {args.pipeline_depth} steps ahead of the oldest uncollected reduce") if sharded else
                           (f"one GPU, no collective; sweeps pipelined {args.pipeline_depth + 1} deep (fmh_hudson_sweep_sharded_begin / _end on a local one-rank communicator)"
                            + (", each step replayed from a captured hipGraph (FMH_GRAPH=1)" if args.graph else "") + (", explicit stream" if user_stream is not None else "")
                            if comm is not None else "one GPU, no collective, one blocking fmh_hudson_sweep per step"),
            "seed": first["seed"],
            "generate_s": first["gen_s"],
            "pack_s": first["pack_s"],
            "layout": args.layout,
            **({"transport_fallback": transport_note} if transport_note else {}),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "fmh::sweep_kernel<2, Summary|Hudson, no-missing, biallelic, " + ("packed>" if packed else "u8>"),
            "kernel_ms_avg": avg_kernel_s * 1e3,
            # HIP events bracket every n-th sweep of the timed region (--timing-sample): the average is over these launches
            "kernel_launches_timed": first.get("timed_launches"), "timing_sample": max(1, args.timing_sample),
            "kernel_sites_per_launch": S,
            "algorithmic_bytes_per_site": b_site,
            # the same sites/s priced at SURVEY 8(d)'s u8-layout figure (H + 56 B/site): what a sweep over u8 rows would
            # have to move per second to keep up - above the HBM peak when the packed layout does its job
            "u8_layout_bytes_per_site": b_site_u8,
            "u8_layout_equivalent_GBs": b_site_u8 * S / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0,
            # measured, same cohort, same process: the sweep over u8 rows (one byte per haplotype) before they were released
            "u8_layout_measured": first.get("u8_reference"),
        },
        "results": {
            "hudson_fst": totals.numerator_sum / totals.denominator_sum if totals.denominator_sum > 1e-12 else None,
            "segregating_sites": [int(totals.pop[0].segregating_sites), int(totals.pop[1].segregating_sites)],
            "sites_with_components": int(totals.sites_with_components),
            "dxy_uncallable_sites": int(totals.dxy_uncallable_sites),
            "pi_sum": [totals.pop[0].pi_sum, totals.pop[1].pi_sum],
        },
    }
    if secondary is not None:
        result["secondary"] = secondary
    if sharded:
        ks = [r["kernel_ms_avg"] for r in per_rank]
        reds = [r["reduce_ms_avg"] for r in per_rank if r["reduce_ms_avg"] is not None]
        result["comm"] = dict(comm_report, per_rank=per_rank, ranks_reporting=len(per_rank),
                              kernel_ms_avg_min_over_ranks=min(ks), kernel_ms_avg_max_over_ranks=max(ks),
                              reduce_ms_avg_max_over_ranks=max(reds) if reds else None,
                              reduce_note="HIP events on the communicator's stream around ncclGroupStart .. ncclGroupEnd of the two 512-byte all-reduces "
                                          "(every timed step); null when the library's RCCL communicator is not the transport")
        result["parity_vs_n1"] = parity_vs_n1(args, first, totals, rank, world, local_rank, lib, masks, poc, N, H)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        bufs = first["bufs"]

        def gpu_check(sample, cpu):
            got_alt = bufs["alt"].to_numpy(np.uint32, 2 * S).reshape(2, S)[:, :sample]
            ok_int = bool(np.array_equal(got_alt, cpu.alt))
            worst = 0.0
            for name in ("dxy", "pi1", "pi2", "num", "den"):
                g = bufs[name].to_numpy(np.float64, S)[:sample]
                c = getattr(cpu, name)
                same_nan = np.array_equal(np.isnan(g), np.isnan(c))
                with np.errstate(invalid="ignore", divide="ignore"):
                    rel = np.nanmax(np.abs(g - c) / np.maximum(np.abs(c), 1e-300)) if sample else 0.0
                worst = max(worst, float(rel) if same_nan else float("inf"))
            return {"alt_counts_bit_exact": ok_int, "per_site_f64_max_rel_err": worst}

        result["cpu_baseline"] = cpu_baseline(args, min(args.cpu_sample_sites, S), first["thr"], poc, first["seed"], gpu_check)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    # the line is out; a run whose RCCL communicator could not be created (summed through torch.distributed instead) is not a clean run
    if transport_note and sharding.Comm.init_stuck:
        sys.stdout.flush()
        os._exit(3)  # a helper thread is still blocked inside ncclCommInitRank: no orderly teardown is possible
    return 3 if transport_note else 0


if __name__ == "__main__":
    sys.exit(main())
