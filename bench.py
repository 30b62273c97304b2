#!/usr/bin/env python3
"""bench.py — variant-sites/sec of the fused "pi + Hudson FST" sweep (BASELINE.json metric).

A step = one pass of the hot path (fmh_hudson_sweep: per-site allele counts for both populations,
per-site pi1/pi2/Dxy/num/den, regional accumulators back on the host) over the synthetic cohort
already resident in HBM.  Default workload = BASELINE config C4 on ONE GPU: 10 M sites x 5 000
haplotypes, 2 populations (50 GB of uint8 genotypes).  With --gpus N every rank owns its own
genomic slab of the same size (region sharding, weak scaling) and the regional accumulators are
combined with one RCCL all-reduce per step (torch.distributed, backend nccl == RCCL).

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
W_OUT_HUDSON = 56      # bytes of per-site results written: 2*(4+4) counts + 5*8 (pi1, pi2, dxy, num, den); SURVEY.md 8(d)


def synthetic_thresholds(sites: int, first_site: int, total_sites: int, seed: int, sigma: float = 0.05) -> np.ndarray:
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


def cpu_baseline(args, thr_full: np.ndarray, poc: np.ndarray, seed: int, gpu_check):
    """The reference algorithm restated in C (oracle/dense_oracle.c), site ranges over host threads,
    on a bounded sample of the same cohort (rows [0, sample) are bit-identical to the GPU's)."""
    from oracle import dense as D

    cores = os.cpu_count() or 1
    sample = min(args.cpu_sample_sites, args.sites)
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
        "cores": cores,
        "kind": "port",
        "sample": f"first {sample} sites x {H} haplotypes of the same cohort, best of 3 passes "
                  f"({best:.3f} s each), C restatement of stats.rs:1367-1470 + 1554-1623 + 3179-3278",
        "parity_vs_gpu": parity,
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=int, default=10_000_000, help="sites per GPU")
    ap.add_argument("--haplotypes", type=int, default=5000)
    ap.add_argument("--cpu-sample-sites", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse the N > 1 path)")
    ap.add_argument("--layout", choices=["packed", "bytes"], default="packed",
                    help="resident layout of the cohort: bit-packed planes (what fmh_matrix_create keeps for alleles 0..3) or the u8 rows")
    ap.add_argument("--u8-reference-steps", type=int, default=5,
                    help="with --layout packed: before the u8 rows are released, time this many sweeps over them (the layout "
                         "north_star names) and report their HBM fraction next to the headline; 0 skips it")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the all-reduce path even with one rank (measures the software cost of the collective step on a one-GPU box)")
    ap.add_argument("--rehearse-on-one-device", action="store_true",
                    help="all ranks share cuda:0 (with --backend gloo): exercises the sharded code path on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_on_one_device:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}",
                  file=sys.stderr)
            return 2
        args.gpus = world

    import torch

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback exists for the product path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from ferromic_amd import _abi, device

    lib = _abi.load()
    S, H = args.sites, args.haplotypes
    if H % 2:
        raise SystemExit("haplotypes must be even (diploid samples)")
    N = H // 2
    seed = (S * world) + N  # reference recipe: seed = variants + samples
    first_site = rank * S
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)  # contiguous equal populations

    thr = synthetic_thresholds(S, first_site, S * world, seed)
    dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=False, max_allele=1, device=local_rank)
    t0 = time.perf_counter()
    dm.generate(seed, first_site, thr, poc, 0)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    if args.layout == "packed":
        dm.pack(release_bytes=False)  # the resident image fmh_matrix_create keeps: one bit per haplotype
    pack_s = time.perf_counter() - t0
    masks = np.stack([(poc == 0), (poc == 1)]).astype(np.uint8)
    groups = device.Groups(dm, masks)

    # per-site tracks stay in HBM (56 B/site): counts + pi1, pi2, dxy, num, den
    bufs = {n: device.DeviceBuffer(local_rank, 8 * S) for n in ("dxy", "pi1", "pi2", "num", "den")}
    bufs["alt"] = device.DeviceBuffer(local_rank, 4 * 2 * S)
    bufs["called"] = device.DeviceBuffer(local_rank, 4 * 2 * S)
    sites = _abi.HudsonSites(None, bufs["dxy"].ptr, bufs["pi1"].ptr, bufs["pi2"].ptr, bufs["num"].ptr,
                             bufs["den"].ptr, bufs["alt"].ptr, bufs["called"].ptr)
    from ferromic_amd import sharding

    state = {"totals": _abi.HudsonTotals()}
    # N > 1: the 20 regional accumulators of a step are summed over the ranks by one RCCL all-reduce that overlaps the
    # next step's sweep (its own stream); fence() collects the last one, so every step's reduce ends inside the timed region
    pipeline = sharding.HudsonTotalsPipeline(dist, "cuda") if dist is not None else None

    def step():
        local = _abi.HudsonTotals()
        _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(local), None))
        if pipeline is not None:
            pipeline.submit(local)
        else:
            state["totals"] = local

    def fence():
        if pipeline is not None:
            merged = pipeline.flush()
            if merged is not None:
                state["totals"] = merged
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        _abi.check(lib.fmh_stream_synchronize(local_rank, None))

    # the same sweep over the u8 rows (one byte per haplotype, the layout the reference and north_star name), timed while the
    # matrix still holds them: its HBM fraction is reported next to the headline; then the rows are released
    u8_reference = None
    if args.layout == "packed":
        if args.u8_reference_steps > 0:
            os.environ["FMH_LAYOUT"] = "bytes"  # read per call by the library: sweeps take the u8 kernels while the rows exist
            local = _abi.HudsonTotals()
            _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(local), None))
            lib.fmh_timing_enable(1)
            lib.fmh_timing_reset()
            for _ in range(args.u8_reference_steps):
                _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(local), None))
            ms8, n8 = C.c_double(), C.c_uint64()
            lib.fmh_timing_read(C.byref(ms8), C.byref(n8))
            lib.fmh_timing_enable(0)
            del os.environ["FMH_LAYOUT"]
            k8 = ms8.value / 1e3 / max(n8.value, 1)
            u8_reference = {"kernel": "fmh::sweep_kernel<2, Summary|Hudson, no-missing, biallelic, u8>", "kernel_ms_avg": k8 * 1e3,
                            "algorithmic_bytes_per_site": H + W_OUT_HUDSON, "achieved": (H + W_OUT_HUDSON) * S / k8 / 1e9,
                            "frac": (H + W_OUT_HUDSON) * S / k8 / 1e9 / HBM_PEAK_GBS, "unit": "GB/s", "sites_per_s_kernel": S / k8,
                            "steps": args.u8_reference_steps,
                            "hudson_fst": local.numerator_sum / local.denominator_sum if local.denominator_sum > 1e-12 else None}
        dm.pack(release_bytes=True)  # from here on the matrix is what fmh_matrix_create leaves: planes only

    for _ in range(args.warmup):
        step()
    lib.fmh_timing_enable(1)
    lib.fmh_timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = C.c_double(), C.c_uint64()
    lib.fmh_timing_read(C.byref(kernel_ms), C.byref(launches))
    lib.fmh_timing_enable(0)

    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    totals = state["totals"]
    total_sites = S * world
    value = total_sites * args.steps / elapsed
    # algorithmic bytes per site of the kernel that runs: the genotype row as it is resident in HBM (one bit per
    # haplotype when packed, one byte in the u8 layout) + the 56 B of per-site tracks it writes
    packed = args.layout == "packed"
    b_site_u8 = H + W_OUT_HUDSON
    b_site = ((H + 7) // 8 if packed else H) + W_OUT_HUDSON
    avg_kernel_s = (kernel_ms.value / 1e3) / max(launches.value, 1)
    achieved = b_site * S / avg_kernel_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"{S}x{H}:{args.layout}")
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    result = {
        "metric": "variant-sites/sec (pi + Hudson FST)",
        "value": value,
        "unit": "sites/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u1" if packed else "u8",
        "data": "synthetic",
        "config": {
            "workload": f"C4 fused per-site pi + Hudson FST sweep: {S} sites x {H} haplotypes per GPU, 2 populations, "
                        "biallelic, no missing data, matrix resident in HBM "
                        + ("bit-packed (1 bit per haplotype, the layout fmh_matrix_create keeps)" if packed else "as u8 rows"),
            "sites_per_gpu": S,
            "haplotypes": H,
            "populations": 2,
            "parallelism": f"region-sharded x{world} (one slab per GPU, one {'RCCL' if args.backend == 'nccl' else args.backend} all-reduce of 20 accumulators per step, overlapped with the next step's sweep)",
            "seed": seed,
            "generate_s": gen_s,
            "pack_s": pack_s,
            "layout": args.layout,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "fmh::sweep_kernel<2, Summary|Hudson, no-missing, biallelic, " + ("packed>" if packed else "u8>"),
            "kernel_ms_avg": avg_kernel_s * 1e3,
            "algorithmic_bytes_per_site": b_site,
            # the same sites/s priced at SURVEY 8(d)'s u8-layout figure (H + 56 B/site): what a sweep over u8 rows would
            # have to move per second to keep up - above the HBM peak when the packed layout does its job
            "u8_layout_bytes_per_site": b_site_u8,
            "u8_layout_equivalent_GBs": b_site_u8 * S / avg_kernel_s / 1e9,
            # measured, same cohort, same process: the sweep over u8 rows (one byte per haplotype) before they were released
            "u8_layout_measured": u8_reference,
        },
        "results": {
            "hudson_fst": totals.numerator_sum / totals.denominator_sum if totals.denominator_sum > 1e-12 else None,
            "segregating_sites": [int(totals.pop[0].segregating_sites), int(totals.pop[1].segregating_sites)],
            "pi_sum": [totals.pop[0].pi_sum, totals.pop[1].pi_sum],
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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

        result["cpu_baseline"] = cpu_baseline(args, thr, poc, seed, gpu_check)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
