#!/usr/bin/env python3
"""bench.py — the polymorphism-scan hot path on N GPUs of one node.

A *step* is one pass of the scan kernel over one device-resident batch of
synthetic alignments (BASELINE.json configs[1] shape: N=50 sequences x L=10 kb,
theta = 0.01, non-coding; `--batch` of them per GPU, distinct, generated on the
device by the deterministic generator, so the working set is several times the
256 MiB Infinity Cache).  Multi-GPU = the reference's `--dir` batch mode:
independent alignments sharded over ranks (weak scaling: every rank scans its
own `--batch` alignments per step), no collective on the data path, ONE RCCL
all-gather of the fixed-size result records at the end, inside the timed region.

Prints one JSON line on rank 0 (contract in the task statement) carrying
`roofline` (algorithmic bytes = N*L per alignment, kernel time from HIP events
recorded on the launch stream) and `cpu_baseline` (the C oracle, one core, on a
bounded sample of the same workload; N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32768, help="alignments per GPU per step (16.6 GB of bases at the default shape)")
    ap.add_argument("--rows", type=int, default=50)
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--theta", type=float, default=0.01)
    ap.add_argument("--stride-align", type=int, default=128)
    ap.add_argument("--variant", type=int, default=0, help="0 = LDS-DMA nt (default), 1 = register staging, 2 = LDS-DMA default cache policy")
    ap.add_argument("--coding", action="store_true", help="scan as in-frame CDS (config 4 shape with --sites 9000)")
    ap.add_argument("--pops", type=int, default=0,
                    help="config 3 shape: every matrix holds rows*pops sequences, scanned as `pops` interleaved row subsets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def cpu_baseline(args, params):
    """Time the C oracle (scalar, one core) on a bounded sample of the workload."""
    from oracle import c_oracle
    from polyfasta_amd import _lib, scan
    n_mat = 128
    stride = scan.row_stride_for(args.sites, args.stride_align)
    sp = _lib.SynthParamsC(seed=2026, first_aln=0, n_aln=n_mat, n_rows=args.rows, n_sites=args.sites,
                           row_stride=stride, seg_thresh=params.seg_thresh, tri_thresh=params.tri_thresh,
                           n_units=len(params.units), coding=0)
    bases = np.zeros(n_mat * args.rows * stride, dtype=np.uint8)
    _lib.check(_lib.load().pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp),
                                               _lib.ptr(np.ascontiguousarray(params.cdf)),
                                               _lib.ptr(np.ascontiguousarray(params.units))))
    c_oracle.scan_many(bases, 4, args.rows, args.sites, stride)  # warm
    done, t0 = 0, time.perf_counter()
    while True:
        s, h = c_oracle.scan_many(bases, n_mat, args.rows, args.sites, stride)
        done += n_mat
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds:
            break
    # the same scalar code on several host threads at once (ctypes releases the GIL): the GPU box
    # gives one GPU's share of the host, 16 cores
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(16, os.cpu_count() or 1))
    per = n_mat // T if n_mat >= T else 1
    slices = [bases[k * per * args.rows * stride:(k + 1) * per * args.rows * stride] for k in range(min(T, n_mat))]

    def worker(sl):
        n, t_end = 0, time.perf_counter() + min(args.cpu_seconds, 5.0)
        while time.perf_counter() < t_end:
            c_oracle.scan_many(sl, per, args.rows, args.sites, stride)
            n += per
        return n
    tm = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(slices)) as ex:
        done_mt = sum(ex.map(worker, slices))
    dt_mt = time.perf_counter() - tm
    multi = {"value": done_mt / dt_mt, "unit": "alignments/s", "cores": len(slices), "kind": "port",
             "sample": f"the same scalar scan on {len(slices)} host threads for {dt_mt:.1f} s"}
    return {"value": done / dt, "unit": "alignments/s", "cores": 1, "kind": "port", "multi_thread": multi,
            "sample": f"{n_mat} distinct synthetic alignments ({args.rows}x{args.sites}, same generator) scanned "
                      f"repeatedly by oracle/pfa_oracle.c for {dt:.1f} s on 1 of {os.cpu_count()} host cores",
            "host_cores": os.cpu_count()}, (s, h)


def measured_traffic(cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this exact
    workload (profiles/*_pmc_summary.json, made by tools/gpu_prof.sh +
    tools/summarize_prof.py: separate --pmc passes, gfx950 correction), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        w = j.get("workload", {})
        if all(w.get(k) == cfg.get(k) for k in ("rows", "sites", "batch_per_gpu", "row_stride", "kernel_variant")) \
                and all((k in w.get("workload", "")) == (k in cfg.get("workload", "")) for k in ("coding", "populations")):
            best = j.get("traffic_bytes_per_launch")
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    dist = None
    if world > 1 or os.environ.get("PFA_FORCE_DIST"):   # PFA_FORCE_DIST: rehearse the RCCL path on one GPU
        import torch
        import torch.distributed as dist
        from polyfasta_amd.dist import stdout_to_stderr
        with stdout_to_stderr():         # RCCL's version banner must not join the one JSON line on stdout
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
    from polyfasta_amd import _lib, scan, synth

    lib = _lib.load()
    _lib.init(local_rank)

    N, L, B = args.rows, args.sites, args.batch
    K = max(args.pops, 1)
    NM = N * K                               # rows of each matrix
    params = synth.make_params(NM, args.theta)
    stride = scan.row_stride_for(L, args.stride_align)
    nbytes = B * NM * stride
    dev = scan.DeviceBuffer(nbytes)
    sp = _lib.SynthParamsC(seed=2026, first_aln=rank * B, n_aln=B, n_rows=NM, n_sites=L, row_stride=stride,
                           seg_thresh=params.seg_thresh, tri_thresh=params.tri_thresh, n_units=len(params.units),
                           coding=int(args.coding))
    _lib.check(lib.pfa_synth_fill_dev(dev.ptr, ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(params.cdf)),
                                      _lib.ptr(np.ascontiguousarray(params.units))))
    descs = np.zeros(B * K, dtype=_lib.DESC_DTYPE)
    descs["base_off"] = np.repeat(np.arange(B, dtype=np.uint64) * np.uint64(NM * stride), K)
    descs["n_rows"], descs["n_sites"], descs["row_stride"], descs["mat_rows"] = N, L, stride, NM
    descs["row_idx_off"] = _lib.PFA_NO_ROW_IDX
    descs["flags"] = _lib.PFA_FLAG_CDS if args.coding else 0
    if os.environ.get("PFA_BENCH_MIX_ONE_CDS"):   # experiment: coding-capable kernel on non-coding units
        descs["flags"][0] = _lib.PFA_FLAG_CDS
    row_idx = None
    if args.pops:                            # population k = rows k, k+K, k+2K, ... (headers pop{1+i%K}_ind{i})
        row_idx = np.concatenate([np.arange(k, NM, K, dtype=np.uint32) for k in range(K)])
        descs["row_idx_off"] = np.tile(np.arange(K, dtype=np.uint32) * N, B)
    table = scan.finalize_layout(descs, row_idx, nbytes)
    plan = scan.Plan(table)

    def sync_all():
        _lib.check(lib.pfa_sync())
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def gather_records(out):
        """the only collective of the job: the fixed-size result records of every rank, to every rank"""
        import torch
        local = torch.from_numpy(out.results.view(np.uint8).reshape(-1).copy()).cuda()
        allr = torch.empty(world * local.numel(), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(allr, local)
        torch.cuda.synchronize()
        return allr.cpu().numpy().view(_lib.RESULT_DTYPE)

    for w in range(args.warmup):
        plan.run(dev, variant=args.variant)
        out = plan.fetch()
        if dist is not None and w == 0:
            gather_records(out)              # RCCL sets up its channels on the first collective of a shape
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(dev, variant=args.variant)
    t_enq = time.perf_counter() - t0
    _lib.check(lib.pfa_sync())
    t_run = time.perf_counter() - t0
    out = plan.fetch()                       # D2H of the integer records (part of the job)
    t_fetch = time.perf_counter() - t0 - t_run
    gathered = None
    if dist is not None:
        gathered = gather_records(out)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kms = plan.kernel_ms(args.steps)
    k_avg_ms = float(kms.mean()) if len(kms) else float("nan")
    alg_bytes = float(B) * NM * L           # every base of every population is read once
    achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0

    if rank == 0:
        res = out.results
        total_aln = world * B * K * args.steps   # scan units (alignment x population) = CSV rows
        line = {
            "metric": "alignments/sec", "value": total_aln / dt, "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "site_columns_per_sec": total_aln * L / dt,
            "config": {"workload": f"BASELINE configs[1] shape as a device-resident batch: {B} distinct alignments "
                                   f"per GPU per step, N={N} x L={L}, theta={args.theta}, "
                                   f"{'coding (--cds)' if args.coding else 'non-coding'}"
                                   + (f", {K} populations of {N} rows per matrix" if args.pops else ""),
                       "rows": N, "sites": L, "batch_per_gpu": B, "row_stride": stride,
                       "parallelism": f"independent alignments sharded over {world} rank(s); one RCCL all-gather of result records"
                                      if world > 1 else "single GPU",
                       "kernel_variant": {0: "lds_dma_nt", 1: "register_staged", 2: "lds_dma_default_policy"}[args.variant],
                       "tiles_per_chunk": plan.tiles_per_chunk, "lds_bytes_per_wave": plan.lds_bytes,
                       "rows_per_chunk": plan.rows_per_chunk},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "pfa::scan_tall_kernel" if plan.rows_per_chunk else "pfa::scan_kernel", "kernel_ms_avg": k_avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_of_measured_copy_6290": achieved / 6290.0,
                         # best plain streaming read measured on this pool (tools/ubench/read_stream.hip)
                         "frac_of_measured_read_6860": achieved / 6860.0},
            "timing_ms": {"enqueue": t_enq * 1e3, "until_kernels_done": t_run * 1e3, "fetch": t_fetch * 1e3,
                          "kernel_ms_min": float(kms.min()), "kernel_ms_max": float(kms.max())},
            "check": {"mean_seg_sites": float(res["seg_sites"].mean()),
                      "sum_H": int(res["H"].sum()),
                      "gathered_records": None if gathered is None else int(len(gathered))},
        }
        line["roofline"]["traffic"] = measured_traffic(line["config"])
        if world == 1 and not args.coding and not args.pops and not args.no_cpu_baseline:   # (not under the profiler)
            # BASELINE configs[1] literally: ONE alignment per launch (latency-bound: 500 KB of bases)
            one = scan.Plan(scan.finalize_layout(descs[:1].copy(), None, nbytes))
            for _ in range(20):
                one.run(dev)
            _lib.check(lib.pfa_sync())
            t1 = time.perf_counter()
            for _ in range(200):
                one.run(dev)
                one.fetch()
            per = (time.perf_counter() - t1) / 200
            line["single_alignment"] = {"us_per_scan_and_fetch": per * 1e6, "kernel_us": float(one.kernel_ms(50).mean()) * 1e3,
                                        "note": "one N x L alignment per launch, bases resident, records fetched after every scan"}
            one.close()
        if world == 1 and not args.no_cpu_baseline and not args.coding and not args.pops:
            cb, (s_cpu, h_cpu) = cpu_baseline(args, params)
            line["cpu_baseline"] = cb
            # the CPU sample is the first 128 alignments of rank 0's batch: integers must agree
            line["check"]["cpu_sample_matches_gpu"] = bool(
                (res["seg_sites"][:len(s_cpu)] == s_cpu).all() and (res["H"][:len(h_cpu)] == h_cpu).all())
        print(json.dumps(line), flush=True)

    plan.close()
    dev.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
