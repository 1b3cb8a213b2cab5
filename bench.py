#!/usr/bin/env python3
"""Benchmark of the RANSAC essential-matrix hot path on MI355X.

Metric (BASELINE.json): correspondence-evaluations/s = matches x hypotheses / wall time of one RANSAC
pass (sample -> eight-point fit -> SED scoring -> selection -> inlier mask), inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload: the configuration the north-star target is quoted on — 50 000 correspondences x 100 000
hypotheses per GPU (BASELINE.json configs[2]); with N GPUs every rank processes its own 100 000
hypotheses of one global Philox stream (weak scaling) and one 16-byte RCCL exchange per step picks the
global best model.  One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BYTES_PER_EVAL = 32.0          # xa, ya, xb, yb as f64, read once per hypothesis
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); ~6290 measured-achievable
THR, MIN_EXTRA = 1.5e-6, 10    # reference apps/config/config.yaml:6-9 (RMS aggregation)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--matches", type=int, default=50_000)
    ap.add_argument("--hypotheses", type=int, default=100_000, help="per GPU")
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay each step as one captured HIP graph (for launch-bound small workloads)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    return ap.parse_args()


def cpu_baseline(corr: np.ndarray, seed: int, target_seconds: float):
    """Times the CPU oracle (numpy eight-point fit + plain-C/OpenMP SED scoring + numpy selection) on a
    bounded number of hypotheses of the same workload.  The oracle is only the thing timed here."""
    from oracle import sfm_oracle as orc

    # build output (if any) goes to stderr: stdout carries exactly one JSON line
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle")], check=True, stdout=sys.stderr)
    lib = C.CDLL(os.path.join(REPO, "oracle", "libsfm_oracle.so"))
    lib.sfm_oracle_score.restype = C.c_int
    lib.sfm_oracle_score.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = corr.shape[0]
    # a 1-GPU box's CPU share is 16 cores even when the host exposes more hardware threads
    threads = int(os.environ.get("SFM_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))

    def run(h, h_begin):
        S = orc.philox_sample_table(seed, h_begin, h, n)
        E, deg, _ = orc.fit_hypotheses(corr, S)
        E = np.ascontiguousarray(E.reshape(h, 9))
        cnt = np.zeros(h, dtype=np.int32)
        s1 = np.zeros(h)
        s2 = np.zeros(h)
        used = lib.sfm_oracle_score(corr.ctypes.data, n, E.ctypes.data, S.ctypes.data, h, THR,
                                    cnt.ctypes.data, s1.ctypes.data, s2.ctypes.data, threads)
        best, err = orc.select_best(orc.aggregate(cnt, s1, s2, orc.RMS), cnt, MIN_EXTRA)
        return used, err

    run(64, 0)  # warm the caches / thread pool
    chunk, done, used = 10_000, 0, 1
    t0 = time.perf_counter()
    while True:
        used, _ = run(chunk, done)
        done += chunk
        elapsed = time.perf_counter() - t0
        if elapsed >= target_seconds or done >= 2_000_000:
            break
    return {
        "value": n * done / elapsed,
        "unit": "correspondence-evals/s",
        "cores": used,
        "kind": "port",
        "sample": f"{done} hypotheses x {n} matches of the same workload in {elapsed:.1f} s: numpy eight-point "
                  f"fit (1 thread) + C/OpenMP SED scoring ({used} threads) + numpy selection",
    }


def measured_valu(n, h):
    """VALU-issue evidence for the dominant kernel from the committed PMC summary (the bound that actually holds:
    the data set is cache-resident) — None for other workload sizes."""
    path = os.path.join(REPO, "profiles", "score_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("matches") != n or rec.get("hypotheses") != h:
        return None
    return rec.get("valu")


def measured_traffic(n, h):
    """HBM bytes per score-kernel launch from the committed rocprofv3 PMC passes (profiles/), if they were
    taken on this workload: 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B) + WRITE_SIZE."""
    path = os.path.join(REPO, "profiles", "score_traffic.json")
    try:
        rec = json.load(open(path))
    except OSError:
        return None
    if rec.get("matches") != n or rec.get("hypotheses") != h:
        return None
    return (2.0 * rec["fetch_size_kib"] + rec["write_size_kib"]) * 1024.0


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # SFM_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal (collective staged through the host)
    backend = os.environ.get("SFM_DIST_BACKEND", "nccl")
    local_device = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_device))
        else:
            dist.init_process_group(backend)

    from structure_from_motion_amd import device, distributed, synthetic
    from structure_from_motion_amd._native import AGG_RMS

    device.require_gpu()
    n, h = args.matches, args.hypotheses
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    engine = distributed.ShardedRansac(corr, h, THR, MIN_EXTRA, AGG_RMS, rank, world)

    # the dominant kernel (SED scoring) is bracketed with events on the stream it is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    score = device.score_sed
    state = {"i": -1}

    def timed_score(*a, **k):
        i = state["i"]
        if i >= 0:
            ev[i][0].record()
        out = score(*a, **k)
        if i >= 0:
            ev[i][1].record()
        return out

    device.score_sed = timed_score

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        engine.step(args.seed + w)
    if args.graph:
        # per-kernel events cannot be recorded inside a replayed graph: time the score kernel on a few
        # eager steps first, then capture
        for s in range(min(args.steps, 10)):
            state["i"] = s
            engine.step(args.seed + 500 + s)
        state["i"] = -1
        ev = ev[:min(args.steps, 10)]
        engine.capture()
        engine.step(args.seed + 999)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        state["i"] = -1 if args.graph else s
        engine.step(args.seed + 1000 + s)
    barrier()
    elapsed = time.perf_counter() - t0
    state["i"] = -1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.cpu()[0])

    best_h, err, E, sample, mask = engine.outcome()
    score_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    evals_per_gpu = float(n) * float(h)
    value = evals_per_gpu * world * args.steps / elapsed
    achieved = evals_per_gpu * BYTES_PER_EVAL / (score_ms * 1e-3) / 1e9 if score_ms > 0 else None

    if rank == 0:
        line = {
            "metric": "correspondence-evals/s (matches x hypotheses)",
            "value": value,
            "unit": "correspondence-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic two-view, {n} correspondences x {h} RANSAC hypotheses per GPU "
                            "(BASELINE.json configs[2]; the configuration the north-star target is quoted on)",
                "matches": n, "hypotheses_per_gpu": h, "global_hypotheses": h * world,
                "sed_inlier_threshold": THR, "min_num_extra_inliers": MIN_EXTRA, "aggregation": "rms",
                "sampler": "philox", "parallelism": f"hypothesis-shard x{world}",
                "launch": "hip-graph" if args.graph else "eager",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "score_sed_filtered_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": measured_traffic(n, h),
                "traffic_unit": "bytes per launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/)",
                "algorithmic_bytes": evals_per_gpu * BYTES_PER_EVAL,
                "valu_issue": measured_valu(n, h),
                "kernel_variant": os.environ.get("SFM_SCORE_KERNEL", "filtered"),
                "kernel_ms": score_ms,
                "note": "achieved = 32 B/eval x matches x hypotheses / avg score-kernel time (HIP events); "
                        "the correspondence set (1.6 MB f64 + 0.8 MB f32 copy) is L2-resident, so physical HBM "
                        "traffic is far lower and the kernel is VALU-issue bound (DESIGN.md)",
            },
            "result": {"best_h": best_h, "error": err, "inliers": int((mask != 0).sum()) if mask is not None else 0},
        }
        if not args.no_cpu_baseline and world == 1:
            corr_host = corr.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(corr_host, args.seed, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
