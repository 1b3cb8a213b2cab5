#!/usr/bin/env python3
"""Benchmark of the RANSAC essential-matrix hot path on MI355X.

Metric (BASELINE.json): correspondence-evaluations/s = matches x hypotheses / wall time of one RANSAC
pass (sample -> eight-point fit -> SED scoring -> selection -> inlier mask), inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload.  N = 1: the configuration the north-star target is quoted on — 50 000 correspondences x 100 000
hypotheses (BASELINE.json configs[2]).  N > 1: BASELINE.json configs[3] — 1 000 000 hypotheses of ONE global
Philox stream split over the N ranks (1 000 000 // N each, 125 000 at N = 8), one 40-byte all-gather per step to
pick the global best model; `--hypotheses H` overrides with H per rank (weak scaling).  One JSON line on stdout
(rank 0).

Roofline.  The correspondence set (1.6 MB) is L2-resident, so HBM bytes do not bound the scoring kernel (`hbm_algorithmic`
— 32 B per evaluation / kernel time — exceeds the peak; `hbm_physical` — FETCH_SIZE x 2 + WRITE_SIZE per launch — is ~5 % of it).
What bounds it is instruction issue at the clock the power manager leaves (1.57 GHz under this load: profiles/r04/README.md):
`roofline.frac` = (VALU wave-instructions of one launch, from the committed rocprofv3 PMC passes, priced at the spec issue rates of
MI355X_MICROARCH.md: 2 cycles per wave64 instruction, 4 for fp64, 8 of vector issue per 16-bit MFMA) / (1024 SIMDs x 2.4 GHz x the
kernel's duration measured live with HIP events around that kernel) — the issue-slot UTILISATION of the executed instruction
stream, not a bound (a fatter kernel scores higher).  The bound beside it is `roofline.fp64_floor`: the run's own true inliers and
sample points (sum of cnt + 8 per hypothesis) must go through the 43-instruction fp64 routine — that work at full rate and full
lane utilisation, as a time and as a fraction of the measured kernel.
"""
import argparse
import ctypes as C
import json
import os
import random
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BYTES_PER_EVAL = 32.0          # xa, ya, xb, yb as f64, read once per hypothesis
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); ~6290 measured-achievable
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMD-32; max clock (MI355X_MICROARCH.md chip table)
CYC_VALU, CYC_F64 = 2.0, 4.0   # spec issue cycles per wave64 instruction: v_fma_f32 2 (SIMD-32); fp64 at half rate
MEASURED_CYC_VALU, MEASURED_CYC_F64, MEASURED_CYC_MFMA = 3.3, 4.16, 32.0   # profiles/r04/micro_power_by_piece.txt (see roofline())
CYC_MFMA = 8.0                 # vector-issue cycles a 32x32x16 16-bit MFMA holds (MI355X_MICROARCH.md cycle constants); it runs 32 on the matrix pipe
FP64_PER_EVAL = 43             # fp64-rate wave-instructions of one evaluation of the exact tier (sfm::sed_inlier + the two sums as compiled:
                               # 20 mul, 16 add, 4 fma, 1 rcp, 2 compares — counted in the disassembly; -ffp-contract=off).  Until
                               # round 4's last kernel change: 59 + 4 (sfm::sed_value with its two IEEE divisions of 13 each)
THR, MIN_EXTRA = 1.5e-6, 10    # reference apps/config/config.yaml:6-9 (RMS aggregation)
C4_TOTAL = 1_000_000           # BASELINE.json configs[3]
COUNTERS = os.path.join(REPO, "profiles", "score_traffic.json")


def scoring_kernel_name(variant, n, h):
    """Which scoring kernel sfm_score_sed launches for a single pair of this size (the library's own rule)."""
    if variant == "exact":
        return "score_sed_exact_kernel"
    from structure_from_motion_amd import _native

    return "score_sed_matrix_kernel" if _native.load().sfm_score_kernel_choice(n, h, 1) == 2 else "score_sed_filtered_kernel"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--matches", type=int, default=50_000)
    ap.add_argument("--hypotheses", type=int, default=None,
                    help="per GPU (default: 100 000 at --gpus 1, 1 000 000 // N at --gpus N)")
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the kernel variants and the through-the-API timings (profiling runs)")
    ap.add_argument("--graph", action="store_true",
                    help="replay each step as one captured HIP graph (for launch-bound small workloads)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    return ap.parse_args()


def cpu_baseline(corr: np.ndarray, seed: int, target_seconds: float):
    """Times the CPU oracle (numpy eight-point fit + plain-C/OpenMP SED scoring + numpy selection) on a
    bounded number of hypotheses of the same workload, on all the cores of the box's CPU share: the fit runs over
    contiguous hypothesis blocks in a pool of `cores` worker processes (BASELINE.md section 3 ii), the scoring loop on
    `cores` OpenMP threads.  The oracle is only the thing timed here.  Both legs are also reported on their own:
    `value` is the whole stage the metric names, `score_leg_value` the H x N loop alone."""
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
    legs = {"fit": 0.0, "score": 0.0, "select": 0.0}

    with orc.FitPool(corr, threads) as pool:   # worker start-up (interpreter + numpy import) is not timed
        def run(h, h_begin, record=True):
            t0 = time.perf_counter()
            S, E, deg, _ = pool.sample_and_fit(seed, h_begin, h)
            E = np.ascontiguousarray(E.reshape(h, 9))
            t1 = time.perf_counter()
            cnt = np.zeros(h, dtype=np.int32)
            s1 = np.zeros(h)
            s2 = np.zeros(h)
            used = lib.sfm_oracle_score(corr.ctypes.data, n, E.ctypes.data, S.ctypes.data, h, THR,
                                        cnt.ctypes.data, s1.ctypes.data, s2.ctypes.data, threads)
            t2 = time.perf_counter()
            best, err = orc.select_best(orc.aggregate(cnt, s1, s2, orc.RMS), cnt, MIN_EXTRA)
            t3 = time.perf_counter()
            if record:
                legs["fit"] += t1 - t0
                legs["score"] += t2 - t1
                legs["select"] += t3 - t2
            return used, err

        run(64 * threads, 0, record=False)  # warm the caches / thread pool / workers
        chunk, done, used = 40_000, 0, 1
        t0 = time.perf_counter()
        while True:
            used, _ = run(chunk, done)
            done += chunk
            elapsed = time.perf_counter() - t0
            if elapsed >= target_seconds or done >= 4_000_000:
                break
    # (i) of BASELINE.md section 3: the same port in ONE process on one thread, ~2 s of work
    t0 = time.perf_counter()
    done1, chunk1 = 0, 1_000
    while True:
        S = orc.philox_sample_table(seed, done1, chunk1, n)
        E, _, _ = orc.fit_hypotheses(corr, S)
        E = np.ascontiguousarray(E.reshape(chunk1, 9))
        cnt, s1, s2 = np.zeros(chunk1, dtype=np.int32), np.zeros(chunk1), np.zeros(chunk1)
        lib.sfm_oracle_score(corr.ctypes.data, n, E.ctypes.data, S.ctypes.data, chunk1, THR, cnt.ctypes.data,
                             s1.ctypes.data, s2.ctypes.data, 1)
        orc.select_best(orc.aggregate(cnt, s1, s2, orc.RMS), cnt, MIN_EXTRA)
        done1 += chunk1
        elapsed1 = time.perf_counter() - t0
        if elapsed1 >= 2.0:
            break
    single = {"value": n * done1 / elapsed1, "cores": 1,
              "sample": f"{done1} hypotheses x {n} matches in {elapsed1:.1f} s: one process, one thread (numpy fit, C scoring loop)"}
    return {
        "value": n * done / elapsed,
        "unit": "correspondence-evals/s",
        "cores": used,
        "single_thread": single,
        "kind": "port",
        "score_leg_value": n * done / legs["score"] if legs["score"] > 0 else None,
        "seconds": {k: round(v, 3) for k, v in legs.items()},
        "sample": f"{done} hypotheses x {n} matches of the same workload in {elapsed:.1f} s: Philox table + numpy "
                  f"eight-point fit over hypothesis blocks in {pool.workers} worker processes ({legs['fit']:.1f} s) + "
                  f"C/OpenMP SED scoring ({used} threads, {legs['score']:.1f} s) + numpy selection; "
                  f"score_leg_value = the H x N loop alone",
    }


def load_counters(n, h):
    """The committed rocprofv3 PMC record of the scoring kernel (tools/collect_counters.sh) and whether it was taken on
    the kernel sources this process runs: (record or None, stale flag)."""
    from structure_from_motion_amd import build

    try:
        rec = json.load(open(COUNTERS))
    except (OSError, ValueError):
        return None, None
    if rec.get("matches") != n or "counters" not in rec or not rec.get("hypotheses"):
        return None, None
    if rec["hypotheses"] != h:
        # same point set, another hypothesis count (the per-rank shard of a multi-GPU run): every per-launch counter of
        # the scoring kernel is proportional to the number of hypotheses — scale, and say so
        factor = float(h) / float(rec["hypotheses"])
        rec = dict(rec, counters={k: (v * factor if k != "profiled_kernel_ms" and k != "GRBM_GUI_ACTIVE" else v)
                                  for k, v in rec["counters"].items()}, scaled_from_hypotheses=rec["hypotheses"])
        rec["counters"].pop("profiled_kernel_ms", None)   # the clock estimate does not carry over
    return rec, rec.get("source_sha") != build.score_source_sha()


def fp64_floor(exact_evals, kernel_ms):
    """The part of the scoring kernel no filter can remove: every true inlier (and every sample point) must go through the
    fp64 routine.  exact_evals x FP64_PER_EVAL wave-instructions / 64 lanes at the spec fp64 rate (4 cycles per wave64
    instruction) on 1024 SIMDs at 2.4 GHz — a floor in ms, and the share of the measured kernel time it explains."""
    if not exact_evals or not kernel_ms:
        return None
    cycles = float(exact_evals) / 64.0 * FP64_PER_EVAL * CYC_F64
    floor_ms = cycles / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
    return {"exact_evaluations": int(exact_evals), "fp64_insts_per_evaluation": FP64_PER_EVAL, "cycles_per_inst": CYC_F64,
            "floor_ms": floor_ms, "frac_of_kernel": floor_ms / kernel_ms,
            "note": "true inliers + sample points of this run (sum of cnt + 8 per hypothesis) x fp64 wave-instructions per "
                    "evaluation x 4 cycles / (64 lanes x 1024 SIMDs x 2.4 GHz): what the kernel would take if it did nothing "
                    "but the unavoidable fp64 evaluations at full lane utilisation"}


def roofline(n, h, kernel_ms, call_ms, variant, exact_evals=None):
    evals = float(n) * float(h)
    if not kernel_ms or kernel_ms <= 0:   # nothing was timed (--steps 0)
        return {"bound": "valu-issue", "kernel_ms": None, "achieved": None, "peak": SIMDS * CLOCK_GHZ, "frac": None,
                "traffic": None, "unit": "G wave-instruction issue cycles/s"}
    seconds = kernel_ms * 1e-3
    algorithmic = evals * BYTES_PER_EVAL
    rec, stale = load_counters(n, h)
    out = {
        "bound": "valu-issue",
        "kernel": scoring_kernel_name(variant, n, h),
        "kernel_ms": kernel_ms,
        "score_call_ms": call_ms,
        "achieved": None, "peak": SIMDS * CLOCK_GHZ, "unit": "G wave-instruction issue cycles/s", "frac": None,
        "traffic": None,
        "hbm_algorithmic": {"bytes_per_launch": algorithmic, "achieved_GBs": algorithmic / seconds / 1e9,
                            "peak_GBs": HBM_PEAK_GBS, "frac": algorithmic / seconds / 1e9 / HBM_PEAK_GBS,
                            "note": "32 B/eval x matches x hypotheses / kernel time; > 1 because the 1.6 MB "
                                    "correspondence set is L2-resident: not a bound for this kernel"},
        "hbm_physical": None,
        "counters_stale": stale,
        "fp64_floor": fp64_floor(exact_evals, kernel_ms),
        "note": "frac = issue-slot utilisation of the EXECUTED instruction stream (not a bound: a fatter kernel scores "
                "higher): (VALU wave-instructions per launch x spec issue cycles: 2 per wave64 instruction, 4 per fp64 "
                "one, 8 of vector issue per 16-bit MFMA) / (1024 SIMDs x 2.4 GHz x kernel time by HIP events around the "
                "kernel); counters from profiles/score_traffic.json (rocprofv3 --pmc, separate passes).  fp64_floor is the "
                "bound: the irreducible fp64 work of this run's true inliers",
    }
    if rec is None or variant != "filtered":
        return out
    if rec.get("kernel_short") and rec["kernel_short"] != out["kernel"]:
        out["counters_stale"] = True   # the committed counters are another kernel's
        return out
    c = rec["counters"]
    f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                      "SQ_INSTS_VALU_TRANS_F64"))
    valu = c["SQ_INSTS_VALU"]
    mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
    mfma_insts = c.get("SQ_INSTS_MFMA", 0.0)
    cycles = (valu - f64) * CYC_VALU + f64 * CYC_F64 + mfma_insts * CYC_MFMA
    out["achieved"] = cycles / seconds / 1e9
    out["frac"] = out["achieved"] / out["peak"]
    out["valu"] = {"insts_per_launch": valu, "fp64_insts": f64, "per_64_evals": valu / (evals / 64.0),
                   "issue_cycles_per_simd": cycles / SIMDS, "mfma_mops_f32": mfma or None, "mfma_insts": mfma_insts or None,
                   "matrix_pipe_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (seconds * CLOCK_GHZ * 1e9)
                                             if c.get("SQ_VALU_MFMA_BUSY_CYCLES") else None)}
    # the same instruction stream at the rates this part was MEASURED to sustain (tools/micro/matrix_step_rates.hip, power mode,
    # 4 waves per SIMD: profiles/r04/micro_power_by_piece.txt): 3.3 cycles per plain vector instruction, 4.16 per fp64 one, and
    # the 32 matrix-pipe cycles of a 16-bit MFMA in full — matrix and vector work of a SIMD add up on this part, they do not overlap
    measured = (valu - f64) * MEASURED_CYC_VALU + f64 * MEASURED_CYC_F64 + mfma_insts * MEASURED_CYC_MFMA
    out["issue_at_measured_rates"] = {
        "ms_at_2p4_GHz": measured / SIMDS / (CLOCK_GHZ * 1e9) * 1e3, "frac_of_kernel": measured / SIMDS / (CLOCK_GHZ * 1e9) / seconds,
        "cycles_per_inst": {"valu": MEASURED_CYC_VALU, "fp64": MEASURED_CYC_F64, "mfma_16bit_32x32x16": MEASURED_CYC_MFMA},
        "note": "what the counted instructions take at the issue rates measured on this part, at the nominal clock: the rest of "
                "the kernel time is the clock the power manager leaves (1.35 kW of a 1.4 kW cap under this load) and idle slots"}
    if "GRBM_GUI_ACTIVE" in c and c.get("profiled_kernel_ms"):
        clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["profiled_kernel_ms"] * 1e-3) / 1e9
        out["valu"]["clock_GHz_under_profiler"] = clock
        out["frac_at_measured_clock"] = out["frac"] * CLOCK_GHZ / clock if clock > 0 else None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["traffic"] = traffic
        out["hbm_physical"] = {"bytes_per_launch": traffic, "achieved_GBs": traffic / seconds / 1e9,
                               "frac": traffic / seconds / 1e9 / HBM_PEAK_GBS,
                               "note": "rocprofv3 PMC: 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B) + "
                                       "WRITE_SIZE, KiB, per launch"}
    out["counters_from"] = {k: rec.get(k) for k in ("git", "source_sha", "abi", "collected", "scaled_from_hypotheses")}
    return out


def api_timings(device_mod):
    """Wall time of the drop-in call itself — lib.epipolar.epipolar_ransac.estimate_essential_mat_with_ransac as
    reference apps/sfm.py:110-119 calls it (Feature lists in, (E, inlier pairs) out) — at BASELINE configs[0]'s scale
    (300 x 2000), configs[1] (5000 x 10000) and configs[2] (50000 x 100000), after one warm-up call each.  Not part
    of `value`."""
    from lib.common.feature import Feature
    from lib.epipolar.eight_point import create_trivial_matches
    from lib.epipolar.epipolar_ransac import estimate_essential_mat_with_ransac
    from lib.ransac.ransac import ErrorAggregationMethod
    from structure_from_motion_amd import synthetic

    out = {}
    saved = {k: os.environ.get(k) for k in ("SFM_SAMPLER", "SFM_SEED")}
    try:
        for name, n, h, sampler in (("c1_300x2000_pyshuffle", 300, 2000, "pyshuffle"),
                                    ("c2_5000x10000_pyshuffle", 5000, 10000, "pyshuffle"),
                                    ("c2_5000x10000_philox", 5000, 10000, "philox"),
                                    ("c2_5000x10000_auto", 5000, 10000, "auto"),
                                    ("c3_50000x100000_auto", 50000, 100000, "auto")):
            pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
            fa = [Feature(x=float(x), y=float(y)) for x, y in pa]
            fb = [Feature(x=float(x), y=float(y)) for x, y in pb]
            matches = create_trivial_matches(n)
            os.environ["SFM_SAMPLER"] = sampler   # "auto" = the explicit size switch (pyshuffle up to 1e7 draws)
            os.environ["SFM_SEED"] = "5"
            times = []
            for rep in range(4):
                random.seed(5)
                t0 = time.perf_counter()
                E, pairs = estimate_essential_mat_with_ransac(
                    K, features_a=fa, features_b=fb, matches=matches, sed_inlier_threshold=THR,
                    error_aggregation_method=ErrorAggregationMethod.RMS, min_num_extra_inliers=MIN_EXTRA,
                    max_iterations=h)
                times.append((time.perf_counter() - t0) * 1e3)
            out[name] = {"ms": min(times[1:]), "first_call_ms": times[0], "inliers": len(pairs)}
        # the DEFAULT call at C3: exact random.shuffle replay — O(H x N) Mersenne-Twister draws on the host, like the
        # reference's own sampling (ransac.py:62); one call, no warm-up repeat (it takes seconds)
        os.environ["SFM_SAMPLER"] = "pyshuffle"
        random.seed(5)
        t0 = time.perf_counter()
        E, pairs = estimate_essential_mat_with_ransac(
            K, features_a=fa, features_b=fb, matches=matches, sed_inlier_threshold=THR,
            error_aggregation_method=ErrorAggregationMethod.RMS, min_num_extra_inliers=MIN_EXTRA, max_iterations=h)
        ms = (time.perf_counter() - t0) * 1e3
        out["c3_50000x100000_pyshuffle"] = {"ms": ms, "calls": 1, "inliers": len(pairs),
                                            "ns_per_element_and_iteration": ms * 1e6 / (float(n) * float(h)),
                                            "note": "default sampler: the reference's cumulative random.shuffle stream, "
                                                    "replayed bit-exactly on the host (csrc/pyshuffle.cpp)"}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def other_configs(torch, device, distributed, synthetic, agg):
    """The other BASELINE.json configurations on the same line (outside `value`): configs[1] C2 = 5 000 x 10 000 on one
    GPU (the lean small pass, us per pass), configs[3]'s per-rank share C4 = 125 000 hypotheses x 50 000 matches (one
    of eight ranks' local pass: sample -> fit -> score -> select, no exchange), configs[4] C5 = 256 pairs x 10 000 x
    2 000 through the batched device pipeline (E + pose vote + triangulation).  Reference call sites: apps/sfm.py:110-119,
    133-138, 181-186."""
    from structure_from_motion_amd import batched

    def wall(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            fn(r)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def scoring_kernel_ms(fn):
        """One more invocation of `fn` with its scoring kernel bracketed by HIP events that travel in the call's launch options
        (sfm_score_options.timing_before / _after): recorded inside the library, immediately around that kernel on its launch
        stream.  `fn(rep, options)` passes the options on to its engine."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        torch.cuda.synchronize()
        times = []
        for rep in range(3):
            fn(rep, device.default_score_options().with_timing(a, b))
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        return float(np.median(times))

    def config_roofline(name, n, h, batch, kernel_ms, exact_evals):
        """Roofline block of one of the other configurations: the fp64 floor from the run's own inlier count, and — when
        profiles/<name>_counters.json (tools/collect_config_counters.sh) holds the kernel's PMC record — the issue-slot
        utilisation and HBM traffic priced like the headline's."""
        block = {"kernel_ms": kernel_ms, "fp64_floor": fp64_floor(exact_evals, kernel_ms),
                 "hbm_algorithmic_frac": (float(n) * h * batch * BYTES_PER_EVAL / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                          if kernel_ms else None)}
        try:
            rec = json.load(open(os.path.join(REPO, "profiles", name + "_counters.json")))
        except (OSError, ValueError):
            return block
        c = rec.get("counters", {})
        if rec.get("matches") != n or rec.get("hypotheses") != h or rec.get("batch", 1) != batch or "SQ_INSTS_VALU" not in c:
            return block
        from structure_from_motion_amd import build

        f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                          "SQ_INSTS_VALU_TRANS_F64"))
        cycles = (c["SQ_INSTS_VALU"] - f64) * CYC_VALU + f64 * CYC_F64 + c.get("SQ_INSTS_MFMA", 0.0) * CYC_MFMA
        block.update({"bound": "valu-issue", "frac": cycles / SIMDS / (CLOCK_GHZ * 1e9) / (kernel_ms * 1e-3) if kernel_ms else None,
                      "valu_insts_per_launch": c["SQ_INSTS_VALU"], "fp64_insts": f64, "mfma_insts": c.get("SQ_INSTS_MFMA"),
                      "traffic": ((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None),
                      "counters_stale": rec.get("source_sha") != build.score_source_sha(),
                      "counters_from": {k: rec.get(k) for k in ("git", "source_sha", "collected", "kernel")}})
        return block

    lib = _native_lib()
    out = {}
    # C2: median over 10 groups of 20 passes
    n, h = 5_000, 10_000
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    eng = distributed.ShardedRansac(corr, h, THR, MIN_EXTRA, agg)
    wall(lambda r: eng.step(50 + r), 20)
    groups = [wall(lambda r, g=g: eng.step(1000 + 20 * g + r), 20) for g in range(10)]
    us = float(np.median(groups)) * 1e6
    res = eng.outcome()
    def timed(step):
        def call(r, options):
            eng.score_options = options
            try:
                step(2000 + r)
            finally:
                eng.score_options = None
        return call

    k_ms = scoring_kernel_ms(timed(eng.step))
    exact = int(eng.ws.cnt.sum().item()) + 8 * h
    out["c2_5000x10000"] = {"pass_us": us, "evals_per_s": n * h / us * 1e6, "passes": 200, "best_h": res.best_h,
                            "inliers": int((res.mask != 0).sum()),
                            "kernel": "score_sed_filtered_kernel (fused small pass, one hypothesis per wave)",
                            "roofline": config_roofline("c2", n, h, 1, k_ms, exact)}
    # C4 per-rank share
    n, total, world = 50_000, C4_TOTAL, 8
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    eng = distributed.ShardedRansac(corr, None, THR, MIN_EXTRA, agg, rank=0, world=world, total_hypotheses=total)
    wall(lambda r: eng.step_local(50 + r), 3)
    sec = wall(lambda r: eng.step_local(1000 + r), 10)
    k_ms = scoring_kernel_ms(timed(eng.step_local))
    exact = int(eng.ws.cnt.sum().item()) + 8 * eng.h
    out["c4_shard_125000x50000"] = {"ms": sec * 1e3, "evals_per_s": n * eng.h / sec,
                                    "note": "rank 0 of 8: local pass over its 125 000 hypotheses of the 1 M stream",
                                    "kernel": ("score_sed_matrix_kernel" if lib.sfm_score_kernel_choice(n, eng.h, 1) == 2
                                               else "score_sed_filtered_kernel"),
                                    "roofline": config_roofline("c4", n, eng.h, 1, k_ms, exact)}
    del eng
    # C5
    B, n, h = 256, 10_000, 2_000
    base = [synthetic.two_view_scene(n, seed=300 + b, outlier_fraction=0.25) for b in range(16)]
    pix_a = device.to_device(np.stack([base[b % 16][0] for b in range(B)]))
    pix_b = device.to_device(np.stack([base[b % 16][1] for b in range(B)]))
    pipe = batched.TwoViewBatch(B, n, h)
    run = lambda r, options=None: pipe.run(pix_a, pix_b, base[0][2], seed=70 + 1000 * r, thr=THR, min_extra=MIN_EXTRA,
                                           aggregation=agg, score_options=options)
    wall(run, 2)
    sec = wall(run, 5)
    ok = sum(r.status == batched.OK for r in pipe.results())
    k_ms = scoring_kernel_ms(lambda r, options: run(100 + r, options))
    exact = int(pipe.ws.cnt.sum().item()) + 8 * h * B
    out["c5_256x10000x2000"] = {"batch_ms": sec * 1e3, "evals_per_s": B * n * h / sec, "pairs_per_s": B / sec,
                                "pairs_ok": ok, "note": "E estimation + pose vote + triangulation, one enqueue",
                                "kernel": ("score_sed_matrix_kernel" if lib.sfm_score_kernel_choice(n, h, B) == 2
                                           else "score_sed_filtered_kernel"),
                                "roofline": config_roofline("c5", n, h, B, k_ms, exact)}
    del pipe
    torch.cuda.empty_cache()
    return out


_PUBLIC = os.path.join("..", "..", "include", "sfm_hip.h")
WIDENED_SOURCES = {   # (translation unit, headers it includes) a widened row's kernels are compiled from: what its counter record is stamped with
    "f1_match_20000x20000_ncc9": (("sfm_match.hip",), ("sfm_common.h", "sfm_math.h", _PUBLIC)),
    "f1_match_600x600_ncc9": (("sfm_match.hip",), ("sfm_common.h", "sfm_math.h", _PUBLIC)),
    "f2_harris_vga": (("sfm_harris.hip",), ("sfm_common.h", _PUBLIC)),
    "f2_harris_1080p": (("sfm_harris.hip",), ("sfm_common.h", _PUBLIC)),
    "f4_refine_50000": (("sfm_refine.hip",), ("sfm_common.h", "sfm_math.h", "sfm_fit.h", _PUBLIC)),
    "pose_tail_c5": (("sfm_pose.hip",), ("sfm_common.h", "sfm_math.h", _PUBLIC)),
}


def widened_sha(name):
    from structure_from_motion_amd import build

    sources, headers = WIDENED_SOURCES[name]
    return build.source_sha(sources, headers=headers)
FP64_PER_DLT = 420   # fp64 wave-instructions of one DLT solve (4 x 4 null vector: Householder QR + inverse iteration) as compiled


def widened_configs(torch):
    """The widened rows of SURVEY.md §8f and the pose tail of configs[4] on the bench line, outside `value`:
    brute-force matcher (reference matching.py:36-118: 20 000 x 20 000 NCC-9 and the demo's 600 x 600), Harris detector
    (harris_detector.py:11-113: VGA and 1080p), local optimisation (refine_kernel, N = 50 000), cheirality + vote + triangulation
    at C5's sizes (eight_point.py:181-242,449-488; triangulation.py:42-62).  Per row: the kernels, their time by HIP events around
    the library call (inputs resident; the Harris rows time the whole call, which uploads the image and reads the corners back),
    a floor where the work is countable, and — when profiles/<row>_counters.json (tools/collect_config_counters.sh) exists — the
    counter-based issue-slot utilisation and HBM traffic, flagged stale when the kernels' sources changed since."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import widened_workloads as ww
    from structure_from_motion_amd import build

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps, (time.perf_counter() - t0) / reps * 1e3   # (stream time, wall time) in ms

    def counters(name, kernel_ms):
        try:
            rec = json.load(open(os.path.join(REPO, "profiles", name + "_counters.json")))
        except (OSError, ValueError):
            return None
        out = {"counters_stale": rec.get("source_sha") != widened_sha(name),
               "counters_from": {k: rec.get(k) for k in ("git", "source_sha", "collected")}, "kernels": {}}
        for kernel, c in rec.get("kernels", {}).items():
            f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                              "SQ_INSTS_VALU_TRANS_F64"))
            entry = {"launches_per_pass": c.get("launches_per_pass"), "trace_avg_us": c.get("trace_avg_us")}
            if "SQ_INSTS_VALU" in c and c.get("trace_avg_us"):
                cycles = (c["SQ_INSTS_VALU"] - f64) * CYC_VALU + f64 * CYC_F64 + c.get("SQ_INSTS_MFMA", 0.0) * CYC_MFMA
                entry.update({"valu_insts_per_launch": c["SQ_INSTS_VALU"], "fp64_insts": f64,
                              "issue_frac": cycles / SIMDS / (CLOCK_GHZ * 1e9) / (c["trace_avg_us"] * 1e-6)})
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c and c.get("trace_avg_us"):
                traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                entry.update({"traffic": traffic, "hbm_frac": traffic / (c["trace_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS})
            out["kernels"][kernel] = entry
        return out

    def fp64_floor_ms(wave_insts):
        return wave_insts * CYC_F64 / SIMDS / (CLOCK_GHZ * 1e9) * 1e3

    out = {}
    for name, build_workload in ww.BUILDERS.items():
        run, info = build_workload()
        reps = {"f1_match_20000x20000_ncc9": 5, "f1_match_600x600_ncc9": 50, "f2_harris_vga": 20, "f2_harris_1080p": 5,
                "f4_refine_50000": 20, "pose_tail_c5": 5}[name]
        gpu_ms, wall_ms = timed(run, reps)
        row = {"kernel": info.get("kernel"), "kernel_ms": gpu_ms}
        if name.startswith("f1_"):
            floor = fp64_floor_ms(info["pairs"] * info["fp64_insts_per_pair"] / 64.0)
            row.update({"features": [info["features_a"], info["features_b"]], "window_elements": info["window_elements"],
                        "pairs_per_s": info["pairs"] / (gpu_ms * 1e-3),
                        "fp64_floor": {"floor_ms": floor, "frac_of_kernel": floor / gpu_ms,
                                       "note": "K separately rounded multiplies + K adds per pair (no FMA: bit-exact window sums) at 4 "
                                               "cycles per fp64 wave-instruction, 64 lanes x 1024 SIMDs x 2.4 GHz"}})
        elif name.startswith("f2_"):
            row.update({"call_ms": wall_ms, "kernel_ms": None, "pixels": info["pixels"], "megapixels_per_s": info["pixels"] / wall_ms / 1e3,
                        "corners": info["corners"],
                        "note": "whole detect_harris_corners call: image upload (uint8, widened on the device), Sobel x / y, cornerness, "
                                "12 NMS rounds + check, compaction, pruning to the leaders, their read-back, the Feature list "
                                "(host-synchronous: wall clock)"})
        elif name.startswith("f4_"):
            row.update({"matches": info["matches"], "rounds": info["rounds"], "inliers_in": info["inliers_in"],
                        "note": "one block per pair runs the whole refit + re-score loop: latency-bound by construction"})
        else:
            stages = {}
            for stage, fn in info["stages"].items():
                ms, _ = timed(fn, reps)
                stages[stage] = {"kernel_ms": ms}
                if stage in info["dlt_solves"]:
                    floor = fp64_floor_ms(info["dlt_solves"][stage] / 64.0 * FP64_PER_DLT)
                    stages[stage]["fp64_floor"] = {"dlt_solves": info["dlt_solves"][stage], "fp64_insts_per_solve": FP64_PER_DLT,
                                                   "floor_ms": floor, "frac_of_kernel": floor / ms}
            row.update({"pairs": info["pairs"], "matches": info["matches"], "inliers": info["inliers"], "stages": stages,
                        "kernel": "cheirality_batched_kernel + pose_vote_kernel + triangulate_selected_kernel"})
        pmc = counters(name, gpu_ms)
        if pmc is not None:
            row["counters"] = pmc
        out[name] = row
        del run, info
        torch.cuda.empty_cache()
    return out


def _native_lib():
    from structure_from_motion_amd import _native

    return _native.load()


def self_launch(args) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and relay their output and exit code.  Nothing in this
    process has touched the GPU yet, and it never will: rank 0 of the children prints the one JSON line."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # stdout carries exactly one JSON line: anything else the ranks' libraries write there (the gloo transport announces
    # its connections on stdout) is passed on to stderr
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for text in child.stdout:
        is_line = text.startswith('{"metric"')
        (sys.stdout if is_line else sys.stderr).write(text)
        (sys.stdout if is_line else sys.stderr).flush()
    return child.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SFM_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal (collective staged through the host)
    backend = os.environ.get("SFM_DIST_BACKEND", "nccl")
    local_device = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_device))
        else:
            dist.init_process_group(backend)

    from structure_from_motion_amd import _native, build, device, distributed, synthetic
    from structure_from_motion_amd._native import AGG_RMS

    device.require_gpu()
    n = args.matches
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    if args.hypotheses is not None:      # explicit per-GPU count: weak scaling
        engine = distributed.ShardedRansac(corr, args.hypotheses, THR, MIN_EXTRA, AGG_RMS, rank, world)
        scaling, partition = "weak", f"{args.hypotheses} hypotheses per GPU"
    elif world == 1:                     # BASELINE configs[2]
        engine = distributed.ShardedRansac(corr, 100_000, THR, MIN_EXTRA, AGG_RMS, rank, world)
        scaling, partition = "weak", "BASELINE.json configs[2]: 100000 hypotheses on one GPU"
    else:                                # BASELINE configs[3]: one global stream of 1 M hypotheses, split
        engine = distributed.ShardedRansac(corr, None, THR, MIN_EXTRA, AGG_RMS, rank, world,
                                           total_hypotheses=C4_TOTAL)
        scaling, partition = "strong", (f"BASELINE.json configs[3]: {C4_TOTAL} hypotheses split over {world} GPUs "
                                        f"({C4_TOTAL // world} per rank)")
    h = engine.h
    total_h = engine.total

    # the dominant kernel is bracketed by HIP events recorded inside sfm_score_sed, immediately around that kernel on
    # its launch stream; a second pair brackets the whole scoring call (workspace preparation + ordering pre-pass)
    def make_events(count):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
        for a, b in evs:   # torch creates the hipEvent at the first record
            a.record()
            b.record()
        return evs

    n_events = max(args.steps, 3)
    kernel_ev, call_ev = make_events(n_events), make_events(n_events)
    score = device.score_sed
    state = {"i": -1, "calls": 0}

    def timed_score(*a, **k):
        # brackets the whole scoring call (workspace preparation + ordering pre-pass + kernel); the lean small pass
        # (n <= 8192, h <= 32768) has no separate scoring call and never comes through here
        i = state["i"]
        if i >= 0:
            call_ev[i][0].record()
        out = score(*a, **k)
        if i >= 0:
            call_ev[i][1].record()
            state["calls"] += 1
        return out

    device.score_sed = timed_score

    def timed_step(i, seed):
        """One engine step with the scoring kernel of that step bracketed by kernel_ev[i]: the events travel in the step's
        launch options and are recorded inside the library immediately around the scoring kernel, on its launch stream — by
        sfm_score_sed_ex and by the scoring launch of sfm_ransac_pass_small / _large alike."""
        state["i"] = i
        engine.score_options = device.default_score_options().with_timing(*kernel_ev[i]) if i >= 0 else None
        try:
            engine.step(seed)
        finally:
            engine.score_options = None
        state["i"] = -1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        engine.step(args.seed + w)

    variants = None
    if world == 1 and not args.no_extras and not args.graph:
        # the same pass with each scoring kernel, 3 steps each: the all-fp64 kernel (every evaluation in fp64) and
        # the default two-tier kernel (conservative fp32 reject filter + the same fp64 routine for the survivors)
        variants = {}
        # and the two-tier kernel with tier 1 on the matrix pipe (the default of this workload); the kernel is chosen through
        # the library's process-wide launch options (sfm_score_set_default_options), which are put back afterwards
        saved_options = device.default_score_options()
        for name, env, kernel in (("exact_f64", "exact", None), ("filtered", "filtered", "filtered"), ("matrix", "filtered", "matrix")):
            os.environ["SFM_SCORE_KERNEL"] = env
            device.set_default_score_options(saved_options if kernel is None else device.ScoreOptions(kernel=kernel))
            engine.step(args.seed + 77)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(3):
                timed_step(s, args.seed + 80 + s)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3
            variants[name] = {"ms_per_step": wall * 1e3, "value": float(n) * h / wall,
                              "kernel_ms": float(np.mean([a.elapsed_time(b) for a, b in kernel_ev[:3]]))}
        os.environ.pop("SFM_SCORE_KERNEL", None)
        device.set_default_score_options(saved_options)

    variant = os.environ.get("SFM_SCORE_KERNEL", "filtered")
    if args.graph:
        # per-kernel events cannot be recorded inside a replayed graph: time the score kernel on a few
        # eager steps first, then capture
        timed = min(args.steps, 10)
        state["calls"] = 0
        for s in range(timed):
            timed_step(s, args.seed + 500 + s)
        torch.cuda.synchronize()
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ev[:timed]])) if timed else None
        call_ms = float(np.mean([a.elapsed_time(b) for a, b in call_ev[:timed]])) if state["calls"] else None
        engine.capture()
        engine.step(args.seed + 999)
    state["calls"] = 0
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        timed_step(-1 if args.graph else s, args.seed + 1000 + s)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.cpu()[0])

    out = engine.outcome()
    exact_evals = int(engine.ws.cnt.sum().item()) + 8 * h   # true inliers + sample points of the last step: all must be scored in fp64
    if not args.graph:
        ev = kernel_ev[:args.steps]
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else None
        # no separate scoring call in a lean small pass: nothing re-recorded the call events
        call_ms = float(np.mean([a.elapsed_time(b) for a, b in call_ev[:args.steps]])) if state["calls"] else None
    value = float(n) * float(total_h) * args.steps / elapsed

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
            "scaling": scaling,
            "vs_baseline": None,
            # every counted decision and every summed error is fp64; the reject filter in front of it only ever discards pairs it
            # proves to be outliers (fp16 / bf16 operands with fp32 accumulation on the matrix pipe, or fp32 on the VALU)
            "dtype": ("f64" if variant != "filtered" else
                      "f64 (+ conservative f16/bf16 matrix-pipe reject filter)"
                      if scoring_kernel_name(variant, n, h) == "score_sed_matrix_kernel" else
                      "f64 (+ conservative f32 reject filter)"),
            "data": "synthetic",
            "config": {
                "workload": f"synthetic two-view, {n} correspondences x {total_h} RANSAC hypotheses "
                            f"({partition})",
                "matches": n, "hypotheses_per_gpu": h, "global_hypotheses": total_h,
                "sed_inlier_threshold": THR, "min_num_extra_inliers": MIN_EXTRA, "aggregation": "rms",
                "sampler": "philox", "parallelism": f"hypothesis-shard x{world}",
                "exchange": "none" if world == 1 else "one all_gather of a 40-byte select record per rank per step",
                "launch": "hip-graph" if args.graph else "eager",
                "library": {"abi": _native.ABI_VERSION, "score_source_sha": build.score_source_sha()},
            },
            "roofline": roofline(n, h, kernel_ms, call_ms, variant, exact_evals),
            "result": {"best_h": out.best_h, "error": out.error,
                       "inliers": int((out.mask != 0).sum()) if out.mask is not None else 0,
                       "n_flagged": out.n_flagged},
        }
        if variants is not None:
            line["variants"] = variants
        if world == 1 and not args.no_extras:
            device.score_sed = score
            line["configs"] = other_configs(torch, device, distributed, synthetic, AGG_RMS)
            line["configs"].update(widened_configs(torch))
            line["api_ms"] = api_timings(device)
        if not args.no_cpu_baseline and world == 1:
            corr_host = corr.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(corr_host, args.seed, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
