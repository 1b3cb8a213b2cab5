#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_counters.sh:

  <out>/pmc_summary.csv      counter, kernel, dispatches, mean value (every kernel of the pass)
  <out>/kernel_stats.csv     the --kernel-trace --stats summary (copied)
  <out>/score_traffic.json   per-launch counters of the dominant kernel (the scoring kernel), stamped with the
                             kernel-source fingerprint / ABI / git revision — bench.py's roofline input

Values are as rocprofv3 reports them (FETCH_SIZE / WRITE_SIZE in KiB; bench.py applies the gfx950 x2 correction to
FETCH_SIZE).  Per-dispatch values are averaged over the timed + warm-up launches of the pass."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out = sys.argv[1]
    args = sys.argv[2:]
    matches = int(args[args.index("--matches") + 1]) if "--matches" in args else 50_000
    hyp = int(args[args.index("--hypotheses") + 1]) if "--hypotheses" in args else 100_000
    values = collections.defaultdict(list)      # (counter, kernel) -> per-dispatch values
    spans = collections.defaultdict(list)       # (pass, kernel) -> dispatch durations in ns
    for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
        pass_name = os.path.relpath(path, out).split(os.sep)[0]
        seen = set()
        for row in csv.DictReader(open(path)):
            kernel = row["Kernel_Name"]
            values[(row["Counter_Name"], kernel)].append(float(row["Counter_Value"]))
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                spans[(pass_name, kernel)].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    with open(os.path.join(out, "pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["counter", "kernel", "dispatches", "mean_value"])
        for (counter, kernel), vals in sorted(values.items()):
            w.writerow([counter, kernel[:96], len(vals), f"{sum(vals) / len(vals):.3f}"])
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    trace_ms = None
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))

    def scoring(name):
        # (score_sed_matrix_kernel<true, ...> is that kernel's cost pre-pass, not the scoring launch)
        return "score_sed_filtered_kernel" in name or ("score_sed_matrix_kernel" in name and "score_sed_matrix_kernel<true" not in name)

    kernels = sorted({k for (_, k) in values if scoring(k)},
                     key=lambda k: -sum(sum(v) for (p, kk), v in spans.items() if kk == k))   # the one the time went to
    if not kernels:
        print("no scoring-kernel dispatches found under", out)
        return
    kernel = kernels[0]
    short = "score_sed_matrix_kernel" if "score_sed_matrix_kernel" in kernel else "score_sed_filtered_kernel"
    if stats:
        for row in csv.DictReader(open(stats[0])):
            if scoring(row["Name"]) and short in row["Name"]:
                trace_ms = float(row["AverageNs"]) * 1e-6
    counters = {c: sum(v) / len(v) for (c, k), v in values.items() if k == kernel}
    sq_span = spans.get(("sq", kernel))
    if sq_span:
        counters["profiled_kernel_ms"] = sum(sq_span) / len(sq_span) * 1e-6
    from structure_from_motion_amd import _native, build

    # the GPU box gets a snapshot without .git: the caller passes the revision (gpurun -- 'SFM_GIT_SHA=$(git rev-parse ...) ...')
    git = os.environ.get("SFM_GIT_SHA", "")
    if not git:
        try:
            git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
        except OSError:
            git = ""
    rec = {
        "kernel": kernel[:120],
        "kernel_short": short,
        "matches": matches,
        "hypotheses": hyp,
        "source_sha": build.score_source_sha(),
        "abi": _native.ABI_VERSION,
        "git": git or None,
        "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
        "command": "tools/collect_counters.sh: rocprofv3 --pmc <one group per pass> -- python3 bench.py " + " ".join(args),
        "kernel_trace_avg_ms": trace_ms,
        "counters": counters,
    }
    json.dump(rec, open(os.path.join(out, "score_traffic.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
