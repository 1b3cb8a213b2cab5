"""rocprofv3 evidence for the widened rows on the bench line (run on the GPU box, from the repo root):

    python3 tools/collect_widened_counters.py gpurun_out/<dir> [workload ...]

For every workload of tools/widened_workloads.py::BUILDERS (or the ones named): one `--kernel-trace --stats` pass and separate
`--pmc` passes (counters never share a run with a trace; FETCH_SIZE and WRITE_SIZE a pass each on gfx950: MI355X_MICROARCH.md)
over `python3 tools/run_config.py <workload> <passes>`, condensed into <out>/<workload>_counters.json — per kernel of the
workload its launches per pass, trace average and per-launch counter means, stamped with git revision and the fingerprint of the
kernels' sources (bench.py: WIDENED_SOURCES) — and <out>/<workload>_kernel_stats.csv.  The files to commit as
profiles/<workload>_counters.json and profiles/r05/<workload>_kernel_stats.csv.  This process never touches the GPU itself:
the profiler starts python3 directly (no shell or env wrapper in between)."""
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
PASSES = {"f1_match_20000x20000_ncc9": 6, "f1_match_600x600_ncc9": 40, "f2_harris_vga": 10, "f2_harris_1080p": 5,
          "f4_refine_50000": 20, "pose_tail_c5": 6}
GROUPS = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
          ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]),
          ("f64", ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]),
          ("wait", ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"])]
OURS = ("pair_summary_kernel", "summary_combine_kernel", "patch_extract_kernel", "correlate_kernel", "cornerness_kernel", "nms_round_kernel",
        "nms_finalize_kernel", "nms_inplace_kernel", "compact_nonzero_kernel", "prune_histogram_kernel", "prune_filter_kernel", "refine_kernel", "cheirality_batched_kernel", "pose_vote_kernel",
        "triangulate_selected_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].strip()


def main():
    out = os.path.abspath(sys.argv[1])
    wanted = sys.argv[2:] or list(PASSES)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    from bench import widened_sha

    for name in wanted:
        passes = PASSES[name]
        runner = [sys.executable, os.path.join(REPO, "tools", "run_config.py"), name, str(passes)]
        work = os.path.join(out, name + "_passes")
        shutil.rmtree(work, ignore_errors=True)
        done = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(work, "trace"), "-o", "p",
                               "--"] + runner, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        print(name, "trace rc", done.returncode, flush=True)
        for group, counters in GROUPS:
            done = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", os.path.join(work, group), "-o", "p", "--"]
                                  + runner, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            print(name, group, "rc", done.returncode, flush=True)
        kernels = collections.defaultdict(dict)
        stats = glob.glob(os.path.join(work, "trace", "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(out, name + "_kernel_stats.csv"))
            for row in csv.DictReader(open(stats[0])):
                k = short(row["Name"])
                if any(k.startswith(o) for o in OURS):
                    kernels[k]["trace_avg_us"] = float(row["AverageNs"]) / 1e3
                    kernels[k]["launches_per_pass"] = int(row["Calls"]) / float(passes)
        values = collections.defaultdict(list)
        for path in glob.glob(os.path.join(work, "*", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                if any(k.startswith(o) for o in OURS):
                    values[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, counter), vals in values.items():
            kernels[k][counter] = sum(vals) / len(vals)
        git = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        rec = {"config": name, "passes": passes, "source_sha": widened_sha(name),
               "git": os.environ.get("SFM_GIT_SHA") or git or None, "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
               "command": "tools/collect_widened_counters.py: rocprofv3 --pmc <one group per pass> -- python3 tools/run_config.py " + name,
               "kernels": dict(kernels)}
        json.dump(rec, open(os.path.join(out, name + "_counters.json"), "w"), indent=1)
        shutil.rmtree(work, ignore_errors=True)
        print(name, {k: {c: round(v, 1) for c, v in d.items() if c in ("trace_avg_us", "launches_per_pass", "SQ_INSTS_VALU", "FETCH_SIZE")}
                     for k, d in kernels.items()}, flush=True)


if __name__ == "__main__":
    main()
