#!/bin/bash
# rocprofv3 evidence for bench.py's `configs` block (run on the GPU box, from the repo root):
#   tools/collect_config_counters.sh gpurun_out/<dir>
# For C2 (5 000 x 10 000, lean small pass) and C5 (256 x 10 000 x 2 000, batched pipeline): one --kernel-trace --stats pass and
# separate --pmc passes over tools/run_config.py, condensed into <out>/<config>_counters.json (the scoring kernel's per-launch
# counters, stamped like score_traffic.json) and <out>/<config>_kernel_stats.csv — the files to commit as profiles/c2_counters.json,
# profiles/c5_counters.json and profiles/rNN/c?_kernel_stats.csv.
OUT=${1:?usage: tools/collect_config_counters.sh gpurun_out/<dir>}
REPO=$(pwd); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp
for cfg in c2 c5; do
  passes=10; [ $cfg = c2 ] && passes=40
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace" -o p -- python3 "$REPO/tools/run_config.py" $cfg $passes > "$OUT/${cfg}_trace.log" 2>&1 || echo "trace $cfg failed"
  for g in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
    set -- $g; name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${cfg}_$name" -o p -- python3 "$REPO/tools/run_config.py" $cfg 3 > "$OUT/${cfg}_$name.log" 2>&1 || echo "pass $cfg $name failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, shutil, sys, time
out = sys.argv[1]
sys.path.insert(0, os.getcwd())
from structure_from_motion_amd import _native, build
shapes = {"c2": (5000, 10000, 1), "c5": (10000, 2000, 256)}
for cfg, (n, h, b) in shapes.items():
    values = collections.defaultdict(list)
    spans = collections.defaultdict(list)
    for path in glob.glob(os.path.join(out, cfg + "_*", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "score_sed_" not in k or "score_sed_matrix_kernel<true" in k:   # (<true, ...>: the cost pre-pass)
                continue
            values[(row["Counter_Name"], k)].append(float(row["Counter_Value"]))
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                spans[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    if not values:
        print(cfg, "no scoring-kernel dispatches found")
        continue
    kernel = max(spans, key=lambda k: sum(spans[k]))
    counters = {c: sum(v) / len(v) for (c, k), v in values.items() if k == kernel}
    stats = glob.glob(os.path.join(out, cfg + "_trace", "**", "*kernel_stats.csv"), recursive=True)
    trace_ms = None
    if stats:
        shutil.copy(stats[0], os.path.join(out, cfg + "_kernel_stats.csv"))
        for row in csv.DictReader(open(stats[0])):
            if row["Name"] == kernel:
                trace_ms = float(row["AverageNs"]) * 1e-6
    rec = {"config": cfg, "kernel": kernel[:140], "matches": n, "hypotheses": h, "batch": b, "source_sha": build.score_source_sha(),
           "abi": _native.ABI_VERSION, "git": os.environ.get("SFM_GIT_SHA") or None, "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
           "command": "tools/collect_config_counters.sh: rocprofv3 --pmc <one group per pass> -- python3 tools/run_config.py " + cfg,
           "kernel_trace_avg_ms": trace_ms, "counters": counters}
    json.dump(rec, open(os.path.join(out, cfg + "_counters.json"), "w"), indent=1)
    print(cfg, kernel[:80], "trace avg ms", trace_ms, {k: round(v) for k, v in counters.items()})
PY
