#!/bin/bash
# rocprofv3 PMC passes over tools/ab_matrix_score.py (run on the GPU box, from the repo root): what bounds the matrix-pipe
# scoring kernel — VALU issue, the texture addresser (scattered gathers of the exact tier), LDS, or waiting.
#   tools/collect_matrix_counters.sh gpurun_out/<dir>
OUT=${1:?usage: tools/collect_matrix_counters.sh gpurun_out/<dir>}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp REPS=3
cd /tmp
run() {
    local name=$1; shift
    echo "== $name: $*"
    # Round 3's "silent hang" was the pass `ta2` (gpurun_out/r03m/ta2.log): FOUR counters of the TA block in one pass —
    # the block has two counter registers per instance on gfx950 — made rocprofiler_create_counter_config fail with
    # "error code 38: Request exceeds the capabilities of the hardware to collect"; rocprofv3 aborted (signal 6) before the
    # program ran and then sat in its own finalisation.  No kernel was involved and the GPU was never touched by that pass.
    # Hence: at most two TA counters (and at most eight SQ counters) per pass below; the limit stays as a guard only.
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$REPO/tools/ab_matrix_score.py" > "$OUT/$name.log" 2>&1 || echo "   (pass failed or timed out: see $OUT/$name.log)"
}
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES
run wait SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr
run ta_stall TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta_waves TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
run grbm GRBM_GUI_ACTIVE
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
rows = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "score_sed_matrix_kernel" not in k and "score_sed_filtered_kernel" not in k:
            continue
        name = "matrix" if "matrix" in k else "filtered"
        d = rows.setdefault((name, r["Counter_Name"]), [])
        d.append(float(r["Counter_Value"]))
with open(os.path.join(out, "matrix_counters.csv"), "w") as f:
    f.write("kernel,counter,launches,mean_per_launch\n")
    for (name, c), v in sorted(rows.items()):
        # one row per (dispatch, counter) after rocprofv3 sums the instances of a dispatch; guard against per-instance rows
        f.write(f"{name},{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(open(os.path.join(out, "matrix_counters.csv")).read())
PY
