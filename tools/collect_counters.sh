#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box, from the repo root):
#   tools/collect_counters.sh <out-dir-under-gpurun_out> [bench args...]
# One --kernel-trace --stats pass and separate --pmc passes (counters never share a run with a trace; FETCH_SIZE and
# WRITE_SIZE need a pass each on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots) over the same bench command, then
# tools/summarize_counters.py condenses them into <out>/score_traffic.json (bench.py's roofline input, stamped with the
# kernel-source fingerprint) and <out>/pmc_summary.csv / kernel_stats.csv (the files to commit under profiles/).
set -e -o pipefail
OUT=${1:?usage: tools/collect_counters.sh gpurun_out/<dir> [bench args]}
shift || true
ARGS=${*:---steps 5 --warmup 2 --no-cpu-baseline --no-extras}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp
run() {  # name, rocprofv3 options...
    local name=$1; shift
    echo "== $name: rocprofv3 $* -- python3 bench.py $ARGS"
    rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
}
# the timing pass runs as long as the default bench (steady clocks: a 7-launch run reads ~10 % slow); counter passes are short
TRACE_ARGS=${TRACE_ARGS:---steps 200 --warmup 10 --no-cpu-baseline --no-extras}
echo "== trace: rocprofv3 --kernel-trace --stats -- python3 bench.py $TRACE_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o p -- python3 "$REPO/bench.py" $TRACE_ARGS > "$OUT/trace.log" 2>&1
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run f64 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
run f32 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
run mfma --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
cd "$REPO"
python3 tools/summarize_counters.py "$OUT" $ARGS
