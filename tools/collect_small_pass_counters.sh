#!/bin/bash
# rocprofv3 PMC evidence for the small pass's scoring kernel (run on the GPU box, from the repo root):
#   tools/collect_small_pass_counters.sh gpurun_out/<dir> [N] [H]
# One --kernel-trace --stats pass and separate --pmc passes (vector-L1 = TCP, L2 = TCC, SQ) over tools/time_small_pass.py
# for each setting of SFM_SCORE_HPW / SFM_SCORE_SYNC (exported before the profiler starts: the program itself follows `--`).
set -e -o pipefail
OUT=${1:?usage: tools/collect_small_pass_counters.sh gpurun_out/<dir> [N] [H]}
export N=${2:-5000} H=${3:-10000} STEPS=${STEPS:-40}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp
run() {  # name, rocprofv3 options...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$REPO/tools/time_small_pass.py" > "$OUT/$name.log" 2>&1
}
for variant in ${VARIANTS:-hpw1 hpw2}; do
    unset SFM_SCORE_HPW SFM_SCORE_SYNC
    case $variant in
        hpw1) export SFM_SCORE_HPW=1 SFM_SCORE_SYNC=0 ;;
        hpw2) export SFM_SCORE_HPW=2 ;;
        hpw1sync) export SFM_SCORE_HPW=1 SFM_SCORE_SYNC=2 ;;
    esac
    echo "== $variant (N=$N H=$H)"
    run ${variant}_trace --kernel-trace --stats
    run ${variant}_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
    run ${variant}_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    run ${variant}_sq --pmc SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
    run ${variant}_lds --pmc SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES
done
cd "$REPO"
python3 tools/summarize_small_pass_counters.py "$OUT" ${VARIANTS:-hpw1 hpw2}
