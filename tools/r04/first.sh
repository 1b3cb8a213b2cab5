#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite on the new bound / hand-off / options, the margin report, and an A/B of the
# range-split hand-off (ACQ_REL arrival vs the relaxed form) plus tier 1 alone (a threshold nothing passes).
set -o pipefail
OUT=gpurun_out/r04_first
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/gputest.log | tee -a $OUT/summary.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_matrix_filter_error_bound_margin" > $OUT/margin.log 2>&1; echo "margin rc=$?" | tee -a $OUT/summary.txt
grep -E "worst|rejected" $OUT/margin.log | tee -a $OUT/summary.txt
for cfg in "50000 100000" "20000 40000" "50000 20000"; do
  set -- $cfg
  echo "acq_rel N=$1 H=$2: $(N=$1 H=$2 REPS=20 timeout -k 10 300 python3 tools/ab_matrix_score.py | tail -1)" | tee -a $OUT/summary.txt
done
echo "tier1-only (thr 1e-14) N=50000 H=100000: $(THR=1e-14 REPS=20 timeout -k 10 300 python3 tools/ab_matrix_score.py | tail -1)" | tee -a $OUT/summary.txt
echo "VALU kernel N=50000 H=100000: $(SFM_SCORE_MATRIX=0 REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py | tail -1)" | tee -a $OUT/summary.txt
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_SPLIT_HANDOFF_RELAXED=1"
for cfg in "50000 100000" "20000 40000" "50000 20000"; do
  set -- $cfg
  echo "relaxed N=$1 H=$2: $(N=$1 H=$2 REPS=20 timeout -k 10 600 python3 tools/ab_matrix_score.py | tail -1)" | tee -a $OUT/summary.txt
done
unset SFM_EXTRA_HIPCC_FLAGS
for cfg in "50000 100000"; do
  set -- $cfg
  echo "acq_rel again N=$1 H=$2: $(N=$1 H=$2 REPS=20 timeout -k 10 600 python3 tools/ab_matrix_score.py | tail -1)" | tee -a $OUT/summary.txt
done
