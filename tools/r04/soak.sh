#!/bin/bash
# Soak of the exact tier's gate (sfm::sed_inlier) in both filtered kernels against the all-fp64 kernel: thresholds ON SED values, +- 1 ulp, extreme scales.
OUT=gpurun_out/${1:-r04_soak}; mkdir -p $OUT
T1=${2:-4000}; T2=${3:-3000}; T3=${4:-500}
SOAK_REPORT=2000 timeout -k 10 900 python3 tools/soak_filter.py $T1 41 > $OUT/soak_valu_filter.txt 2>&1; echo "VALU-filter kernel rc=$?: $(tail -1 $OUT/soak_valu_filter.txt)" | tee -a $OUT/summary.txt
SOAK_REPORT=2000 SFM_SCORE_MATRIX=1 timeout -k 10 900 python3 tools/soak_filter.py $T2 42 > $OUT/soak_matrix_forced.txt 2>&1; echo "matrix kernel forced rc=$?: $(tail -1 $OUT/soak_matrix_forced.txt)" | tee -a $OUT/summary.txt
SOAK_N_MIN=8192 SOAK_H_MIN=2048 SOAK_H_MAX=8000 SOAK_REPORT=200 timeout -k 10 900 python3 tools/soak_filter.py $T3 43 > $OUT/soak_matrix_large.txt 2>&1; echo "matrix kernel, thousands of hypotheses rc=$?: $(tail -1 $OUT/soak_matrix_large.txt)" | tee -a $OUT/summary.txt
SOAK_N_MIN=32768 SOAK_N_MAX=45000 SOAK_H_MIN=2048 SOAK_H_MAX=5000 SOAK_REPORT=100 SFM_SCORE_MATRIX=1 SFM_SCORE_SPLIT=8 timeout -k 10 900 python3 tools/soak_filter.py ${5:-600} 44 > $OUT/soak_matrix_replay.txt 2>&1; echo "matrix kernel, eight ranges over >= 32768 points (the pre-pass is replayed) rc=$?: $(tail -1 $OUT/soak_matrix_replay.txt)" | tee -a $OUT/summary.txt
