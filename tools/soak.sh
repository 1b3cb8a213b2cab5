#!/bin/bash
# Soak of the two-tier scoring kernels against the all-fp64 kernel (tools/soak_filter.py) in the three configurations of
# profiles/r04/soak_exact_tier_gate.txt, on the GPU box from the repo root:
#   tools/soak.sh gpurun_out/<dir> [trials_valu] [trials_matrix] [trials_wide]
# 1 the VALU-filter kernel; 2 the matrix-pipe kernel forced onto the same small problems; 3 the matrix-pipe kernel on tens of
# thousands of points x thousands of hypotheses (cost order, eight ranges, the replayed pre-pass).  The launch options come from
# the SFM_SCORE_* variables, which the Python binding translates once at load.
OUT=${1:?usage: tools/soak.sh gpurun_out/<dir> [trials...]}; mkdir -p "$OUT"
T1=${2:-100000}; T2=${3:-100000}; T3=${4:-20000}
timeout -k 10 900 python tools/soak_filter.py $T1 1 > "$OUT/soak_valu_filter.txt" 2>&1; echo "VALU-filter kernel rc=$?: $(tail -1 "$OUT/soak_valu_filter.txt")"
SFM_SCORE_MATRIX=1 timeout -k 10 900 python tools/soak_filter.py $T2 2 > "$OUT/soak_matrix_forced.txt" 2>&1; echo "matrix kernel forced rc=$?: $(tail -1 "$OUT/soak_matrix_forced.txt")"
SFM_SCORE_MATRIX=1 SFM_SCORE_SPLIT=8 SOAK_N_MIN=33000 SOAK_N_MAX=40000 SOAK_H_MIN=1000 SOAK_H_MAX=3000 SOAK_REPORT=1000 \
  timeout -k 10 900 python tools/soak_filter.py $T3 3 > "$OUT/soak_matrix_wide.txt" 2>&1; echo "matrix kernel, thousands of hypotheses rc=$?: $(tail -1 "$OUT/soak_matrix_wide.txt")"
