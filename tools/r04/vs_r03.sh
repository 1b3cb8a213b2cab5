#!/bin/bash
OUT=gpurun_out/${1:-r04_vs_r03}; mkdir -p $OUT
for cfg in "50000 100000 1.5e-6 -" "50000 100000 1e-14 -" "50000 100000 1.5e-6 0" "50000 125000 1.5e-6 -" "20000 40000 1.5e-6 -" "50000 20000 1.5e-6 -" "16000 16000 1.5e-6 1" "16000 16000 1.5e-6 0" "8192 25000 1.5e-6 1" "8192 25000 1.5e-6 0" "30000 7000 1.5e-6 1" "30000 7000 1.5e-6 0" "65536 4096 1.5e-6 1" "65536 4096 1.5e-6 0" "20000 10000 1.5e-6 1" "20000 10000 1.5e-6 0"; do
  set -- $cfg
  if [ "$4" = "-" ]; then unset SFM_SCORE_MATRIX; else export SFM_SCORE_MATRIX=$4; fi
  N=$1 H=$2 THR=$3 REPS=10 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep "library" | tee -a $OUT/vs_r03.txt
done
