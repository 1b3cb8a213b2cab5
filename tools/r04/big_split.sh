#!/bin/bash
OUT=gpurun_out/${1:-r04_big_split}; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for h in 250000 500000; do
  for env in "" "SFM_SCORE_SPLIT=8" "SFM_SCORE_SPLIT=16"; do
    echo "[$env] $(env $env N=50000 H=$h THR=1.5e-6 REPS=5 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'this library' | sed -e 's/this library *//; s/MATRIX=-: counts differing 0; //')" | tee -a $OUT/big_split.txt
  done
done
