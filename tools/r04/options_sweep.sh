#!/bin/bash
# Launch options re-measured on the final kernel (no rebuild: the package reads SFM_SCORE_* into the default options at import).
OUT=gpurun_out/${1:-r04_options}; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for env in "" "SFM_SCORE_PERSISTENT=1" "SFM_SCORE_SPLIT=16" "SFM_SCORE_SPLIT=4" "SFM_SCORE_XCD=0" ""; do
  for cfg in "50000 100000" "50000 125000" "20000 40000"; do
    set -- $cfg
    echo "[$env] $(env $env N=$1 H=$2 THR=1.5e-6 REPS=15 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'this library' | sed -e 's/this library *//; s/MATRIX=-: counts differing 0; //')" | tee -a $OUT/options.txt
  done
done
