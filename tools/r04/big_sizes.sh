#!/bin/bash
# The per-rank shares of the 1 M-hypothesis configuration at 2 and 4 ranks (500 000 and 250 000 hypotheses x 50 000 points) on one GPU:
# counts against the all-fp64 kernel, kernel time; then the bench step at those sizes.
OUT=gpurun_out/${1:-r04_big}; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for h in 250000 500000; do
  N=50000 H=$h THR=1.5e-6 REPS=5 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'library' | sed -e 's/round-3 library/previous commit/' | tee -a $OUT/big.txt
  python bench.py --hypotheses $h --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/bench_$h.err | cut -c1-260 | tee -a $OUT/big.txt
done
