#!/bin/bash
# Same-box A/B of the scoring call: tools/r04/base/libsfm_hip_prev.so (the previous commit, cross-built) against the current library.
OUT=gpurun_out/${1:-r04_vs_prev}; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for cfg in "50000 100000 1.5e-6 -" "50000 125000 1.5e-6 -" "20000 40000 1.5e-6 -" "50000 20000 1.5e-6 -" "8192 25000 1.5e-6 1" "10000 2000 1.5e-6 1"; do
  set -- $cfg
  if [ "$4" = "-" ]; then unset SFM_SCORE_MATRIX; else export SFM_SCORE_MATRIX=$4; fi
  N=$1 H=$2 THR=$3 REPS=15 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'library' | sed -e 's/round-3 library/previous commit/' | tee -a $OUT/vs_prev.txt
done
unset SFM_SCORE_MATRIX
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filtered or matrix or score or fused or pass or full_size" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json
