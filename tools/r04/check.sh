#!/bin/bash
# parity subset + same-box A/B against round 3's library
OUT=gpurun_out/${1:-r04_check}; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix or filtered_score or full_size or determinism or margin" > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/tests.log | tee -a $OUT/summary.txt
for cfg in "50000 100000 1.5e-6 -" "50000 100000 1e-14 -" "50000 125000 1.5e-6 -" "20000 40000 1.5e-6 -" "50000 20000 1.5e-6 -" "8192 25000 1.5e-6 1" "30000 7000 1.5e-6 1" "20000 10000 1.5e-6 1" "16000 16000 1.5e-6 1"; do
  set -- $cfg
  if [ "$4" = "-" ]; then unset SFM_SCORE_MATRIX; else export SFM_SCORE_MATRIX=$4; fi
  echo "$(N=$1 H=$2 THR=$3 REPS=10 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'library' | sed 's/counts differing 0; //; s/scoring kernel/k/; s/whole scoring call/call/' | tr '\n' '|')" | tee -a $OUT/vs_r03.txt
done
