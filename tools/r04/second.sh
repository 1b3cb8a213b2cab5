#!/bin/bash
set -o pipefail
OUT=gpurun_out/r04_second
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -s -k "margin or wide_coordinates or rccl or matrix" > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
grep -E "worst|passed|failed|rccl" $OUT/tests.log | tee -a $OUT/summary.txt
bash tools/r04/ab.sh $OUT "base|" "ahead1|-DSFM_MATRIX_AHEAD=1" "ahead2|-DSFM_MATRIX_AHEAD=2" "ahead2pops1|-DSFM_MATRIX_AHEAD=2 -DSFM_MATRIX_POPS=1" "base|"
THR=1e-14 SIZES="50000 100000" bash tools/r04/ab.sh $OUT "base|" "ahead1|-DSFM_MATRIX_AHEAD=1" "ahead2|-DSFM_MATRIX_AHEAD=2"
