#!/bin/bash
OUT=gpurun_out/${1:-r04_final_checks}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/gputest.log | tee -a $OUT/summary.txt
echo "C5: $(timeout -k 10 300 python3 tools/time_c5.py 2>&1 | tail -1)" | tee -a $OUT/summary.txt
