#!/bin/bash
OUT=gpurun_out/${1:-r04_c5split}; mkdir -p $OUT
for split in default 2 3 4 6 9 12; do
  if [ $split = default ]; then unset SFM_SCORE_SPLIT; else export SFM_SCORE_SPLIT=$split; fi
  echo "split=$split: $(timeout -k 10 300 python3 tools/time_c5.py 2>&1 | tail -1)" | tee -a $OUT/c5.txt
done
unset SFM_SCORE_SPLIT
echo "VALU kernel: $(SFM_SCORE_MATRIX=0 timeout -k 10 300 python3 tools/time_c5.py 2>&1 | tail -1)" | tee -a $OUT/c5.txt
