#!/bin/bash
OUT=gpurun_out/${1:-r04_timeline}; mkdir -p $OUT
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_STAMPS=1"
python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || { echo "build failed"; tail $OUT/build.log; exit 1; }
for thr in 1e-14 1.5e-6; do
  for split in 8 16; do
    echo "=== thr=$thr split=$split" | tee -a $OUT/timeline.txt
    THR=$thr SFM_SCORE_MATRIX=1 SFM_SCORE_SPLIT=$split timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | tee -a $OUT/timeline.txt
  done
done
echo "=== thr=1e-14 split=8 N=12512" | tee -a $OUT/timeline.txt
N=12512 THR=1e-14 SFM_SCORE_MATRIX=1 SFM_SCORE_SPLIT=8 timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | tee -a $OUT/timeline.txt
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
