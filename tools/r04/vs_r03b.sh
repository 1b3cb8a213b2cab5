#!/bin/bash
OUT=gpurun_out/${1:-r04_vs_r03b}; mkdir -p $OUT
for pers in 1 0; do
export SFM_SCORE_PERSISTENT=$pers
for cfg in "50000 100000 1.5e-6 -" "50000 125000 1.5e-6 -" "20000 40000 1.5e-6 -" "50000 20000 1.5e-6 -" "8192 25000 1.5e-6 1" "30000 7000 1.5e-6 1" "20000 10000 1.5e-6 1"; do
  set -- $cfg
  if [ "$4" = "-" ]; then unset SFM_SCORE_MATRIX; else export SFM_SCORE_MATRIX=$4; fi
  echo "persistent=$pers $(N=$1 H=$2 THR=$3 REPS=10 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'library' | tr '\n' '|')" | tee -a $OUT/vs_r03.txt
done
done
unset SFM_SCORE_PERSISTENT SFM_SCORE_MATRIX
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_STAMPS=1"
python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_stamps.log 2>&1
for cfg in "50000 100000" "20000 40000"; do
  set -- $cfg
  echo "=== N=$1 H=$2" | tee -a $OUT/timeline.txt
  N=$1 H=$2 timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
