#!/bin/bash
OUT=gpurun_out/${1:-r04_fold}; mkdir -p $OUT
LIB=structure_from_motion_amd/csrc/libsfm_hip.so
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix or filtered_score or full_size or determinism" > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/tests.log | tee -a $OUT/summary.txt
run() {  # name, flags
  if [ "$1" = base ]; then
    cp tools/r04/base/libsfm_hip.so $LIB; : > structure_from_motion_amd/csrc/libsfm_hip.flags; unset SFM_EXTRA_HIPCC_FLAGS
  else
    export SFM_EXTRA_HIPCC_FLAGS="$2"
    python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > $OUT/build_$1.log 2>&1 || { echo "build failed: $1"; return; }
  fi
  for cfg in "50000 100000 1e-14" "50000 100000 1.5e-6" "20000 40000 1.5e-6" "50000 20000 1.5e-6" "50000 125000 1.5e-6" "12512 100000 1e-14"; do
    set -- $cfg
    echo "$NAME [$FLAGS] N=$1 H=$2 thr=$3: $(N=$1 H=$2 THR=$3 REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -2 | sed 's/N=[0-9]* H=[0-9]* MATRIX=- SPLIT=-: //; s/max rel diff of the sums [0-9.e-]* [0-9.e-]*//' | tr '\n' ' ')" | tee -a $OUT/ab.txt
  done
}
NAME=base FLAGS="" run base ""
NAME=default FLAGS="" run default ""
for split in 4 16; do
  echo "default split=$split: $(SFM_SCORE_SPLIT=$split REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -1)" | tee -a $OUT/ab.txt
done
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_STAMPS=1"
python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_stamps.log 2>&1
for thr in 1e-14 1.5e-6; do
  echo "=== thr=$thr (default ranges)" | tee -a $OUT/timeline.txt
  THR=$thr STAMPS_OUT=$OUT/stamps_$thr.npy timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
