#!/bin/bash
# occupancy / sample fix-up / E-at-bursts variants of the matrix-pipe kernel, same box; base = the library of the round's first commit
OUT=gpurun_out/${1:-r04_occ}; mkdir -p $OUT
LIB=structure_from_motion_amd/csrc/libsfm_hip.so
run() {  # name, flags
  if [ "$1" = base ]; then
    cp tools/r04/base/libsfm_hip.so $LIB; : > structure_from_motion_amd/csrc/libsfm_hip.flags; unset SFM_EXTRA_HIPCC_FLAGS
  else
    export SFM_EXTRA_HIPCC_FLAGS="$2"
    python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_$1.log 2>&1 || { echo "build failed: $1"; return; }
  fi
  for thr in 1e-14 1.5e-6; do
    echo "$1 [$2] thr=$thr: $(THR=$thr REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -2 | sed 's/N=[0-9]* H=[0-9]* MATRIX=- SPLIT=-: //; s/max rel diff of the sums [0-9.e-]* [0-9.e-]*//' | tr '\n' ' ')" | tee -a $OUT/occ.txt
  done
}
run base ""
run default ""
run e_in_regs "-DSFM_MATRIX_E_IN_REGISTERS=1"
run occ5 "-DSFM_MATRIX_OCC=5"
run occ6cap16 "-DSFM_MATRIX_OCC=6 -DSFM_MATRIX_CAP=16"
run occ5ahead2 "-DSFM_MATRIX_OCC=5 -DSFM_MATRIX_AHEAD=2"
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_STAMPS=1"
python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_stamps.log 2>&1
for thr in 1e-14 1.5e-6; do
  echo "=== thr=$thr (default ranges)" | tee -a $OUT/timeline.txt
  THR=$thr STAMPS_OUT=$OUT/stamps_$thr.npy timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | tee -a $OUT/timeline.txt
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
