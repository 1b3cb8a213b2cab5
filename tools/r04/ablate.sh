#!/bin/bash
# Where does tier 1 of the matrix-pipe kernel spend its time?  Measurement builds (SFM_MATRIX_ABLATE bits: wrong results) on a
# threshold nothing passes (tier 1 alone) and on the bench threshold, same box.
OUT=gpurun_out/${1:-r04_ablate}; mkdir -p $OUT
run() {  # name, flags
  export SFM_EXTRA_HIPCC_FLAGS="$2"
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_$1.log 2>&1 || { echo "build failed: $1"; return; }
  for thr in 1e-14 1.5e-6; do
    echo "$1 [$2] thr=$thr: $(THR=$thr REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -2 | sed 's/N=[0-9]* H=[0-9]* MATRIX=- SPLIT=-: //; s/max rel diff of the sums [0-9.e-]* [0-9.e-]*//' | tr '\n' ' ')" | tee -a $OUT/ablate.txt
  done
}
run default ""
run ahead2 "-DSFM_MATRIX_AHEAD=2"
run noloads "-DSFM_MATRIX_ABLATE=1"
run onemfma "-DSFM_MATRIX_ABLATE=2"
run onebit "-DSFM_MATRIX_ABLATE=4"
run nopush "-DSFM_MATRIX_ABLATE=8"
run noloads_onemfma_onebit "-DSFM_MATRIX_ABLATE=7"
run all "-DSFM_MATRIX_ABLATE=15"
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
