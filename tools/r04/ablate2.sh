#!/bin/bash
OUT=gpurun_out/${1:-r04_ablate2}; mkdir -p $OUT
run() {  # name, flags
  export SFM_EXTRA_HIPCC_FLAGS="$2"
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_$1.log 2>&1 || { echo "build failed: $1"; return; }
  for thr in 1e-14; do
    echo "$1 [$2] thr=$thr: $(THR=$thr REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -1 | tr '\n' ' ')" | tee -a $OUT/ablate.txt
  done
}
run default ""
run noloads "-DSFM_MATRIX_ABLATE=1"
run onemfma "-DSFM_MATRIX_ABLATE=2"
run nopush "-DSFM_MATRIX_ABLATE=8"
run noloads_onemfma "-DSFM_MATRIX_ABLATE=3"
run occ3 "-DSFM_MATRIX_OCC=3"
run occ2 "-DSFM_MATRIX_OCC=2"
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
