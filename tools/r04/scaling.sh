#!/bin/bash
# fixed cost per wave vs cost per step of the matrix-pipe kernel: point count varied at fixed hypotheses and ranges
OUT=gpurun_out/${1:-r04_scaling}; mkdir -p $OUT
run() {  # name, flags
  export SFM_EXTRA_HIPCC_FLAGS="$2"
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build_$1.log 2>&1 || { echo "build failed: $1"; return; }
  for n in 12512 25024 50048; do
    for split in 8 16; do
      echo "$1 [$2] N=$n split=$split thr=1e-14: $(N=$n THR=1e-14 SFM_SCORE_MATRIX=1 SFM_SCORE_SPLIT=$split REPS=10 timeout -k 10 300 python3 tools/ab_matrix_score.py 2>&1 | tail -1 | tr '\n' ' ')" | tee -a $OUT/scaling.txt
    done
  done
}
run default ""
run all "-DSFM_MATRIX_ABLATE=15"
run nosubnormals "-DSFM_MATRIX_ABLATE=16"
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
