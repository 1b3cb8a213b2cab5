#!/bin/bash
# Same-box A/B of scoring-kernel builds: tools/r04/ab.sh OUTDIR "name|extra hipcc flags" ...   (name "base" = the library of
# the round's first commit, cross-built into tools/r04/base/).  Sizes: SIZES="N H;N H" (default: the bench workload + two more).
OUT=$1; shift
mkdir -p $OUT
SIZES=${SIZES:-"50000 100000;20000 40000;50000 20000"}
LIB=structure_from_motion_amd/csrc/libsfm_hip.so
for v in "$@"; do
  name=${v%%|*}; flags=${v#*|}
  if [ "$name" = base ]; then
    cp tools/r04/base/libsfm_hip.so $LIB; : > structure_from_motion_amd/csrc/libsfm_hip.flags; unset SFM_EXTRA_HIPCC_FLAGS
  else
    export SFM_EXTRA_HIPCC_FLAGS="$flags"
    touch structure_from_motion_amd/csrc/sfm_score_matrix.h   # stale: rebuild with these flags
  fi
  IFS=';' read -ra S <<< "$SIZES"
  for cfg in "${S[@]}"; do
    set -- $cfg
    echo "$name [$flags] N=$1 H=$2 thr=${THR:-1.5e-6}: $(N=$1 H=$2 REPS=${REPS:-20} timeout -k 10 600 python3 tools/ab_matrix_score.py 2>&1 | tail -2 | tr '\n' ' ')" | tee -a $OUT/ab.txt
  done
done
unset SFM_EXTRA_HIPCC_FLAGS
touch structure_from_motion_amd/csrc/sfm_score_matrix.h
python3 -c "from structure_from_motion_amd import build; build.build_all()" > /dev/null 2>&1   # leave the default build behind
