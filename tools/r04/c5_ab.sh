#!/bin/bash
# C5 (256 x 10 000 x 2 000) and the single-pair sizes by build: tools/r04/c5_ab.sh OUT "flags" ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for flags in "$@"; do
  if [ "$flags" = "-" ]; then export SFM_EXTRA_HIPCC_FLAGS=""; else export SFM_EXTRA_HIPCC_FLAGS="$flags"; fi
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || { echo "build failed: $flags"; continue; }
  echo "[$flags] $(timeout -k 10 300 python3 tools/time_c5.py 2>&1 | tail -1)" | tee -a $OUT/c5.txt
  echo "[$flags] $(N=50000 H=100000 THR=1.5e-6 REPS=10 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'this library' | sed -e 's/this library *//; s/MATRIX=-: counts differing 0; //')" | tee -a $OUT/c5.txt
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
