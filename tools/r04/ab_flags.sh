#!/bin/bash
# Same-box A/B of builds: tools/r04/ab_flags.sh OUTDIR "flags A" "flags B" ...   ("-" = the default build)
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export OLD_LIB=libsfm_hip_prev.so
for flags in "$@"; do
  if [ "$flags" = "-" ]; then export SFM_EXTRA_HIPCC_FLAGS=""; else export SFM_EXTRA_HIPCC_FLAGS="$flags"; fi
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || { echo "build failed: $flags"; tail -5 $OUT/build.log; continue; }
  for cfg in "50000 100000 1.5e-6" "20000 40000 1.5e-6" "50000 20000 1.5e-6" "50000 100000 1e-14"; do
    set -- $cfg
    echo "[$flags] $(N=$1 H=$2 THR=$3 REPS=15 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'this library' | sed -e 's/this library *//; s/MATRIX=-: counts differing 0; //')" | tee -a $OUT/ab.txt
  done
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
