#!/bin/bash
# Per-wave stamps of the final kernel at the bench size (default options): where the slots are empty.
OUT=gpurun_out/${1:-r04_timeline_final}; mkdir -p $OUT
export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_STAMPS=1"
python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || { echo "build failed"; tail $OUT/build.log; exit 1; }
THR=1.5e-6 timeout -k 10 300 python3 tools/r04/matrix_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
