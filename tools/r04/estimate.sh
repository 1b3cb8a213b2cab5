#!/bin/bash
OUT=gpurun_out/${1:-r04_estimate}; mkdir -p $OUT
for steps in 128 64 96; do
  export SFM_EXTRA_HIPCC_FLAGS="-DSFM_MATRIX_ESTIMATE_STEPS=$steps"
  python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > $OUT/build_$steps.log 2>&1 || { echo "build failed"; continue; }
  for rep in 1 2; do
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$steps.json 2> $OUT/bench_$steps.err
  python -c "
import json
d=json.load(open('$OUT/bench_$steps.json'))
print('estimate steps=$steps value %.3e ms/step %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $OUT/summary.txt
  done
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
