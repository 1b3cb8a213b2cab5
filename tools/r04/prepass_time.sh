#!/bin/bash
# Kernel times of one bench step by build (rocprofv3 kernel trace over python bench.py --steps 20): tools/r04/prepass_time.sh OUT "flags" ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for flags in "$@"; do
  if [ "$flags" = "-" ]; then export SFM_EXTRA_HIPCC_FLAGS=""; tag=default; else export SFM_EXTRA_HIPCC_FLAGS="$flags"; tag=$(echo "$flags" | tr -c 'A-Za-z0-9=_' '_'); fi
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || { echo "build failed: $flags"; continue; }
  unset SFM_EXTRA_HIPCC_FLAGS_TMP
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "[$flags]"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]: print("   ", r['Name'][:60].ljust(60), r['Calls'], "%.1f us" % (float(r['AverageNs'])/1e3))
PY
done
unset SFM_EXTRA_HIPCC_FLAGS
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
