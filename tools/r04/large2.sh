#!/bin/bash
OUT=gpurun_out/${1:-r04_large2}; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_pass" > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/tests.log | tee -a $OUT/summary.txt
for lp in 1 0 1 0; do
  SFM_LARGE_PASS=$lp python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_lp$lp.json 2> $OUT/bench_lp$lp.err
  python -c "
import json
d=json.load(open('$OUT/bench_lp$lp.json'))
print('large_pass=$lp value %.3e ms/step %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $OUT/summary.txt
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/trace -o p -- python3 $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/trace.log 2>&1; cd $OLDPWD
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python - <<PY | tee -a $OUT/summary.txt
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:9]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us  {float(r['TotalDurationNs'])/tot*100:5.1f} %")
PY
