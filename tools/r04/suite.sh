#!/bin/bash
OUT=gpurun_out/${1:-r04_suite}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/gputest.log | tee -a $OUT/summary.txt
python bench.py --steps 100 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/${1:-r04_suite}/summary.txt
import json,sys,os
p=os.path.join("gpurun_out", sys.argv[1] if len(sys.argv)>1 else "r04_suite","bench.json")
PY
python -c "
import json
d=json.load(open('$OUT/bench.json'))
print('value %.3e ms/step %.3f kernel_ms %.3f frac %s floor %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['fp64_floor']))
print('variants', {k:(round(v['ms_per_step'],3), round(v['kernel_ms'],3)) for k,v in d.get('variants',{}).items()})
for k,v in d.get('configs',{}).items(): print(k, {kk:vv for kk,vv in v.items() if kk not in ('note',)})
print('api', d.get('api_ms'))
print('cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k!='sample'})
" 2>&1 | tee -a $OUT/summary.txt
