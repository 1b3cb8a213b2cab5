#!/bin/bash
# Power draw, power cap and clocks of the card while the bench workload runs (rocm-smi read-only queries).
OUT=gpurun_out/${1:-r04_power}; mkdir -p $OUT
rocm-smi --showmaxpower --showpower --showclocks --showperflevel > $OUT/idle.txt 2>&1
rocm-smi --showpowerplay > /dev/null 2>&1
python3 bench.py --steps 30000 --warmup 100 --no-cpu-baseline --no-extras > $OUT/bench_long.json 2> $OUT/bench_long.err &
BENCH=$!
sleep 25   # build check + first import + warmup
for i in $(seq 1 24); do
  echo "--- sample $i" >> $OUT/load.txt
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge|hotspot)" >> $OUT/load.txt
  sleep 1
done
wait $BENCH
echo "bench rc=$?" >> $OUT/load.txt
cut -c1-300 $OUT/bench_long.json
grep -E "Power|sclk" $OUT/load.txt | sort | uniq -c | sort -rn | head -20
cat $OUT/idle.txt | grep -E "Power|sclk|Max" 
