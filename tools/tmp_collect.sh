set -o pipefail
tools/collect_counters.sh gpurun_out/r03_counters > gpurun_out/r03_counters.log 2>&1 || { tail -20 gpurun_out/r03_counters.log; exit 1; }
tail -5 gpurun_out/r03_counters.log
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
N=5000 H=10000 STEPS=300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r03_c2trace -o p -- python3 $REPO/tools/time_small_pass.py > $REPO/gpurun_out/r03_c2trace.log 2>&1
N=300 H=2000 STEPS=300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r03_demotrace -o p -- python3 $REPO/tools/time_small_pass.py > $REPO/gpurun_out/r03_demotrace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r03_c5trace -o p -- python3 $REPO/tools/time_c5.py > $REPO/gpurun_out/r03_c5trace.log 2>&1
cd $REPO
tail -1 gpurun_out/r03_c2trace.log gpurun_out/r03_demotrace.log gpurun_out/r03_c5trace.log
find gpurun_out/r03_c2trace gpurun_out/r03_c5trace gpurun_out/r03_demotrace -name "*kernel_stats.csv"
