mkdir -p gpurun_out/r03_sync
SIZES="5000:10000 8000:9000 7000:6000 3000:10000 2000:5000 8192:4200"
SFM_SCORE_SYNC=0 OUT=gpurun_out/r03_sync/ref.npz timeout -k 5 200 python -u tools/ab_small_score.py $SIZES 2>&1 | grep "us/pass" | sed "s/^/sync 0: /" | tee gpurun_out/r03_sync/timing.log
for sync in 1 2 4 8 16; do
SFM_SCORE_SYNC=$sync OUT=gpurun_out/r03_sync/new.npz timeout -k 5 200 python -u tools/ab_small_score.py $SIZES > gpurun_out/r03_sync/new_$sync.log 2>&1; grep "us/pass" gpurun_out/r03_sync/new_$sync.log | sed "s/^/sync $sync: /" | tee -a gpurun_out/r03_sync/timing.log
grep -q "Memory access fault" gpurun_out/r03_sync/new_$sync.log && exit 9
python tools/ab_small_score.py --compare gpurun_out/r03_sync/new.npz gpurun_out/r03_sync/ref.npz | grep -v same; [ ${PIPESTATUS[0]} -eq 0 ] || echo "MISMATCH at sync $sync"
done
