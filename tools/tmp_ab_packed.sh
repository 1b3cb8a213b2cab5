mkdir -p gpurun_out/r03_packed
for rep in 1 2; do
python tools/time_score.py 2>&1 | grep -v amdgpu.ids | sed "s/^/plain: /"
done | tee gpurun_out/r03_packed/plain.log
SFM_EXTRA_HIPCC_FLAGS=-DSFM_SCORE_PACKED=1 python -m structure_from_motion_amd.build > gpurun_out/r03_packed/build.log 2>&1 || { tail gpurun_out/r03_packed/build.log; exit 1; }
export SFM_EXTRA_HIPCC_FLAGS=-DSFM_SCORE_PACKED=1
for rep in 1 2; do
python tools/time_score.py 2>&1 | grep -v amdgpu.ids | sed "s/^/packed: /"
done | tee gpurun_out/r03_packed/packed.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -k "filtered_score or fused_small_pass or full_size_properties" 2>&1 | tail -5 | tee gpurun_out/r03_packed/parity.log
python tools/time_c5.py 2>&1 | grep -v amdgpu.ids | sed "s/^/packed: /" | tee -a gpurun_out/r03_packed/packed.log
N=5000 H=10000 python tools/time_small_pass.py 2>&1 | grep -v amdgpu.ids | sed "s/^/packed: /" | tee -a gpurun_out/r03_packed/packed.log
N=8000 H=30000 python tools/time_small_pass.py 2>&1 | grep -v amdgpu.ids | sed "s/^/packed: /" | tee -a gpurun_out/r03_packed/packed.log
