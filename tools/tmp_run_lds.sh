mkdir -p gpurun_out/r03_lds2
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
SIZES="300:2000 600:300 5000:10000"
SFM_SPLIT_UNITS=0 SFM_SMALL_SCORE=stream OUT=gpurun_out/r03_lds2/ref0.npz STEPS=20 timeout -k 5 90 python -u tools/ab_small_score.py $SIZES > gpurun_out/r03_lds2/ref0.log 2>&1; grep "us/pass" gpurun_out/r03_lds2/ref0.log
for hpw in 2 1 4; do
SFM_LDS_HPW=$hpw SFM_SMALL_SCORE=lds OUT=gpurun_out/r03_lds2/new0.npz STEPS=20 timeout -k 5 90 python -u tools/ab_small_score.py $SIZES > gpurun_out/r03_lds2/new0_$hpw.log 2>&1; echo "hpw $hpw rc=$?"; grep "us/pass" gpurun_out/r03_lds2/new0_$hpw.log; fault gpurun_out/r03_lds2/new0_$hpw.log
python tools/ab_small_score.py --compare gpurun_out/r03_lds2/new0.npz gpurun_out/r03_lds2/ref0.npz | grep -v same; [ ${PIPESTATUS[0]} -eq 0 ] || exit 3
done
SIZES="5000:10000 8000:30000 8000:9000 2000:5000 1000:32768 8192:32768 64:4097 777:5 5000:20000 3000:10000 1001:11000 450:10241 7000:6000"
SFM_SPLIT_UNITS=0 SFM_SMALL_SCORE=stream OUT=gpurun_out/r03_lds2/ref.npz timeout -k 5 200 python -u tools/ab_small_score.py $SIZES 2>&1 | grep "us/pass" | tee gpurun_out/r03_lds2/ref.log
for hpw in 2 1 4; do
SFM_LDS_HPW=$hpw SFM_SMALL_SCORE=lds OUT=gpurun_out/r03_lds2/new.npz timeout -k 5 200 python -u tools/ab_small_score.py $SIZES > gpurun_out/r03_lds2/new_$hpw.log 2>&1; echo "hpw $hpw rc=$?"; grep "us/pass\|rror" gpurun_out/r03_lds2/new_$hpw.log; fault gpurun_out/r03_lds2/new_$hpw.log
python tools/ab_small_score.py --compare gpurun_out/r03_lds2/new.npz gpurun_out/r03_lds2/ref.npz | grep -v same; echo "compare rc=${PIPESTATUS[0]}"
done
for v in "SFM_LDS_HPW=2 SFM_LDS_UNITS=1" "SFM_LDS_HPW=2 SFM_LDS_UNITS=2" "SFM_LDS_HPW=2 SFM_LDS_UNITS=3" "SFM_LDS_HPW=2 SFM_LDS_UNITS=4" "SFM_LDS_HPW=4 SFM_LDS_UNITS=2" "SFM_LDS_HPW=4 SFM_LDS_UNITS=4" "SFM_LDS_HPW=4 SFM_LDS_UNITS=6"; do
  env $v SFM_SMALL_SCORE=lds timeout -k 5 100 python -u tools/ab_small_score.py 5000:10000 8000:30000 2>&1 | grep "us/pass" | sed "s/^/$v: /"
done | tee gpurun_out/r03_lds2/variants.log
