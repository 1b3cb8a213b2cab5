#!/bin/bash
# A/B of the exact tier's division: two IEEE divisions (SFM_SED_EXACT_DIVISION=1) against one refined reciprocal (default), same box.
OUT=gpurun_out/${1:-r04_fastdiv}; mkdir -p $OUT
for flags in "-DSFM_SED_EXACT_DIVISION=1" ""; do
  export SFM_EXTRA_HIPCC_FLAGS="$flags"
  python3 -c "from structure_from_motion_amd import build; build.build(force=True)" > $OUT/build.log 2>&1 || exit 1
  for cfg in "50000 100000 1.5e-6 -" "50000 125000 1.5e-6 -" "20000 40000 1.5e-6 -" "50000 100000 1.5e-6 0" "5000 10000 1.5e-6 0"; do
    set -- $cfg
    if [ "$4" = "-" ]; then unset SFM_SCORE_MATRIX; else export SFM_SCORE_MATRIX=$4; fi
    echo "[$flags] $(N=$1 H=$2 THR=$3 REPS=15 timeout -k 10 300 python3 tools/r04/time_r03_lib.py 2>&1 | grep 'library' | tr '\n' '|')" | tee -a $OUT/fastdiv.txt
  done
done
unset SFM_EXTRA_HIPCC_FLAGS SFM_SCORE_MATRIX
python3 -c "from structure_from_motion_amd import build; build.build_all(force=True)" > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filtered or matrix or score or fused or pass" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
