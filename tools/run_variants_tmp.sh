cd /root/repo
V=structure_from_motion_amd/csrc/variants
L=structure_from_motion_amd/csrc/libsfm_hip.so
cp $L /tmp/keep.so
cp $V/stats.so $L
REPS=3 timeout -k 10 120 python3 -u tools/ab_matrix_score.py 2>&1 | grep -E "utilisation"
cp /tmp/keep.so $L
for cfg in "50000 100000" "50000 20000" "50000 125000" "20000 40000"; do
  set -- $cfg
  for m in 0 1; do
    N=$1 H=$2 SFM_SCORE_MATRIX=$m REPS=9 timeout -k 10 120 python3 -u tools/ab_matrix_score.py 2>&1 | grep -v amdgpu.ids
  done
done
