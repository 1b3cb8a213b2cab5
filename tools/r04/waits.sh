#!/bin/bash
# What the waves of the matrix-pipe kernel wait on: SQ wait / active breakdown (quad-cycles), one rocprofv3 --pmc pass per
# group (<= 8 SQ counters), tier 1 alone (THR=1e-14) and the bench threshold.   tools/r04/waits.sh gpurun_out/<dir>
OUT=${1:?out dir}; REPO=$(pwd); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp REPS=3
cd /tmp
for thr in 1e-14 1.5e-6; do
  export THR=$thr
  for g in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "b SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "c SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE SQ_CYCLES"; do
    set -- $g; name=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${thr}_$name" -o p -- python3 "$REPO/tools/ab_matrix_score.py" > "$OUT/${thr}_$name.log" 2>&1 || echo "pass $thr $name failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
rows = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    tag = os.path.relpath(f, out).split(os.sep)[0].rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "score_sed_matrix_kernel<false>" not in k:
            continue
        rows.setdefault((tag, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "waits.csv"), "w") as f:
    f.write("threshold,counter,launches,mean_per_launch\n")
    for (tag, c), v in sorted(rows.items()):
        f.write(f"{tag},{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(open(os.path.join(out, "waits.csv")).read())
PY
