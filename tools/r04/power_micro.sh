#!/bin/bash
# Socket power while each piece of a tier-1 step (and the exact tier's arithmetic) runs alone at 4 waves per SIMD: the micro-benchmark's
# POWER mode prints a wall-clock window per variant, this script samples rocm-smi beside it and averages the samples inside each window.
OUT=gpurun_out/${1:-r04_power_micro}; mkdir -p $OUT
SECS=${2:-5}
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed -e 's/.*: //' | tr '\n' ' ')"; done ) > $OUT/samples.txt &
SAMPLER=$!
timeout -k 10 300 tools/micro/bin/matrix_step_rates power $SECS > $OUT/windows.txt 2>&1
kill $SAMPLER
python3 - $OUT <<'PY'
import sys, re
out = sys.argv[1]
samples = []
for line in open(out + "/samples.txt"):
    f = line.split()
    try:
        t = float(f[0]); watts = [float(x) for x in f[1:] if re.fullmatch(r"\d+\.\d+", x)]
        mhz = [int(m) for m in re.findall(r"\((\d+)Mhz\)", line)]
        if watts: samples.append((t, watts[-1], mhz[0] if mhz else 0))
    except Exception:
        pass
print(f"{len(samples)} power samples")
for line in open(out + "/windows.txt"):
    m = re.match(r"POWER (.*?)\s+window (\d+\.\d+) (\d+\.\d+)(.*)", line)
    if not m: continue
    name, a, b, rest = m.group(1), float(m.group(2)), float(m.group(3)), m.group(4)
    inside = [(w, c) for t, w, c in samples if a + 1.5 <= t <= b - 0.2]   # (the first 1.5 s: the reading is an average over a window)
    if inside:
        print(f"{name:42s} {sum(w for w, _ in inside) / len(inside):7.0f} W (n={len(inside)}, max {max(w for w, _ in inside):.0f}), smi sclk {sum(c for _, c in inside) / len(inside):.0f} MHz {rest.strip()}")
    else:
        print(f"{name:42s} no samples {rest.strip()}")
PY
