"""Build recipe for libsfm_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsfm_hip.so")
SOURCES = ["sfm_kernels.hip", "sfm_score.hip", "sfm_refine.hip", "sfm_pose.hip", "sfm_shard.hip", "sfm_match.hip", "sfm_harris.hip", "pyshuffle.cpp"]
# every header a translation unit can include: all of csrc/*.h plus the public C ABI header
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "sfm_hip.h")]
# -ffp-contract=off: multiply/add round separately (parity with the NumPy elementwise semantics of the
# reference); fused ops appear only where the source spells fma().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def score_source_sha() -> str:
    """Fingerprint of the sources the scoring kernels are compiled from; profiles/score_traffic.json is stamped
    with it so that bench.py can tell when the committed PMC counters were taken on an older kernel."""
    import hashlib

    h = hashlib.sha256()
    for name in ("sfm_score.hip", "sfm_math.h", "sfm_common.h"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    for flag in FLAGS:
        h.update(flag.encode())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the kernels + C ABI into csrc/libsfm_hip.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
