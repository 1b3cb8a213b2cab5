"""Build recipe for libsfm_hip.so (hipcc, gfx950 only, in-tree)."""
import contextlib
import fcntl
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


def _extra_flags():
    return os.environ.get("SFM_EXTRA_HIPCC_FLAGS", "").split()   # experiment builds (e.g. -DSFM_SCORE_E_IN_VGPR=1)


def source_sha(sources, extra=None, headers=None) -> str:
    """Fingerprint of the translation units `sources` (names under csrc/), the headers they include (``headers=None``: every
    csrc/*.h and the public header) and the compiler flags (``extra=None``: the flags the in-tree library was built with) —
    what a committed counter record of one of their kernels is stamped with."""
    import hashlib

    if extra is None:
        extra = built_flags()
    h = hashlib.sha256()
    for name in list(sources) + (HEADERS if headers is None else list(headers)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    for flag in FLAGS + list(extra):
        h.update(flag.encode())
    return h.hexdigest()[:16]


def score_source_sha(extra=None) -> str:
    """Fingerprint of what the scoring kernels are compiled from — sfm_score.hip, every csrc/*.h it can include, the
    public header and the compiler flags INCLUDING the extra flags of an experiment build; profiles/score_traffic.json
    is stamped with it so that bench.py can tell when the committed PMC counters were taken on another kernel.
    ``extra=None`` reads the flags the in-tree library was built with (its stamp file), so a bench running an experiment
    build reports that build's fingerprint, not the default's."""
    import hashlib

    if extra is None:
        extra = built_flags()
    h = hashlib.sha256()
    for name in ["sfm_score.hip"] + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    for flag in FLAGS + list(extra):
        h.update(flag.encode())
    return h.hexdigest()[:16]


STAMP_PATH = os.path.join(CSRC, "libsfm_hip.flags")


def built_flags():
    """The extra hipcc flags the in-tree libsfm_hip.so was built with ([] for the default build or an unstamped one)."""
    try:
        with open(STAMP_PATH) as f:
            return f.read().split()
    except OSError:
        return []


OPS_LIB_PATH = os.path.join(CSRC, "libsfm_torch_ops.so")
OPS_SOURCE = os.path.join(CSRC, "sfm_torch_ops.cpp")


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout: the ranks of a multi-GPU launch import the package together, and a stale
    library must be rebuilt once, not by every rank into the same file."""
    fd = None
    try:
        fd = os.open(os.path.join(CSRC, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
        fcntl.flock(fd, fcntl.LOCK_EX)
    except OSError:
        fd = fd if fd is None else (os.close(fd) or None)   # read-only tree: build unlocked
    try:
        yield
    finally:
        if fd is not None:
            fcntl.flock(fd, fcntl.LOCK_UN)
            os.close(fd)


def _compile(cmd, target: str, verbose: bool) -> None:
    """Run the compiler into a private file next to `target`, then rename: readers never see a half-written library."""
    tmp = f"{target}.tmp{os.getpid()}"
    cmd = [tmp if c == target else c for c in cmd]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def build_ops(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/sfm_torch_ops.cpp — TORCH_LIBRARY(sfm_hip): the PyTorch-ROCm custom-op registration above the
    C ABI — into csrc/libsfm_torch_ops.so.  Host code only (g++), linked against libsfm_hip.so and torch's libraries."""
    deps = [OPS_SOURCE, os.path.join(CSRC, "..", "..", "include", "sfm_hip.h"), os.path.abspath(__file__)]

    def fresh() -> bool:
        return os.path.exists(OPS_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(OPS_LIB_PATH) for d in deps)

    if not force and fresh():
        return OPS_LIB_PATH
    import torch

    troot = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           # torch's ROCm headers are the hipified ones: they expect the platform macros of a HIP build
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{troot}/include", f"-I{troot}/include/torch/csrc/api/include", f"-I{rocm}/include",
           OPS_SOURCE, "-o", OPS_LIB_PATH, f"-L{CSRC}", "-l:libsfm_hip.so", "-Wl,-rpath,$ORIGIN",
           f"-L{troot}/lib", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch"]
    with _build_lock():
        if force or not fresh():   # another process may have built it while this one waited
            _compile(cmd, OPS_LIB_PATH, verbose)
    return OPS_LIB_PATH


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > built for d in deps):
        return True
    return built_flags() != _extra_flags()   # an experiment build left behind (or wanted now): the flags differ


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the kernels + C ABI into csrc/libsfm_hip.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = _extra_flags()
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    with _build_lock():
        if force or _stale():   # another process may have built it while this one waited
            _compile(cmd, LIB_PATH, verbose)
            with open(STAMP_PATH, "w") as f:   # what this library was built with: _stale() and score_source_sha() read it
                f.write(" ".join(extra))
    return LIB_PATH


HOSTFAST_PATH = os.path.join(CSRC, "_sfm_hostfast.so")
HOSTFAST_SOURCE = os.path.join(CSRC, "hostfast.c")


def build_hostfast(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/hostfast.c — the CPython helper for the bulk Feature-list <-> array conversions of the drop-in
    boundary (host code only, plain gcc) — into csrc/_sfm_hostfast.so."""
    import sysconfig

    def fresh() -> bool:
        return os.path.exists(HOSTFAST_PATH) and os.path.getmtime(HOSTFAST_SOURCE) <= os.path.getmtime(HOSTFAST_PATH)

    if not force and fresh():
        return HOSTFAST_PATH
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", f"-I{sysconfig.get_paths()['include']}",
           HOSTFAST_SOURCE, "-o", HOSTFAST_PATH]
    with _build_lock():
        if force or not fresh():
            _compile(cmd, HOSTFAST_PATH, verbose)
    return HOSTFAST_PATH


def build_all(force: bool = False, verbose: bool = False):
    """libsfm_hip.so (kernels + C ABI), libsfm_torch_ops.so (torch custom ops above it), _sfm_hostfast.so (host helper)."""
    return build(force, verbose), build_ops(force, verbose), build_hostfast(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
