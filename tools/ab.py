"""Same-box A/B runner (round 5; replaces the one-off shell scripts of tools/r04/).

Every comparison cited under profiles/r05/ is of TREES: complete source trees of this repository, each with its own built
libraries, kept under tools/tmp/trees/<name>/ (git-ignored, but they travel to the GPU box with the snapshot).  Nothing is ever
built over the in-tree libraries, so an interrupted run leaves the product build untouched, and a baseline is reproducible from
a commit id.

  in the build container (no GPU):
    python tools/ab.py prepare NAME [--commit REV] [--flags "-DSFM_X=1 ..."]
        NAME from REV (default: the working tree as it is now), compiled with the extra hipcc flags
  on the GPU box (through gpurun):
    python tools/ab.py run --trees A,B[,C] --configs c3,c5 [--steps 50] [--reps 3] [--options k=v,...] [--out FILE]
        alternates A, B, C, A, B, C ... (one child process per measurement, tools/steptime.py of THIS tree importing from the
        named one), prints and writes the table: median / min ms per pass per (config, tree)
    python tools/ab.py profile --tree A --config c3 [--steps 20] --out DIR
        rocprofv3 --kernel-trace --stats around the same loop; leaves DIR/<tree>_<config>_kernel_stats.csv and prints the
        per-kernel averages (the profiler starts the program itself: no shell, no env wrapper between them)
"""
import argparse
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THR = None
TREES = os.path.join(REPO, "tools", "tmp", "trees")
KEEP = ("structure_from_motion_amd", "include", "lib", "oracle", "apps")   # what a tree needs to be imported and built


def tree_dir(name):
    return REPO if name in ("here", ".") else os.path.join(TREES, name)


def prepare(args):
    dst = tree_dir(args.name)
    if dst == REPO:
        raise SystemExit("'here' is the working tree itself")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    if args.commit:
        tar = subprocess.Popen(["git", "-C", REPO, "archive", args.commit] + list(KEEP), stdout=subprocess.PIPE)
        subprocess.run(["tar", "-x", "-C", dst], stdin=tar.stdout, check=True)
        if tar.wait() != 0:
            raise SystemExit(f"git archive {args.commit} failed")
    else:
        for d in KEEP:
            shutil.copytree(os.path.join(REPO, d), os.path.join(dst, d),
                            ignore=shutil.ignore_patterns("*.so", "*.o", "__pycache__", "*.flags", "_ref", "*.tmp*"))
    env = dict(os.environ, SFM_EXTRA_HIPCC_FLAGS=args.flags or "")
    subprocess.run([sys.executable, "-m", "structure_from_motion_amd.build"], cwd=dst, env=env, check=True,
                   stdout=subprocess.DEVNULL)
    with open(os.path.join(dst, "TREE.json"), "w") as f:
        rev = subprocess.run(["git", "-C", REPO, "rev-parse", args.commit or "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", REPO, "status", "--porcelain"], capture_output=True, text=True).stdout.strip())
        json.dump({"name": args.name, "commit": rev, "from_working_tree": not args.commit, "working_tree_dirty": dirty and not args.commit,
                   "flags": args.flags or ""}, f)
    print("prepared", dst)


def describe(name):
    try:
        meta = json.load(open(os.path.join(tree_dir(name), "TREE.json")))
        return f"{meta['commit'][:9]}{'+wt' if meta['from_working_tree'] else ''} [{meta['flags']}]"
    except OSError:
        return "working tree" if tree_dir(name) == REPO else "?"


def measure(tree, config, steps, options):
    cmd = [sys.executable, os.path.join(REPO, "tools", "steptime.py"), "--config", config, "--tree", tree_dir(tree), "--steps", str(steps)]
    if options:
        cmd += ["--options", options]
    if THR is not None:
        cmd += ["--thr", THR]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    for line in done.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{tree} {config}: no result\n{done.stdout[-2000:]}\n{done.stderr[-4000:]}")


def run(args):
    trees = args.trees.split(",")
    lines = [f"same-box A/B: trees {', '.join(f'{t} = {describe(t)}' for t in trees)}; {args.steps} passes per measurement, "
             f"{args.reps} alternating repetitions; options '{args.options}'"]
    for config in args.configs.split(","):
        results = {t: [] for t in trees}
        for rep in range(args.reps):
            for t in trees:
                r = measure(t, config, args.steps, args.options)
                results[t].append(r)
                print(f"  {config} {t} rep {rep}: {r['ms_per_pass']:.4f} ms  {({k: v for k, v in r.items() if k in ('best_h', 'inliers', 'pairs_ok')})}",
                      flush=True)
        base = statistics.median(r["ms_per_pass"] for r in results[trees[0]])
        for t in trees:
            ms = [r["ms_per_pass"] for r in results[t]]
            med = statistics.median(ms)
            lines.append(f"{config:5s} {t:24s} median {med:8.4f} ms  min {min(ms):8.4f}  ({(med / base - 1) * 100:+5.1f} % vs {trees[0]})  "
                         f"outcome {({k: v for k, v in results[t][-1].items() if k in ('best_h', 'inliers', 'pairs_ok')})}")
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
        with open(args.out, "a") as f:
            f.write(text + "\n\n")


def profile(args):
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    work = os.path.join(out, f"{args.tree}_{args.config}_trace")
    shutil.rmtree(work, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", work, "-o", "p", "--", sys.executable,
           os.path.join(REPO, "tools", "steptime.py"), "--config", args.config, "--tree", tree_dir(args.tree), "--steps", str(args.steps)]
    if args.options:
        cmd += ["--options", args.options]
    if THR is not None:
        cmd += ["--thr", THR]
    env = dict(os.environ, TMPDIR="/tmp")
    done = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    stats = glob.glob(os.path.join(work, "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        raise SystemExit(f"no kernel stats\n{done.stdout[-2000:]}\n{done.stderr[-3000:]}")
    target = os.path.join(out, f"{args.tree}_{args.config}_kernel_stats.csv")
    shutil.copy(stats[0], target)
    shutil.rmtree(work, ignore_errors=True)
    rows = list(csv.DictReader(open(target)))
    passes = args.steps + 5
    total = 0.0
    print(f"{args.tree} ({describe(args.tree)}) {args.config}: per-kernel averages, {passes} passes (5 warm-up)")
    for r in rows:
        calls = int(r["Calls"])
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if calls < args.steps or "copyBuffer" in name:   # (copies: the uploads / read-backs around the loop, not per-pass work)
            continue
        per_pass = float(r["TotalDurationNs"]) / passes / 1e3
        total += per_pass
        print(f"  {name[-70:]:70s} {calls / passes:5.1f} x {float(r['AverageNs']) / 1e3:9.1f} us = {per_pass:9.1f} us per pass")
    print(f"  kernels per pass: {total:.1f} us;  {[l for l in done.stdout.splitlines() if l.startswith('{')][-1:]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("prepare")
    p.add_argument("name")
    p.add_argument("--commit", default=None)
    p.add_argument("--flags", default="")
    p = sub.add_parser("run")
    p.add_argument("--trees", required=True)
    p.add_argument("--configs", default="c3")
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--options", default="")
    p.add_argument("--out", default=None)
    p = sub.add_parser("profile")
    p.add_argument("--tree", default="here")
    p.add_argument("--config", default="c3")
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--options", default="")
    p.add_argument("--out", required=True)
    ap.add_argument("--thr", default=None, help="inlier threshold passed to tools/steptime.py (1e-14: tier 1 alone)")
    a = ap.parse_args()
    THR = a.thr
    {"prepare": prepare, "run": run, "profile": profile}[a.cmd](a)
