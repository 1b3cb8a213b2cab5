"""The reference's OWN test modules, unmodified and read where they lie, run against this package.

Only in the build container: `/root/reference` does not travel, so the test is skipped wherever it is absent (the GPU
box).  Each module's source is executed at run time in a fresh interpreter whose `lib` package is this repo's
(`lib/` re-exports `structure_from_motion_amd`); nothing of it is stored here.  Host-only modules must pass outright —
they are the reference's statement of the drop-in contract for `lib.ransac` (generic callables), `lib.feature_matching`
(matching / util), `lib.blur`, `lib.data_utils`.  Modules whose functions are HIP kernels here (`lib.common.correlate`,
`ncc`, `ssd`) cannot compute without a GPU: there they must fail with this package's loud "no CPU fallback" error and
nothing else; their cases are restated for the GPU in tests/test_gpu_matching.py and tests/test_gpu_harris.py.
`lib/epipolar/tests/test_epipolar.py` and `lib/harris/tests/test_harris_detector.py` need cv2 / tkinter, which this image
does not have: their cases are restated in tests/test_gpu_api.py.
"""
import json
import os
import subprocess
import sys

import pytest

REFERENCE = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOST_ONLY = {
    "lib/blur/tests/test_gaussian.py": 2,
    "lib/data_utils/tests/test_middlebury_utils.py": 1,
    "lib/feature_matching/tests/test_matching.py": 2,
    "lib/feature_matching/tests/test_util.py": 1,
    "lib/ransac/tests/test_ransac.py": 2,
}
DEVICE_BACKED = {
    "lib/common/tests/test_correlate.py": 2,
    "lib/feature_matching/tests/test_ncc.py": 3,
    "lib/feature_matching/tests/test_ssd.py": 1,
}

RUNNER = r'''
import inspect, json, os, sys, types, unittest
repo, reference, modules = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, repo)
import lib
assert lib.__file__.startswith(repo), lib.__file__
report = {}
for rel in modules:
    path = os.path.join(reference, rel)
    module = types.ModuleType("reference_" + os.path.basename(rel)[:-3])
    module.__file__ = path
    module.__package__ = os.path.dirname(rel).replace("/", ".")   # relative imports resolve inside THIS repo's lib
    sys.modules[module.__name__] = module
    exec(compile(open(path).read(), path, "exec"), module.__dict__)
    result = unittest.TestResult()
    unittest.defaultTestLoader.loadTestsFromModule(module).run(result)
    passed = result.testsRun - len(result.failures) - len(result.errors)
    problems = [trace.strip().splitlines()[-1] for _, trace in result.failures + result.errors]
    for name, fn in list(module.__dict__.items()):
        if name.startswith("test_") and inspect.isfunction(fn):   # pytest-style functions (hypothesis-wrapped too)
            try:
                fn()
                passed += 1
            except Exception as exc:  # noqa: BLE001 - reported to the parent
                problems.append(f"{type(exc).__name__}: {exc}")
    report[rel] = {"passed": passed, "problems": problems}
assert sys.modules["lib"].__file__.startswith(repo)
print("REPORT " + json.dumps(report))
'''


def run_reference_modules(modules):
    env = dict(os.environ, MPLBACKEND="Agg")
    out = subprocess.run([sys.executable, "-c", RUNNER, REPO, REFERENCE, *modules], capture_output=True, text=True,
                         cwd=REPO, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("REPORT ")][-1]
    return json.loads(line[len("REPORT "):])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is only present in the build container")
def test_reference_host_side_tests_pass_unmodified():
    report = run_reference_modules(sorted(HOST_ONLY))
    for rel, expected in HOST_ONLY.items():
        assert report[rel]["problems"] == [], (rel, report[rel])
        assert report[rel]["passed"] == expected, (rel, report[rel])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is only present in the build container")
def test_reference_device_backed_tests_fail_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: these modules compute")
    report = run_reference_modules(sorted(DEVICE_BACKED))
    for rel, cases in DEVICE_BACKED.items():
        assert report[rel]["passed"] == 0 and len(report[rel]["problems"]) == cases, (rel, report[rel])
        assert all("no CPU fallback" in p for p in report[rel]["problems"]), (rel, report[rel])
