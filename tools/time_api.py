"""Wall time of the drop-in call (reference apps/sfm.py:110-119) by stage, at a chosen size.  N, H, SAMPLER from the
environment.  Shows where a call spends its time once the GPU pass itself is microseconds to milliseconds."""
import cProfile
import os
import pstats
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lib.common.feature import Feature  # noqa: E402
from lib.epipolar.eight_point import create_trivial_matches  # noqa: E402
from lib.epipolar.epipolar_ransac import estimate_essential_mat_with_ransac  # noqa: E402
from lib.ransac.ransac import ErrorAggregationMethod  # noqa: E402
from structure_from_motion_amd import synthetic  # noqa: E402

n, h = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000))
os.environ.setdefault("SFM_SAMPLER", "philox")
os.environ.setdefault("SFM_SEED", "5")
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
fa = [Feature(x=float(x), y=float(y)) for x, y in pa]
fb = [Feature(x=float(x), y=float(y)) for x, y in pb]
matches = create_trivial_matches(n)


def call():
    random.seed(5)
    return estimate_essential_mat_with_ransac(K, features_a=fa, features_b=fb, matches=matches, sed_inlier_threshold=1.5e-6,
                                              error_aggregation_method=ErrorAggregationMethod.RMS, min_num_extra_inliers=10,
                                              max_iterations=h)


call()
times = []
for _ in range(5):
    t0 = time.perf_counter()
    E, pairs = call()
    times.append((time.perf_counter() - t0) * 1e3)
print(f"n={n} h={h} sampler={os.environ['SFM_SAMPLER']}: {min(times):.2f} ms per call (inliers {len(pairs)})")
prof = cProfile.Profile()
prof.enable()
call()
prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(14)
