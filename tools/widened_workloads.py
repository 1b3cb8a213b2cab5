"""The workloads of the widened rows of SURVEY.md §8f as plain callables, shared by bench.py (its `configs` block: kernel time by
HIP events around the library calls) and tools/run_config.py (the same loops under rocprofv3 for the committed counters):

  f1  brute-force window matcher (reference lib/feature_matching/matching.py:36-118): 20 000 x 20 000 features NCC-9 through the
      fused summaries (sfm_match_summary: pair_summary_kernel + summary_combine_kernel), and the demo's 600 x 600
  f2  Harris detector (reference lib/harris/harris_detector.py:11-113): VGA and 1080p, 600 corners
  f4  local optimisation (no reference counterpart): refine_kernel, one round at N = 50 000
  pose  the tail of BASELINE configs[4]: cheirality of 4 poses, vote and triangulation for 256 pairs x 10 000 correspondences
      (reference eight_point.py:181-242,449-488; triangulation.py:42-62)

Each builder returns (run, info): run() enqueues ONE pass on the current stream; info describes the workload."""
import numpy as np
import torch

from structure_from_motion_amd import _native, batched, device, synthetic
from structure_from_motion_amd._native import AGG_RMS, MATCH_NCC

THR, MIN_EXTRA = 1.5e-6, 10


def photo_like(height, width, seed):
    """A smooth textured uint8 image (white noise blurred twice by a 1-2-1 kernel: neighbouring windows correlate)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(height, width)).astype(np.float64)
    for _ in range(2):
        img[1:-1] = (img[:-2] + 2.0 * img[1:-1] + img[2:]) / 4.0
        img[:, 1:-1] = (img[:, :-2] + 2.0 * img[:, 1:-1] + img[:, 2:]) / 4.0
    return np.clip(img, 0, 255).astype(np.uint8)


def matcher(n_a, n_b, window=9, seed=3):
    """sfm_match_summary on patches extracted once (the |A| x |B| score loop of matching.py:55-65 + heap summaries)."""
    from structure_from_motion_amd.feature_matching import _device_match

    lib = _native.load()
    H, W = 480, 640
    ia, ib = photo_like(H, W, seed), photo_like(H, W, seed + 1)
    rng = np.random.default_rng(seed)
    fa = np.column_stack([rng.integers(0, W, n_a), rng.integers(0, H, n_a)]).astype(np.float64)
    fb = np.column_stack([rng.integers(0, W, n_b), rng.integers(0, H, n_b)]).astype(np.float64)
    (pa, qa, oka, nA), (pb, qb, okb, nB), K, metric = _device_match._extract_patches(MATCH_NCC, ia, ib, device.to_device(fa),
                                                                                     device.to_device(fb), window)
    dev = pa.device
    best = torch.empty((nA,), dtype=torch.float64, device=dev)
    arg = torch.empty((nA,), dtype=torch.int32, device=dev)
    second = torch.empty((nA,), dtype=torch.float64, device=dev)
    ws_bytes = int(lib.sfm_match_summary_workspace_bytes(nA, nB))
    ws = torch.empty((ws_bytes // 8,), dtype=torch.float64, device=dev)

    def run():
        _native.check(lib.sfm_match_summary(metric, pa.data_ptr(), pa.shape[1], pb.data_ptr(), pb.shape[1], qa.data_ptr(), qb.data_ptr(),
                                            oka.data_ptr(), okb.data_ptr(), nA, nB, K, ws.data_ptr(), ws_bytes, best.data_ptr(),
                                            arg.data_ptr(), second.data_ptr(), device._stream()), "sfm_match_summary")

    return run, {"features_a": nA, "features_b": nB, "window_elements": K, "pairs": float(nA) * nB,
                 "kernel": "pair_summary_kernel<NCC, LDS-DMA staging> + summary_combine_kernel",
                 # the window sum: K multiplies + K adds per pair, separately rounded (bit-exact against the oracle: no FMA)
                 "fp64_insts_per_pair": 2 * K}


def harris(height, width, corners=600, seed=5):
    from structure_from_motion_amd.harris import harris_detector

    image = photo_like(height, width, seed)

    def run():
        return harris_detector.detect_harris_corners(image, num_corners=corners)   # (uploads the image, reads the corners back)

    return run, {"height": height, "width": width, "corners": corners, "pixels": float(height) * width,
                 "kernel": "correlate_kernel x 2 + cornerness_kernel + nms_round_kernel x 12 + nms_finalize + compact_nonzero + prune_histogram x 2 + prune_filter"}


def refine(n=50_000, h=2_000, rounds=1):
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K).reshape(1, n, 4)
    ws = device.RansacWorkspace(1, n, h)
    ws.run(corr, THR, n // 10, AGG_RMS, philox=(5, 0, 1))   # (a gate of n / 10 extra inliers: the winner is a model of the scene, ~28 000 inliers)
    best = ws.result[:, 1].clamp(min=0)
    E = ws.E[torch.arange(1, device=best.device), best].contiguous()
    err = ws.result.view(torch.float64)[:, 2].contiguous()
    inliers = int((ws.mask[0] != 0).sum())
    state = {}

    def run():
        state["out"] = device.refine_inliers(corr, E, ws.mask, err, THR, AGG_RMS, rounds)

    return run, {"matches": n, "rounds": rounds, "inliers_in": inliers, "kernel": "refine_kernel (one 512-thread block per pair)"}


def pose_tail(batch=256, n=10_000, h=2_000):
    """The three launches behind the RANSAC pass of BASELINE configs[4], on the winners of a real batched pass."""
    import ctypes as C

    lib = _native.load()
    base = [synthetic.two_view_scene(n, seed=300 + b, outlier_fraction=0.25) for b in range(16)]
    pix_a = device.to_device(np.stack([base[b % 16][0] for b in range(batch)]))
    pix_b = device.to_device(np.stack([base[b % 16][1] for b in range(batch)]))
    pipe = batched.TwoViewBatch(batch, n, h)
    pipe.run(pix_a, pix_b, base[0][2], seed=70, thr=THR, min_extra=MIN_EXTRA, aggregation=AGG_RMS)
    torch.cuda.synchronize()
    K = np.ascontiguousarray(base[0][2], dtype=np.float64)
    inliers = int((pipe.ws.mask != 0).sum())

    def cheirality():
        _native.check(lib.sfm_cheirality_batched(pipe.corr.data_ptr(), n, batch, pipe.poses.data_ptr(), pipe.ws.mask.data_ptr(), 50.0,
                                                 pipe.passes.data_ptr(), device._stream()), "sfm_cheirality_batched")

    def vote():
        _native.check(lib.sfm_pose_vote(pipe.passes.data_ptr(), n, batch, pipe.skip.data_ptr(), pipe.votes.data_ptr(),
                                        pipe.best_pose.data_ptr(), device._stream()), "sfm_pose_vote")

    def triangulate():
        _native.check(lib.sfm_triangulate_selected(pix_a.data_ptr(), pix_b.data_ptr(), n, batch, K.ctypes.data_as(C.c_void_p),
                                                   pipe.poses.data_ptr(), pipe.best_pose.data_ptr(), pipe.passes.data_ptr(),
                                                   pipe.X.data_ptr(), pipe.valid.data_ptr(), device._stream()),
                      "sfm_triangulate_selected")

    def run():
        cheirality()
        vote()
        triangulate()

    info = {"pairs": batch, "matches": n, "inliers": inliers, "keepalive": (pipe, pix_a, pix_b, K),
            "stages": {"cheirality_batched_kernel": cheirality, "pose_vote_kernel": vote, "triangulate_selected_kernel": triangulate},
            # one DLT solve (4 x 4 null vector by Householder QR + inverse iteration) per inlier and antipodal pose pair, and one
            # per triangulated point: ~420 fp64 instructions each as compiled
            "dlt_solves": {"cheirality_batched_kernel": 2.0 * inliers}}
    return run, info


BUILDERS = {
    "f1_match_20000x20000_ncc9": lambda: matcher(20_000, 20_000, 9),
    "f1_match_600x600_ncc9": lambda: matcher(600, 600, 9),
    "f2_harris_vga": lambda: harris(480, 640),
    "f2_harris_1080p": lambda: harris(1080, 1920),
    "f4_refine_50000": lambda: refine(50_000),
    "pose_tail_c5": lambda: pose_tail(),
}
