"""Round-2 experiment: tier 1 of the scoring kernel on the matrix pipe (tools/micro/mfma_filter.hip) against the
production VALU test, both without compaction / tier 2.  Run on the GPU box:

    hipcc ... tools/micro/mfma_filter.hip -o tools/micro/bin/libmfma_filter.so   (see tools/micro/build.sh)
    python3 tools/time_mfma_filter.py

Prints kernel times and checks that neither filter ever rejects a pair whose fp64 SED is <= thr."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from structure_from_motion_amd import device as dev, synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "micro", "bin", "libmfma_filter.so"))
lib.mfma_table_floats.restype = C.c_int64
lib.mfma_table_floats.argtypes = [C.c_int64]
P, I64, D = C.c_void_p, C.c_int64, C.c_double
lib.filter_prepare.argtypes = [P, I64, D, P, P, P]
lib.valu_filter_count.argtypes = [P, I64, P, I64, D, P, P]
lib.mfma_filter_count.argtypes = [P, P, I64, P, I64, D, P, P, I64, P]

n, h, thr = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000)), float(os.environ.get("THR", 1.5e-6))
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)
table = torch.zeros((lib.mfma_table_floats(n),), dtype=torch.float32, device=corr.device)
st = torch.cuda.current_stream().cuda_stream
assert lib.filter_prepare(corr.data_ptr(), n, thr, ws.data_ptr(), table.data_ptr(), st) == 0
surv_v = torch.zeros((h,), dtype=torch.int32, device=corr.device)
surv_m = torch.zeros((h,), dtype=torch.int32, device=corr.device)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_v = timed(lambda: lib.valu_filter_count(ws.data_ptr(), n, E.data_ptr(), h, thr, surv_v.data_ptr(), st))
t_m = timed(lambda: lib.mfma_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, 0, st))
t_o = timed(lambda: lib.mfma_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, -1, st))
print(f"MFMA chain alone (8 x v_mfma_f32_32x32x2_f32 per 32 x 32 tile, operand loads, one compare): {t_o:.3f} ms")
lib.mfma_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, 0, st)
cnt, s1, s2 = dev.score_sed(corr, E, S, thr, exact_only=True)
cnt = cnt[0].cpu().numpy().astype(np.int64)
sv, sm = surv_v.cpu().numpy().astype(np.int64), surv_m.cpu().numpy().astype(np.int64)
print(f"N={n} H={h}: VALU tier 1 only {t_v:.3f} ms ({n * h / t_v / 1e6:.0f} G evals/s), MFMA tier 1 only {t_m:.3f} ms "
      f"({n * h / t_m / 1e6:.0f} G evals/s), ratio {t_v / t_m:.2f}x")
print(f"survivors / true inliers: VALU {sv.sum() / (cnt.sum() + 8 * h):.4f}, MFMA {sm.sum() / (cnt.sum() + 8 * h):.4f}")
# the 8 sample points are inliers of their own model but not counted in cnt; a sample may repeat no point, so
# survivors >= cnt + (sample points surviving) >= cnt
print("hypotheses with survivors < inliers: VALU", int((sv < cnt).sum()), " MFMA", int((sm < cnt).sum()))
assert np.all(sv >= cnt), "VALU filter rejected an inlier"
assert np.all(sm >= cnt), "MFMA filter rejected an inlier"
# per-point containment on the first hypotheses: every exact inlier (sed <= thr) must survive the MFMA filter
mh = 96
tiles = (n + 31) // 32
masks = torch.zeros((mh, tiles), dtype=torch.int64, device=corr.device)
assert lib.mfma_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), masks.data_ptr(), mh, st) == 0
mk = masks.cpu().numpy().astype(np.uint64)
bits = ((mk[:, :, None] >> np.arange(32, dtype=np.uint64)[None, None, :]) & np.uint64(1)).reshape(mh, tiles * 32)[:, :n].astype(bool)
worst = 0
for k in range(mh):
    sed = dev.sed_values(corr[0], E[0, k]).cpu().numpy()
    inl = sed <= thr
    assert np.all(bits[k][inl]), f"hypothesis {k}: MFMA filter rejected an exact inlier"
    assert bits[k].sum() == surv_m[k].item(), (k, bits[k].sum(), surv_m[k].item())
    worst = max(worst, bits[k].sum() - inl.sum())
print(f"containment ok on {mh} hypotheses (layout check: mask populations == counts); max extra survivors {worst}")
