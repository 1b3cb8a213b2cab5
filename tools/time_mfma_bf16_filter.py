"""Round-3 experiment: tier 1 of the scoring kernel on the bf16 matrix pipe (tools/micro/mfma_bf16_filter.hip) against the
production VALU test, both without compaction / tier 2.  Run on the GPU box:

    tools/micro/build.sh && python3 tools/time_mfma_bf16_filter.py

Prints what one v_mfma_f32_32x32x16_bf16 does with its sixteen products (accumulation probe), kernel times, the survivor
ratios, and checks that the filter never rejects a pair whose fp64 SED is <= thr."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from structure_from_motion_amd import device as dev, synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "micro", "bin", "libmfma_bf16_filter.so"))
old = C.CDLL(os.path.join(HERE, "micro", "bin", "libmfma_filter.so"))
P, I64, D = C.c_void_p, C.c_int64, C.c_double
lib.bf16_table_bytes.restype = I64
lib.bf16_table_bytes.argtypes = [I64]
lib.bf16_filter_scale.restype = D
lib.bf16_filter_scale.argtypes = [D]
lib.bf16_prepare.argtypes = [P, I64, D, P, P, P]
lib.bf16_filter_count.argtypes = [P, P, I64, P, I64, D, P, P, I64, P]
lib.bf16_probe.argtypes = [P, P, C.c_float, P, C.c_int, P]
old.filter_prepare.argtypes = [P, I64, D, P, P, P]
old.valu_filter_count.argtypes = [P, I64, P, I64, D, P, P]
old.mfma_table_floats.restype = I64
old.mfma_table_floats.argtypes = [I64]

device = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def probe(a, b, c=0.0, half=0):
    ta = torch.tensor(a, dtype=torch.float32, device=device)
    tb = torch.tensor(b, dtype=torch.float32, device=device)
    out = torch.zeros((1,), dtype=torch.float32, device=device)
    assert lib.bf16_probe(ta.data_ptr(), tb.data_ptr(), c, out.data_ptr(), half, st) == 0
    torch.cuda.synchronize()
    return float(out.item())


ones = [1.0] * 16
cases = [
    ("2^23 + 15 x 1", [2.0 ** 23] + [1.0] * 15, ones, 0.0),
    ("15 x 1 + 2^23 (big term last)", [1.0] * 15 + [2.0 ** 23], ones, 0.0),
    ("2^24 + 15 x 1", [2.0 ** 24] + [1.0] * 15, ones, 0.0),
    ("2^26 + 15 x 3", [2.0 ** 26] + [3.0] * 15, ones, 0.0),
    ("2^30 + 15 x 64", [2.0 ** 30] + [64.0] * 15, ones, 0.0),
    ("c = 2^24, 16 x 1", ones, ones, 2.0 ** 24),
    ("c = 2^26, 16 x 3", [3.0] * 16, ones, 2.0 ** 26),
    ("2^20 - 2^20 + 14 x 2^-10", [2.0 ** 20, -2.0 ** 20] + [2.0 ** -10] * 14, ones, 0.0),
    ("2^40 - 2^40 + 14 x 1", [2.0 ** 40, -2.0 ** 40] + [1.0] * 14, ones, 0.0),
    ("c = -2^30, 2^30 + 15 x 1", [2.0 ** 30] + [1.0] * 15, ones, -(2.0 ** 30)),
    ("1.5 * 1.5 x 16 (products with 4 significant bits)", [1.5] * 16, [1.5] * 16, 0.0),
    ("255/128 * 255/128 x 16 (full 16-bit products)", [255.0 / 128] * 16, [255.0 / 128] * 16, 0.0),
    ("2^23 + 15 x (255/128)^2", [2.0 ** 23] + [255.0 / 128] * 15, [1.0] + [255.0 / 128] * 15, 0.0),
]
print("accumulation probe: one v_mfma_f32_32x32x16_bf16, row 0 x column 0")
for name, a, b, c in cases:
    exact = float(np.sum(np.array(a, dtype=np.float64) * np.array(b, dtype=np.float64)) + c)
    got = probe(a, b, c)
    mag = float(np.sum(np.abs(np.array(a, dtype=np.float64) * np.array(b, dtype=np.float64))) + abs(c))
    print(f"  {name:52s} exact {exact!r:24s} got {got!r:24s} error / (2^-23 sum|terms|) = {abs(got - exact) / (mag * 2.0 ** -23):.3f}")
print("the same with v_mfma_f32_32x32x16_f16 (operands that fit fp16)")
half_cases = [
    ("2^13 * 2^10 + 15 x 1", [2.0 ** 13] + [1.0] * 15, [2.0 ** 10] + [1.0] * 15, 0.0),
    ("2^14 * 2^12 + 15 x 3", [2.0 ** 14] + [3.0] * 15, [2.0 ** 12] + [1.0] * 15, 0.0),
    ("2^13 * 2^10 + 15 x (2047/1024)^2", [2.0 ** 13] + [2047.0 / 1024] * 15, [2.0 ** 10] + [2047.0 / 1024] * 15, 0.0),
    ("(2047/1024)^2 x 16 (22-bit products)", [2047.0 / 1024] * 16, [2047.0 / 1024] * 16, 0.0),
    ("subnormal operands 2^-20 * 2^10 x 16", [2.0 ** -20] * 16, [2.0 ** 10] * 16, 0.0),
    ("c = -2^23, 2^13 * 2^10 + 15 x 1.5", [2.0 ** 13] + [1.5] * 15, [2.0 ** 10] + [1.0] * 15, -(2.0 ** 23)),
]
for name, a, b, c in half_cases:
    exact = float(np.sum(np.array(a, dtype=np.float64) * np.array(b, dtype=np.float64)) + c)
    got = probe(a, b, c, 1)
    mag = float(np.sum(np.abs(np.array(a, dtype=np.float64) * np.array(b, dtype=np.float64))) + abs(c))
    print(f"  {name:52s} exact {exact!r:24s} got {got!r:24s} error / (2^-23 sum|terms|) = {abs(got - exact) / (mag * 2.0 ** -23):.3f}")
worst = 0.0
rng = np.random.default_rng(2)
for trial in range(2000):
    a = np.round(rng.uniform(1, 2, 16) * 1024) / 1024 * 2.0 ** rng.integers(-8, 13, size=16) * rng.choice([-1, 1], 16)
    b = np.round(rng.uniform(1, 2, 16) * 1024) / 1024 * 2.0 ** rng.integers(-8, 10, size=16) * rng.choice([-1, 1], 16)
    c = float(np.float32(rng.normal() * 2.0 ** rng.integers(-10, 24)))
    exact = float(np.sum(a.astype(np.float64) * b.astype(np.float64)) + c)
    mag = float(np.sum(np.abs(a * b)) + abs(c))
    worst = max(worst, abs(probe(a.tolist(), b.tolist(), c, 1) - exact) / (mag * 2.0 ** -23))
print(f"  2000 random fp16 cases: worst error / (2^-23 sum|terms|) = {worst:.3f}   (the bound assumes <= 17)")
rng = np.random.default_rng(1)
worst = 0.0
for trial in range(2000):
    ex = rng.integers(-20, 20, size=16)
    a = np.round(rng.uniform(1, 2, 16) * 128) / 128 * 2.0 ** ex * rng.choice([-1, 1], 16)
    b = np.round(rng.uniform(1, 2, 16) * 128) / 128 * rng.choice([-1, 1], 16)
    c = float(np.float32(rng.normal() * 2.0 ** rng.integers(-20, 20)))
    exact = float(np.sum(a.astype(np.float64) * b.astype(np.float64)) + c)
    mag = float(np.sum(np.abs(a * b)) + abs(c))
    got = probe(a.tolist(), b.tolist(), c)
    worst = max(worst, abs(got - exact) / (mag * 2.0 ** -23))
print(f"  2000 random cases (exponents spread over 2^40): worst error / (2^-23 sum|terms|) = {worst:.3f}   (the bound assumes <= 17)")

n, h, thr = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000)), float(os.environ.get("THR", 1.5e-6))
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)
ws_old = dev.score_workspace(n, h, 1, corr.device)
table = torch.zeros((lib.bf16_table_bytes(n),), dtype=torch.uint8, device=corr.device)
table_old = torch.zeros((old.mfma_table_floats(n),), dtype=torch.float32, device=corr.device)
assert lib.bf16_prepare(corr.data_ptr(), n, thr, ws.data_ptr(), table.data_ptr(), st) == 0
assert old.filter_prepare(corr.data_ptr(), n, thr, ws_old.data_ptr(), table_old.data_ptr(), st) == 0
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


t_v = timed(lambda: old.valu_filter_count(ws_old.data_ptr(), n, E.data_ptr(), h, thr, surv_v.data_ptr(), st))
t_m = timed(lambda: lib.bf16_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, 0, st))
t_o = timed(lambda: lib.bf16_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, -1, st))
print(f"matrix work alone (3 x v_mfma_f32_32x32x16_bf16 per 32 x 32 tile, operand loads, one compare): {t_o:.3f} ms")
lib.bf16_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), None, 0, st)
cnt, s1, s2 = dev.score_sed(corr, E, S, thr, exact_only=True)
cnt = cnt[0].cpu().numpy().astype(np.int64)
sv, sm = surv_v.cpu().numpy().astype(np.int64), surv_m.cpu().numpy().astype(np.int64)
print(f"N={n} H={h}: VALU tier 1 only {t_v:.3f} ms ({n * h / t_v / 1e6:.0f} G evals/s), bf16 MFMA tier 1 only {t_m:.3f} ms "
      f"({n * h / t_m / 1e6:.0f} G evals/s), ratio {t_v / t_m:.2f}x")
print(f"survivors / true inliers: VALU {sv.sum() / (cnt.sum() + 8 * h):.4f}, bf16 MFMA {sm.sum() / (cnt.sum() + 8 * h):.4f}; "
      f"survivors / evaluations: VALU {sv.sum() / (n * h):.4f}, bf16 MFMA {sm.sum() / (n * h):.4f}")
print("hypotheses with survivors < inliers: VALU", int((sv < cnt).sum()), " bf16 MFMA", int((sm < cnt).sum()))
assert np.all(sv >= cnt), "VALU filter rejected an inlier"
assert np.all(sm >= cnt), "MFMA filter rejected an inlier"
# per-point containment on the first hypotheses: every exact inlier (sed <= thr) must survive the MFMA filter
mh = 256
tiles = (n + 31) // 32
masks = torch.zeros((mh, tiles), dtype=torch.int32, device=corr.device)
assert lib.bf16_filter_count(table.data_ptr(), ws.data_ptr(), n, E.data_ptr(), h, thr, surv_m.data_ptr(), masks.data_ptr(), mh, st) == 0
mk = masks.cpu().numpy().astype(np.uint32)
bits = ((mk[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & np.uint32(1)).reshape(mh, tiles * 32)[:, :n].astype(bool)
worst = 0
closest = np.inf
for k in range(mh):
    sed = dev.sed_values(corr[0], E[0, k]).cpu().numpy()
    inl = sed <= thr
    assert np.all(bits[k][inl]), f"hypothesis {k}: MFMA filter rejected an exact inlier"
    assert bits[k].sum() == surv_m[k].item(), (k, bits[k].sum(), surv_m[k].item())
    worst = max(worst, bits[k].sum() - inl.sum())
    rejected = sed[~bits[k]]
    if rejected.size:
        closest = min(closest, float(rejected.min()) / thr)
print(f"containment ok on {mh} hypotheses (layout check: mask populations == counts); max extra survivors {worst}; "
      f"smallest sed / thr among rejected pairs {closest:.3f}")
