"""Diagnostic: dump device fit results for offline accuracy analysis (vs mpmath)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import sfm_oracle as orc
from structure_from_motion_amd import device as dev
out = {}
for n, h, seed in [(1000, 257, 7), (5000, 2000, 5)]:
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=6)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(seed, 0, h, n)
    lam = torch.empty((1, h), dtype=torch.float64, device="cuda")
    E, flags = dev.fit_eight_point(dev.to_device(corr).reshape(1, n, 4), dev.to_device(S, torch.int32).reshape(1, h, 8), lambda2=lam)
    out[f"E_{n}_{h}"] = E.cpu().numpy().reshape(h, 3, 3)
    out[f"lam_{n}_{h}"] = lam.cpu().numpy()[0]
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/fit_dump.npz", **out)
print("dumped")
