"""Batched two-view geometry, entirely on the device: for B independent image pairs run RANSAC
essential-matrix estimation, pose recovery (decomposition + cheirality vote) and triangulation of the
surviving inliers — the sequence of reference ``apps/sfm.py:110-186`` — with no host round trip between
the stages.  Pairs are independent, so this also shards over GPUs with no collective.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import ctypes as C
import numpy as np
import torch

from . import _native, device
from ._native import check

F64 = torch.float64

# per-pair status codes
OK = 0
NO_MODEL = 1            # RANSAC found no hypothesis with enough inliers (ValueError in the reference)
DEGENERATE_SAMPLE = 2   # a sampled eight-tuple was degenerate (EightPointCalculationError, strict policy)
BAD_ESSENTIAL = 3       # smallest singular value of E not ~0 (eight_point.py:268-271)
NO_POSE = 4             # no candidate pose passes the cheirality vote (eight_point.py:233-236)


@dataclass
class PairResult:
    status: int
    E: Optional[np.ndarray]            # (3,3)
    best_h: int
    inlier_order: Optional[np.ndarray]  # indices in the reference's list order: sample, then survivors
    R: Optional[np.ndarray]
    t: Optional[np.ndarray]
    votes: Optional[np.ndarray]        # (4,) quirk votes
    pose_mask: Optional[np.ndarray]    # positions within inlier_order that pass the chosen pose
    points: Optional[np.ndarray]       # (M,3) triangulated points of those, in that order


class TwoViewBatch:
    """Pre-allocated pipeline for ``batch`` pairs x ``n`` correspondences x ``hypotheses`` samples."""

    def __init__(self, batch: int, n: int, hypotheses: int, device_=None):
        dev = device_ or device.require_gpu()
        self.batch, self.n, self.h = batch, n, hypotheses
        self.ws = device.RansacWorkspace(batch, n, hypotheses, dev)
        self.corr = torch.empty((batch, n, 4), dtype=F64, device=dev)
        self.E_best = torch.empty((batch, 9), dtype=F64, device=dev)
        self.poses = torch.empty((batch, 4, 12), dtype=F64, device=dev)
        self.decomp_status = torch.empty((batch,), dtype=torch.int32, device=dev)
        self.passes = torch.empty((batch, 4, n), dtype=torch.uint8, device=dev)
        self.skip = torch.empty((batch,), dtype=torch.int32, device=dev)
        self.votes = torch.empty((batch, 4), dtype=torch.int32, device=dev)
        self.best_pose = torch.empty((batch,), dtype=torch.int32, device=dev)
        self.X = torch.empty((batch, n, 3), dtype=F64, device=dev)
        self.valid = torch.empty((batch, n), dtype=torch.uint8, device=dev)
        self._pix = None
        self._K = None
        self._mask = self.ws.mask   # inlier mask the pose stage works from (refined one with local optimisation)
        self._refined = False

    def run(self, pix_a: torch.Tensor, pix_b: torch.Tensor, K, seed: int, thr: float, min_extra: float,
            aggregation: int, distance_threshold: float = 50.0, seed_stride: int = 1,
            local_optimisation: int = 0, score_options=None) -> None:
        """Enqueue the whole pipeline.  pix_a, pix_b: [B,N,2] f64 device tensors (pixel coordinates);
        pair b samples with Philox seed ``seed + b*seed_stride``.  ``local_optimisation=k`` (extension, off by
        default) refits every winner on all its inliers up to k times before pose recovery; the inlier lists
        then come back in index order (a refined model has no sample to put first).  ``score_options``: launch options of
        the scoring launch of this call (``device.ScoreOptions``, timing events included)."""
        lib = _native.load()
        B, N = self.batch, self.n
        st = device._stream()
        ws = self.ws
        self._pix, self._K = (pix_a, pix_b), np.ascontiguousarray(K, dtype=np.float64)
        device.normalize_correspondences(pix_a, pix_b, K, out=self.corr)
        ws.run(self.corr, thr, min_extra, aggregation, philox=(seed, 0, seed_stride), options=score_options)
        # winner's E and first sample index per pair (torch indexing = device memory plumbing only)
        best = ws.result[:, 1].clamp(min=0)
        rows = torch.arange(B, device=best.device)
        self.E_best.copy_(ws.E[rows, best])
        self._refined = local_optimisation > 0
        if self._refined:
            err = ws.result.view(F64)[:, 2]
            E_ref, self._mask, _ = device.refine_inliers(self.corr, self.E_best, ws.mask, err, thr, aggregation,
                                                         local_optimisation)
            self.E_best.copy_(E_ref)
            self.skip.copy_(self._mask.argmax(dim=1))   # list position 0 = lowest inlier index (vote quirk Q9)
        else:
            self._mask = ws.mask
            self.skip.copy_(ws.S[rows, best, 0])
        device.decompose_essential(self.E_best, out=(self.poses, self.decomp_status))
        check(lib.sfm_cheirality_batched(self.corr.data_ptr(), N, B, self.poses.data_ptr(), self._mask.data_ptr(),
                                         float(distance_threshold), self.passes.data_ptr(), st),
              "sfm_cheirality_batched")
        check(lib.sfm_pose_vote(self.passes.data_ptr(), N, B, self.skip.data_ptr(), self.votes.data_ptr(),
                                self.best_pose.data_ptr(), st), "sfm_pose_vote")
        check(lib.sfm_triangulate_selected(pix_a.data_ptr(), pix_b.data_ptr(), N, B,
                                           self._K.ctypes.data_as(C.c_void_p), self.poses.data_ptr(),
                                           self.best_pose.data_ptr(), self.passes.data_ptr(), self.X.data_ptr(),
                                           self.valid.data_ptr(), st), "sfm_triangulate_selected")

    def results(self) -> List[PairResult]:
        """Copy everything back (synchronises) and arrange it in the reference's list orders."""
        recs = device.read_select(self.ws.result)
        E = self.E_best.cpu().numpy().reshape(-1, 3, 3)
        S = self.ws.S.cpu().numpy()
        mask = self._mask.cpu().numpy()
        poses = self.poses.cpu().numpy()
        dstat = self.decomp_status.cpu().numpy()
        votes = self.votes.cpu().numpy()
        best_pose = self.best_pose.cpu().numpy()
        passes = self.passes.cpu().numpy()
        X = self.X.cpu().numpy()
        out = []
        for b, rec in enumerate(recs):
            if rec.n_flagged:
                out.append(PairResult(DEGENERATE_SAMPLE, None, -1, None, None, None, None, None, None))
                continue
            if rec.best_h < 0:
                out.append(PairResult(NO_MODEL, None, -1, None, None, None, None, None, None))
                continue
            if self._refined:
                order = np.nonzero(mask[b])[0]
            else:
                sample = S[b, rec.best_h].astype(np.int64)
                order = np.concatenate([sample, np.nonzero(mask[b] == 1)[0]])
            if dstat[b] != 0:
                out.append(PairResult(BAD_ESSENTIAL, E[b], int(rec.best_h), order, None, None, None, None, None))
                continue
            if best_pose[b] < 0:
                out.append(PairResult(NO_POSE, E[b], int(rec.best_h), order, None, None, votes[b], None, None))
                continue
            p = poses[b, best_pose[b]]
            keep = passes[b, best_pose[b], order] != 0
            out.append(PairResult(OK, E[b], int(rec.best_h), order, p[:9].reshape(3, 3).copy(), p[9:].copy(),
                                  votes[b], np.nonzero(keep)[0], X[b, order[keep]]))
        return out
