"""Multi-GPU RANSAC: hypotheses shard across ranks, one tiny exchange picks the global best model.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).  Rank r fits and scores hypotheses ``[r*H, (r+1)*H)`` of the global counter-based sample stream
against its own replica of the correspondences (32 B x N, at most a few MB); the only data-path
exchange is 16 bytes per rank:

    key  = all_reduce(MIN) of the winner's error bits   (non-negative f64 bits are monotone as int64;
                                                         "no model" is INT64_MAX)
    h    = all_reduce(MIN) of the winner's global index, masked to ranks whose key equals the global key

which reproduces the sequential rule "strictly lower error wins, earliest first" (reference
``lib/ransac/ransac.py:83``) for any sharding.  Every rank then re-derives the winner locally
(``finalize``): the sample is a pure function of (seed, h), so no E / mask broadcast is needed.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

from ._native import INT64_MAX

NO_MODEL_KEY = INT64_MAX


def reduce_best(key: torch.Tensor, best_h: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """key, best_h: int64 tensors of equal shape (one entry per image pair) holding this rank's winner
    (``sfm_select_result.key`` / ``.best_h`` with global indices; best_h = -1 and key = INT64_MAX if the
    rank found no model).  Returns the global (key, best_h), identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return key.clone(), best_h.clone()
    gkey = _all_reduce_min(key, group)
    sentinel = torch.full_like(best_h, INT64_MAX)
    cand = torch.where((key == gkey) & (best_h >= 0), best_h, sentinel)
    cand = _all_reduce_min(cand, group)
    gbest = torch.where(cand == sentinel, torch.full_like(cand, -1), cand)
    return gkey, gbest


def _all_reduce_min(t: torch.Tensor, group=None) -> torch.Tensor:
    """MIN all-reduce of a small int64 tensor.  With the "nccl" backend (RCCL) the device tensor is reduced in
    place over xGMI; with "gloo" (CPU tests, or a rehearsal with several ranks sharing one GPU) device tensors
    are staged through the host."""
    out = t.clone()
    if out.is_cuda and dist.get_backend(group) == "gloo":
        host = out.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
        out.copy_(host)
    else:
        dist.all_reduce(out, op=dist.ReduceOp.MIN, group=group)
    return out


def reduce_flagged(first_flagged: torch.Tensor, n_flagged: torch.Tensor, group=None):
    """Global (lowest flagged hypothesis index, number of flagged hypotheses) so that every rank raises
    the same EightPointCalculationError."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return first_flagged.clone(), n_flagged.clone()
    first = first_flagged.clone()
    dist.all_reduce(first, op=dist.ReduceOp.MIN, group=group)
    total = n_flagged.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return first, total


def shard_range(total_hypotheses: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of hypothesis indices owned by ``rank``: (begin, count)."""
    per = -(-total_hypotheses // world)
    begin = min(total_hypotheses, rank * per)
    return begin, max(0, min(total_hypotheses, begin + per) - begin)


class ShardedRansac:
    """RANSAC-E over ``world`` GPUs: rank r owns hypotheses [r*H, (r+1)*H) of the Philox stream.

    ``step`` enqueues sample -> fit -> score -> select on the local shard, the 16-byte exchange, and the
    local re-derivation of the global winner (sample, E, inlier mask) — no host synchronisation.
    """

    def __init__(self, corr: torch.Tensor, hypotheses_per_rank: int, thr: float, min_extra: float,
                 aggregation: int, rank: int = 0, world: int = 1, group=None):
        from . import device

        self.device_api = device
        self.corr = corr.reshape(1, -1, 4)
        self.n = self.corr.shape[1]
        self.h = hypotheses_per_rank
        self.thr, self.min_extra, self.aggregation = thr, min_extra, aggregation
        self.rank, self.world, self.group = rank, world, group
        dev = corr.device
        self.ws = device.RansacWorkspace(1, self.n, self.h, dev)
        # winner re-derivation buffers (one hypothesis)
        self.win_S = torch.empty((1, 1, 8), dtype=torch.int32, device=dev)
        self.win_E = torch.empty((1, 1, 9), dtype=torch.float64, device=dev)
        self.win_flags = torch.empty((1, 1), dtype=torch.int32, device=dev)
        self.win_record = torch.zeros((1, 5), dtype=torch.int64, device=dev)
        if world == 1:  # the local record is the global one: views, no copies
            self.global_key, self.global_best = self.ws.result[:, 0], self.ws.result[:, 1]
        else:
            self.global_key = torch.empty((1,), dtype=torch.int64, device=dev)
            self.global_best = torch.empty((1,), dtype=torch.int64, device=dev)

        self.graph = None
        self.seed_dev = None

    def _local_pass(self, seed) -> None:
        """sample -> fit -> score -> select on this rank's shard (+ mask when the winner is local).
        ``seed=None`` reads the seed from ``self.seed_dev`` (the form a HIP graph can replay)."""
        begin = self.rank * self.h
        source = (self.seed_dev if seed is None else seed, begin, 1)  # sampled inside the fit kernel
        if self.world == 1:
            # single GPU: the winner is local — mask straight from the shard's own E / S
            self.ws.run(self.corr, self.thr, self.min_extra, self.aggregation, h_offset=0, with_mask=True, philox=source)
        else:
            self.ws.run(self.corr, self.thr, self.min_extra, self.aggregation, h_offset=begin, with_mask=False,
                        philox=source)

    def capture(self) -> None:
        """Record the local pass once into a HIP graph; later ``step`` calls rewrite one seed word in device
        memory and replay it (one graph launch instead of ~a dozen kernel launches — what bounds small
        problems such as the demo's N~300 x H=2000).  Only single-GPU passes are captured whole; with
        world > 1 the graph covers the local pass and the exchange stays eager."""
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.corr.device)
        side = torch.cuda.Stream(device=self.corr.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._local_pass(None)  # warm-up outside capture (lazy module loads, workspace sizing)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._local_pass(None)
        self.graph = graph

    def step(self, seed: int) -> None:
        d = self.device_api
        if self.graph is not None:
            s = seed & (2**64 - 1)
            self.seed_dev.fill_(s - 2**64 if s >= 2**63 else s)
            self.graph.replay()
        else:
            self._local_pass(seed)
        if self.world == 1:
            return
        key, best = self.ws.result[:, 0].contiguous(), self.ws.result[:, 1].contiguous()
        gkey, gbest = reduce_best(key, best, self.group)
        self.global_key.copy_(gkey)
        self.global_best.copy_(gbest)
        # every rank re-derives the winner from (seed, h*): same code path -> bit-identical E
        d.sample_philox_at(seed, self.global_best, self.n, out=self.win_S)
        d.fit_eight_point(self.corr, self.win_S, self.win_E, self.win_flags)
        self.win_record[:, 0] = gkey
        self.win_record[:, 1] = torch.where(gbest >= 0, torch.zeros_like(gbest), gbest)
        d.inlier_mask(self.corr, self.win_E, self.win_S, self.win_record, self.thr, self.ws.mask)

    def outcome(self):
        """Host copy of the global winner (synchronises): (best_h, error, E (3,3), sample, mask)."""
        import numpy as np

        best = int(self.global_best.cpu()[0])
        if best < 0:
            return -1, float("inf"), None, None, None
        err = float(self.global_key.view(torch.float64).cpu()[0])
        if self.world == 1:
            E = self.ws.E[0, best].cpu().numpy().reshape(3, 3)
            sample = self.ws.S[0, best].cpu().numpy().astype(np.int64)
        else:
            E = self.win_E.cpu().numpy().reshape(3, 3)
            sample = self.win_S.cpu().numpy().reshape(8).astype(np.int64)
        return best, err, E, sample, self.ws.mask.cpu().numpy()[0]
