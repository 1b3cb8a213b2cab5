"""Multi-GPU RANSAC: hypotheses shard across ranks, ONE tiny collective picks the global best model.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).  Rank r fits and scores its contiguous block of the global counter-based sample stream
(``shard_range``) against its own replica of the correspondences (32 B x N, at most a few MB).  The only
data-path exchange per pass is one ``all_gather`` of each rank's 40-byte ``sfm_select_result`` — written by the
selection kernel itself with global hypothesis indices — after which every rank folds the ``world`` records
locally (``sfm_fold_select_records``: lowest error key, then lowest global index; flag statistics min / sum):

* that reproduces the sequential rule "strictly lower error wins, earliest first" (reference
  ``lib/ransac/ransac.py:83-86``) for any partition of the hypotheses;
* degenerate samples (``lib/epipolar/eight_point.py:415-421``, raised through ``ransac.py:65``) are seen by all
  ranks alike, so ``SFM_DEGENERATE=raise|skip`` acts exactly as in the single-GPU drop-in path;
* every rank re-derives the winner locally from (seed, h*) — the sample is a pure function of them — so no
  E / mask broadcast is needed.

Image-pair batches (BASELINE config 5) shard over pairs with no collective at all.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _native
from ._native import INT64_MAX, SelectResult, check

NO_MODEL_KEY = INT64_MAX
RECORD_WORDS = C.sizeof(SelectResult) // 8   # the record viewed as int64 words


def shard_range(total_hypotheses: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of hypothesis indices owned by ``rank``: (begin, count)."""
    per = -(-total_hypotheses // world)
    begin = min(total_hypotheses, rank * per)
    return begin, max(0, min(total_hypotheses, begin + per) - begin)


def _distributed(group) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def gather_records(record: torch.Tensor, out: Optional[torch.Tensor] = None, group=None,
                   force_collective: bool = False) -> torch.Tensor:
    """The ONE collective of a sharded pass.  record: int64 [batch, 5] view of this rank's sfm_select_result
    records -> int64 [world, batch, 5], identical on every rank.  With the "nccl" backend (RCCL) the device tensor
    is gathered in place over xGMI; with "gloo" (CPU tests, or a rehearsal with several ranks sharing one GPU)
    device tensors are staged through the host.  A world of one copies without a collective unless
    ``force_collective`` (an initialised process group of one rank: the call RCCL sees at N > 1, executable on one GPU)."""
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    if out is None:
        out = torch.empty((world,) + tuple(record.shape), dtype=record.dtype, device=record.device)
    if force_collective and not initialised:
        raise RuntimeError("gather_records(force_collective=True) needs an initialised process group")
    if world == 1 and not force_collective:
        out[0].copy_(record)
        return out
    if record.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host.view(-1), record.cpu().reshape(-1), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out.view(-1), record.reshape(-1), group=group)
    return out


def fold_records(gathered: torch.Tensor, global_record: Optional[torch.Tensor] = None,
                 best_h: Optional[torch.Tensor] = None, single: Optional[torch.Tensor] = None):
    """gathered int64 [world, batch, 5] -> (global_record [batch,5], best_h [batch], single [batch,5]) by
    ``sfm_fold_select_records`` — a kernel on the current stream for device tensors, the same routine on the
    calling thread for host tensors (CPU process-group tests)."""
    lib = _native.load()
    world, batch, words = gathered.shape
    assert words == RECORD_WORDS and gathered.dtype == torch.int64 and gathered.is_contiguous()
    dev = gathered.device
    if global_record is None:
        global_record = torch.empty((batch, words), dtype=torch.int64, device=dev)
    if best_h is None:
        best_h = torch.empty((batch,), dtype=torch.int64, device=dev)
    if single is None:
        single = torch.empty((batch, words), dtype=torch.int64, device=dev)
    if gathered.is_cuda:
        check(lib.sfm_fold_select_records(gathered.data_ptr(), world, batch, global_record.data_ptr(),
                                          best_h.data_ptr(), single.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "sfm_fold_select_records")
    else:
        check(lib.sfm_fold_select_records_host(gathered.data_ptr(), world, batch, global_record.data_ptr(),
                                               best_h.data_ptr(), single.data_ptr()),
              "sfm_fold_select_records_host")
    return global_record, best_h, single


def read_records(record: torch.Tensor):
    """Host copies of int64 [batch, 5] records as ``SelectResult`` structures (synchronises)."""
    raw = record.cpu().numpy().tobytes()
    size = C.sizeof(SelectResult)
    return [SelectResult.from_buffer_copy(raw[i * size:(i + 1) * size]) for i in range(record.shape[0])]


def degenerate_policy(policy: Optional[str] = None) -> str:
    policy = (policy or os.environ.get("SFM_DEGENERATE", "raise")).lower()
    if policy not in ("raise", "skip"):
        raise ValueError(f"SFM_DEGENERATE must be 'raise' or 'skip', got {policy!r}")
    return policy


class ShardedOutcome(NamedTuple):
    best_h: int                    # global index of the winning hypothesis, -1 if none
    error: float                   # its aggregated inlier error (+inf if none)
    E: Optional[np.ndarray]        # (3,3)
    sample: Optional[np.ndarray]   # (8,) indices of its sample
    mask: Optional[np.ndarray]     # (N,) uint8: 1 survivor, 2 sample point
    n_flagged: int                 # degenerate samples among ALL ranks' hypotheses
    first_flagged: int             # lowest global index of one, -1 if none


class ShardedRansac:
    """RANSAC-E over ``world`` GPUs.  Either ``hypotheses_per_rank`` (rank r owns [r*H, (r+1)*H) of the Philox
    stream: weak scaling) or ``total_hypotheses`` (the global count is split by ``shard_range``: BASELINE
    config 4, 1 M hypotheses over 8 GPUs).

    ``step`` enqueues sample -> fit -> score -> select on the local shard, the 40-byte all-gather, the fold and
    the local re-derivation of the global winner (sample, E, inlier mask) — no host synchronisation.
    """

    def __init__(self, corr: torch.Tensor, hypotheses_per_rank: Optional[int], thr: float, min_extra: float,
                 aggregation: int, rank: int = 0, world: int = 1, group=None,
                 total_hypotheses: Optional[int] = None, force_exchange: bool = False):
        from . import device

        self.device_api = device
        self.corr = corr.reshape(1, -1, 4)
        self.n = self.corr.shape[1]
        if total_hypotheses is not None:
            self.h_begin, self.h = shard_range(total_hypotheses, rank, world)
            self.total = total_hypotheses
        else:
            self.h_begin, self.h = rank * hypotheses_per_rank, hypotheses_per_rank
            self.total = hypotheses_per_rank * world
        self.thr, self.min_extra, self.aggregation = thr, min_extra, aggregation
        self.rank, self.world, self.group = rank, world, group
        # force_exchange: a world of ONE still runs the multi-rank path — global indices from the selection kernel, the
        # all-gather as a real collective on the device tensor, the fold, the winner re-derived from (seed, h*) — so that the
        # code RCCL sees at N > 1 can be executed and checked on one GPU (tests/test_gpu_api.py)
        self.exchange = world > 1 or force_exchange
        self.force_exchange = force_exchange
        dev = corr.device
        self.ws = device.RansacWorkspace(1, self.n, self.h, dev)  # h == 0 (more ranks than hypotheses) is a valid empty shard
        if not self.exchange:  # the local record is the global one: views, no copies, no exchange
            self.global_record = self.ws.result
            self.global_best = self.ws.result[:, 1]
        else:
            words = RECORD_WORDS
            self.gathered = torch.empty((world, 1, words), dtype=torch.int64, device=dev)
            self.global_record = torch.empty((1, words), dtype=torch.int64, device=dev)
            self.global_best = torch.empty((1,), dtype=torch.int64, device=dev)
            self.win_record = torch.empty((1, words), dtype=torch.int64, device=dev)
            # winner re-derivation buffers (one hypothesis)
            self.win_S = torch.empty((1, 1, 8), dtype=torch.int32, device=dev)
            self.win_E = torch.empty((1, 1, 9), dtype=torch.float64, device=dev)
            self.win_flags = torch.empty((1, 1), dtype=torch.int32, device=dev)
        self.graph = None
        self.seed_dev = None
        self.score_options = None   # launch options of the local pass' scoring launch (device.ScoreOptions; None: process defaults)

    def _local_pass(self, seed) -> None:
        """sample -> fit -> score -> select on this rank's shard (+ mask when the winner is local).
        ``seed=None`` reads the seed from ``self.seed_dev`` (the form a HIP graph can replay)."""
        source = (self.seed_dev if seed is None else seed, self.h_begin, 1)  # sampled inside the fit kernel
        if not self.exchange:
            # single GPU: the winner is local — mask straight from the shard's own E / S
            self.ws.run(self.corr, self.thr, self.min_extra, self.aggregation, h_offset=0, with_mask=True,
                        philox=source, options=self.score_options)
        else:
            self.ws.run(self.corr, self.thr, self.min_extra, self.aggregation, h_offset=self.h_begin,
                        with_mask=False, philox=source, options=self.score_options)

    def capture(self) -> None:
        """Record the local pass once into a HIP graph; later ``step`` calls rewrite one seed word in device
        memory and replay it.  With world > 1 the graph covers the local pass and the exchange stays eager.
        Every buffer the captured kernels touch is allocated before capture begins (``RansacWorkspace`` and
        ``seed_dev``): nothing is allocated from the graph's private pool."""
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.corr.device)
        self.ws._fit_workspace(self.score_options)   # sized for this engine's launch options BEFORE capture ...
        self.ws.frozen = True                        # ... and never reallocated afterwards: the graph holds its address
        side = torch.cuda.Stream(device=self.corr.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._local_pass(None)  # warm-up outside capture (lazy module loads)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        def allocations():  # host-side counter of the caching allocator (torch's own capture set-up allocates too:
            return torch.cuda.memory_stats(self.corr.device).get("allocation.all.allocated", 0)  # count only our pass)

        with torch.cuda.graph(graph):
            before = allocations()
            self._local_pass(None)
            grown = allocations() - before
        if grown:
            # a tensor allocated during capture lives in the graph's private pool and is recycled by later
            # replays: every buffer of the pass must exist before capture (graph replay)
            raise RuntimeError(f"ShardedRansac.capture: {grown} device allocation(s) happened inside graph capture")
        self.graph = graph

    def step(self, seed: int) -> None:
        self.step_local(seed)
        if self.exchange:
            # the one collective: 40 B per rank
            gather_records(self.ws.result, self.gathered, self.group, force_collective=self.force_exchange)
            self.finish(seed)

    def step_local(self, seed: int) -> None:
        """This rank's share of a pass: sample -> fit -> score -> select over its own hypotheses; leaves the
        rank's select record (global indices) in ``self.ws.result``."""
        if self.graph is not None:
            s = seed & (2**64 - 1)
            self.seed_dev.fill_(s - 2**64 if s >= 2**63 else s)
            self.graph.replay()
        else:
            self._local_pass(seed)

    def finish(self, seed: int, gathered: Optional[torch.Tensor] = None) -> None:
        """Fold the gathered records and re-derive the global winner locally.  ``gathered`` ([world, 1, 5] int64)
        defaults to what ``step`` all-gathered; tests that run several virtual ranks on one GPU pass the stacked
        records of those ranks instead."""
        d = self.device_api
        if gathered is not None:
            self.gathered.copy_(gathered)
        fold_records(self.gathered, self.global_record, self.global_best, self.win_record)
        # every rank re-derives the winner from (seed, h*): same code path -> bit-identical E
        d.sample_philox_at(seed, self.global_best, self.n, out=self.win_S)
        d.fit_eight_point(self.corr, self.win_S, self.win_E, self.win_flags)
        d.inlier_mask(self.corr, self.win_E, self.win_S, self.win_record, self.thr, self.ws.mask)

    def outcome(self, policy: Optional[str] = None) -> ShardedOutcome:
        """Host copy of the global winner (synchronises).  A degenerate sample anywhere in the global hypothesis
        range raises ``EightPointCalculationError`` on every rank alike unless the policy (argument, else
        ``SFM_DEGENERATE``) is "skip" — the reference aborts the whole call (eight_point.py:415-421 through
        ransac.py:65), exactly as the single-GPU drop-in path does."""
        rec = read_records(self.global_record)[0]
        first = -1 if rec.first_flagged == INT64_MAX else int(rec.first_flagged)
        n_flagged = int(rec.n_flagged)
        if n_flagged and degenerate_policy(policy) == "raise":
            from .epipolar.eight_point import EightPointCalculationError

            raise EightPointCalculationError(
                "More than one eigenvalue of Y.T @ Y is small. Cannot confidently estimate"
                f" fundamental matrix. (hypothesis {first}, {n_flagged} in total)")
        best = int(rec.best_h)
        if best < 0:
            return ShardedOutcome(-1, float("inf"), None, None, None, n_flagged, first)
        if not self.exchange:
            E = self.ws.E[0, best].cpu().numpy().reshape(3, 3)
            sample = self.ws.S[0, best].cpu().numpy().astype(np.int64)
        else:
            E = self.win_E.cpu().numpy().reshape(3, 3)
            sample = self.win_S.cpu().numpy().reshape(8).astype(np.int64)
        mask = self.device_api.checked_mask(self.ws.mask.cpu().numpy()[0])
        return ShardedOutcome(best, float(rec.best_err), E, sample, mask, n_flagged, first)
