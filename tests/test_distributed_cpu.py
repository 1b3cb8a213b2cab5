"""World-size-2 `gloo` tests (CPU) of the cross-rank model selection used by the sharded RANSAC path:
the two MIN all-reduces must reproduce the sequential "strictly lower error wins, earliest first" rule of
reference lib/ransac/ransac.py:83 for any split of the hypotheses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sfm_oracle as orc
from structure_from_motion_amd import distributed
from structure_from_motion_amd._native import INT64_MAX


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _local_select(err, cnt, min_extra, offset):
    """What sfm_select_best leaves in (key, best_h) for one shard: error bits and global index."""
    best, e = orc.select_best(err, cnt, min_extra)
    if best < 0:
        return INT64_MAX, -1
    return int(np.float64(e).view(np.int64)), best + offset


def _worker(rank, world, port, cases, out_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys, bests, firsts, nflags = [], [], [], []
        for err, cnt, min_extra, flagged in cases:
            h = len(err) // world
            lo = rank * h
            k, b = _local_select(err[lo:lo + h], cnt[lo:lo + h], min_extra, lo)
            keys.append(k)
            bests.append(b)
            fl = [i for i in flagged if lo <= i < lo + h]
            firsts.append(min(fl) if fl else INT64_MAX)
            nflags.append(len(fl))
        gkey, gbest = distributed.reduce_best(torch.tensor(keys, dtype=torch.int64),
                                              torch.tensor(bests, dtype=torch.int64))
        gfirst, gn = distributed.reduce_flagged(torch.tensor(firsts, dtype=torch.int64),
                                                torch.tensor(nflags, dtype=torch.int64))
        out_queue.put((rank, gkey.tolist(), gbest.tolist(), gfirst.tolist(), gn.tolist()))
    finally:
        dist.destroy_process_group()


def _cases():
    rng = np.random.default_rng(0)
    cases = []
    for trial in range(6):
        h = 64
        err = rng.random(h)
        cnt = rng.integers(0, 30, h).astype(np.int32)
        flagged = []
        if trial == 1:      # exact tie across ranks: the earlier (rank 0) index must win
            err[5] = err[40] = 1e-9
            cnt[5] = cnt[40] = 29
        if trial == 2:      # best lives on rank 1
            err[50] = 1e-12
            cnt[50] = 29
        if trial == 3:      # rank 0 has no gated model at all
            cnt[:32] = 0
        if trial == 4:      # nobody has a model
            cnt[:] = 0
        if trial == 5:      # NaN / inf never win, flags reported
            err[3] = np.nan
            err[35] = np.inf
            flagged = [7, 33, 60]
        cases.append((err, cnt, 10, flagged))
    return cases


def test_reduce_best_two_ranks_gloo():
    cases = _cases()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cases, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1:] == results[1][1:]  # identical on every rank
    _, gkey, gbest, gfirst, gn = results[0]
    for i, (err, cnt, min_extra, flagged) in enumerate(cases):
        best, e = orc.select_best(err, cnt, min_extra)  # the sequential rule over all hypotheses
        assert gbest[i] == best
        if best >= 0:
            assert np.int64(gkey[i]).view(np.float64) == e
        else:
            assert gkey[i] == INT64_MAX
        assert gn[i] == len(flagged)
        assert gfirst[i] == (min(flagged) if flagged else INT64_MAX)


def test_reduce_is_identity_without_process_group():
    key = torch.tensor([5, INT64_MAX], dtype=torch.int64)
    best = torch.tensor([3, -1], dtype=torch.int64)
    k, b = distributed.reduce_best(key, best)
    assert k.tolist() == key.tolist() and b.tolist() == best.tolist()


def test_shard_range_partitions():
    for total, world in [(100, 8), (1_000_000, 8), (7, 8), (0, 2), (100_000, 1)]:
        spans = [distributed.shard_range(total, r, world) for r in range(world)]
        covered = []
        for begin, count in spans:
            covered.extend(range(begin, begin + count))
        assert covered == list(range(total))


def test_error_bits_are_monotone_as_int64():
    x = np.sort(np.abs(np.random.default_rng(1).normal(size=1000)) * 10.0 ** np.random.default_rng(2).integers(-300, 300, 1000))
    bits = x.view(np.int64)
    assert np.all(np.diff(bits) >= 0) and bits.max() < INT64_MAX
    assert np.float64(np.inf).view(np.int64) < INT64_MAX
