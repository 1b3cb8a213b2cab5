"""World-size-2 and -4 `gloo` tests (CPU) of the cross-rank model selection used by the sharded RANSAC path:
ONE all-gather of each rank's 40-byte select record + the local fold (`sfm_fold_select_records`) must reproduce
the sequential "strictly lower error wins, earliest first" rule of reference lib/ransac/ransac.py:83-86 for any
split of the hypotheses, and carry the degenerate-sample statistics (lib/epipolar/eight_point.py:415-421) to
every rank so that all of them raise — or skip — alike."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sfm_oracle as orc
from structure_from_motion_amd import _native, distributed
from structure_from_motion_amd._native import INT64_MAX, SelectResult


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _record_words(key, best_h, err, first_flagged, n_flagged, cnt):
    rec = SelectResult(key, best_h, err, first_flagged, n_flagged, cnt)
    return torch.frombuffer(bytearray(bytes(rec)), dtype=torch.int64).clone()


def _local_record(err, cnt, min_extra, begin, flagged):
    """What sfm_select_best leaves for one shard (global indices via h_offset = begin).  Flagged hypotheses never
    compete (select_*_kernel drops them)."""
    cnt = cnt.copy()
    err = err.copy()
    for i in flagged:
        err[i - begin] = np.inf
    best, e = orc.select_best(err, cnt, min_extra)
    first = min(flagged) if flagged else INT64_MAX
    if best < 0:
        return _record_words(INT64_MAX, -1, float("inf"), first, len(flagged), 0)
    return _record_words(int(np.float64(e).view(np.int64)), best + begin, float(e), first, len(flagged), int(cnt[best]))


def _worker(rank, world, port, cases, out_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        records = []
        for err, cnt, min_extra, flagged in cases:
            begin, count = distributed.shard_range(len(err), rank, world)
            mine = [i for i in flagged if begin <= i < begin + count]
            records.append(_local_record(err[begin:begin + count], cnt[begin:begin + count], min_extra, begin, mine))
        local = torch.stack(records)                       # [batch, 5]: one "image pair" per case
        gathered = distributed.gather_records(local)       # the ONE collective
        glob, best_h, single = distributed.fold_records(gathered)
        out_queue.put((rank, glob.tolist(), best_h.tolist(), single.tolist()))
    finally:
        dist.destroy_process_group()


def _cases(h=64):
    rng = np.random.default_rng(0)
    cases = []
    for trial in range(8):
        err = rng.random(h)
        cnt = rng.integers(0, 30, h).astype(np.int32)
        flagged = []
        if trial == 1:      # exact tie across ranks: the earlier index must win
            err[5] = err[40] = 1e-9
            cnt[5] = cnt[40] = 29
        if trial == 2:      # best lives on the last rank
            err[h - 3] = 1e-12
            cnt[h - 3] = 29
        if trial == 3:      # the first half has no gated model at all
            cnt[:h // 2] = 0
        if trial == 4:      # nobody has a model
            cnt[:] = 0
        if trial == 5:      # NaN / inf never win, flags reported from several ranks
            err[3] = np.nan
            err[35] = np.inf
            flagged = [7, 33, 60]
        if trial == 6:      # the would-be winner is a degenerate sample on one rank: it must not compete
            err[50] = 0.0
            cnt[50] = 29
            flagged = [50]
        if trial == 7:      # zero error (a perfect model) ties with -0.0 handled upstream; earliest wins
            err[20] = err[21] = 0.0
            cnt[20] = cnt[21] = 15
        cases.append((err, cnt, 10, flagged))
    return cases


@pytest.mark.parametrize("world", [2, 4])
def test_gather_and_fold_gloo(world):
    cases = _cases()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    for other in results[1:]:
        assert other[1:] == results[0][1:]  # identical on every rank
    _, glob, best_h, single = results[0]
    size = C.sizeof(SelectResult)
    for i, (err, cnt, min_extra, flagged) in enumerate(cases):
        masked = err.copy()
        masked[flagged] = np.inf
        best, e = orc.select_best(masked, cnt, min_extra)  # the sequential rule over all hypotheses
        rec = SelectResult.from_buffer_copy(np.asarray(glob[i], dtype=np.int64).tobytes()[:size])
        one = SelectResult.from_buffer_copy(np.asarray(single[i], dtype=np.int64).tobytes()[:size])
        assert rec.best_h == best and best_h[i] == best
        assert one.best_h == (0 if best >= 0 else -1) and one.key == rec.key
        if best >= 0:
            assert rec.best_err == e and np.int64(rec.key).view(np.float64) == e
            assert rec.best_cnt == cnt[best]
        else:
            assert rec.key == INT64_MAX and rec.best_err == float("inf")
        assert rec.n_flagged == len(flagged)
        assert rec.first_flagged == (min(flagged) if flagged else INT64_MAX)


def test_fold_is_identity_for_one_rank():
    recs = torch.stack([_record_words(int(np.float64(0.5).view(np.int64)), 3, 0.5, 9, 2, 17),
                        _record_words(INT64_MAX, -1, float("inf"), INT64_MAX, 0, 0)])
    gathered = distributed.gather_records(recs)   # no process group: world 1
    assert gathered.shape == (1, 2, 5)
    glob, best_h, single = distributed.fold_records(gathered)
    assert glob.tolist() == recs.tolist()
    assert best_h.tolist() == [3, -1]
    assert [r.best_h for r in distributed.read_records(single)] == [0, -1]


def test_fold_saturates_flag_count_and_rejects_bad_sizes(native_lib):
    recs = torch.stack([_record_words(INT64_MAX, -1, float("inf"), 5 + r, 0x7FFFFFF0, 0) for r in range(3)])
    glob, _, _ = distributed.fold_records(recs.reshape(3, 1, 5).contiguous())
    rec = distributed.read_records(glob)[0]
    assert rec.n_flagged == 0x7FFFFFFF and rec.first_flagged == 5
    assert native_lib.sfm_fold_select_records_host(None, 0, 1, None, None, None) == -1
    assert native_lib.sfm_fold_select_records_host(None, 2, 1, None, None, None) == -1
    assert native_lib.sfm_fold_select_records_host(None, 2, 0, None, None, None) == 0


def test_degenerate_policy_names():
    assert distributed.degenerate_policy("skip") == "skip"
    assert distributed.degenerate_policy("RAISE") == "raise"
    with pytest.raises(ValueError):
        distributed.degenerate_policy("ignore")


def test_shard_range_partitions():
    for total, world in [(100, 8), (1_000_000, 8), (7, 8), (0, 2), (100_000, 1)]:
        spans = [distributed.shard_range(total, r, world) for r in range(world)]
        covered = []
        for begin, count in spans:
            covered.extend(range(begin, begin + count))
        assert covered == list(range(total))
    assert [distributed.shard_range(1_000_000, r, 8) for r in (0, 7)] == [(0, 125_000), (875_000, 125_000)]


def test_error_bits_are_monotone_as_int64():
    x = np.sort(np.abs(np.random.default_rng(1).normal(size=1000)) * 10.0 ** np.random.default_rng(2).integers(-300, 300, 1000))
    bits = x.view(np.int64)
    assert np.all(np.diff(bits) >= 0) and bits.max() < INT64_MAX
    assert np.float64(np.inf).view(np.int64) < INT64_MAX
