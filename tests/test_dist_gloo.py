"""world_size-2 (and 3) CPU test of the N > 1 exchange protocol with the gloo backend: every
rank runs the local WTA of its disparity shard (CPU oracle standing in for k_wta), packs the
keys exactly as the device does, ONE all_gather, signed minimum -> must equal the unsharded
DispSel::CVSelect result.  The GPU flavour of the same steps is bench.py --gpus N (RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, D, H, W, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import psm_oracle_py as O
        from shard_model import pack_keys, shard_bounds, unpack_disp
        rng = np.random.default_rng(123)                       # same volume on every rank
        vol = rng.integers(0, 5, size=(2, D, H, W)).astype(np.float32)   # many ties
        vol[0, :, 0, 0] = np.nan
        vol[1, min(3, D - 1), 1, 1] = -0.0
        d0, d1 = shard_bounds(D, world)[rank]
        keys = np.empty((2, H, W), np.int64)
        for side in range(2):
            if d1 > d0:
                c, d = O.wta_partial(vol[side, d0:d1], d0, d1)
            else:
                c, d = np.full((H, W), np.inf, np.float32), np.zeros((H, W), np.int32)
            keys[side] = pack_keys(c, d)
        local = torch.from_numpy(keys.reshape(-1))
        gathered = torch.empty(world * local.numel(), dtype=torch.int64)
        dist.all_gather_into_tensor(gathered, local)            # exchange flavour 1: all-gather + min
        best = gathered.view(world, -1).min(dim=0).values.numpy()
        maps = unpack_disp(best).reshape(2, H, W)
        reduced = local.clone()
        dist.all_reduce(reduced, op=dist.ReduceOp.MIN)          # exchange flavour 2: all-reduce(min)
        maps2 = unpack_disp(reduced.numpy()).reshape(2, H, W)
        ref = np.stack([O.wta(vol[0]), O.wta(vol[1])])
        q.put((rank, bool(np.array_equal(maps, ref)) and bool(np.array_equal(maps2, ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,D", [(2, 16), (3, 7), (2, 1)])
def test_allgather_min_reproduces_wta(world, D):
    from oracle import psm_oracle_py as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world * 10 + D
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, 9, 13, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_pack_keys_order():
    from shard_model import pack_keys, unpack_disp
    c = np.array([np.inf, 1.0, 1.0, -2.0, 0.0, -0.0, 3e-39], np.float32)
    d = np.array([0, 5, 4, 9, 7, 6, 2], np.int32)
    k = pack_keys(c, d)
    order = np.argsort(k, kind="stable")
    assert list(order) == [3, 5, 4, 6, 2, 1, 0]     # -2 < (+-0: d=6 < d=7) < denormal < (1.0: d=4 < d=5) < inf
    assert np.array_equal(unpack_disp(k), d.astype(np.uint8))


def _stripe_worker(rank, world, port, H, W, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from primestereomatch_amd import stripes
        rng = np.random.default_rng(7)                          # the same "whole-image" maps on every rank
        full = rng.integers(0, 256, size=(2, H, W), dtype=np.uint8)
        R, y0, y1 = stripes.stripe_bounds(H, world, rank)
        mine = torch.full((2 * H * W + 4,), 255, dtype=torch.uint8)       # rows outside the stripe: garbage, never sent
        mine[:2 * H * W].view(2, H, W)[:, y0:y1] = torch.from_numpy(full[:, y0:y1].copy())
        send = torch.zeros(2 * R * W, dtype=torch.uint8)
        recv = torch.zeros(world * 2 * R * W, dtype=torch.uint8)
        stripes.pack_stripe(mine, y0, y1, send, H, W, R)
        dist.all_gather_into_tensor(recv, send)                 # the one exchange step of bench.py --shard rows
        out = torch.zeros(2 * H * W + 4, dtype=torch.uint8)
        stripes.assemble(recv, world, H, W, R, out)
        q.put((rank, bool(np.array_equal(out[:2 * H * W].view(2, H, W).numpy(), full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 12), (3, 10), (2, 7), (4, 9)])
def test_row_stripe_gather_rebuilds_the_maps(world, H):
    """bench.py --shard rows: aligned stripes of ceil(H / N) rows, one all_gather, a transpose - also when H % N != 0, when
    the last rank's stripe is shorter, and when it is empty (N = 4, H = 9: 3-row stripes; bench.py itself refuses that)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world * 20 + H
    procs = [ctx.Process(target=_stripe_worker, args=(r, world, port, H, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(res[r] for r in range(world)), res


def test_stripe_bounds_tile_the_image():
    from primestereomatch_amd import stripes
    for H in (1, 7, 288, 375, 1080, 2160):
        for G in (1, 2, 3, 4, 8):
            rows = []
            for g in range(G):
                R, y0, y1 = stripes.stripe_bounds(H, G, g)
                assert 0 <= y0 <= y1 <= H and y1 - y0 <= R and (y0 == g * R or y0 == H)
                rows += list(range(y0, y1))
            assert rows == list(range(H))


def _exchange_worker(rank, world, port, q):
    """primestereomatch_amd.exchange.Exchange over gloo, CPU tensors standing in for device tensors: the frame-pipelined
    protocol of bench.py (two alternating key tensors, the collective of frame i finished inside step i+1) for both
    exchanges, against the unsharded result."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from primestereomatch_amd import stripes
        from primestereomatch_amd.exchange import Exchange
        ex = Exchange(torch, dist, "gloo")
        H, W, frames = 10, 7, 5
        n = 2 * H * W
        ok = True
        # ---- disparity shards: per frame a key tensor per rank; the minimum over ranks one step later ----
        rng = np.random.default_rng(5)
        allkeys = rng.integers(-2**40, 2**40, size=(frames, world, n), dtype=np.int64)    # every rank knows all (to check)
        kbuf = [torch.empty(n, dtype=torch.int64) for _ in range(2)]
        pending, got = [], []
        for f in range(frames):
            kb = kbuf[f & 1]
            kb.copy_(torch.from_numpy(allkeys[f, rank]))
            while pending:                                  # finish_pending(): frame f-1's collective -> its result
                w, b = pending.pop(0)
                w.wait()
                got.append(b.clone().numpy())
            pending.append((ex.all_reduce_min(kb, async_op=True), kb))
        w, b = pending.pop(0); w.wait(); got.append(b.clone().numpy())
        ok &= all(np.array_equal(got[f], allkeys[f].min(axis=0)) for f in range(frames))
        gat = torch.empty(world * n, dtype=torch.int64)
        ex.all_gather(gat, torch.from_numpy(allkeys[0, rank].copy()))
        ok &= bool(np.array_equal(gat.view(world, n).numpy(), allkeys[0]))
        # ---- row stripes: the finished rows of both maps, gathered asynchronously, assembled one step later ----
        full = rng.integers(0, 256, size=(frames, 2, H, W), dtype=np.uint8)
        R, y0, y1 = stripes.stripe_bounds(H, world, rank)
        mbuf = [torch.zeros(n + 4, dtype=torch.uint8) for _ in range(2)]
        send, recv = torch.zeros(2 * R * W, dtype=torch.uint8), torch.zeros(world * 2 * R * W, dtype=torch.uint8)
        pending, maps = [], []
        for f in range(frames):
            mb = mbuf[f & 1]
            mb.fill_(255)
            mb[:n].view(2, H, W)[:, y0:y1] = torch.from_numpy(full[f][:, y0:y1].copy())
            while pending:
                w, b = pending.pop(0)
                w.wait()
                stripes.assemble(recv, world, H, W, R, b)
                maps.append(b[:n].clone().numpy().reshape(2, H, W))
            stripes.pack_stripe(mb, y0, y1, send, H, W, R)
            pending.append((ex.all_gather(recv, send, async_op=True), mb))
        w, b = pending.pop(0); w.wait(); stripes.assemble(recv, world, H, W, R, b); maps.append(b[:n].clone().numpy().reshape(2, H, W))
        ok &= all(np.array_equal(maps[f], full[f]) for f in range(frames))
        # ---- two frames in flight per rank (bench.py --gpus N, round 6): frame f goes to slot f % 2 - a slot has its own two map
        # tensors, its own send / receive buffers and its own pending list, and finishes a frame's exchange inside its NEXT frame's
        # step (two steps later); the collectives of the two slots interleave in the same order on every rank ----
        F = 2
        frames2 = 7
        full2 = rng.integers(0, 256, size=(frames2, 2, H, W), dtype=np.uint8)
        keys2 = rng.integers(-2**40, 2**40, size=(frames2, world, n), dtype=np.int64)

        class Slot:
            def __init__(self):
                self.mbuf = [torch.zeros(n + 4, dtype=torch.uint8) for _ in range(2)]
                self.kbuf = [torch.empty(n, dtype=torch.int64) for _ in range(2)]
                self.send, self.recv = torch.zeros(2 * R * W, dtype=torch.uint8), torch.zeros(world * 2 * R * W, dtype=torch.uint8)
                self.pending, self.n = [], 0
        slots = [Slot() for _ in range(F)]
        done_maps, done_keys = {}, {}

        def finish(sl):
            while sl.pending:
                f_, wm, mb_, wk, kb_ = sl.pending.pop(0)
                wm.wait(); wk.wait()
                stripes.assemble(sl.recv, world, H, W, R, mb_)
                done_maps[f_] = mb_[:n].clone().numpy().reshape(2, H, W)
                done_keys[f_] = kb_.clone().numpy()
        for f in range(frames2):
            sl = slots[f % F]
            mb, kb = sl.mbuf[sl.n & 1], sl.kbuf[sl.n & 1]
            sl.n += 1
            mb.fill_(255)
            mb[:n].view(2, H, W)[:, y0:y1] = torch.from_numpy(full2[f][:, y0:y1].copy())      # "the filter" of frame f on this slot
            kb.copy_(torch.from_numpy(keys2[f, rank]))
            finish(sl)                                                                        # this slot's previous frame (f - F)
            stripes.pack_stripe(mb, y0, y1, sl.send, H, W, R)
            sl.pending.append((f, ex.all_gather(sl.recv, sl.send, async_op=True), mb, ex.all_reduce_min(kb, async_op=True), kb))
            assert len(done_maps) == max(0, f + 1 - F)                                        # exactly F frames are ever in flight
        for sl in slots:
            finish(sl)
        ok &= sorted(done_maps) == list(range(frames2))
        ok &= all(np.array_equal(done_maps[f], full2[f]) and np.array_equal(done_keys[f], keys2[f].min(axis=0)) for f in range(frames2))
        ok &= ex.max_float(float(rank)) == float(world - 1) and ex.collectives == 2 * frames + 1 + 2 * frames2
        # per-rank diagnostics of the bench line (compute / collective ms of every rank, on every rank)
        per = ex.gather_floats([10.0 + rank, 0.5 * rank])
        ok &= per == [[10.0 + r_, 0.5 * r_] for r_ in range(world)]
        ex.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_staged_exchange_frame_pipeline(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, 29850 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(res[r] for r in range(world)), res
