"""N>1 path on CPU: two processes (gloo), each tracking its own shard of streams with the kernel-logic
emulator through the C++ host (sgx_tracker_*), then the same collectives bench.py uses (max-over-ranks time, gather of the
packed per-frame records to rank 0).  Checks: shards are disjoint, rank 0 ends up with every rank's records bit for bit,
the library's one-kernel packer equals the tensor-slice packer."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from sg_slam_amd import synth, dist as sdist
    from sg_slam_amd.capi import SgxLib
    from sg_slam_amd.tracker import TrackerBatch
    from sg_slam_amd.tracker_native import TrackerNative
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = SgxLib(os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so'))
    cam = dict(synth.TUM3); S = 1
    gen = synth.PlaneStream(seed=1234)
    offs = sdist.stream_offsets(rank, S)
    tr = TrackerBatch(lib, S, cam, xp='numpy'); nat = TrackerNative(lib, S, cam, dynamic_mask=False, pipelined=False)
    tr.set_initial_pose(np.stack([gen.Tcw(o) for o in offs])); nat.set_initial_pose(np.stack([gen.Tcw(o) for o in offs]))
    held = []
    for t in range(3):
        fr = [gen.frame(o + t) for o in offs]
        g, d = np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]); held.append((g, d))
        tr.step(g, d); nat.step(g, d)
    n, nm, ninl = tr.last_counts()
    rec = sdist.gather_frame_records(dist, torch.from_numpy(tr.Tcw[1].copy()), torch.from_numpy(ninl.copy()), torch.from_numpy(nm.copy()))
    # config-5 record gather as specified: keypoints + descriptors + pose of the last frame, one gather to rank 0 per step
    G = sdist.FrameRecordGather(dist, S, nat.cap, 'cpu', async_stream=False)
    c = tr.cur
    r_native = G.submit_tracker(nat); G.wait()                        # the library's packer
    got = G.unpack(r_native) if rank == 0 else None
    r_torch = G.submit(torch.from_numpy(tr.n[c].copy()), torch.from_numpy(tr.keys[c].copy()), torch.from_numpy(tr.desc[c].copy()), torch.from_numpy(tr.Tcw[1].copy())); G.wait()
    assert (r_native is None) == (rank != 0) and (r_torch is None) == (rank != 0)
    if rank == 0: assert (r_native == r_torch).all()                 # one-kernel packer == tensor-slice packer (and native tracker == Python orchestration)
    mine = dict(n=tr.n[c].copy(), keys=tr.keys[c].copy(), desc=tr.desc[c].copy(), Tcw=tr.Tcw[1].copy())
    tmax = sdist.max_over_ranks(dist, float(rank + 1), 'cpu')
    tot = sdist.sum_over_ranks(dist, [1.0, float(ninl.sum())], 'cpu')
    q.put((rank, offs, rec.numpy(), tmax, tot, got, mine))
    dist.destroy_process_group()


def test_two_rank_sharded_tracking(emu):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda r: r[0])
    for p in ps: p.join(60)
    (r0, o0, rec0, tm0, tot0, got0, mine0), (r1, o1, rec1, tm1, tot1, got1, mine1) = res
    assert set(o0).isdisjoint(o1)                         # weak scaling: different streams per rank
    assert rec0.shape == (2, 1, 18) and (rec0 == rec1).all()    # every rank holds every rank's records
    assert tm0 == tm1 == 2.0 and tot0[0] == 2.0
    assert not np.allclose(rec0[0], rec0[1])              # the two shards really tracked different streams
    assert (rec0[:, :, 16] > 100).all()                   # inliers: both shards tracked
    # the gathered frame records: rank 0 holds every rank's keypoints / descriptors / pose of the step, bit for bit; the other rank receives nothing
    assert got1 is None
    for got in (got0,):
        for r, mine in enumerate((mine0, mine1)):
            n = int(mine['n'][0])
            assert got['n'][r, 0] == n and n > 500
            assert (got['keys'][r, 0, :n] == mine['keys'][0, :n]).all() and (got['desc'][r, 0, :n] == mine['desc'][0, :n]).all()
            assert (got['Tcw'][r, 0].view(np.uint32) == mine['Tcw'][0].view(np.uint32)).all()
