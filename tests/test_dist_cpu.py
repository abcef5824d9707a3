"""world_size-2 gloo test of the N>1 path: frame sharding, result gather and the max-over-ranks timing rule (no GPU needed)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = importlib.import_module("fast-livo2_amd.frames")
    mine = frames.frames_for_rank(n_frames, rank, world)
    local = np.array([[f, f * f, 100 + f] for f in mine], float).reshape(len(mine), 3)     # stand-in for per-frame results
    allres = frames.gather_results(local, n_frames, dist)
    tmax = frames.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    q.put((rank, allres.tolist(), tmax, mine))
    dist.destroy_process_group()


def test_frame_sharding_and_gather_world2():
    world, n_frames = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[f, f * f, 100 + f] for f in range(n_frames)]
    seen = []
    for rank, allres, tmax, mine in outs:
        assert allres == expect
        assert tmax == 2.0                      # MAX over ranks of (1 + rank)
        seen += mine
    assert sorted(seen) == list(range(n_frames))


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    frames = importlib.import_module("fast-livo2_amd.frames")
    assert frames.frames_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    a = np.arange(6.0).reshape(3, 2)
    assert np.array_equal(frames.gather_results(a, 3), a)
    assert frames.max_over_ranks(3.5) == 3.5
