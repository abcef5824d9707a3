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


# ---- C5 path (frames.run_frames_sharded + gather) at world size 2, with a stand-in context: no GPU, the same sharding / packing / gather code as bench.py's c5_leg ----
class _Obj:
    pass


class _FakeState:
    @staticmethod
    def from_pose(R, t, P, inv_expo=1.0):
        s = _Obj(); s.R, s.t, s.P, s.inv_expo = np.asarray(R, float), np.asarray(t, float), np.asarray(P, float), inv_expo
        return s


class _FakeCtx:
    """deterministic stand-in for fast-livo2_amd.Context: results are a function of the frame's content only"""
    def set_scan(self, xyz, cfg):
        self.xyz = np.asarray(xyz)

    def _state(self, prior, salt):
        st = _Obj()
        if hasattr(prior, "rot"):                    # a posterior handed on as the next prior (the visual update starts from the LiDAR posterior)
            p = _Obj(); p.R, p.t, p.P, p.inv_expo = np.array(prior.rot).reshape(3, 3), np.array(prior.pos), np.array(prior.cov).reshape(19, 19), prior.inv_expo
            prior = p
        st.rot = list((prior.R + salt).ravel()); st.pos = list(prior.t + salt); st.inv_expo = prior.inv_expo
        st.vel = [salt, 0.0, 0.0]; st.bg = [0.0] * 3; st.ba = [0.0] * 3; st.grav = [0.0, 0.0, -9.81]; st.cov = list((prior.P * (1.0 + salt)).ravel())
        return st

    def lidar_update(self, prior, prop, cfg):
        r = _Obj(); r.n_iters = 3; r.state = self._state(prior, float(self.xyz.sum()) * 1e-6)
        r.iter_sums = [_Obj() for _ in range(3)]
        for k, it in enumerate(r.iter_sums):
            it.n_eff = len(self.xyz) - k
        return r, None

    def set_frame(self, img, pos, warp, sl, ie):
        self.m = len(pos)

    # the whole-frame API (livo2_frame_update_async / _fetch): two frames in flight, the visual update chained to the LiDAR posterior
    def _frame_in(self, xyz, prior, cfg, img, pos, warp, sl, ie, vcfg):
        return (np.asarray(xyz), prior, len(pos)), None, len(pos), 1

    def new_frame_results(self):
        return None, None

    def frame_enqueue(self, fin):
        if not hasattr(self, "_q"):
            self._q = []
        assert len(self._q) < 2
        self._q.append(fin)

    def frame_update_async(self, xyz, prior, cfg, img, pos, warp, sl, ie, vcfg):
        self.frame_enqueue(self._frame_in(xyz, prior, cfg, img, pos, warp, sl, ie, vcfg)[0])

    def frame_update_fetch(self, into=None):
        xyz, prior, m = self._q.pop(0)
        self.xyz, self.m = xyz, m
        lres, _ = self.lidar_update(prior, prior, None)
        vres, _ = self.visual_update(lres.state, lres.state, None)
        return lres, vres

    def visual_update(self, prior, prop, cfg):
        v = _Obj(); v.n_steps = 2; v.state = self._state(prior, 1e-3 * self.m)
        v.steps = [_Obj(), _Obj()]; v.steps[0].error = 5.0; v.steps[1].error = 4.0 + self.m
        return v, None


def _fake_frames(n):
    rng = np.random.default_rng(0)
    out = []
    for f in range(n):
        vs = _Obj(); vs.img = np.zeros((4, 4), np.uint8); vs.pos = rng.normal(size=(3 + f % 4, 3)); vs.warp_patch = None; vs.search_levels = None; vs.inv_expo_list = None
        vs.R_prior, vs.t_prior, vs.P = np.eye(3), rng.normal(size=3), np.eye(19) * 1e-3
        out.append(dict(xyz=rng.normal(size=(10 + f, 3)).astype(np.float32), R_prior=np.eye(3), t_prior=rng.normal(size=3), P=np.eye(19) * 1e-4, vs=vs))
    return out


def _c5_worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = importlib.import_module("fast-livo2_amd.frames")
    fr = _fake_frames(n_frames)
    recs, evals = frames.run_frames_sharded([_FakeCtx(), _FakeCtx(), _FakeCtx()] if rank else _FakeCtx(), _FakeState, fr, None, None, rank, world)   # rank 1: three contexts, one thread each
    allrec = frames.gather_results(recs, n_frames, dist)
    ev = frames.gather_results(np.array([[float(evals)]]), world, dist)
    dist.barrier()
    q.put((rank, allrec.tolist(), float(ev.sum())))
    dist.destroy_process_group()


def test_c5_frames_sharded_and_gathered_world2():
    sys.path.insert(0, ROOT)
    frames = importlib.import_module("fast-livo2_amd.frames")
    world, n_frames = 2, 9
    single, ev1 = frames.run_frames_sharded(_FakeCtx(), _FakeState, _fake_frames(n_frames), None, None, 0, 1)
    assert single.shape == (n_frames, frames.RESULT_DOUBLES)
    piped, evk = frames.run_frames_sharded([_FakeCtx() for _ in range(4)], _FakeState, _fake_frames(n_frames), None, None, 0, 1)      # contexts in threads: same records
    assert np.array_equal(piped, single) and evk == ev1
    for cs in (_FakeCtx(), [_FakeCtx() for _ in range(3)]):                                                                           # one library call per frame, two in flight
        api, eva = frames.run_frames_sharded(cs, _FakeState, _fake_frames(n_frames), None, None, 0, 1, frame_api=True)
        assert np.array_equal(api, single) and eva == ev1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_c5_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allrec, ev in outs:
        assert np.array_equal(np.array(allrec), single)          # every rank holds the single-rank results, frame by frame, bit for bit
        assert ev == ev1


# ---- bench.py's c5 leg itself at world size 2 (gloo): rank 0 generates the frames and the others wait for its file, frames cycle through the distinct ones,
# ---- per-rank rates and the all-gather are reported, the gathered copy is checked on rank 0 — with stand-in contexts, no GPU
class _FakeCtx2(_FakeCtx):
    def __init__(self, device=0):
        pass

    def upload_map(self, fm):
        pass

    def synchronize(self):
        pass

    def close(self):
        pass


def _bench_c5_worker(rank, world, port, cache_dir, q):
    sys.path.insert(0, ROOT)
    os.environ["TMPDIR"] = cache_dir
    import tempfile
    tempfile.tempdir = cache_dir
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    frames = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    fake = type("FakeLivo2", (), {"Context": _FakeCtx2, "State": _FakeState})
    out = bench.c5_leg(_FakeCtx2(), fake, frames, cfgs, dist, "cpu", rank, world, 3, 4, "c1", dist.barrier, torch)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_bench_c5_leg_world2(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bench_c5_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len([f for f in os.listdir(tmp_path) if f.startswith("livo2_c5_c1_f3")]) == 1          # generated once (rank 0), read by rank 1
    o = outs[0]
    assert o["frames"] == 8 and o["frames_per_rank"] == 4 and o["distinct_frames"] == 3
    assert len(o["frames_per_s_per_rank"]) == 2 and len(o["frames_per_s_per_rank_one_context"]) == 2
    assert o["all_gather_ms"] >= 0 and o["frames_per_s"] > 0 and o["frames_per_s_one_context"] > 0
    assert o["gathered_copy_check"] == {"frames_recomputed_on_rank0": 2, "mismatches": 0, "pipelined_records_equal_one_context_records": True,
                                        "frame_api_records_equal_one_context_records": True}
    assert o["frames_per_s_frame_api"] > 0 and o["frames_per_s_frame_api_one_context"] > 0 and len(o["frames_per_s_per_rank_frame_api"]) == 2
    assert outs[1]["gathered_copy_check"] is None
