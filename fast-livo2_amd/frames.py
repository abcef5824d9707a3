"""Frame-level sharding across the GPUs of one node: frames are independent once captured, so rank r takes frames
r, r+world, ... and the only exchange is the gather of the tiny per-frame results (state increment, n_eff, ...).  No data-path
collective.  Works with any torch.distributed backend ("nccl" == RCCL on the GPUs, "gloo" in the CPU tests)."""
import numpy as np


def frames_for_rank(n_frames, rank, world):
    return list(range(rank, n_frames, world))


def gather_results(local, n_frames, dist=None, device="cpu"):
    """local: float64 [n_local, K] results of frames_for_rank(n_frames, rank, world) in that order.
    Returns float64 [n_frames, K] in frame order on every rank (one all_gather of a padded buffer)."""
    local = np.ascontiguousarray(local, np.float64).reshape(len(local), -1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(local) == n_frames
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    K = local.shape[1] if local.size else 0
    kk = torch.tensor([K], dtype=torch.int64, device=device)
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    K = int(kk.item())
    per = (n_frames + world - 1) // world
    buf = torch.zeros((per, K), dtype=torch.float64, device=device)
    if len(local):
        buf[: len(local)] = torch.from_numpy(local).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = np.zeros((n_frames, K))
    for r in range(world):
        idx = frames_for_rank(n_frames, r, world)
        res[idx] = out[r][: len(idx)].cpu().numpy()
    return res


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX of a python float over all ranks (the bench's timing rule)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- whole frames, transfers included --------------------------------------------------------------------------------------
RESULT_DOUBLES = 2 * (25 + 361) + 4         # per frame: LiDAR posterior (25 scalars + P) + n_iters + n_eff of the last iteration, visual posterior + n_steps + last error


def pack_result(lres, vres):
    """the per-frame record that travels to rank 0: what LIVMapper keeps of a frame (state_ after handleLIO, state after handleVIO) plus the loop counters"""
    def st(s):
        return np.concatenate([np.array(s.rot), np.array(s.pos), [s.inv_expo], np.array(s.vel), np.array(s.bg), np.array(s.ba), np.array(s.grav), np.array(s.cov)])
    n_it = int(lres.n_iters)
    last_err = float(vres.steps[vres.n_steps - 1].error) if vres.n_steps > 0 else 0.0
    return np.concatenate([st(lres.state), [n_it, float(lres.iter_sums[n_it - 1].n_eff) if n_it > 0 else 0.0], st(vres.state), [float(vres.n_steps), last_err]])


def run_frame(ctx, State, frame, cfg, vcfg):
    """One LIO + VIO frame through the C ABI as LIVMapper::handleLIO / handleVIO drive it (LIVMapper.cpp:336-482, 281-334): the down-sampled scan goes up
    (H2D + per-scan precompute), StateEstimation from the frame's prior, the image + visual sub-map go up, computeJacobianAndUpdateEKF runs on the SHARED state:
    `state` is the LiDAR posterior (LIVMapper.cpp:135, 371) and so is `state_propagat` (processImu re-assigns it from _state before the VIO step,
    LIVMapper.cpp:256); the two results come back (D2H).  The map stays resident.  The frame's image / sub-map are generated at the pose the scan was
    taken from (scenarios.synth.frame_sequence), so the chained update is well posed."""
    vs = frame["vs"]
    prior = State.from_pose(frame["R_prior"], frame["t_prior"], frame["P"], inv_expo=getattr(vs, "tau_prior", 1.0))
    ctx.set_scan(frame["xyz"], cfg)
    lres, _ = ctx.lidar_update(prior, prior, cfg)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    vres, _ = ctx.visual_update(lres.state, lres.state, vcfg)
    return pack_result(lres, vres), int(lres.n_iters) * len(frame["xyz"]) + int(vres.n_steps) * 64 * len(vs.pos)


def prepare_frames(ctx, State, frames, cfg, vcfg, indices=None):
    """the livo2_frame_in structs of the frames (pointers into the caller's arrays) + result blocks, built ahead of a timed loop: in the reference's C++ this is
    a handful of pointer assignments per frame; in Python it is numpy / ctypes bookkeeping that has nothing to do with the library"""
    idx = list(range(len(frames))) if indices is None else list(indices)
    prep = []
    for f in idx:
        fr = frames[f]; vs = fr["vs"]
        prior = State.from_pose(fr["R_prior"], fr["t_prior"], fr["P"], inv_expo=getattr(vs, "tau_prior", 1.0))
        fin, keep, M, L = ctx._frame_in(fr["xyz"], prior, cfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg)
        prep.append([f, fin, keep] + list(ctx.new_frame_results()))
    return prep


def run_prepared(ctx, frames, prep):
    """two frames in flight on one context over prepared frames; returns (records, evaluations) like run_frames_pipelined"""
    pending = []
    for k, (f, fin, keep, lres, vres) in enumerate(prep):
        ctx.frame_enqueue(fin)
        pending.append(k)
        if len(pending) == 2:
            j = pending.pop(0)
            prep[j][3], prep[j][4] = ctx.frame_update_fetch(into=(prep[j][3], prep[j][4]))
    for j in pending:
        prep[j][3], prep[j][4] = ctx.frame_update_fetch(into=(prep[j][3], prep[j][4]))
    recs = np.zeros((len(prep), RESULT_DOUBLES))
    evals = 0
    for k, (f, fin, keep, lres, vres) in enumerate(prep):
        recs[k] = pack_result(lres, vres)
        evals += int(lres.n_iters) * len(frames[f]["xyz"]) + int(vres.n_steps) * 64 * len(frames[f]["vs"].pos)
    return recs, evals


def run_frames_pipelined(ctx, State, frames, cfg, vcfg, indices=None):
    """The same frames through livo2_frame_update_async / _fetch on ONE context, two frames in flight: frame k + 1 is enqueued (scan and image staged, every launch of
    both updates queued behind frame k's on the stream) before frame k's results are fetched, so the host's work for one frame hides behind the GPU's work for the
    other; the visual update takes the LiDAR posterior on the device.  Records identical to run_frame's (tests/test_c5_gpu.py)."""
    idx = list(range(len(frames))) if indices is None else list(indices)
    recs = np.zeros((len(idx), RESULT_DOUBLES))
    evals, pending = 0, []

    def enqueue(f):
        fr = frames[f]; vs = fr["vs"]
        prior = State.from_pose(fr["R_prior"], fr["t_prior"], fr["P"], inv_expo=getattr(vs, "tau_prior", 1.0))
        ctx.frame_update_async(fr["xyz"], prior, cfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg)
        return prior                                                   # (kept alive until the fetch)

    def fetch(k, f):
        nonlocal evals
        lres, vres = ctx.frame_update_fetch()
        recs[k] = pack_result(lres, vres)
        evals += int(lres.n_iters) * len(frames[f]["xyz"]) + int(vres.n_steps) * 64 * len(frames[f]["vs"].pos)
    for k, f in enumerate(idx):
        pending.append((k, f, enqueue(f)))
        if len(pending) == 2:
            kk, ff, _ = pending.pop(0)
            fetch(kk, ff)
    for kk, ff, _ in pending:
        fetch(kk, ff)
    return recs, evals


def run_prepared_sharded(ctxs, frames, preps):
    """preps[j] = prepare_frames(ctxs[j], ..., share j of the rank's frames): every context pipelines its share from its own host thread.  Returns per context (records, evaluations)."""
    import threading
    out, errors = [None] * len(ctxs), []

    def work(j):
        try:
            out[j] = run_prepared(ctxs[j], frames, preps[j])
        except BaseException as exc:
            errors.append(exc)
    if len(ctxs) == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(j,)) for j in range(len(ctxs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    if errors:
        raise errors[0]
    return out


def run_frames_sharded(ctx, State, frames, cfg, vcfg, rank, world, frame_api=False):
    """rank's share (frames rank, rank + world, ...) in order; returns (records [n_local, RESULT_DOUBLES], residual evaluations done).
    `ctx` may be a list of K contexts of the rank's GPU (each with the map resident): frame k of the share then runs on context k % K from its own host thread, so
    that the H2D of one frame, the updates of another and the D2H of a third overlap on the contexts' streams (the C ABI calls release the GIL).  The records do
    not depend on K: a context's results depend on its inputs only.  frame_api: every frame is ONE library call (livo2_frame_update_async / _fetch, two frames in
    flight per context) instead of the four calls of run_frame — same records."""
    mine = frames_for_rank(len(frames), rank, world)
    recs = np.zeros((len(mine), RESULT_DOUBLES))
    ctxs = list(ctx) if isinstance(ctx, (list, tuple)) else [ctx]
    if frame_api and (len(ctxs) == 1 or len(mine) < 2):
        return run_frames_pipelined(ctxs[0], State, frames, cfg, vcfg, mine)
    if len(ctxs) == 1 or len(mine) < 2:
        evals = 0
        for k, f in enumerate(mine):
            recs[k], e = run_frame(ctxs[0], State, frames[f], cfg, vcfg)
            evals += e
        return recs, evals
    import threading
    K = len(ctxs)
    evals, errors = [0] * K, []

    def work(j):
        try:
            if frame_api:                                                  # each context pipelines its own share, two frames in flight
                ks = list(range(j, len(mine), K))
                r, evals[j] = run_frames_pipelined(ctxs[j], State, frames, cfg, vcfg, [mine[k] for k in ks])
                recs[ks] = r
                return
            for k in range(j, len(mine), K):
                recs[k], e = run_frame(ctxs[j], State, frames[mine[k]], cfg, vcfg)
                evals[j] += e
        except BaseException as exc:                                   # re-raised on the caller's thread
            errors.append(exc)
    th = [threading.Thread(target=work, args=(j,)) for j in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    return recs, sum(evals)
