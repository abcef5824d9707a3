"""Parity on EXACTLY what bench.py times: the C4 frame bench.c4_frame(4, 200000, 4000) builds (0.1 m voxel-grid filter, 3.1 n test rays,
random thinning — not the scene of tests/test_full_size_gpu.py), the configuration structs bench.py builds (fast-livo2_amd.configs, MP_PROC_NUM = 4), the priors of
bench.frame_priors, and the three execution forms the bench line reports: one frame in flight (headline), the same 8 updates in lockstep per launch, and
independent contexts running concurrently.  Every form against the oracle, with the float decisions bit-equal."""
import importlib
import threading

import numpy as np
import pytest

import bench
from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
F = 8


@pytest.fixture(scope="module")
def c4(livo2):
    sc, vs = bench.c4_frame(4, 200000, 4000)
    assert len(sc.xyz) == 200000 and len(vs.pos) == 4000
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    lid, vis = bench.frame_priors(livo2, synth, sc, vs, F, seed=100)       # rank 0's priors (bench.py: seed = 100 + rank)
    return sc, vs, cfgs.lidar_cfg(sc), cfgs.visual_cfg(vs, mp_proc_num=4), lid, vis


def _opod(orc, st):
    s = orc.StatePOD()
    for k in ("rot", "pos", "vel", "bg", "ba", "grav", "cov"):
        getattr(s, k)[:] = list(getattr(st, k))
    s.inv_expo = st.inv_expo
    return s


@pytest.fixture(scope="module")
def oracle_results(orc, c4):
    sc, vs, cfg, vcfg, lid, vis = c4
    om = orc.OracleMap.from_flat(sc.fmap)
    ocfg = orc.lidar_cfg(sc.cfg, sc.extR, sc.extT, num_threads=4)
    ovcfg = orc.visual_cfg(vs, num_threads=4)
    out = []
    for f in range(F):
        l = orc.lidar_state_estimation(om, ocfg, sc.xyz, _opod(orc, lid[f]), _opod(orc, lid[f]))
        v = orc.visual_update(ovcfg, vs, _opod(orc, vis[f]), _opod(orc, vis[f]))
        out.append((l, v))
    return out


def _check_lidar(res, pts, ref, sc, prior):
    assert res.n_iters == ref["n_iters"]
    if pts is not None:
        assert np.array_equal(pts["match_plane"], ref["match_plane"]) and np.array_equal(pts["dis_to_plane"], ref["dis"])
    for it in range(res.n_iters):
        assert res.iter_sums[it].n_eff == ref["trace"][it].n_eff
    so, sp = H.orc.state_arrays(ref["state"]), H.orc.state_arrays(res.state)
    R0, t0 = np.array(prior.rot).reshape(3, 3), np.array(prior.pos)
    dx_ref = np.concatenate([so["t"] - t0, (R0.T @ so["R"] - np.eye(3)).ravel()])
    dx_gpu = np.concatenate([sp["t"] - t0, (R0.T @ sp["R"] - np.eye(3)).ravel()])
    assert H.relerr(dx_gpu, dx_ref) < 1e-7 and H.relerr(sp["P"], so["P"]) < 1e-8          # contract: 1e-5


def _check_visual(res, errors, ref):
    assert [(s.level, s.iteration, s.accepted, s.n_meas, s.error) for s in res.steps[:res.n_steps]] == [(t.level, t.iteration, t.accepted, t.n_meas, t.error) for t in ref["trace"]]
    if errors is not None:
        assert np.array_equal(errors, ref["errors"])
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-8 and d["t"] < 1e-8 and d["P"] < 1e-7 and d["inv_expo"] < 1e-8, d


def test_headline_frames_match_oracle(ctx, c4, oracle_results):
    """one frame in flight: the launch sequence of bench.C4.enqueue_step, frame by frame"""
    sc, vs, cfg, vcfg, lid, vis = c4
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    n0, t0 = ctx.counter("visual_persistent_launches"), ctx.counter("visual_persistent_timeouts")
    for f in range(F):
        res, pts = ctx.lidar_update(lid[f], lid[f], cfg, want=("match_plane", "dis_to_plane"))
        _check_lidar(res, pts, oracle_results[f][0], sc, lid[f])
        vres, errors = ctx.visual_update(vis[f], vis[f], vcfg)
        _check_visual(vres, errors, oracle_results[f][1])
    assert ctx.counter("visual_persistent_launches") == n0 + F and ctx.counter("visual_persistent_timeouts") == t0      # the resident grid ran AND finished (a grid that gives up is re-run per step)


def test_lockstep_batch_equals_single_updates_and_oracle(ctx, c4, oracle_results):
    """extra.c4_lockstep: the same 8 frame updates as ONE batch per launch, against the 8 single updates and the oracle.  Visual: bit-equal (same bodies, same
    per-frame reduction order).  LiDAR: the batch grid uses 64-point blocks (single scans: 256), so its partial sums are added in another order — identical
    decisions (iteration counts, n_eff), states to rounding."""
    sc, vs, cfg, vcfg, lid, vis = c4
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ctx.set_option("visual_persistent", 0)                         # the batch kernels share the per-step bodies: bit-equality is with that sequence
    try:
        single = [(ctx.lidar_update(lid[f], lid[f], cfg)[0], ctx.visual_update(vis[f], vis[f], vcfg)) for f in range(F)]
    finally:
        ctx.set_option("visual_persistent", 1)
    ctx.batch_set_scans([sc.xyz] * F, cfg)
    ctx.visual_batch_set_frames([(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)] * F)
    rb = ctx.batch_update(lid, lid, cfg)
    vb = ctx.visual_batch_update(vis, vis, vcfg)
    for f in range(F):
        assert rb[f].n_iters == single[f][0].n_iters
        assert [rb[f].iter_sums[i].n_eff for i in range(rb[f].n_iters)] == [single[f][0].iter_sums[i].n_eff for i in range(rb[f].n_iters)]
        d = H.state_diff(rb[f].state, single[f][0].state)
        assert d["R"] < 1e-12 and d["t"] < 1e-12 and d["P"] < 1e-11, d
        assert vb[f].n_steps == single[f][1][0].n_steps and bytes(vb[f].state) == bytes(single[f][1][0].state)
        _check_lidar(rb[f], None, oracle_results[f][0], sc, lid[f])
        _check_visual(vb[f], None, oracle_results[f][1])


def test_concurrent_contexts_give_the_same_bits(livo2, ctx, c4):
    """extra.c4_concurrent_chains: K contexts (own stream, own host thread) running whole frames at the same time == one context"""
    sc, vs, cfg, vcfg, lid, vis = c4
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ref = [(ctx.lidar_update(lid[f], lid[f], cfg)[0], ctx.visual_update(vis[f], vis[f], vcfg)[0]) for f in range(F)]
    ctxs = [livo2.Context(0) for _ in range(2)]
    for c in ctxs:
        c.upload_map(sc.fmap); c.set_scan(sc.xyz, cfg)
        c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    bad = []

    def work(c):
        try:
            for rep in range(3):
                for f in range(F):
                    r = c.lidar_update(lid[f], lid[f], cfg)[0]
                    v = c.visual_update(vis[f], vis[f], vcfg)[0]
                    if bytes(r.state) != bytes(ref[f][0].state) or bytes(v.state) != bytes(ref[f][1].state) or v.n_steps != ref[f][1].n_steps:
                        bad.append((rep, f))
        except BaseException as exc:                     # (an exception in a thread would otherwise only be a pytest warning)
            bad.append(repr(exc))
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    for c in ctxs:
        c.close()
    assert not bad, bad


def test_block_order_of_the_residual_launch_does_not_change_a_bit(ctx, c4):
    """option "lidar_block_order": k_lidar_residual starts the chunks that lived longest in the previous launch first (the order is written by the solve of the previous
    iteration; 784 chunks at C4 = more than one round of blocks, so it is active here).  Chunk index, not launch slot, selects points and partial row: same bytes."""
    sc, vs, cfg, vcfg, lid, vis = c4
    ctx.upload_map(sc.fmap)
    out = {}
    for on in (1, 0, 1):
        ctx.set_option("lidar_block_order", on)
        ctx.set_scan(sc.xyz, cfg)
        rs = []
        for rep in range(2):                     # second pass: the order left by the previous update of the same scan is in use from the first iteration on
            for f in range(3):
                r, pts = ctx.lidar_update(lid[f], lid[f], cfg, want=("match_plane", "dis_to_plane"))
                rs.append((bytes(r.state), r.n_iters, pts["match_plane"].tobytes(), pts["dis_to_plane"].tobytes(), bytes(r.iter_sums[0])))
        out.setdefault(on, []).append(rs)
    ctx.set_option("lidar_block_order", 1)
    assert out[1][0] == out[0][0] == out[1][1]
    assert out[1][0][:3] == out[1][0][3:]


def test_fused_iteration_launch_equals_the_two_launch_sequence(ctx, c4):
    """option "lidar_fused_iteration" (round 5): one launch per ESIKF iteration — the last block of the residual grid to publish its row reduces all rows in
    reduce_partials_block's order and solves — against k_lidar_residual + k_lidar_solve: every byte of the result block (state, P, per-iteration sums and
    solutions, counters) and the per-point outputs, with and without the block order, update after update on the same scan (tickets re-zeroed by the header)."""
    sc, vs, cfg, vcfg, lid, vis = c4
    ctx.upload_map(sc.fmap)
    out = {}
    for fused, order in ((1, 1), (0, 1), (1, 0), (0, 0), (1, 1)):
        ctx.set_option("lidar_fused_iteration", fused); ctx.set_option("lidar_block_order", order)
        ctx.set_scan(sc.xyz, cfg)
        n0 = ctx.counter("lidar_fused_launches")
        rs = []
        for rep in range(2):
            for f in range(3):
                r, pts = ctx.lidar_update(lid[f], lid[f], cfg, want=("match_plane", "dis_to_plane"))
                rs.append((bytes(r), pts["match_plane"].tobytes(), pts["dis_to_plane"].tobytes()))
        assert (ctx.counter("lidar_fused_launches") > n0) == bool(fused)
        out.setdefault((fused, order), []).append(rs)
    ctx.set_option("lidar_fused_iteration", 0); ctx.set_option("lidar_block_order", 1)
    ref = out[(0, 0)][0]
    assert all(rs == ref for v in out.values() for rs in v)


def _meaningful(r):
    """the result block without the per-iteration slots the update did not reach (they keep whatever an earlier update left there)"""
    return (bytes(r.state), r.n_iters, r.converged, bytes(r.position_last), [bytes(r.iter_sums[i]) for i in range(r.n_iters)], [bytes(r.iter_solution[i]) for i in range(r.n_iters)])


def test_fused_iteration_small_scans_and_fixed_iteration_loops(ctx, livo2):
    """the same equality where the grid is a single round of blocks (8 ... 64 chunks: ticket chunks - 2 and chunks - 1 may be drawn by blocks that started together),
    for an empty scan, and for the benchmark loop of more iterations than the header has ticket slots (livo2_lidar_iterations_async, mode 2)."""
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    for n in (0, 1, 300, 2047, 2048, 16000):
        sc = synth.lidar_scenario(seed=31 + n % 7, n_points=max(n, 64), downsample=0.1)
        xyz = sc.xyz[:n]
        cfg = cfgs.lidar_cfg(sc)
        st = livo2.State.from_pose(sc.R_prior, sc.t_prior, sc.P)
        ctx.upload_map(sc.fmap)
        res = {}
        for fused in (1, 0):
            ctx.set_option("lidar_fused_iteration", fused)
            ctx.set_scan(xyz, cfg)
            a = [_meaningful(ctx.lidar_update(st, st, cfg)[0]) for _ in range(3)]
            ctx.lidar_iterations_async(st, st, cfg, 37); ctx.synchronize()
            b = bytes(ctx.lidar_update_fetch().state)
            res[fused] = (a, b)
        ctx.set_option("lidar_fused_iteration", 0)
        assert res[1] == res[0], n
        assert res[1][0][0] == res[1][0][1] == res[1][0][2]
