"""BASELINE.json configs[4] ("C5": batched distinct frames) on the GPU against the oracle — the path bench.py's c5 leg and `--gpus N` run
(fast-livo2_amd/frames.py::run_frames_sharded: per frame scan H2D + per-scan precompute, full LiDAR update from the frame's prior, image + sub-map H2D, full
visual update STARTING FROM THE LiDAR POSTERIOR, results D2H) with ONE real Context and with THREE (one host thread and one stream each, the 3-context form
the bench times, where a frame's resident visual grid meets the LiDAR kernels and the rocPRIM sorts of the other contexts).
Every 776-double record is compared with the oracle running the same chain (reference shape: LIVMapper.cpp:336-482 handleLIO, 281-334 handleVIO; the VIO update
reads the state the LIO update left, LIVMapper.cpp:135-136, 371; vio.cpp:1799-1810): loop counters and n_eff equal, posterior states to 1e-7, P to 1e-6
(the visual posterior inherits the LiDAR posterior's ~1e-9 difference as its prior)."""
import importlib

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


class _Sc:
    def __init__(self, cfg, extR, extT):
        self.cfg, self.extR, self.extT = cfg, extR, extT


def _oracle_records(orc, frames_mod, fmap, lio_cfg, extR, extT, seq, mp_proc_num):
    om = orc.OracleMap.from_flat(fmap)
    ocfg = orc.lidar_cfg(lio_cfg, extR, extT)
    recs, trails = [], []
    for fr in seq:
        vs = fr["vs"]
        prior = orc.make_state(fr["R_prior"], fr["t_prior"], fr["P"], inv_expo=vs.tau_prior)
        lref = orc.lidar_state_estimation(om, ocfg, fr["xyz"], prior, prior, want_points=False)
        vref = orc.visual_update(orc.visual_cfg(vs, num_threads=mp_proc_num), vs, lref["state"], lref["state"])
        n_it = lref["n_iters"]

        def st(s):
            return np.concatenate([np.array(s.rot), np.array(s.pos), [s.inv_expo], np.array(s.vel), np.array(s.bg), np.array(s.ba), np.array(s.grav), np.array(s.cov)])
        tr = vref["trace"]
        recs.append(np.concatenate([st(lref["state"]), [n_it, float(lref["trace"][n_it - 1].n_eff) if n_it else 0.0], st(vref["state"]),
                                    [float(len(tr)), float(tr[-1].error) if tr else 0.0]]))
        trails.append((fr, lref, vref))
    return np.array(recs), trails


def _compare(got, want, frames_mod):
    K = 25 + 361
    assert got.shape == want.shape == (len(want), frames_mod.RESULT_DOUBLES)
    for f in range(len(want)):
        g, w = got[f], want[f]
        # loop counters, n_eff of the last LiDAR iteration, visual step count: decisions, equal
        assert g[K] == w[K] and g[K + 1] == w[K + 1], (f, g[K:K + 2], w[K:K + 2])
        assert g[2 * K + 2] == w[2 * K + 2], (f, g[2 * K + 2], w[2 * K + 2])
        assert g[2 * K + 3] == w[2 * K + 3], (f, "last float frame error", g[2 * K + 3], w[2 * K + 3])
        for base, tol_x, tol_P in ((0, 1e-7, 1e-7), (K + 2, 1e-7, 1e-6)):
            gx, wx = g[base:base + 25], w[base:base + 25]
            assert np.abs(gx - wx).max() < tol_x, (f, base, np.abs(gx - wx).max())
            gP, wP = g[base + 25:base + K], w[base + 25:base + K]
            assert np.linalg.norm(gP - wP) <= tol_P * np.linalg.norm(wP), (f, base, np.linalg.norm(gP - wP) / np.linalg.norm(wP))


@pytest.mark.parametrize("shape", ["c1", "c4"])
def test_c5_frames_one_and_three_contexts_match_the_oracle(livo2, orc, shape):
    frames_mod = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    if shape == "c1":                                                         # avia-like: ~10 k points after the 0.1 m filter + 350 patches per frame
        fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(6)
    else:                                                                     # C4-shaped: 200 000 post-filter points + 4 000 patches per frame (bench.py's c5 "c4" frames, two of them)
        fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(2, n_raw=620000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_points=1600000, n_patches=4000,
                                                              max_points=200000)
        assert all(len(f["xyz"]) == 200000 and len(f["vs"].pos) == 4000 for f in seq)
    cfg = cfgs.lidar_cfg(_Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    want, trails = _oracle_records(orc, frames_mod, fmap, lio_cfg, extR, extT, seq, 4)
    # the chain is well posed: the LiDAR posterior is near the true pose and the visual update (started there) stays there
    for fr, lref, vref in trails:
        a = orc.state_arrays(vref["state"])
        assert np.linalg.norm(a["t"] - fr["t_true"]) < 0.02 and np.linalg.norm(a["R"] - fr["R_true"]) < 5e-3
        assert lref["n_iters"] >= 2 and len(vref["trace"]) >= 4
    ctxs = [livo2.Context(0) for _ in range(3)]
    try:
        for c in ctxs:
            c.upload_map(fmap)
        one, ev1 = frames_mod.run_frames_sharded(ctxs[0], livo2.State, seq, cfg, vcfg, 0, 1)
        three, ev3 = frames_mod.run_frames_sharded(ctxs, livo2.State, seq, cfg, vcfg, 0, 1)
        # sharded over two "ranks" (what --gpus 2 does, here on one device): rank r takes frames r, r + 2, ...
        r0, _ = frames_mod.run_frames_sharded(ctxs[:2], livo2.State, seq, cfg, vcfg, 0, 2)
        r1, _ = frames_mod.run_frames_sharded(ctxs[2], livo2.State, seq, cfg, vcfg, 1, 2)
        assert sum(c.counter("visual_persistent_launches") for c in ctxs) > 0          # the resident-grid form ran next to the other contexts' kernels
        assert sum(c.counter("visual_persistent_timeouts") for c in ctxs) == 0          # ... and none of those grids gave up (it would be re-run per step, with the same result)
        # the whole frame as ONE call (livo2_frame_update_async / _fetch), two frames in flight on one context, the LiDAR posterior handed to the visual update on the device
        piped, evp = frames_mod.run_frames_pipelined(ctxs[1], livo2.State, seq, cfg, vcfg)
        twice, _ = frames_mod.run_frames_pipelined(ctxs[1], livo2.State, seq + seq, cfg, vcfg)
    finally:
        for c in ctxs:
            c.close()
    _compare(one, want, frames_mod)
    _compare(three, want, frames_mod)
    assert np.array_equal(one, three) and ev1 == ev3                                  # a context's results depend on its inputs only
    sharded = np.zeros_like(one); sharded[0::2] = r0; sharded[1::2] = r1
    assert np.array_equal(sharded, one)
    assert np.array_equal(piped, one) and evp == ev1                                   # same kernels, same bits: only host work was removed
    assert np.array_equal(twice[: len(seq)], one) and np.array_equal(twice[len(seq):], one)


def test_frame_update_argument_errors_and_empty_submap(livo2, orc):
    frames_mod = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(2)
    cfg = cfgs.lidar_cfg(_Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    c = livo2.Context(0)
    try:
        fr = seq[0]; vs = fr["vs"]
        prior = livo2.State.from_pose(fr["R_prior"], fr["t_prior"], fr["P"], inv_expo=vs.tau_prior)
        args = (fr["xyz"], prior, cfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg)
        with pytest.raises(livo2.Livo2Error) as e:
            c.frame_update(*args)                                                   # no map
        assert e.value.code == livo2.abi.ERR_NO_MAP
        c.upload_map(fmap)
        with pytest.raises(livo2.Livo2Error) as e:
            c.frame_update_fetch()                                                  # nothing in flight
        assert e.value.code == livo2.abi.ERR_INVALID
        c.frame_update_async(*args); c.frame_update_async(*args)
        with pytest.raises(livo2.Livo2Error) as e:
            c.frame_update_async(*args)                                             # a third frame without a fetch
        assert e.value.code == livo2.abi.ERR_INVALID
        a, b = c.frame_update_fetch(), c.frame_update_fetch()
        assert bytes(a[0].state) == bytes(b[0].state) and bytes(a[1].state) == bytes(b[1].state)
        # the four separate calls give the same bits
        rec, _ = frames_mod.run_frame(c, livo2.State, fr, cfg, vcfg)
        assert np.array_equal(rec, frames_mod.pack_result(*a))
        # an empty sub-map: computeJacobianAndUpdateEKF returns at once (vio.cpp:786), the visual result is the LiDAR posterior
        l0, v0 = c.frame_update(fr["xyz"], prior, cfg, vs.img, vs.pos[:0], vs.warp_patch[:0], vs.search_levels[:0], vs.inv_expo_list[:0], vcfg)
        assert v0.n_steps == 0 and bytes(v0.state) == bytes(l0.state) and bytes(l0.state) == bytes(a[0].state)
    finally:
        c.close()


def test_a_frame_behind_a_resident_grid_that_gave_up_is_rerun_per_step(livo2, orc):
    """round 5: the exchange buffers of the resident visual grid carry no launch tags any more; a grid that gives up half-way leaves a mark on the device, and a launch
    that is ALREADY enqueued behind it gives up at once instead of reading words of unfinished steps.  Two frames in flight, the first one loses a block (test hook):
    its fetch reports the error (its inputs are gone), the second frame is re-run per step by its fetch and equals the undisturbed result; afterwards the resident
    grid is back."""
    frames_mod = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(2)
    cfg = cfgs.lidar_cfg(_Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    c = livo2.Context(0)
    try:
        c.upload_map(fmap)
        args = []
        for fr in seq:
            vs = fr["vs"]
            prior = livo2.State.from_pose(fr["R_prior"], fr["t_prior"], fr["P"], inv_expo=vs.tau_prior)
            args.append((fr["xyz"], prior, cfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg))
        want = [c.frame_update(*a) for a in args]
        t0, l0 = c.counter("visual_persistent_timeouts"), c.counter("visual_persistent_launches")
        c.set_option("visual_persistent_debug_timeout", 1)
        c.frame_update_async(*args[0])
        c.set_option("visual_persistent_debug_timeout", 0)
        c.frame_update_async(*args[1])
        with pytest.raises(livo2.Livo2Error):
            c.frame_update_fetch()
        got = c.frame_update_fetch()
        assert bytes(got[0].state) == bytes(want[1][0].state) and bytes(got[1].state) == bytes(want[1][1].state) and got[1].n_steps == want[1][1].n_steps
        assert c.counter("visual_persistent_timeouts") == t0 + 2 and c.counter("visual_persistent_launches") == l0 + 2
        for _ in range(20):                                                          # (the back-off after two time-outs: 8 + 16 updates on the per-step path)
            again = c.frame_update(*args[0])
            assert bytes(again[1].state) == bytes(want[0][1].state)
        again = [c.frame_update(*a) for a in args for _ in range(6)]
        assert c.counter("visual_persistent_launches") > l0 + 2 and c.counter("visual_persistent_timeouts") == t0 + 2
    finally:
        c.close()
