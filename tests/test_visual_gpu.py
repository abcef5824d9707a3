"""GPU parity: visual photometric pass and the full computeJacobianAndUpdateEKF loop vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vs_small():
    return synth.visual_scenario(seed=3, n_patches=300)


@pytest.mark.parametrize("level", [3, 0])
@pytest.mark.parametrize("exposure", [True, False])
def test_iterate_matches_oracle(ctx, livo2, orc, vs_small, level, exposure):
    vs = vs_small
    ocfg = orc.visual_cfg(vs, exposure=exposure)
    pcfg = H.visual_cfg_product(vs, exposure=exposure)
    ocur, _ = H.states(vs, orc.StatePOD)
    pcur, _ = H.states(vs, livo2.State)
    ref = orc.visual_iterate(ocfg, vs, level, ocur)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    sums, errors, z, Hs = ctx.visual_iterate(level, pcur, pcfg, rows=True)
    assert sums.n_meas == ref["n_meas"] == 64 * len(vs.pos)
    # residuals: same float32 bilinear samples, same double expression -> identical
    assert np.array_equal(z, ref["z"]), f"max |dz| = {np.abs(z - ref['z']).max()}"
    assert H.relerr(Hs, ref["H"]) < 1e-14
    # float patch_error += res*res in pixel order, float error += patch_error in patch order (vio.cpp:1563,1624,1634): the same chains on the device
    assert np.array_equal(errors, ref["errors"])
    assert sums.error == ref["error"]
    # moment-factorised H^T H / H^T z vs the dense row-by-row sums of the oracle
    assert H.relerr(np.array(sums.HtH).reshape(7, 7), ref["HtH"]) < 1e-11
    assert H.relerr(np.array(sums.Htz), ref["Htz"]) < 1e-10
    # without rows requested the production kernel variant must give the same sums
    sums2, errors2, _, _ = ctx.visual_iterate(level, pcur, pcfg, rows=False)
    assert np.array_equal(np.array(sums2.HtH), np.array(sums.HtH)) and np.array_equal(errors2, errors)


def test_full_update_matches_oracle(ctx, livo2, orc, vs_small):
    vs = vs_small
    ocfg = orc.visual_cfg(vs)
    pcfg = H.visual_cfg_product(vs)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert res.n_steps == len(ref["trace"]), (res.n_steps, len(ref["trace"]))
    for k in range(res.n_steps):
        a, b = res.steps[k], ref["trace"][k]
        assert (a.level, a.iteration, a.accepted, a.n_meas) == (b.level, b.iteration, b.accepted, b.n_meas), k
        assert a.error == b.error
        if a.accepted:
            assert H.relerr(np.array(a.HtH), np.array(b.HtH)) < 1e-8
            assert H.relerr(np.array(a.solution), np.array(b.solution)) < 1e-6
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8 and d["inv_expo"] < 1e-9, d
    assert H.relerr(np.array(res.G).reshape(19, 19), ref["G"]) < 1e-7
    assert H.relerr(np.array(res.Rcw).reshape(3, 3), ref["Rcw"]) < 1e-12 and H.relerr(np.array(res.Pcw), ref["Pcw"]) < 1e-9
    assert np.array_equal(errors, ref["errors"])


@pytest.mark.parametrize("threads,M", [(4, 300), (4, 4000), (3, 1001), (7, 5), (1, 2000)])
def test_frame_error_follows_the_openmp_static_partition(ctx, livo2, orc, threads, M):
    """error <= last_error compares FLOAT sums whose order is set by MP_PROC_NUM threads' static blocks (vio.cpp:1554): same bits as the oracle for every partition,
    including more threads than patches and M % threads != 0; whole update: identical accept / revert sequence and errors."""
    vs = synth.visual_scenario(seed=40 + threads, n_patches=M)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    pcfg = H.visual_cfg_product(vs, mp_proc_num=threads)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    for level in (3, 0):
        ref = orc.visual_iterate(orc.visual_cfg(vs, num_threads=threads), vs, level, ocur)
        sums, errors, _, _ = ctx.visual_iterate(level, pcur, pcfg)
        assert np.array_equal(errors, ref["errors"]) and sums.error == ref["error"] and sums.n_meas == ref["n_meas"]
    ref = orc.visual_update(orc.visual_cfg(vs, num_threads=threads), vs, ocur, oprop)
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert [(res.steps[k].level, res.steps[k].iteration, res.steps[k].accepted, res.steps[k].error) for k in range(res.n_steps)] == \
           [(t.level, t.iteration, t.accepted, t.error) for t in ref["trace"]]
    assert np.array_equal(errors, ref["errors"])
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d


def test_radtan_camera_of_the_avia_config(ctx, livo2, orc):
    """cam->world2cam of config/camera_pinhole.yaml (cam_d0..d3 non-zero; vio.cpp:1574): the distorted projection moves every patch anchor, z stays bit-identical."""
    d = synth.AVIA_RADTAN
    vs = synth.visual_scenario(seed=9, n_patches=500, distortion=d)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ocfg, pcfg = orc.visual_cfg(vs, distortion=d), H.visual_cfg_product(vs, distortion=d)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    for level in (2, 0):
        ref = orc.visual_iterate(ocfg, vs, level, ocur)
        plain = orc.visual_iterate(orc.visual_cfg(vs), vs, level, ocur)
        assert not np.array_equal(ref["z"], plain["z"])                       # the distortion terms matter on this scene
        sums, errors, z, Hs = ctx.visual_iterate(level, pcur, pcfg, rows=True)
        assert np.array_equal(z, ref["z"]) and np.array_equal(errors, ref["errors"]) and sums.error == ref["error"]
        assert H.relerr(Hs, ref["H"]) < 1e-14 and H.relerr(np.array(sums.HtH).reshape(7, 7), ref["HtH"]) < 1e-11
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert [(res.steps[k].level, res.steps[k].iteration, res.steps[k].accepted) for k in range(res.n_steps)] == [(t.level, t.iteration, t.accepted) for t in ref["trace"]]
    dd = H.state_diff(res.state, ref["state"])
    assert dd["R"] < 1e-9 and dd["t"] < 1e-9 and dd["P"] < 1e-8, dd
    # the true pose is the optimum of this scene: the update moves the prior towards it
    assert np.linalg.norm(np.array(res.state.pos) - vs.t_true) < np.linalg.norm(vs.t_prior - vs.t_true)


def test_equidistant_camera_of_the_hilti22_config(ctx, livo2, orc):
    """cam_model: EquidistantCamera (config/camera_fisheye_HILTI22.yaml, k1..k4; livo2_cam.distortion = 2) through world2cam of the forward update (vio.cpp:1574).
    The model contains atan(): device and host libm may differ in the last bit of the projected pixel, which a float32 bilinear weight feels in rare pixels —
    residuals are compared to 1e-3 grey levels with >= 99.9 % of them bit-identical, every decision identical."""
    k = synth.HILTI_EQUIDISTANT
    vs = synth.visual_scenario(seed=9, n_patches=500)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ocfg, pcfg = orc.visual_cfg(vs, equidistant=k, num_threads=4), H.visual_cfg_product(vs, equidistant=k, mp_proc_num=4)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    for level in (2, 0):
        ref = orc.visual_iterate(ocfg, vs, level, ocur)
        plain = orc.visual_iterate(orc.visual_cfg(vs), vs, level, ocur)
        assert not np.array_equal(ref["z"], plain["z"])
        sums, errors, z, Hs = ctx.visual_iterate(level, pcur, pcfg, rows=True)
        assert np.abs(z - ref["z"]).max() < 1e-3 and (z == ref["z"]).mean() > 0.999
        assert np.allclose(errors, ref["errors"], rtol=1e-4)
        assert H.relerr(Hs, ref["H"]) < 1e-6 and H.relerr(np.array(sums.HtH).reshape(7, 7), ref["HtH"]) < 1e-6
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    for persistent in (1, 0):
        ctx.set_option("visual_persistent", persistent)
        res, errors = ctx.visual_update(pcur, pprop, pcfg)
        assert [(res.steps[j].level, res.steps[j].iteration, res.steps[j].accepted) for j in range(res.n_steps)] == [(t.level, t.iteration, t.accepted) for t in ref["trace"]]
        dd = H.state_diff(res.state, ref["state"])
        assert dd["R"] < 1e-7 and dd["t"] < 1e-7 and dd["P"] < 1e-7, dd
    ctx.set_option("visual_persistent", 1)
    bad = H.visual_cfg_product(vs); bad.cam.distortion = 3
    with pytest.raises(Exception):
        ctx.visual_update(pcur, pprop, bad)                        # unknown camera model: LIVO2_ERR_INVALID


def test_batched_frames_equal_single_updates(ctx, livo2, orc):
    """livo2_visual_batch_*: B independent updates in lockstep grids produce the bits of B separate livo2_visual_update calls (ragged sizes, an empty frame)."""
    sizes = [700, 0, 64, 1500, 9]
    frames = [synth.visual_scenario(seed=60 + k, n_patches=max(m, 1)) for k, m in enumerate(sizes)]
    cfg = H.visual_cfg_product(frames[0], mp_proc_num=4)
    st = [H.states(vs, livo2.State)[0] for vs in frames]
    singles = []
    for vs, m, s in zip(frames, sizes, st):
        ctx.set_frame(vs.img, vs.pos[:m], vs.warp_patch[:m], vs.search_levels[:m], vs.inv_expo_list[:m])
        singles.append(ctx.visual_update(s, s, cfg))
    ctx.visual_batch_set_frames([(vs.img, vs.pos[:m], vs.warp_patch[:m], vs.search_levels[:m], vs.inv_expo_list[:m]) for vs, m in zip(frames, sizes)])
    res = ctx.visual_batch_update(st, st, cfg)
    for k, (r, (single, _)) in enumerate(zip(res, singles)):
        assert r.n_steps == single.n_steps, k
        assert bytes(r.state) == bytes(single.state) and bytes(r.Rcw) == bytes(single.Rcw) and bytes(r.Pcw) == bytes(single.Pcw), k
        assert sizes[k] == 0 or bytes(r.G) == bytes(single.G), k       # (total_points == 0: no update, G is whatever the previous frame left)
        for j in range(r.n_steps):                               # (entries past n_steps are whatever an earlier update left there)
            assert bytes(r.steps[j]) == bytes(single.steps[j]), (k, j)
    # against the oracle, frame by frame
    for k in (0, 3):
        ocur, oprop = H.states(frames[k], orc.StatePOD)
        ref = orc.visual_update(orc.visual_cfg(frames[k], num_threads=4), frames[k], ocur, oprop)
        assert [(res[k].steps[j].level, res[k].steps[j].accepted, res[k].steps[j].error) for j in range(res[k].n_steps)] == [(t.level, t.accepted, t.error) for t in ref["trace"]]


@pytest.mark.parametrize("M,threads,kw", [(3000, 4, {}), (4000, 4, {}), (300, 1, {}), (33, 4, {}), (1, 1, {}), (9000, 4, {}), (2000, 3, dict(exposure=False)),
                                          (1000, 4, dict(distortion=synth.AVIA_RADTAN)), (500, 4, dict(max_iterations=1)), (700, 2, dict(rot_sigma_deg=0.25))])
def test_persistent_update_gives_the_same_bits_as_the_per_step_launches(livo2, ctx, M, threads, kw):
    """k_visual_update_persistent (default path): the whole computeJacobianAndUpdateEKF as one resident grid, one grid barrier per (level, iteration), the reduction /
    error chain / accept-revert / solve run redundantly by every block.  Same residual body, same order of additions, same solve => the same bits as the
    launch-per-step sequence: state, covariance, G, Rcw / Pcw, every recorded step, errors[].  Repeated: a stale read between workgroups would make it flaky.
    M = 9000 > 32 x 256: blocks own several patch groups (rows are summed in another order there: tolerance instead of bits)."""
    skw = {k: v for k, v in kw.items() if k == "rot_sigma_deg"}
    ckw = {k: v for k, v in kw.items() if k != "rot_sigma_deg"}
    vs = synth.visual_scenario(seed=90 + threads + M % 7, n_patches=M, **skw)
    cfg = H.visual_cfg_product(vs, mp_proc_num=threads, **ckw)
    cur, prop = H.states(vs, livo2.State)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ctx.set_option("visual_persistent", 0)
    try:
        ref, ref_err = ctx.visual_update(cur, prop, cfg)
    finally:
        ctx.set_option("visual_persistent", 1)
    n0, t0 = ctx.counter("visual_persistent_launches"), ctx.counter("visual_persistent_timeouts")
    for _ in range(10):
        res, err = ctx.visual_update(cur, prop, cfg)
        assert res.n_steps == ref.n_steps
        assert [(res.steps[j].level, res.steps[j].iteration, res.steps[j].accepted, res.steps[j].error) for j in range(res.n_steps)] == \
               [(ref.steps[j].level, ref.steps[j].iteration, ref.steps[j].accepted, ref.steps[j].error) for j in range(ref.n_steps)]
        assert np.array_equal(err, ref_err)
        if M <= 32 * 256:
            assert bytes(res.state) == bytes(ref.state) and bytes(res.G) == bytes(ref.G) and bytes(res.Rcw) == bytes(ref.Rcw) and bytes(res.Pcw) == bytes(ref.Pcw)
            assert all(bytes(res.steps[j]) == bytes(ref.steps[j]) for j in range(ref.n_steps))
        else:
            d = H.state_diff(res.state, ref.state)
            assert d["R"] < 1e-12 and d["t"] < 1e-12 and d["P"] < 1e-12, d
    # (a resident grid that gives up is re-run per step by the fetch with the RIGHT result: only the counters tell that the resident path is what ran — round 5)
    assert ctx.counter("visual_persistent_launches") == n0 + 10 and ctx.counter("visual_persistent_fallbacks") == 0 and ctx.counter("visual_persistent_timeouts") == t0


def test_persistent_updates_of_changing_shape_share_the_exchange_buffers(livo2):
    """round 5: the exchange buffers of the resident grid hold plain values, 'empty' is an all-ones word, four buffers rotate with a step number that runs on from
    launch to launch and every launch empties two steps ahead — also the slots that an OLDER, larger sub-map wrote.  Updates of very different size and length
    (one level x one iteration = a single step; many steps) alternate on one fresh context: every one equals the launch-per-step sequence bit for bit."""
    c = livo2.Context(0)
    try:
        cases = []
        for k, (M, L, kw) in enumerate([(4000, 4, {}), (40, 4, {}), (1700, 4, dict(max_iterations=1)), (4000, 4, {}), (300, 1, dict(max_iterations=1)), (3100, 3, {}),
                                        (17, 4, {}), (2500, 2, dict(max_iterations=2)), (4000, 1, dict(max_iterations=1)), (4000, 4, {})]):
            vs = synth.visual_scenario(seed=300 + k, n_patches=M, L=L)
            cfg = H.visual_cfg_product(vs, mp_proc_num=4, **kw)
            cur, prop = H.states(vs, livo2.State)
            cases.append((vs, cfg, cur, prop))
        refs = []
        c.set_option("visual_persistent", 0)
        for vs, cfg, cur, prop in cases:
            c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
            refs.append(c.visual_update(cur, prop, cfg))
        c.set_option("visual_persistent", 1)
        for rep in range(3):
            for (vs, cfg, cur, prop), (ref, ref_err) in zip(cases, refs):
                c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
                res, err = c.visual_update(cur, prop, cfg)
                assert res.n_steps == ref.n_steps and bytes(res.state) == bytes(ref.state) and bytes(res.G) == bytes(ref.G) and np.array_equal(err, ref_err)
                assert all(bytes(res.steps[j]) == bytes(ref.steps[j]) for j in range(ref.n_steps))
        assert c.counter("visual_persistent_launches") == 3 * len(cases) and c.counter("visual_persistent_timeouts") == 0
    finally:
        c.close()


def test_persistent_update_matches_oracle_with_reverts(livo2, ctx, orc):
    """the persistent path against the oracle on scenes where levels end by a rejected step (state restored from old_state, vio.cpp:1677-1681)"""
    seen = False
    for seed in range(20, 26):
        vs = synth.visual_scenario(seed=seed, n_patches=200, rot_sigma_deg=0.25)
        ocur, oprop = H.states(vs, orc.StatePOD)
        ref = orc.visual_update(orc.visual_cfg(vs, num_threads=4), vs, ocur, oprop)
        cur, prop = H.states(vs, livo2.State)
        ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        res, err = ctx.visual_update(cur, prop, H.visual_cfg_product(vs, mp_proc_num=4))
        assert [(res.steps[j].level, res.steps[j].iteration, res.steps[j].accepted, res.steps[j].error) for j in range(res.n_steps)] == \
               [(t.level, t.iteration, t.accepted, t.error) for t in ref["trace"]]
        seen = seen or any(not t.accepted for t in ref["trace"])
        d = H.state_diff(res.state, ref["state"])
        assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d
        assert np.array_equal(err, ref["errors"])
    assert seen


def test_concurrent_persistent_contexts(livo2):
    """four contexts, each on its own stream and host thread, run persistent updates at the same time: admission keeps the resident grids within the device
    (the ones that do not fit take the per-step path), nothing deadlocks, every result equals the single-context one"""
    import threading
    vs = synth.visual_scenario(seed=61, n_patches=4000)
    cfg = H.visual_cfg_product(vs, mp_proc_num=4)
    cur, prop = H.states(vs, livo2.State)
    ctxs = [livo2.Context(0) for _ in range(4)]
    for c in ctxs:
        c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ref, ref_err = ctxs[0].visual_update(cur, prop, cfg)
    bad = []
    def work(c):
        try:
            for _ in range(25):
                res, err = c.visual_update(cur, prop, cfg)
                if bytes(res.state) != bytes(ref.state) or not np.array_equal(err, ref_err):
                    bad.append(1)
        except BaseException as exc:                     # (an exception in a thread would otherwise only be a pytest warning)
            bad.append(repr(exc))
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    used = sum(c.counter("visual_persistent_launches") for c in ctxs)
    for c in ctxs:
        c.close()
    assert not bad and used >= 26


def test_a_resident_grid_that_loses_a_block_is_rerun_per_step_with_the_same_result(livo2, ctx):
    """advisor (round 3): admission of the resident grid is per process; if anything keeps a block off the device every block gives up — and used to overwrite
    ctl->cur / cov / G with garbage, the fetch returning a hard error.  Now a grid that timed out commits nothing and livo2_visual_update_fetch re-runs the update
    as the launch-per-step sequence from the inputs kept at enqueue.  The test hook drops the last block of the grid and shortens the wait to 2 ms."""
    vs = synth.visual_scenario(seed=77, n_patches=1200)
    cfg = H.visual_cfg_product(vs, mp_proc_num=4)
    cur, prop = H.states(vs, livo2.State)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ref, ref_err = ctx.visual_update(cur, prop, cfg)
    t0, f0 = ctx.counter("visual_persistent_timeouts"), ctx.counter("visual_persistent_launches")
    ctx.set_option("visual_persistent_debug_timeout", 1)
    try:
        for _ in range(3):
            res, err = ctx.visual_update(cur, prop, cfg)
            assert bytes(res.state) == bytes(ref.state) and bytes(res.G) == bytes(ref.G) and res.n_steps == ref.n_steps and np.array_equal(err, ref_err)
    finally:
        ctx.set_option("visual_persistent_debug_timeout", 0)
    assert ctx.counter("visual_persistent_timeouts") == t0 + 3 and ctx.counter("visual_persistent_launches") == f0 + 3
    res, err = ctx.visual_update(cur, prop, cfg)                    # and the resident grid works again afterwards
    assert bytes(res.state) == bytes(ref.state) and ctx.counter("visual_persistent_timeouts") == t0 + 3


@pytest.mark.parametrize("M,threads", [(4000, 4), (2000, 1), (9000, 4), (9000, 1), (6001, 6), (8000, 10)])
def test_frame_error_on_groups_of_lanes_gives_the_bits_of_the_one_lane_chains(livo2, ctx, M, threads):
    """round 6: the frame error's float chains (vio.cpp:1554, 1634) run on groups of 32 lanes (float_chain.hpp) when every OpenMP thread's block holds >= 768 patch
    errors (>= 768 since the break-even was measured); option "visual_error_waves" = 0 keeps one lane per thread.  Same records, same errors[], on the resident grid and on the launch-per-step sequence
    (M = 9000: the blocks reach beyond the staging area; one thread: its chain on all 64 lanes of one wave, several passes; 6 threads: three chain waves; 10 threads: five)."""
    vs = synth.visual_scenario(seed=400 + threads, n_patches=M)
    cfg = H.visual_cfg_product(vs, mp_proc_num=threads)
    cur, prop = H.states(vs, livo2.State)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    out = {}
    try:
        for persistent in (0, 1):
            for waves in (0, 1):
                ctx.set_option("visual_persistent", persistent); ctx.set_option("visual_error_waves", waves)
                out[persistent, waves] = ctx.visual_update(cur, prop, cfg)
    finally:
        ctx.set_option("visual_persistent", 1); ctx.set_option("visual_error_waves", 1)
    for persistent in (0, 1):
        (a, ea), (b, eb) = out[persistent, 0], out[persistent, 1]
        assert a.n_steps == b.n_steps and np.array_equal(ea, eb)
        assert bytes(a.state) == bytes(b.state) and bytes(a.G) == bytes(b.G)
        assert all(bytes(a.steps[j]) == bytes(b.steps[j]) for j in range(a.n_steps)), persistent
