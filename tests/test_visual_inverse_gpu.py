"""GPU parity of the inverse-compositional visual update (vio/inverse_composition_en; reference src/vio.cpp:1327-1518) vs the oracle."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vs_inv():
    return synth.visual_inverse_scenario(seed=5, n_patches=300)


def _upload(ctx, vs):
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    ctx.set_reference(vs.ref_imgs, vs.ref_img_idx, vs.ref_px, vs.ref_f, vs.ref_R, vs.ref_pos)


@pytest.mark.parametrize("level", [3, 1, 0])
def test_inverse_iterate_matches_oracle(ctx, livo2, orc, vs_inv, level):
    vs = vs_inv
    ocfg = orc.visual_cfg(vs, inverse=True)
    pcfg = H.visual_cfg_product(vs, inverse=True)
    ocur, _ = H.states(vs, orc.StatePOD)
    pcur, _ = H.states(vs, livo2.State)
    ref = orc.visual_iterate_inverse(ocfg, vs, level, ocur)
    _upload(ctx, vs)
    sums, errors, z, Hs = ctx.visual_iterate(level, pcur, pcfg, rows=True)
    assert sums.n_meas == ref["n_meas"] == 64 * len(vs.pos)
    assert np.array_equal(z, ref["z"]), np.abs(z - ref["z"]).max()          # all-float residual expression: bit-identical
    assert H.relerr(Hs[:, :6], ref["H"]) < 1e-13 and not np.any(Hs[:, 6])
    assert np.allclose(errors, ref["errors"], rtol=2e-6)
    assert abs(sums.error - ref["error"]) <= 2e-6 * abs(ref["error"])
    HtH = np.array(sums.HtH).reshape(7, 7)
    assert H.relerr(HtH[:6, :6], ref["HtH"]) < 1e-11 and not np.any(HtH[6]) and not np.any(HtH[:, 6])
    assert H.relerr(np.array(sums.Htz)[:6], ref["Htz"]) < 1e-10
    sums2, errors2, _, _ = ctx.visual_iterate(level, pcur, pcfg, rows=False)
    assert np.array_equal(np.array(sums2.HtH), np.array(sums.HtH)) and np.array_equal(errors2, errors)


def test_inverse_iterate_with_radtan_camera(ctx, livo2, orc, vs_inv):
    """cam->world2cam with the avia camera's distortion coefficients inside updateStateInverse (vio.cpp:1447): the kernel's radtan branch."""
    vs = vs_inv
    ocfg = orc.visual_cfg(vs, inverse=True, distortion=synth.AVIA_RADTAN)
    pcfg = H.visual_cfg_product(vs, inverse=True, distortion=synth.AVIA_RADTAN)
    ocur, _ = H.states(vs, orc.StatePOD)
    pcur, _ = H.states(vs, livo2.State)
    pin = orc.visual_iterate_inverse(orc.visual_cfg(vs, inverse=True), vs, 0, ocur)
    ref = orc.visual_iterate_inverse(ocfg, vs, 0, ocur)
    assert not np.array_equal(ref["z"], pin["z"])                            # the distortion moves the sampling positions
    _upload(ctx, vs)
    sums, errors, z, Hs = ctx.visual_iterate(0, pcur, pcfg, rows=True)
    assert sums.n_meas == ref["n_meas"]
    assert np.array_equal(z, ref["z"]), np.abs(z - ref["z"]).max()
    assert H.relerr(Hs[:, :6], ref["H"]) < 1e-13
    assert np.allclose(errors, ref["errors"], rtol=2e-6)


def test_inverse_full_update_matches_oracle(ctx, livo2, orc, vs_inv):
    vs = vs_inv
    ocfg = orc.visual_cfg(vs, inverse=True)
    pcfg = H.visual_cfg_product(vs, inverse=True)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    _upload(ctx, vs)
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert res.n_steps == len(ref["trace"])
    for k in range(res.n_steps):
        a, b = res.steps[k], ref["trace"][k]
        assert (a.level, a.iteration, a.accepted, a.n_meas) == (b.level, b.iteration, b.accepted, b.n_meas), k
        assert abs(a.error - b.error) <= 4e-6 * abs(b.error)
        if a.accepted:
            assert H.relerr(np.array(a.solution), np.array(b.solution)) < 1e-6
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d
    assert H.relerr(np.array(res.G).reshape(19, 19), ref["G"]) < 1e-7
    assert np.allclose(errors, ref["errors"], rtol=1e-5)


def test_inverse_needs_reference(ctx, livo2, vs_inv):
    vs = vs_inv
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)       # invalidates earlier references
    pcur, pprop = H.states(vs, livo2.State)
    with pytest.raises(livo2.Livo2Error) as e:
        ctx.visual_update(pcur, pprop, H.visual_cfg_product(vs, inverse=True))
    assert e.value.code == livo2.abi.ERR_INVALID


def test_inverse_update_with_equidistant_camera(ctx, livo2, orc, vs_inv):
    """vk::EquidistantCamera (config/camera_fisheye_HILTI22.yaml) inside updateStateInverse's cam->world2cam (vio.cpp:1447)"""
    vs = vs_inv
    ocfg = orc.visual_cfg(vs, inverse=True, equidistant=synth.HILTI_EQUIDISTANT)
    pcfg = H.visual_cfg_product(vs, inverse=True, equidistant=synth.HILTI_EQUIDISTANT)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_iterate_inverse(ocfg, vs, 0, ocur)
    _upload(ctx, vs)
    sums, errors, z, Hs = ctx.visual_iterate(0, pcur, pcfg, rows=True)
    assert sums.n_meas == ref["n_meas"]
    assert np.abs(z - ref["z"]).max() < 1e-3 and (z == ref["z"]).mean() > 0.999
    full = orc.visual_update(ocfg, vs, ocur, oprop)
    res, _ = ctx.visual_update(pcur, pprop, pcfg)
    assert [(res.steps[j].level, res.steps[j].accepted) for j in range(res.n_steps)] == [(t.level, t.accepted) for t in full["trace"]]
    d = H.state_diff(res.state, full["state"])
    assert d["R"] < 1e-7 and d["t"] < 1e-7, d


def test_inverse_in_the_frame_call_equals_the_separate_calls(ctx, livo2, orc, vs_inv):
    """livo2_frame_in.reference (round 6): a whole LIO + VIO frame with vio/inverse_composition_en — scan, StateEstimation, image + sub-map + reference patches,
    updateStateInverse from the LiDAR posterior — in one call gives the bits of the separate calls, and a forward-compositional frame afterwards does not see the
    reference patches of this one."""
    vs = vs_inv
    sc = synth.lidar_scenario(seed=61, n_points=4000, downsample=0.1)
    pcfg = H.lidar_cfg_product(sc)
    vcfg = H.visual_cfg_product(vs, inverse=True)
    prior = orc.make_state(sc.R_prior, sc.t_prior, sc.P, cls=livo2.State)
    ctx.upload_map(sc.fmap)
    # separate calls: LIO, then VIO on the shared state (iterate = prior = the LiDAR posterior)
    ctx.set_scan(sc.xyz, pcfg)
    lres, _ = ctx.lidar_update(prior, prior, pcfg)
    _upload(ctx, vs)
    vres, _ = ctx.visual_update(lres.state, lres.state, vcfg)
    ref = (vs.ref_imgs, vs.ref_img_idx, vs.ref_px, vs.ref_f, vs.ref_R, vs.ref_pos)
    fl, fv = ctx.frame_update(sc.xyz, prior, pcfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg, reference=ref)
    assert bytes(fl.state) == bytes(lres.state) and fl.n_iters == lres.n_iters
    assert fv.n_steps == vres.n_steps >= 4 and bytes(fv.state) == bytes(vres.state)
    assert np.linalg.norm(np.array(fv.state.pos) - np.array(lres.state.pos)) > 0          # the visual update moved the state
    with pytest.raises(livo2.Livo2Error):                                                  # the inverse form without reference patches is refused
        ctx.frame_update(sc.xyz, prior, pcfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, vcfg)
    # a forward-compositional frame on the same context: same as its own separate calls
    fcfg = H.visual_cfg_product(vs)
    ctx.set_scan(sc.xyz, pcfg)
    lres2, _ = ctx.lidar_update(prior, prior, pcfg)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    vres2, _ = ctx.visual_update(lres2.state, lres2.state, fcfg)
    fl2, fv2 = ctx.frame_update(sc.xyz, prior, pcfg, vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list, fcfg)
    assert bytes(fv2.state) == bytes(vres2.state) and bytes(fv2.state) != bytes(fv.state)


@pytest.mark.parametrize("M,seed", [(300, 5), (1000, 6), (37, 7), (4000, 8)])
def test_inverse_on_the_resident_grid_gives_the_bits_of_the_per_step_launches(ctx, livo2, M, seed):
    """round 6: updateStateInverse + precomputeReferencePatches run on the resident grid too (k_visual_update_persistent<true>: the reference patches of a block's own
    patch groups at the head of every level, then the INV form of the shared wave body).  Same body, same rows, same order of additions as the launch-per-step
    sequence (option visual_persistent_inverse = 0) => the same bits: state, covariance, G, every recorded step, errors[].  Counters: the resident grid is what ran."""
    vs = synth.visual_inverse_scenario(seed=seed, n_patches=M)
    cfg = H.visual_cfg_product(vs, inverse=True)
    cur, prop = H.states(vs, livo2.State)
    _upload(ctx, vs)
    ctx.set_option("visual_persistent_inverse", 0)
    try:
        n_before = ctx.counter("visual_persistent_launches")
        ref, ref_err = ctx.visual_update(cur, prop, cfg)
        assert ctx.counter("visual_persistent_launches") == n_before
    finally:
        ctx.set_option("visual_persistent_inverse", 1)
    n0, t0, f0 = ctx.counter("visual_persistent_launches"), ctx.counter("visual_persistent_timeouts"), ctx.counter("visual_persistent_fallbacks")
    for _ in range(5):
        _upload(ctx, vs)
        res, err = ctx.visual_update(cur, prop, cfg)
        assert res.n_steps == ref.n_steps > 0
        assert np.array_equal(err, ref_err)
        assert bytes(res.state) == bytes(ref.state) and bytes(res.G) == bytes(ref.G) and bytes(res.Rcw) == bytes(ref.Rcw) and bytes(res.Pcw) == bytes(ref.Pcw)
        assert all(bytes(res.steps[j]) == bytes(ref.steps[j]) for j in range(ref.n_steps))
    assert ctx.counter("visual_persistent_launches") == n0 + 5 and ctx.counter("visual_persistent_fallbacks") == f0 and ctx.counter("visual_persistent_timeouts") == t0


@pytest.mark.parametrize("M,seed", [(37, 11), (1500, 12)])
def test_inverse_full_update_matches_oracle_at_other_sizes(ctx, livo2, orc, M, seed):
    """the resident-grid form (default) against the oracle for a sub-map that does not fill a wave's four patch slots / one that spans ~94 blocks"""
    vs = synth.visual_inverse_scenario(seed=seed, n_patches=M)
    ocfg = orc.visual_cfg(vs, inverse=True)
    pcfg = H.visual_cfg_product(vs, inverse=True)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    _upload(ctx, vs)
    n0 = ctx.counter("visual_persistent_launches")
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert ctx.counter("visual_persistent_launches") == n0 + 1
    assert res.n_steps == len(ref["trace"])
    for k in range(res.n_steps):
        a, b = res.steps[k], ref["trace"][k]
        assert (a.level, a.iteration, a.accepted, a.n_meas) == (b.level, b.iteration, b.accepted, b.n_meas), k
        assert abs(a.error - b.error) <= 4e-6 * abs(b.error)
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d
    assert np.allclose(errors, ref["errors"], rtol=1e-5)


def test_batched_inverse_frames_equal_single_updates(ctx, livo2, orc):
    """livo2_visual_batch_* with inverse_composition_en (round 6; livo2_visual_batch_set_references): B independent updateStateInverse-based updates in lockstep grids
    produce the bits of B separate livo2_visual_update calls (ragged sizes, different numbers of reference images); one of them also against the oracle."""
    sizes = [400, 37, 900, 5]
    frames = [synth.visual_inverse_scenario(seed=70 + k, n_patches=m) for k, m in enumerate(sizes)]
    cfg = H.visual_cfg_product(frames[0], inverse=True)
    st = [H.states(vs, livo2.State)[0] for vs in frames]
    singles = []
    for vs, s in zip(frames, st):
        _upload(ctx, vs)
        singles.append(ctx.visual_update(s, s, cfg))
    ctx.visual_batch_set_frames([(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list) for vs in frames])
    with pytest.raises(livo2.Livo2Error):
        ctx.visual_batch_update(st, st, cfg)                       # no references yet
    ctx.visual_batch_set_references([(vs.ref_imgs, vs.ref_img_idx, vs.ref_px, vs.ref_f, vs.ref_R, vs.ref_pos) for vs in frames])
    for _ in range(2):
        res = ctx.visual_batch_update(st, st, cfg)
        for k, (r, (single, _)) in enumerate(zip(res, singles)):
            assert r.n_steps == single.n_steps > 0, k
            assert bytes(r.state) == bytes(single.state) and bytes(r.Rcw) == bytes(single.Rcw) and bytes(r.Pcw) == bytes(single.Pcw) and bytes(r.G) == bytes(single.G), k
            for j in range(r.n_steps):
                assert bytes(r.steps[j]) == bytes(single.steps[j]), (k, j)
    k = 2
    ocur, oprop = H.states(frames[k], orc.StatePOD)
    ref = orc.visual_update(orc.visual_cfg(frames[k], inverse=True), frames[k], ocur, oprop)
    assert [(res[k].steps[j].level, res[k].steps[j].accepted) for j in range(res[k].n_steps)] == [(t.level, t.accepted) for t in ref["trace"]]
    d = H.state_diff(res[k].state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d
    # the forward form on the same batch still works (the references are simply not read)
    fcfg = H.visual_cfg_product(frames[0])
    fres = ctx.visual_batch_update(st, st, fcfg)
    assert all(r.n_steps > 0 for r in fres)
