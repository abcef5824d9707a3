"""Pins the oracle's restatement of the retrieval tail (oracle/orc_warp.hpp, reference src/vio.cpp:203-350, 698-767) with independent numpy
evaluations: the plane-induced warp by explicit ray/plane intersection, the affine warp by re-projection at the reference depth, the current
patch by scenarios.synth.sample_patch (numpy getImagePatch), the photometric error and the NCC by their definitions."""
import numpy as np

from oracle import orc
from scenarios import synth


def _proj(cam, p):
    return np.array([cam["fx"] * p[0] / p[2] + cam["cx"], cam["fy"] * p[1] / p[2] + cam["cy"]])


def _ray(cam, px):
    f = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
    return f / np.linalg.norm(f)


def _check(rs, ref, idxs):
    cam = rs.cam
    for i in idxs:
        R_ref, t_ref = rs.ref_R[i].reshape(3, 3), rs.ref_t[i]
        R_cr = rs.R_cur @ R_ref.T
        t_cr = rs.t_cur - R_cr @ t_ref
        pf = R_ref @ rs.pos[i] + t_ref
        A = np.zeros((2, 2))
        if rs.cfg["normal_en"]:
            n = R_ref @ rs.normal[i]; n /= np.linalg.norm(n)
            hit = lambda f: f * (n @ pf) / (n @ f)                      # ray through the reference camera centre meets the plane n.X = n.pf
            p0 = _proj(cam, R_cr @ pf + t_cr)
            for k, d in enumerate(((4.0, 0.0), (0.0, 4.0))):
                A[:, k] = (_proj(cam, R_cr @ hit(_ray(cam, rs.ref_px[i] + d)) + t_cr) - p0) / 4.0
        else:
            ref_pos = -R_ref.T @ t_ref
            xyz = rs.ref_f[i] * np.linalg.norm(ref_pos - rs.pos[i])
            p0 = _proj(cam, R_cr @ xyz + t_cr)
            step = 4.0 * (1 << int(rs.ref_level[i]))
            for k, d in enumerate(((step, 0.0), (0.0, step))):
                f = _ray(cam, rs.ref_px[i] + d)
                A[:, k] = (_proj(cam, R_cr @ (f * xyz[2] / f[2]) + t_cr) - p0) / 4.0
        np.testing.assert_allclose(ref["A"][i].reshape(2, 2), A, rtol=1e-8, atol=1e-9)
        D, lvl = np.linalg.det(A), 0
        while D > 3.0 and lvl < 2:
            lvl, D = lvl + 1, D * 0.25
        assert ref["search_level"][i] == lvl
        # level-0 warped patch: bilinear samples of the reference image at A^-1 offsets
        Ainv = np.linalg.inv(A)
        img_ref = rs.ref_imgs[rs.ref_img_idx[i]].astype(np.float64)
        w0 = ref["patch_wrap"][i, 0].reshape(8, 8)
        for (y, x) in ((0, 0), (3, 5), (7, 7)):
            u, v = Ainv @ (np.array([x - 4.0, y - 4.0]) * (1 << lvl)) + rs.ref_px[i]
            if u < 0 or v < 0 or u >= cam["width"] - 1 or v >= cam["height"] - 1:
                assert w0[y, x] == 0
                continue
            xi, yi = int(np.floor(u)), int(np.floor(v))
            sx, sy = u - xi, v - yi
            val = (1 - sx) * (1 - sy) * img_ref[yi, xi] + (1 - sx) * sy * img_ref[yi + 1, xi] + sx * (1 - sy) * img_ref[yi, xi + 1] + sx * sy * img_ref[yi + 1, xi + 1]
            assert abs(w0[y, x] - val) < 2e-2 * (1 + abs(val)) * 1e-1 + 0.05, (i, y, x)    # float pixel coordinates: ~1e-4 px
        pc = _proj(cam, rs.R_cur @ rs.pos[i] + rs.t_cur)
        buf = synth.sample_patch(rs.img, pc, 1).astype(np.float64).ravel()
        wr = ref["patch_wrap"][i, 0].astype(np.float64)
        err = ((rs.ref_inv_expo[i] * wr - rs.inv_expo_cur * buf) ** 2).sum()
        assert abs(ref["error"][i] - err) <= 1e-5 * err + 1e-3
        a, b = wr - wr.mean(), buf - buf.mean()
        assert abs(ref["ncc"][i] - (a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum() + 1e-10)) < 1e-9
        ok = (not (rs.cfg["ncc_en"] and ref["ncc"][i] < rs.cfg["ncc_thre"])) and not (ref["error"][i] > rs.cfg["outlier_threshold"] * 64)
        assert ref["accepted"][i] == int(ok)


def test_homography_variant():
    rs = synth.retrieve_scenario(seed=31, n_cand=200, normal_en=True)
    ref = orc.warp_candidates(rs)
    _check(rs, ref, range(0, 200, 3))


def test_affine_variant_with_ncc_gate():
    rs = synth.retrieve_scenario(seed=32, n_cand=200, normal_en=False, ncc_en=True, ncc_thre=0.9)
    ref = orc.warp_candidates(rs)
    _check(rs, ref, range(0, 200, 3))


def test_radtan_cam2world_inverts_world2cam_and_matches_numpy():
    """The distorted vk::PinholeCamera::cam2world (OpenCV's undistortPoints restated: float32 pixel in, five fixed-point iterations in double, float32
    normalised point out) against a plain numpy evaluation of the same published iteration, and as the inverse of world2cam on the avia camera."""
    from oracle import orc
    cam = dict(synth.AVIA["cam"]); cam["d"] = synth.AVIA_RADTAN
    rng = np.random.default_rng(5)
    uv = np.stack([rng.uniform(0, cam["width"] - 1, 400), rng.uniform(0, cam["height"] - 1, 400)], 1)
    f, px = orc.cam_roundtrip(cam, uv)
    assert np.allclose(np.linalg.norm(f, axis=1), 1.0, atol=1e-15)
    assert np.abs(px - uv).max() < 2e-3                       # float32 pixel / float32 normalised point: ~1e-4 px; the iteration itself converges to 1e-9
    d = np.array(cam["d"])
    u32, v32 = uv[:, 0].astype(np.float32).astype(np.float64), uv[:, 1].astype(np.float32).astype(np.float64)
    x0, y0 = (u32 - cam["cx"]) * (1.0 / cam["fx"]), (v32 - cam["cy"]) * (1.0 / cam["fy"])
    x, y = x0.copy(), y0.copy()
    for _ in range(5):
        r2 = x * x + y * y
        ic = 1.0 / (1.0 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2)
        dx = 2.0 * d[2] * x * y + d[3] * (r2 + 2.0 * x * x)
        dy = d[2] * (r2 + 2.0 * y * y) + 2.0 * d[3] * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    g = np.stack([x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64), np.ones_like(x)], 1)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    assert np.abs(f - g).max() < 1e-15
    f0, _ = orc.cam_roundtrip(dict(synth.AVIA["cam"]), uv)     # the pinhole bearing differs by the distortion: up to a few pixels' worth near the corners
    assert 1e-4 < np.abs(f - f0).max() < 2e-2


def test_radtan_camera_changes_the_warps():
    rs = synth.retrieve_scenario(seed=24, n_cand=300, normal_en=True)
    a = orc.warp_candidates(rs)
    rs.cam = dict(rs.cam); rs.cam["d"] = synth.AVIA_RADTAN
    b = orc.warp_candidates(rs)
    assert np.abs(a["A"] - b["A"]).max() > 1e-4 and not np.array_equal(a["patch_wrap"], b["patch_wrap"])
    assert 0.3 < b["accepted"].mean() <= 1.0
