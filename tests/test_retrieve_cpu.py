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
