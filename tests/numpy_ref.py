"""Independent float64 numpy re-derivation of the ESIKF measurement update, written from the formulas, not
from the oracle's code.  Used as a second opinion on the oracle (tests/test_oracle_cpu.py): given the oracle's DISCRETE decisions
(matched plane per point) it recomputes residuals, R^-1, H rows, H^T R^-1 H, H^T R^-1 z and the Kalman solution."""
import numpy as np

from scenarios import synth


def so3_log(R):
    tr = np.trace(R)
    theta = 0.0 if tr > 3.0 - 1e-6 else np.arccos(0.5 * (tr - 1))
    K = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * K if abs(theta) < 0.001 else 0.5 * theta / np.sin(theta) * K


def so3_exp(v):
    n = np.linalg.norm(v)
    if n <= 1e-5:
        return np.eye(3)
    k = v / n
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * (K @ K)


def lidar_sums(fm, xyz, match_plane, R, t, R_prop, t_prop, extR, extT, dept_err, beam_err):
    """H^T R^-1 H (6x6), H^T R^-1 z (6) and per-point r, R_inv, H rows for the given matched planes (A.1 of SURVEY.md)."""
    m = match_plane >= 0
    pl = xyz[m].astype(np.float64)
    idx = match_plane[m]
    n, c = fm.plane_normal[idx], fm.plane_center[idx]
    S = fm.plane_var[idx].reshape(-1, 6, 6)
    d = fm.plane_d[idx].astype(np.float64)
    p_i = pl @ extR.T + extT
    p_w = (p_i @ R.T + t).astype(np.float32).astype(np.float64)           # float32 world point (Q1)
    r = (np.einsum("ij,ij->i", n, p_w) + d).astype(np.float32).astype(np.float64)   # float32 residual (Q5)
    q = p_i @ R_prop.T + t_prop                                            # prior pose, un-rounded
    J = np.concatenate([q - c, -n], 1)
    sig = np.einsum("ia,iab,ib->i", J, S, J)
    Cb = synth.body_cov(pl, dept_err, beam_err)                            # z==0 patch is inside
    RE = R_prop @ extR
    var = RE @ Cb @ RE.T
    nvn = np.einsum("ia,iab,ib->i", n, var, n)
    Rinv = 1.0 / (0.001 + sig + nvn)
    A = np.cross(p_i, n @ R)                                               # [p_i]x R^T n
    H = np.concatenate([A, n], 1)
    HtH = (H * Rinv[:, None]).T @ H
    Htz = (H * Rinv[:, None]).T @ (-r)
    return dict(HtH=HtH, Htz=Htz, r=r, Rinv=Rinv, H=H, mask=m)


def boxminus(Ra, ta, Rb, tb):
    """[Log(Rb^T Ra), ta - tb] (pose part of StatesGroup::operator-, common_lib.h:194-206)"""
    return so3_log(Rb.T @ Ra), ta - tb


def esikf_solution(HtH, Htz, k, P, vec, sign=+1, scale=1.0):
    """The reference's form: K1 = (H_T_H + (P/scale)^-1)^-1 by two full inversions (voxel_map.cpp:468-472 / vio.cpp:1661-1667)."""
    Hf = np.zeros((19, 19)); Hf[:k, :k] = HtH
    K1 = np.linalg.inv(Hf + np.linalg.inv(P / scale))
    G = np.zeros((19, 19)); G[:, :k] = K1[:, :k] @ HtH
    sol = sign * (K1[:, :k] @ Htz) + vec - G[:, :k] @ vec[:k]
    return sol, G


def visual_rows(vs, level, R, t, tau, exposure=True):
    """z, H_sub (M*64 x 7) from the formulas of SURVEY.md A.2 using scenarios.synth.sample_patch for the float32 sampling."""
    Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    Rcw = Rci @ R.T
    Pcw = -Rci @ R.T @ t + Pci
    Pic = -Rci.T @ Pci
    skew = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    Jdp_dR = -Rci @ skew(Pic)
    fx, fy, cx, cy = vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"]
    M = len(vs.pos)
    z = np.zeros(M * 64); H = np.zeros((M * 64, 7))
    img = vs.img
    for i in range(M):
        pf = Rcw @ vs.pos[i] + Pcw
        pc = np.array([fx * pf[0] / pf[2] + cx, fy * pf[1] / pf[2] + cy])
        Jpi = np.array([[fx / pf[2], 0, -fx * pf[0] / pf[2] ** 2], [0, fy / pf[2], -fy * pf[1] / pf[2] ** 2]])
        sc = 1 << (level + int(vs.search_levels[i]))
        cur = synth.sample_patch(img, pc, sc)
        f32 = np.float32
        # gradients need the SAME integer anchor as the centre sample: shift the image instead of the pixel
        def shifted(dx, dy):
            return synth.sample_patch(np.roll(np.roll(img, -dy * sc, 0), -dx * sc, 1), pc, sc)
        du = f32(0.5) * (shifted(1, 0) - shifted(-1, 0))
        dv = f32(0.5) * (shifted(0, 1) - shifted(0, -1))
        g = np.stack([du.astype(np.float64).ravel(), dv.astype(np.float64).ravel()], 1) * tau * np.float64(f32(1.0) / f32(sc))
        JdR = g @ Jpi @ skew(pf) @ Rci + (-g @ Jpi) @ Jdp_dR
        Jdt = (-g @ Jpi) @ Rcw
        c = cur.astype(np.float64).ravel()
        res = tau * c - vs.inv_expo_list[i] * vs.warp_patch[i, level].astype(np.float64)
        z[i * 64:(i + 1) * 64] = res
        H[i * 64:(i + 1) * 64, :3] = JdR
        H[i * 64:(i + 1) * 64, 3:6] = Jdt
        if exposure:
            H[i * 64:(i + 1) * 64, 6] = c
    return z, H
