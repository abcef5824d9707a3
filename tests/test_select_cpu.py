"""Pins the oracle's selection half of retrieveFromVisualSparseMap (oracle/orc_select.hpp, reference src/vio.cpp:352-486, 598-635) with a
vectorised numpy evaluation of the same definitions (sets instead of hash maps, argmin instead of the running `<=`)."""
import numpy as np

from oracle import orc
from scenarios import synth


def _numpy_select(ss):
    cam = ss.cam
    def proj(p):
        pc = p @ ss.R_cur.T + ss.t_cur
        return pc, np.stack([cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"], cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]], 1)
    def inframe(px):
        x, y = px[:, 0].astype(np.int64), px[:, 1].astype(np.int64)
        return (x >= ss.border) & (x < cam["width"] - ss.border) & (y >= ss.border) & (y < cam["height"] - ss.border)
    loc = np.floor(ss.pg / 0.5).astype(np.int64)
    loc = np.where(loc < 0, loc - 1, loc)
    scan_vox = set(map(tuple, loc))
    pc, px = proj(ss.pg)
    depth = np.zeros((cam["height"], cam["width"]), np.float32)
    ok = (pc[:, 2] > 0) & inframe(np.nan_to_num(px))
    for i in np.nonzero(ok)[0]:                                      # last writer wins
        depth[int(px[i, 1]), int(px[i, 0])] = np.float32(pc[i, 2])
    in_vox = np.array([tuple(k) in scan_vox for k in ss.keys])
    vc, vpx = proj(ss.pos)
    cand = in_vox & (ss.active > 0) & ~(vc[:, 2] < 0)
    fov = cand & inframe(np.nan_to_num(vpx))
    length = ss.grid_n_width * ss.grid_n_height
    cell = (vpx[:, 1] / ss.grid_size).astype(np.int64) * ss.grid_n_width + (vpx[:, 0] / ss.grid_size).astype(np.int64)
    cam_pos = -ss.R_cur.T @ ss.t_cur
    dist = np.linalg.norm(cam_pos - ss.pos, axis=1).astype(np.float32)
    cell_point = np.full(length, -1); cell_dist = np.full(length, 10000.0, np.float32); disc = np.zeros(length, int)
    for c in np.unique(cell[fov]):
        idx = np.nonzero(fov & (cell == c))[0]
        j = idx[np.argmin(dist[idx])]
        cell_point[c], cell_dist[c] = j, dist[j]
        u0, v0 = int(vpx[j, 0]), int(vpx[j, 1])
        win = depth[v0 - 4:v0 + 5, u0 - 4:u0 + 5].astype(np.float64).copy()
        win[4, 4] = 0.0
        disc[c] = int(np.any((win != 0) & (np.abs(vc[j, 2] - win) > 0.5)))
    return cell_point, cell_dist, disc, fov, depth


def test_select_matches_numpy():
    ss = synth.select_scenario(seed=72, n_pg=6000, n_vis=4000)
    o = orc.visual_select(ss)
    cp, cd, disc, fov, depth = _numpy_select(ss)
    assert np.array_equal(o["depth_img"], depth)
    assert np.array_equal(o["in_fov"].astype(bool), fov)
    assert np.array_equal(o["cell_point"], cp) and np.array_equal(o["cell_dist"], cd)
    assert np.array_equal(o["cell_type"] == 1, cp >= 0)
    assert np.array_equal(o["discont"], disc)
    assert (cp >= 0).sum() > 100 and disc.sum() > 5 and fov.sum() > 300


def test_negative_axis_key_mismatch_is_reproduced():
    """A scan point at x = -0.3 looks into voxel -2 (floor(-0.6) = -1, then another -1) while a visual point at x = -0.3 is filed under
    -1 ((int64)(-0.6f - 1)): they never meet, but a visual point at x = -0.7 (filed under -2) does."""
    ss = synth.select_scenario(seed=73, n_pg=10, n_vis=2)
    ss.R_cur, ss.t_cur = np.eye(3), np.zeros(3)
    ss.pg = np.array([[-0.3, 0.2, 5.0]])
    ss.pos = np.array([[-0.3, 0.2, 5.0], [-0.7, 0.2, 5.0]])
    ss.keys = synth.feat_map_key_np(ss.pos)
    ss.active = np.ones(2, np.uint8)
    assert list(ss.keys[:, 0]) == [-1, -2]
    o = orc.visual_select(ss)
    assert list(o["in_fov"]) == [0, 1]
    assert np.array_equal(orc.feat_map_keys(ss.pos), ss.keys)
