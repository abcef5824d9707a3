"""Oracle pinning for the middle of retrieveFromVisualSparseMap (reference src/vio.cpp:644-735, src/visual_point.cpp:57-95): the reference-patch
choice against hand-made known answers and a numpy second opinion, the warp_map reuse of the !normal_en branch, and the chained oracle
(selection -> choice -> tail) against its own stages."""
import numpy as np
import pytest

from oracle import orc
from scenarios import synth


def _mini(normal_en, pos, obs_per_point, ref_patch=None, ninit=None, R_cur=None, t_cur=None):
    """A scenario-like object with one grid cell per point (cell c selects point c)."""
    class S:
        pass
    cs, sel = S(), S()
    n = len(pos)
    sel.pos = np.asarray(pos, float); sel.R_cur = np.eye(3) if R_cur is None else R_cur; sel.t_cur = np.zeros(3) if t_cur is None else t_cur
    cs.sel = sel
    cs.cfg = dict(normal_en=int(normal_en))
    offs = np.zeros(n + 1, np.int32)
    ids, Rs, ts, patches = [], [], [], []
    for i, obs in enumerate(obs_per_point):
        for (oid, R, t, patch) in obs:
            ids.append(oid); Rs.append(np.asarray(R, float).ravel()); ts.append(np.asarray(t, float)); patches.append(np.asarray(patch, np.float32))
        offs[i + 1] = offs[i] + len(obs)
    cs.obs_offset, cs.obs_id = offs, np.array(ids, np.int32)
    cs.obs_R, cs.obs_t, cs.obs_patch = np.array(Rs).reshape(-1, 9), np.array(ts).reshape(-1, 3), np.array(patches, np.float32).reshape(-1, 64)
    cs.normal_initialized = np.ones(n, np.uint8) if ninit is None else np.asarray(ninit, np.uint8)
    cs.ref_patch = np.full(n, -1, np.int32) if ref_patch is None else np.asarray(ref_patch, np.int32)
    return cs


def _score_f32(patches, ids, a):
    """float32 serial restatement of vio.cpp:666-684 for observation a"""
    err, count = np.float32(0), 0
    for b in range(len(ids)):
        if ids[b] == ids[a]:
            continue
        for k in range(64):
            d = np.float32(patches[a][k] - patches[b][k])
            err = np.float32(err + np.float32(d * d))
        count += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.float32(err) / np.float32(count)


def test_choice_by_patches_known_answers():
    rng = np.random.default_rng(1)
    I, z = np.eye(3), np.zeros(3)
    base = rng.uniform(0, 255, 64).astype(np.float32)
    mk = lambda off: base + np.float32(off)
    # point 0: one observation -> it is chosen and becomes ref_patch
    # point 1: three frames; the middle patch is closest to the other two
    # point 2: preset ref_patch is kept although another observation would score better
    # point 3: two observations of ONE frame + one of another: same-id pairs do not count against each other
    # point 4: all observations carry one id -> 0/0, skipped, ref_patch stays -1
    # point 5: normal not initialised -> skipped
    # point 6: depth-discontinuous cell -> skipped
    obs = [[(7, I, z, mk(0))],
           [(1, I, z, mk(0)), (2, I, z, mk(10)), (3, I, z, mk(25))],
           [(1, I, z, mk(0)), (2, I, z, mk(10)), (3, I, z, mk(25))],
           [(1, I, z, mk(0)), (1, I, z, mk(100)), (2, I, z, mk(90))],
           [(4, I, z, mk(0)), (4, I, z, mk(5))],
           [(1, I, z, mk(0)), (2, I, z, mk(1))],
           [(1, I, z, mk(0)), (2, I, z, mk(1))]]
    off = np.cumsum([0] + [len(o) for o in obs])
    cs = _mini(True, np.zeros((7, 3)), obs, ref_patch=[-1, -1, off[2] + 2, -1, -1, -1, -1], ninit=[1, 1, 1, 1, 1, 0, 1])
    cell_point = np.arange(7, dtype=np.int32)
    discont = np.array([0, 0, 0, 0, 0, 0, 1], np.int32)
    cell_obs, rp = orc.choose_ref(cs, cell_point, discont)
    assert cell_obs.tolist() == [0, off[1] + 1, off[2] + 2, off[3] + 1, -1, -1, -1]
    # point 3: obs 0 scores 90^2, obs 1 scores 10^2 (only the frame-2 patch counts), obs 2 scores (90^2 + 10^2) / 2
    assert rp.tolist() == [0, off[1] + 1, off[2] + 2, off[3] + 1, -1, -1, -1]


def test_choice_by_patches_numpy_second_opinion():
    rng = np.random.default_rng(2)
    I, z = np.eye(3), np.zeros(3)
    obs = []
    for i in range(60):
        m = int(rng.integers(2, 9))
        ids = rng.integers(0, 4, m)
        obs.append([(int(ids[k]), I, z, rng.uniform(0, 255, 64).astype(np.float32)) for k in range(m)])
    cs = _mini(True, np.zeros((60, 3)), obs)
    cell_obs, rp = orc.choose_ref(cs, np.arange(60, dtype=np.int32), np.zeros(60, np.int32))
    for i in range(60):
        b = cs.obs_offset[i]
        ids = [o[0] for o in obs[i]]
        patches = [o[3] for o in obs[i]]
        scores = [_score_f32(patches, ids, a) for a in range(len(ids))]
        best, arg = np.float32(np.finfo(np.float32).max), -1
        for a, sc in enumerate(scores):
            if sc < best:
                best, arg = sc, a
        assert cell_obs[i] == (b + arg if arg >= 0 else -1), i
        assert rp[i] == cell_obs[i]


def test_close_view_obs():
    # camera at the origin looking at a point 5 m ahead; observations from cameras at various angles around the point
    pos = np.array([[0.0, 0.0, 5.0]])
    def cam_at(c):                       # T_f_w_ with identity rotation: t = -c
        return (np.eye(3), -np.asarray(c, float))
    p64 = np.zeros(64, np.float32)
    deg = np.deg2rad
    def at_angle(a, r=4.0):              # a camera seen from the point under angle a to the current camera's direction
        return cam_at(pos[0] + r * np.array([np.sin(a), 0.0, -np.cos(a)]))
    cases = [([at_angle(deg(70)), at_angle(deg(20)), at_angle(deg(5)), at_angle(deg(5))], 2),      # best cosine, the FIRST of two equal ones
             ([at_angle(deg(61)), at_angle(deg(75))], -1),                                          # more than 60 degrees off
             ([at_angle(deg(59.5))], 0),
             ([at_angle(deg(120)), at_angle(deg(100))], -1)]                                        # negative cosines never replace the start value 0
    for cams, want in cases:
        obs = [[(k, R, t, p64) for k, (R, t) in enumerate(cams)]]
        cs = _mini(False, pos, obs)
        cell_obs, rp = orc.choose_ref(cs, np.zeros(1, np.int32), np.zeros(1, np.int32))
        assert cell_obs[0] == want
        assert rp[0] == -1                # the !normal_en branch never touches ref_patch
    # numpy second opinion with rotated reference frames
    rng = np.random.default_rng(3)
    R_cur = synth.rot_from_rpy(0.1, -0.2, 0.3); t_cur = np.array([0.4, -0.1, 0.2])
    framepos = -R_cur.T @ t_cur
    P = rng.uniform(-5, 5, (80, 3))
    obs, want = [], []
    for i in range(80):
        m = int(rng.integers(1, 7))
        row, cosines = [], []
        for k in range(m):
            R = synth.so3_exp(rng.normal(0, 1.0, 3)); c = P[i] + rng.normal(0, 3.0, 3)
            row.append((k, R, -R @ c, p64))
            a, b = framepos - P[i], c - P[i]
            cosines.append(float(a @ b / np.linalg.norm(a) / np.linalg.norm(b)))
        best, arg = 0.0, 0
        for k, cv in enumerate(cosines):
            if cv > best + 1e-12:
                best, arg = cv, k
        want.append(arg if best >= 0.5 else -1)
        obs.append(row)
    cs = _mini(False, P, obs, R_cur=R_cur, t_cur=t_cur)
    cell_obs, _ = orc.choose_ref(cs, np.arange(80, dtype=np.int32), np.zeros(80, np.int32))
    got = [int(o - cs.obs_offset[i]) if o >= 0 else -1 for i, o in enumerate(cell_obs)]
    assert got == want


def test_warp_map_reuses_first_warp_per_frame_id():
    rs = synth.retrieve_scenario(seed=33, n_cand=300, normal_en=False)
    plain = orc.warp_candidates(rs)
    rs.ref_id = np.arange(300, dtype=np.int32)              # every feature from its own frame: nothing is reused
    uniq = orc.warp_candidates(rs)
    for k in ("accepted", "search_level", "error", "A"):
        assert np.array_equal(plain[k], uniq[k]), k
    rs.ref_id = (np.arange(300) % 7).astype(np.int32)       # seven frames: candidates 0..6 lead, the rest reuse their warp
    shared = orc.warp_candidates(rs)
    for i in range(300):
        assert np.array_equal(shared["A"][i], plain["A"][i % 7]) and shared["search_level"][i] == plain["search_level"][i % 7]
    assert (shared["error"][7:] != plain["error"][7:]).any()  # the reused warp changes the patches, hence the errors
    rs2 = synth.retrieve_scenario(seed=33, n_cand=300, normal_en=True)
    a = orc.warp_candidates(rs2)
    rs2.ref_id = rs.ref_id
    b = orc.warp_candidates(rs2)                             # normal_en: warp_map is not used at all
    assert np.array_equal(a["A"], b["A"]) and np.array_equal(a["error"], b["error"])


@pytest.mark.parametrize("normal_en", [True, False])
def test_chain_is_its_stages(normal_en):
    cs = synth.retrieve_chain_scenario(seed=85, n_pg=6000, n_vis=8000, normal_en=normal_en)
    o = orc.visual_retrieve(cs)
    sel = o["sel"]
    ok = (sel["cell_point"] >= 0) & (sel["discont"] == 0)
    assert (o["cell_obs"][~ok] == -1).all()
    pts = sel["cell_point"][o["cand_cell"]]
    assert (cs.normal_initialized[pts] == 1).all()
    assert ((o["cand_obs"] >= cs.obs_offset[pts]) & (o["cand_obs"] < cs.obs_offset[pts + 1])).all()
    assert len(o["cand_cell"]) > 20 and 0 < len(o["sub_point"]) <= len(o["cand_cell"])
    if normal_en:
        assert np.array_equal(o["ref_patch"][pts], o["cand_obs"])                       # chosen = remembered
        untouched = np.ones(len(cs.ref_patch), bool); untouched[pts] = False
        assert np.array_equal(o["ref_patch"][untouched], cs.ref_patch[untouched])
        again, rp2 = orc.choose_ref(cs, sel["cell_point"], sel["discont"], ref_patch=o["ref_patch"])
        assert np.array_equal(again, o["cell_obs"]) and np.array_equal(rp2, o["ref_patch"])   # idempotent once remembered
    else:
        assert np.array_equal(o["ref_patch"], cs.ref_patch)
        assert (cs.obs_img_idx[o["cand_obs"]] != 5).all()                                 # the side-view frame is never chosen
