"""The one-scene live chain through the C++ shim (fast-livo2_amd/host/live_chain.cpp) against the oracle running the same chain (oracle/live_chain.py) — the data
flow VERDICT r03 found missing: the VIO update starts from the LIO posterior through the shared `state` (LIVMapper.cpp:135-136, 256-257, 371; vio.cpp:1799-1810),
`pg` and new_frame_->T_f_w_ come from that posterior, the map the next frame reads is the one this frame updated (LIVMapper.cpp:413-426), the next frame is
propagated from the VIO posterior.  Both shim modes: "full" (pv_list_ / ptpl_list_ kept on the host as the reference keeps them, pg uploaded from pv_list_) and
"lean" (nothing per point on the host; pg read where the map update left it on the GPU).
Compared per frame: effct_feat_num_, the size and the members of the visual sub-map (decisions: equal), LIO and VIO posterior states (1e-7 / P 1e-6: a frame
inherits the ~1e-8 differences of the planes re-fitted on the device, tests/test_plane_fit_gpu.py, and hands them on)."""
import os
import subprocess

import numpy as np
import pytest

from scenarios import live_inputs as LI

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fast-livo2_amd", "lib", "live_chain")
K = 25 + 361


@pytest.fixture(scope="module")
def chain(tmp_path_factory):
    from oracle import live_chain as OC
    live = LI.make_live(**LI.SIZES["test"])
    d = str(tmp_path_factory.mktemp("live"))
    LI.write_live_dir(d, live)
    recs, _ = OC.run(live)
    return d, live, recs, OC.pack(recs)


@pytest.mark.parametrize("mode", ["full", "lean"])
def test_one_scene_chain_matches_the_oracle(chain, mode):
    d, live, recs, want = chain
    r = subprocess.run([EXE, d] + (["lean"] if mode == "lean" else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    F = len(recs)
    got = np.fromfile(os.path.join(d, "live_states.bin")).reshape(F, 2, K)
    counts = np.fromfile(os.path.join(d, "live_counts.bin"), np.int32).reshape(F, 2)
    sub_pos = np.fromfile(os.path.join(d, "live_sub_pos.bin")).reshape(-1, 3)
    at = 0
    for f, rec in enumerate(recs):
        assert counts[f, 0] == rec["n_eff"], (f, counts[f, 0], rec["n_eff"])
        assert counts[f, 1] == rec["n_sub"], (f, counts[f, 1], rec["n_sub"])
        assert np.array_equal(sub_pos[at:at + rec["n_sub"]], live["cs"][f].sel.pos[rec["sub_point"]]), f      # the same visual points, in the same (grid) order
        at += rec["n_sub"]
        for which in (0, 1):
            g, w = got[f, which], want[f, which]
            assert np.abs(g[:25] - w[:25]).max() < 1e-7, (mode, f, which, np.abs(g[:25] - w[:25]).max())
            assert np.linalg.norm(g[25:] - w[25:]) < 1e-6 * np.linalg.norm(w[25:]), (mode, f, which)
        # the chain is a chain: the VIO posterior differs from the LIO posterior it started from, and the next LIO prior comes from it
        assert np.abs(got[f, 1, :12] - got[f, 0, :12]).max() > 1e-6
        assert np.linalg.norm(rec["vio"]["t"] - live["t_true"][f]) < 0.03
    assert at == len(sub_pos)
    print(r.stdout)


def test_full_and_lean_give_the_same_states(chain):
    d, live, recs, want = chain
    outs = []
    for args in ([], ["lean"]):
        r = subprocess.run([EXE, d] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(os.path.join(d, "live_states.bin")))
    assert np.array_equal(outs[0], outs[1])


def test_retrieve_and_update_as_one_member_equals_the_two_calls(chain):
    """lean runs vio.cpp:1808 + 1810 through VIOManager::retrieveAndUpdate (the update is enqueued before visual_submap's host lists are built); with
    LIVO2_LIVE_SPLIT_VIO=1 the same program makes the two separate calls.  Same states, same sub-maps (members, order), same counts, bit for bit."""
    d, live, recs, want = chain
    outs = []
    for mode_args, env_add in ((["lean"], {}), (["lean"], {"LIVO2_LIVE_SPLIT_VIO": "1"}), (["lean"], {"LIVO2_LIVE_SPLIT_VIO": "1", "LIVO2_LIVE_SYNC_MAP": "1"}),
                               ([], {}), ([], {"LIVO2_LIVE_SPLIT_VIO": "1", "LIVO2_LIVE_SYNC_MAP": "1"})):      # (the form with the host point lists uses both as well)
        r = subprocess.run([EXE, d] + mode_args, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_add))
        assert r.returncode == 0, r.stderr
        outs.append([np.fromfile(os.path.join(d, n)) for n in ("live_states.bin", "live_sub_pos.bin")] + [np.fromfile(os.path.join(d, "live_counts.bin"), np.int32)])
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)


@pytest.fixture(scope="module")
def grow_chain(tmp_path_factory):
    from oracle import live_chain as OC
    live = LI.make_live(**LI.SIZES["test_grow"])
    d = str(tmp_path_factory.mktemp("live_grow"))
    LI.write_live_dir(d, live)
    recs, _ = OC.run(live)
    return d, live, recs, OC.pack(recs)


@pytest.mark.parametrize("mode", ["full", "lean"])
def test_growing_visual_map_through_the_incremental_mirror(grow_chain, mode):
    """ONE visual map that the (scripted) map maintenance changes after every frame — new points, new observations pushed to the front of obs_, deletions, ref_patch /
    normal changes, a new reference image (scenarios/visual_map_growth.py) — replayed on the shim's objects through insertPointIntoVoxelMap / markPointDirty /
    erasePointFromVoxelMap; syncFeatMap brings the device mirror up to date with ONE full upload before frame 0 and O(changes) deltas afterwards
    (livo2_visual_map_apply).  Against the oracle that gets the WHOLE map re-flattened per frame: same sub-maps (members, order), same states."""
    d, live, recs, want = grow_chain
    r = subprocess.run([EXE, d] + (["lean"] if mode == "lean" else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    F = len(recs)
    got = np.fromfile(os.path.join(d, "live_states.bin")).reshape(F, 2, K)
    counts = np.fromfile(os.path.join(d, "live_counts.bin"), np.int32).reshape(F, 2)
    sub_pos = np.fromfile(os.path.join(d, "live_sub_pos.bin")).reshape(-1, 3)
    at = 0
    for f, rec in enumerate(recs):
        assert counts[f, 0] == rec["n_eff"] and counts[f, 1] == rec["n_sub"] > 100, (f, counts[f], rec["n_eff"], rec["n_sub"])
        assert np.array_equal(sub_pos[at:at + rec["n_sub"]], rec["sub_pos"]), f                      # the same visual points, in the same (grid) order
        at += rec["n_sub"]
        for which in (0, 1):
            g, w = got[f, which], want[f, which]
            assert np.abs(g[:25] - w[:25]).max() < 1e-7, (mode, f, which, np.abs(g[:25] - w[:25]).max())
            assert np.linalg.norm(g[25:] - w[25:]) < 1e-6 * np.linalg.norm(w[25:]), (mode, f, which)
        assert np.linalg.norm(rec["vio"]["t"] - live["t_true"][f]) < 0.03
    assert at == len(sub_pos)
    assert ("%d delta syncs, 1 full" % (F - 1)) in r.stdout, r.stdout
    assert ("visual map %d points / %d observations" % (recs[-1]["map_points"], recs[-1]["map_obs"])) in r.stdout, r.stdout
    # new points of later frames do get selected: the sub-map of the last frame holds points that did not exist when frame 0 arrived
    n_base = len(live["cs"][0].sel.pos)
    assert (recs[-1]["sub_point"] >= n_base).sum() > 0
    print(r.stdout)
