"""TEST INFRASTRUCTURE (oracle side): the one-scene live chain of scenarios/live_inputs.py on the CPU oracle — the sequence of calls LIVMapper makes per frame
with the data flow of the reference (all file:line relative to /root/reference):
  processImu       state_propagat = _state (+ motion: stand-in for the IMU propagation)                    src/LIVMapper.cpp:250-257
  handleLIO        StateEstimation(state_propagat); _state = state_                                        src/LIVMapper.cpp:370-371
                   pv_list_[i].point_w / .var from the POSTERIOR, UpdateVoxelMap(pv_list_), _pv_list        src/LIVMapper.cpp:413-426
  handleVIO        state = &_state (shared), state_propagat = _state (processImu ran again)                src/LIVMapper.cpp:135-136, 256
                   updateFrameState(*state) -> new_frame_->T_f_w_                                          src/vio.cpp:1799-1800, 1690-1697
                   retrieveFromVisualSparseMap(img, _pv_list, ...)                                         src/vio.cpp:1808
                   computeJacobianAndUpdateEKF(img) -> _state                                              src/vio.cpp:1810
  next frame       propagated from the VIO posterior.
Only tests/ and bench.py's cpu_baseline leg import this."""
import time

import numpy as np

from oracle import orc
from scenarios import synth


def posterior_points(xyz, body_cov, cross_mat, R, t, P, extR, extT):
    """LIVMapper.cpp:413-423: world points (float32 cloud, transformLidar 637-653) and covariances of the scan at the posterior"""
    pl = np.asarray(xyz, np.float64)
    pw = ((pl @ extR.T + extT) @ R.T + t).astype(np.float32).astype(np.float64)
    RE = R @ extR
    cb, X = body_cov.reshape(-1, 3, 3), cross_mat.reshape(-1, 3, 3)
    var = RE @ cb @ RE.T + X @ P[0:3, 0:3] @ X.transpose(0, 2, 1) + P[3:6, 3:6]
    return pw, var


def frame_pose(R, t, vs):
    """updateFrameState (vio.cpp:1690-1697): Rcw = Rci Rwi^T, Pcw = -Rci Rwi^T Pwi + Pci"""
    Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    return Rci @ R.T, -Rci @ R.T @ t + Pci


def run(live, lib=None, num_threads=1, timing=False):
    """Returns per frame dict(lio = state arrays after StateEstimation, vio = state arrays after the visual update, n_iters, n_eff, n_sub, steps); with
    timing=True also the four stage times in seconds."""
    lib = lib or orc.load()
    c, extR, extT, vs, L = live["c"], live["extR"], live["extT"], live["vs"], live["L"]
    om = orc.OracleMap.build(live["pw0"], live["var0"].reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"], lib)
    cfg = orc.lidar_cfg(c, extR, extT, num_threads=num_threads)
    post = orc.state_arrays(orc.make_state(live["R0"], live["t0"], live["P0"]))
    out = []
    grow = live.get("grow")
    gm = None
    if grow:                                                              # ONE visual map that the scripted maintenance changes after every frame
        from scenarios.visual_map_growth import GrowingMap
        gm = GrowingMap(live["cs"][0])
    for k, xyz in enumerate(live["scans"]):
        mo = live["motion"][k]
        prop = orc.make_state(post["R"] @ mo[:9].reshape(3, 3), post["t"] + mo[9:], post["P"] + np.diag(live["q"]), inv_expo=post["inv_expo"], vel=post["vel"],
                              bg=post["bg"], ba=post["ba"], grav=post["grav"])
        t0 = time.perf_counter()
        r = orc.lidar_state_estimation(om, cfg, xyz, prop, prop, want_points=True)
        ta = time.perf_counter()
        lio = orc.state_arrays(r["state"])
        pw, var = posterior_points(xyz, r["body_cov"], r["cross_mat"], lio["R"], lio["t"], lio["P"], extR, extT)
        om.update(pw, var.reshape(-1, 9))
        tb = time.perf_counter()
        if gm is not None:
            ck, g2c = gm.flat()                                           # the whole map as it stands when frame k arrives (CSR: what a full upload would carry)
            ck.img = grow["imgs"][k]
        else:
            ck = live["cs"][k]
        ck.sel.pg = pw
        ck.sel.R_cur, ck.sel.t_cur = frame_pose(lio["R"], lio["t"], vs)
        ck.inv_expo_cur = lio["inv_expo"]                                 # retrieve reads state->inv_expo_time (vio.cpp:745-752)
        ret = orc.visual_retrieve(ck, lib)
        tc = time.perf_counter()
        keep = ret["tail"]["accepted"] != 0
        sub = type("Sub", (), {})()
        for name in ("cam", "Rcl", "Pcl", "extR", "extT"):
            setattr(sub, name, getattr(vs, name))
        sub.cfg = dict(vs.cfg, patch_pyrimid_level=L)
        sub.img, sub.pos = ck.img, ck.sel.pos[ret["sub_point"]]
        sub.warp_patch, sub.search_levels, sub.inv_expo_list = ret["tail"]["patch_wrap"][keep], ret["tail"]["search_level"][keep], ck.obs_inv_expo[ret["sub_obs"]]
        steps = []
        if len(sub.pos):
            v = orc.visual_update(orc.visual_cfg(sub, num_threads=num_threads), sub, r["state"], r["state"], lib)
            vio = orc.state_arrays(v["state"])
            steps = [(t.level, t.iteration, t.accepted, t.error) for t in v["trace"]]
        else:
            vio = lio
        td = time.perf_counter()
        n_it = r["n_iters"]
        rec = dict(lio=lio, vio=vio, n_iters=n_it, n_eff=int(r["trace"][n_it - 1].n_eff) if n_it else 0, n_sub=int(keep.sum()), steps=steps, sub_point=ret["sub_point"],
                   sub_pos=ck.sel.pos[ret["sub_point"]])
        if gm is not None:
            # pt->ref_patch as the retrieval left it (vio.cpp:660-661, 689-690), then the maintenance that closes processFrame
            c2g = gm.c2g(g2c)
            gm.set_ref_patch(np.where(ret["ref_patch"] >= 0, c2g[np.maximum(ret["ref_patch"], 0)], -1))
            if k < len(grow["scripts"]):
                gm.apply(grow["scripts"][k])
            rec["map_points"], rec["map_obs"] = gm.n_points, gm.n_obs
        if timing:
            rec["stage_s"] = (ta - t0, tb - ta, tc - tb, td - tc)
        out.append(rec)
        post = vio
    return out, om


def pack(recs):
    """[F][2][25 + 361]: LIO posterior then VIO posterior per frame, in the layout live_chain writes to live_states.bin (livo2_state order)"""
    def st(a):
        return np.concatenate([a["R"].ravel(), a["t"], [a["inv_expo"]], a["vel"], a["bg"], a["ba"], a["grav"], a["P"].ravel()])
    return np.array([[st(r["lio"]), st(r["vio"])] for r in recs])
