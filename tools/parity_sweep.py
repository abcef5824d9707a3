#!/usr/bin/env python
"""Parity sweep on a GPU box: full LiDAR and visual ESIKF updates through the C ABI against the oracle over many seeded scenarios.
Prints one line per scenario and a summary (matched-plane flips, float32 residual mismatches, worst accumulated delta-x relative error,
worst covariance relative error).  Usage: python tools/parity_sweep.py [n_lidar_seeds] [n_visual_seeds] > profiles/rNN_parity_sweep.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402  (checker only)
from scenarios import synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    nl = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    livo2 = importlib.import_module("fast-livo2_amd")
    ctx = livo2.Context(0)
    tot = dict(dec=0, flips=0, dis=0, dx=0.0, P=0.0, iters_differ=0)
    print("# LiDAR: seed points matched iters flips residual_mismatch dx_rel P_rel")
    for k in range(nl):
        seed = 300 + k
        ext = None if k % 3 else synth.rot_from_rpy(0.05 * k, -0.03 * k, 0.02 * k)
        sc = synth.lidar_scenario(seed=seed, n_points=20000, downsample=0.1, n_boxes=4 + k % 6, rot_sigma_deg=0.2 + 0.1 * (k % 5), pos_sigma=0.01 + 0.01 * (k % 4), extR=ext)
        om = orc.OracleMap.from_flat(sc.fmap)
        ocur, oprop = H.states(sc, orc.StatePOD)
        pcur, pprop = H.states(sc, livo2.State)
        ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, ocur, oprop)
        pcfg = H.lidar_cfg_product(sc)
        ctx.upload_map(sc.fmap)
        ctx.set_scan(sc.xyz, pcfg)
        res, pts = ctx.lidar_update(pcur, pprop, pcfg, want=("match_plane", "dis_to_plane"))
        flips = int((pts["match_plane"] != ref["match_plane"]).sum())
        dis = int((pts["dis_to_plane"] != ref["dis"]).sum())
        so, sp = orc.state_arrays(ref["state"]), orc.state_arrays(res.state)
        dx_ref = np.concatenate([so["t"] - sc.t_prior, (sc.R_prior.T @ so["R"] - np.eye(3)).ravel()])
        dx_gpu = np.concatenate([sp["t"] - sc.t_prior, (sc.R_prior.T @ sp["R"] - np.eye(3)).ravel()])
        dx, dP = H.relerr(dx_gpu, dx_ref), H.relerr(sp["P"], so["P"])
        tot["dec"] += len(sc.xyz) * res.n_iters
        tot["flips"] += flips
        tot["dis"] += dis
        tot["dx"] = max(tot["dx"], dx)
        tot["P"] = max(tot["P"], dP)
        tot["iters_differ"] += int(res.n_iters != ref["n_iters"])
        print(seed, len(sc.xyz), int((ref["match_plane"] >= 0).sum()), res.n_iters, flips, dis, f"{dx:.2e}", f"{dP:.2e}", flush=True)
    print(f"# LiDAR summary: {tot['dec']} point-iterations, {tot['flips']} matched-plane flips, {tot['dis']} float32 residual mismatches, "
          f"{tot['iters_differ']} iteration-count differences, worst dx rel {tot['dx']:.2e}, worst P rel {tot['P']:.2e}")
    print("# visual: seed patches steps steps_equal dR dt dtau dP_rel")
    worst = dict(R=0.0, t=0.0, P=0.0)
    steps_bad, rows = 0, 0
    for k in range(nv):
        seed = 400 + k
        vs = synth.visual_scenario(seed=seed, n_patches=2000, rot_sigma_deg=0.03 + 0.01 * (k % 4))
        ocur, oprop = H.states(vs, orc.StatePOD)
        pcur, pprop = H.states(vs, livo2.State)
        exposure = bool(k % 2 == 0)
        ref = orc.visual_update(orc.visual_cfg(vs, exposure=exposure), vs, ocur, oprop)
        ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        res, _ = ctx.visual_update(pcur, pprop, H.visual_cfg_product(vs, exposure=exposure))
        a = [(t.level, t.iteration, t.accepted, t.n_meas) for t in ref["trace"]]
        b = [(res.steps[i].level, res.steps[i].iteration, res.steps[i].accepted, res.steps[i].n_meas) for i in range(res.n_steps)]
        d = H.state_diff(res.state, ref["state"])
        steps_bad += int(a != b)
        rows += sum(s[3] for s in a)
        for key in ("R", "t", "P"):
            worst[key] = max(worst[key], d[key])
        print(seed, len(vs.pos), len(a), int(a == b), f"{d['R']:.2e}", f"{d['t']:.2e}", f"{d['inv_expo']:.2e}", f"{d['P']:.2e}", flush=True)
    print(f"# visual summary: {rows} scalar residual rows, {steps_bad} scenarios with a different accept/revert sequence, worst dR {worst['R']:.2e}, "
          f"dt {worst['t']:.2e}, dP rel {worst['P']:.2e}")
    ctx.close()


if __name__ == "__main__":
    main()
