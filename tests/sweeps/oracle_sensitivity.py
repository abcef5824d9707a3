#!/usr/bin/env python
"""How far can the reference's OWN build move the answer?  ("weak" #1; CPU only, no GPU, no /root/reference at run time.)

The parity tests pin the HIP path to ONE realisation of the reference arithmetic: oracle/ compiled `-O2 -ffp-contract=off`, every long sum strictly
left-to-right.  The reference binary is a different realisation: `-O3 -march=native` with GCC's default `-ffp-contract=fast` (CMakeLists.txt:34)
and Eigen's GEMM / GEMV order of additions for `Hsub_T_R_inv * Hsub`, `Hsub_T_R_inv * meas_vec` (voxel_map.cpp:464-466) and `H_sub^T H_sub`,
`H_sub^T z` (vio.cpp:1660-1662).  This script runs the same seeded scenarios through four builds of the oracle

    A  golden        -O2 -ffp-contract=off, sequential sums               (the parity checker)
    B  fast          -O3 -march=native -funroll-loops (contraction on), sequential sums
    C  fast+eigen    B with the Eigen-like order (orc_math.hpp SumModel: kc-panelled FMA chains, 4-lane packets)      kc = 256
    D  golden+eigen  A with the Eigen-like order, kc = 128

and reports, against A: matched-plane flips of the LAST iteration's match list, iteration-count changes, spread of the accumulated update
`x_final [-] x_prior` (relative) for the LiDAR update; accept/revert sequence changes, per-step float error differences and pose spread for the visual update.
Usage: python tests/sweeps/oracle_sensitivity.py [n_lidar] [n_visual] > profiles/r02_oracle_sensitivity.txt
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from scenarios import synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def variants():
    gold = orc.load()
    fast = orc.load(orc.build("fast", out_dir=tempfile.mkdtemp(prefix="orc_fast_")))
    return [("A golden", gold, (0, 256, 4)), ("B fast", fast, (0, 256, 4)), ("C fast+eigen(kc=256,4 lanes)", fast, (1, 256, 4)), ("D golden+eigen(kc=128,4 lanes)", gold, (1, 128, 4))]


def dx_of(sc, st):
    s = orc.state_arrays(st)
    return np.concatenate([s["t"] - sc.t_prior, (sc.R_prior.T @ s["R"] - np.eye(3)).ravel()])


def main():
    nl = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    V = variants()
    print("# oracle sensitivity: variants", "; ".join(v[0] for v in V))
    print("# LiDAR StateEstimation (tests/sweeps/parity_sweep.py scenarios): seed points | per variant vs A: flips(last-iteration match list) iters dx_rel P_rel")
    tot = {v[0]: dict(flips=0, iters=0, dx=0.0, P=0.0, dec=0) for v in V[1:]}
    for k in range(nl):
        seed = 300 + k
        ext = None if k % 3 else synth.rot_from_rpy(0.05 * k, -0.03 * k, 0.02 * k)
        sc = synth.lidar_scenario(seed=seed, n_points=20000, downsample=0.1, n_boxes=4 + k % 6, rot_sigma_deg=0.2 + 0.1 * (k % 5), pos_sigma=0.01 + 0.01 * (k % 4), extR=ext)
        res = []
        for name, lib, sm in V:
            lib.orc_set_sum_model(*sm)
            om = orc.OracleMap.from_flat(sc.fmap, lib)
            cur, prop = H.states(sc, orc.StatePOD)
            res.append(orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, cur, prop))
            lib.orc_set_sum_model(0, 256, 4)
        a = res[0]
        dxa, Pa = dx_of(sc, a["state"]), orc.state_arrays(a["state"])["P"]
        cols = []
        for (name, _, _), r in zip(V[1:], res[1:]):
            flips = int((r["match_plane"] != a["match_plane"]).sum())
            dx, dP = H.relerr(dx_of(sc, r["state"]), dxa), H.relerr(orc.state_arrays(r["state"])["P"], Pa)
            t = tot[name]
            t["flips"] += flips; t["iters"] += int(r["n_iters"] != a["n_iters"]); t["dx"] = max(t["dx"], dx); t["P"] = max(t["P"], dP); t["dec"] += len(sc.xyz) * a["n_iters"]
            cols.append(f"{flips} {r['n_iters']} {dx:.1e} {dP:.1e}")
        print(seed, len(sc.xyz), a["n_iters"], "|", " | ".join(cols), flush=True)
    for name, t in tot.items():
        print(f"# LiDAR summary, {name} vs A: {t['flips']} matched-plane flips in the final match lists ({t['dec']} point-iterations run), {t['iters']} iteration-count changes, "
              f"worst dx rel {t['dx']:.2e} (contract 1e-5), worst P rel {t['P']:.2e}")
    print("# visual computeJacobianAndUpdateEKF: seed patches steps | per variant vs A: same_accept_sequence max_rel_error_diff dR dt")
    vt = {v[0]: dict(seq=0, err=0.0, R=0.0, t=0.0) for v in V[1:]}
    for k in range(nv):
        seed = 400 + k
        vs = synth.visual_scenario(seed=seed, n_patches=2000, rot_sigma_deg=0.03 + 0.01 * (k % 4))
        exposure = bool(k % 2 == 0)
        res = []
        for name, lib, sm in V:
            lib.orc_set_sum_model(*sm)
            cur, prop = H.states(vs, orc.StatePOD)
            res.append(orc.visual_update(orc.visual_cfg(vs, exposure=exposure), vs, cur, prop, lib))
            lib.orc_set_sum_model(0, 256, 4)
        a = res[0]
        seq_a = [(t.level, t.iteration, t.accepted) for t in a["trace"]]
        cols = []
        for (name, _, _), r in zip(V[1:], res[1:]):
            seq = [(t.level, t.iteration, t.accepted) for t in r["trace"]]
            same = seq == seq_a
            ed = max((abs(x.error - y.error) / max(abs(y.error), 1e-30) for x, y in zip(r["trace"], a["trace"])), default=0.0)
            d = H.state_diff(r["state"], a["state"])
            t = vt[name]
            t["seq"] += int(not same); t["err"] = max(t["err"], ed); t["R"] = max(t["R"], d["R"]); t["t"] = max(t["t"], d["t"])
            cols.append(f"{int(same)} {ed:.1e} {d['R']:.1e} {d['t']:.1e}")
        print(seed, len(vs.pos), len(seq_a), "|", " | ".join(cols), flush=True)
    for name, t in vt.items():
        print(f"# visual summary, {name} vs A: {t['seq']} scenarios with a different accept/revert sequence, worst relative float-error difference per step {t['err']:.2e}, "
              f"worst dR {t['R']:.2e}, worst dt rel {t['t']:.2e}")


if __name__ == "__main__":
    main()
