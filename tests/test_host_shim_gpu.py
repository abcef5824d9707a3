"""The C++ host shim (fast-livo2_amd/host): reference-shaped containers -> C ABI -> HIP, driven like LIVMapper drives the reference
managers, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "fast-livo2_amd", "lib", "shim_demo")


def _state_vec(s):
    return np.frombuffer(bytes(s), dtype=np.float64).copy()


def test_shim_builds_and_links():
    subprocess.run(["make", "-C", os.path.join(ROOT, "fast-livo2_amd", "host")], check=True, capture_output=True)
    assert os.path.exists(DEMO)
    out = subprocess.run(["ldd", DEMO], capture_output=True, text=True).stdout
    assert "liblivo2_host.so" in out and "liblivo2_hip.so" in out and "not found" not in out


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path, orc, livo2):
    d = str(tmp_path)
    sc = synth.lidar_scenario(seed=17, n_points=2500, downsample=0.1)
    fm = sc.fmap
    for name in ("root_key", "root_node", "root_center", "root_quarter", "node_plane", "node_child", "plane_normal", "plane_center", "plane_var", "plane_d", "plane_radius"):
        np.ascontiguousarray(getattr(fm, name)).tofile(os.path.join(d, name + ".bin"))
    sc.xyz.tofile(os.path.join(d, "xyz.bin"))
    c = sc.cfg
    np.concatenate([[c["max_iterations"], c["max_layer"], c["sigma_num"], c["dept_err"], c["beam_err"], c["voxel_size"]], sc.extR.ravel(), sc.extT]).astype(np.float64).tofile(os.path.join(d, "lidar_cfg.bin"))
    ocur, oprop = H.states(sc, orc.StatePOD)
    _state_vec(ocur).tofile(os.path.join(d, "state_in.bin")); _state_vec(oprop).tofile(os.path.join(d, "state_prop.bin"))
    vs = synth.visual_scenario(seed=18, n_patches=100)
    vs.img.tofile(os.path.join(d, "img.bin"))
    np.concatenate([[vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"], vs.cam["width"], vs.cam["height"], vs.cfg["img_point_cov"], vs.cfg["patch_pyrimid_level"],
                     vs.cfg["max_iterations"], 1.0], vs.Rcl.ravel(), vs.Pcl, vs.extR.ravel(), vs.extT]).astype(np.float64).tofile(os.path.join(d, "vis_cfg.bin"))
    vs.pos.astype(np.float64).tofile(os.path.join(d, "vis_pos.bin")); vs.warp_patch.astype(np.float32).tofile(os.path.join(d, "vis_warp.bin"))
    vs.search_levels.astype(np.int32).tofile(os.path.join(d, "vis_search.bin")); vs.inv_expo_list.astype(np.float64).tofile(os.path.join(d, "vis_invexpo.bin"))
    vcur, vprop = H.states(vs, orc.StatePOD)
    _state_vec(vcur).tofile(os.path.join(d, "vis_state_in.bin")); _state_vec(vprop).tofile(os.path.join(d, "vis_state_prop.bin"))

    r = subprocess.run([DEMO, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr

    om = orc.OracleMap.from_flat(fm)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, ocur, oprop)
    out = np.fromfile(os.path.join(d, "out_state.bin"))
    refv = _state_vec(ref["state"])
    assert np.allclose(out[:25], refv[:25], rtol=0, atol=1e-9) and H.relerr(out[25:], refv[25:]) < 1e-8
    assert int(np.fromfile(os.path.join(d, "out_effct.bin"), dtype=np.int32)[0]) == int((ref["match_plane"] >= 0).sum())
    assert np.array_equal(np.fromfile(os.path.join(d, "out_ptpl_dis.bin"), dtype=np.float32), ref["dis"][ref["match_plane"] >= 0])      # ptpl_list_ keeps point order
    assert np.array_equal(np.fromfile(os.path.join(d, "out_pv_normal.bin")).reshape(-1, 3), ref["normal"])
    assert H.relerr(np.fromfile(os.path.join(d, "out_pv_var.bin")).reshape(-1, 9), ref["var"]) < 1e-13

    vref = orc.visual_update(orc.visual_cfg(vs), vs, vcur, vprop)
    vout = np.fromfile(os.path.join(d, "vis_out_state.bin"))
    vrefv = _state_vec(vref["state"])
    assert np.allclose(vout[:25], vrefv[:25], rtol=0, atol=1e-9) and H.relerr(vout[25:], vrefv[25:]) < 1e-8
    assert np.allclose(np.fromfile(os.path.join(d, "vis_out_errors.bin"), dtype=np.float32), vref["errors"], rtol=1e-5)
    assert H.relerr(np.fromfile(os.path.join(d, "vis_out_G.bin")).reshape(19, 19), vref["G"]) < 1e-7
