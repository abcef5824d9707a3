"""CPU half of the shipped-configuration matrix (tests/test_shipped_configs_gpu.py): the profiles of scenarios/shipped_configs.py quote the reference's yaml values,
and for each of avia / NTU_VIRAL / HILTI22 / MARS_LVIG the oracle runs the LIO update and the VIO update with that configuration's knobs together and converges
towards the true pose.  (The reference itself is unbuildable in this image — SURVEY.md §8c — so there is no reference build to compare with: parity unpinned.)"""
import os

import numpy as np
import pytest

from scenarios import shipped_configs as SC
from scenarios import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(SC.PROFILES)


def test_profiles_quote_the_yaml_values():
    p = SC.PROFILES
    assert p["NTU_VIRAL"]["vio"]["patch_pyrimid_level"] == 3 and p["NTU_VIRAL"]["lio"]["beam_err"] == 0.01
    assert p["HILTI22"]["lio"]["voxel_size"] == 0.4 and p["HILTI22"]["lio"]["min_eigen_value"] == 1e-4 and p["HILTI22"]["lio"]["max_points_num"] == 100
    assert p["HILTI22"]["vio"]["img_point_cov"] == 1000 and p["HILTI22"]["camera"]["model"] == "equidistant" and not np.allclose(p["HILTI22"]["extrinsic_R"], np.eye(3))
    assert np.isclose(np.linalg.det(p["HILTI22"]["extrinsic_R"]), 1.0)
    assert p["MARS_LVIG"]["lio"]["voxel_size"] == 2.0 and p["MARS_LVIG"]["lio"]["min_eigen_value"] == 0.005 and p["MARS_LVIG"]["camera"]["cam"]["width"] == 612
    assert p["avia"]["camera"]["cam"] == synth.AVIA["cam"]


@pytest.mark.parametrize("profile", NAMES)
def test_oracle_runs_every_profile(orc, profile):
    s = SC.lio_scene(profile, seed=700 + NAMES.index(profile), n_map=20000, n_scan=2500)
    c = s["cfg"]
    om = orc.OracleMap.build(s["pw0"], s["var0"], c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    fm = om.export(c["voxel_size"], c["max_layer"])
    assert fm.n_planes > (40 if c["voxel_size"] > 1.0 else 300)
    sc = synth.LidarScenario(fm, s["xyz"], s["R_true"], s["t_true"], s["R_prior"], s["t_prior"], s["P"], s["extR"], s["extT"], c)
    a = orc.lidar_state_estimation(orc.OracleMap.from_flat(fm), orc.lidar_cfg(c, s["extR"], s["extT"]), sc.xyz, *H.states(sc, orc.StatePOD))
    assert a["n_iters"] >= 2 and a["trace"][0].n_eff > 200
    assert np.linalg.norm(np.array(a["state"].pos) - s["t_true"]) < np.linalg.norm(s["t_prior"] - s["t_true"])
    vs = SC.visual_scene(profile, seed=800 + NAMES.index(profile), n_patches=150)
    va = orc.visual_update(orc.visual_cfg(vs, num_threads=1, **SC.cam_kw(profile)), vs, *H.states(vs, orc.StatePOD))
    assert {t.level for t in va["trace"]} == set(range(SC.PROFILES[profile]["vio"]["patch_pyrimid_level"]))
    assert np.linalg.norm(np.array(va["state"].pos) - vs.t_true) < np.linalg.norm(vs.t_prior - vs.t_true)
