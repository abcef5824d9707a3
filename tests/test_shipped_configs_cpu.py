"""CPU half of the shipped-configuration matrix (tests/test_shipped_configs_gpu.py): the scenario builders of scenarios/shipped_configs.py run, and for each of
avia / NTU_VIRAL / HILTI22 / MARS_LVIG the oracle equals the reference's OWN translation units (oracle/_ref/libref.so: voxel_map.cpp, vio.cpp compiled unmodified)
on the LIO update and the VIO update with that configuration's knobs together — the pin the GPU comparison against the oracle rests on."""
import os

import numpy as np
import pytest

from scenarios import shipped_configs as SC
from scenarios import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(SC.PROFILES)


@pytest.fixture(scope="module")
def refbuild(orc):
    path = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref.so was not built")
    return orc.load(path)


def test_profiles_quote_the_yaml_values():
    p = SC.PROFILES
    assert p["NTU_VIRAL"]["vio"]["patch_pyrimid_level"] == 3 and p["NTU_VIRAL"]["lio"]["beam_err"] == 0.01
    assert p["HILTI22"]["lio"]["voxel_size"] == 0.4 and p["HILTI22"]["lio"]["min_eigen_value"] == 1e-4 and p["HILTI22"]["lio"]["max_points_num"] == 100
    assert p["HILTI22"]["vio"]["img_point_cov"] == 1000 and p["HILTI22"]["camera"]["model"] == "equidistant" and not np.allclose(p["HILTI22"]["extrinsic_R"], np.eye(3))
    assert np.isclose(np.linalg.det(p["HILTI22"]["extrinsic_R"]), 1.0)
    assert p["MARS_LVIG"]["lio"]["voxel_size"] == 2.0 and p["MARS_LVIG"]["lio"]["min_eigen_value"] == 0.005 and p["MARS_LVIG"]["camera"]["cam"]["width"] == 612
    assert p["avia"]["camera"]["cam"] == synth.AVIA["cam"]


@pytest.mark.parametrize("profile", NAMES)
def test_oracle_equals_the_reference_build(orc, refbuild, profile):
    s = SC.lio_scene(profile, seed=700 + NAMES.index(profile), n_map=20000, n_scan=2500)
    c = s["cfg"]
    om = orc.OracleMap.build(s["pw0"], s["var0"], c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    fm = om.export(c["voxel_size"], c["max_layer"])
    assert fm.n_planes > (40 if c["voxel_size"] > 1.0 else 300)
    sc = synth.LidarScenario(fm, s["xyz"], s["R_true"], s["t_true"], s["R_prior"], s["t_prior"], s["P"], s["extR"], s["extT"], c)
    ocfg = orc.lidar_cfg(c, s["extR"], s["extT"])
    a = orc.lidar_state_estimation(orc.OracleMap.from_flat(fm), ocfg, sc.xyz, *H.states(sc, orc.StatePOD))
    b = orc.lidar_state_estimation(orc.OracleMap.from_flat(fm, refbuild), ocfg, sc.xyz, *H.states(sc, orc.StatePOD))
    assert a["n_iters"] == b["n_iters"] >= 2 and [t.n_eff for t in a["trace"]] == [t.n_eff for t in b["trace"]]
    assert np.array_equal(a["match_plane"], b["match_plane"]) and np.array_equal(a["dis"], b["dis"]) and np.array_equal(a["pw"], b["pw"])
    d = H.state_diff(a["state"], b["state"])
    assert d["R"] < 1e-12 and d["t"] < 1e-12 and d["P"] < 1e-10, d
    vs = SC.visual_scene(profile, seed=800 + NAMES.index(profile), n_patches=150)
    kw = SC.cam_kw(profile)
    va = orc.visual_update(orc.visual_cfg(vs, num_threads=1, **kw), vs, *H.states(vs, orc.StatePOD))
    vb = orc.visual_update(orc.visual_cfg(vs, num_threads=1, **kw), vs, *H.states(vs, orc.StatePOD), lib=refbuild)
    # (unmodified reference code exposes no per-step trace: the per-patch float errors of the last evaluated step, the final state / covariance and G pin the whole run)
    assert {t.level for t in va["trace"]} == set(range(SC.PROFILES[profile]["vio"]["patch_pyrimid_level"]))
    assert np.array_equal(va["errors"], vb["errors"])
    d = H.state_diff(va["state"], vb["state"])
    assert d["R"] < 1e-12 and d["t"] < 1e-12 and d["P"] < 1e-10, d
    assert H.relerr(va["G"], vb["G"]) < 1e-10
