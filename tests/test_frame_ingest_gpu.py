"""Head and tail of a frame as single launches (fast-livo2_amd/csrc/frame_kernels.hpp, round 5): the one-block scan preparation (Morton keys, LDS radix sort, gather,
calcBodyCov — voxel_map.cpp:15-34, 349-360) against the launch sequence it replaces (k_morton_keys -> rocPRIM radix sort -> k_gather_xyz -> k_body_cov), the
single-launch frame input (pinned block read by the kernel / one H2D into an arena) against one copy command per array, and the result slot written by a kernel
against three D2H copies.  Only the NUMBER of commands on the stream may change: every comparison here is bit for bit, and the oracle comparison of the same paths
is tests/test_c5_gpu.py / tests/test_lidar_gpu.py (which run with the defaults, i.e. through the new launches)."""
import importlib

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


class _Sc:
    def __init__(self, cfg, extR, extT):
        self.cfg, self.extR, self.extT = cfg, extR, extT


_SC = []


def _scenario():
    if not _SC:
        _SC.append(synth.lidar_scenario(seed=41, n_points=41000, downsample=0.0, map_rays_factor=4))
        assert len(_SC[0].xyz) == 41000
    return _SC[0]


def _lidar_bits(ctx, livo2, sc, xyz, pcfg, want=("match_plane", "dis_to_plane", "body_cov", "h_row")):
    pcur, pprop = H.states(sc, livo2.State)
    ctx.set_scan(xyz, pcfg)
    res, out = ctx.lidar_update(pcur, pprop, pcfg, want=want)
    return bytes(res), {k: np.array(v).tobytes() for k, v in out.items()}


# sizes around the block-sort tiers (1024 threads x 4 / 8 / 16 items), a partial last tier, one scan above the one-block limit, tiny scans
@pytest.mark.parametrize("n", [1, 63, 1000, 4096, 4097, 8192, 8200, 12345, 16384, 16385, 40000])
def test_one_block_scan_preparation_equals_the_launch_sequence(livo2, n):
    sc = _scenario()
    xyz = np.ascontiguousarray(sc.xyz[:n])
    assert len(xyz) == n
    pcfg = H.lidar_cfg_product(sc)
    ctx = livo2.Context(0)
    try:
        ctx.upload_map(sc.fmap)
        got = {}
        for fused, ingest in ((0, 0), (1, 0), (0, 2), (1, 2), (1, 1)):
            ctx.set_option("scan_small_fused", fused); ctx.set_option("frame_ingest", ingest)
            before = ctx.counter("scan_small_launches"), ctx.counter("frame_zero_copy_launches")
            got[(fused, ingest)] = _lidar_bits(ctx, livo2, sc, xyz, pcfg)
            after = ctx.counter("scan_small_launches"), ctx.counter("frame_zero_copy_launches")
            assert after[0] - before[0] == (1 if (fused and n <= 16384) else 0), (fused, ingest, before, after)       # the path under test is the one that ran
            assert after[1] - before[1] == (1 if ingest == 2 else 0), (fused, ingest, before, after)
        base = got[(0, 0)]
        for key, val in got.items():
            assert val[0] == base[0], ("result block differs", n, key)
            for name in base[1]:
                assert val[1][name] == base[1][name], (name, n, key)
    finally:
        ctx.close()


def test_duplicate_cells_keep_the_stable_order(livo2):
    """many points per Morton cell (ties): the block sort must keep equal keys in scan order exactly like the library's stable sort — the partial sums (and so the bits of
    HtH) depend on it"""
    sc = _scenario()
    rng = np.random.default_rng(5)
    base = sc.xyz[rng.integers(0, 40, size=9000)]                  # 40 distinct points, each ~225 times
    xyz = np.ascontiguousarray(base + rng.normal(0, 1e-4, base.shape).astype(np.float32))
    pcfg = H.lidar_cfg_product(sc)
    ctx = livo2.Context(0)
    try:
        ctx.upload_map(sc.fmap)
        ctx.set_option("scan_small_fused", 0); ctx.set_option("frame_ingest", 0)
        a = _lidar_bits(ctx, livo2, sc, xyz, pcfg)
        ctx.set_option("scan_small_fused", 1); ctx.set_option("frame_ingest", 2)
        b = _lidar_bits(ctx, livo2, sc, xyz, pcfg)
        assert a == b
    finally:
        ctx.close()


def test_frame_api_every_input_and_output_form_gives_the_same_records(livo2):
    frames_mod = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(5)
    cfg = cfgs.lidar_cfg(_Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    ctx = livo2.Context(0)
    try:
        ctx.upload_map(fmap)
        recs = {}
        for fused, ingest, publish in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 2, 0), (1, 2, 1), (0, 2, 1), (0, 1, 1)):
            ctx.set_option("scan_small_fused", fused); ctx.set_option("frame_ingest", ingest); ctx.set_option("frame_publish", publish)
            c0 = {k: ctx.counter(k) for k in ("scan_small_launches", "frame_ingest_launches", "frame_zero_copy_launches", "frame_publish_launches", "visual_persistent_timeouts")}
            recs[(fused, ingest, publish)], _ = frames_mod.run_frames_pipelined(ctx, livo2.State, seq + seq[::-1], cfg, vcfg)
            c1 = {k: ctx.counter(k) for k in c0}
            nfr = 2 * len(seq)
            assert c1["frame_publish_launches"] - c0["frame_publish_launches"] == (nfr if publish else 0)
            assert c1["frame_ingest_launches"] - c0["frame_ingest_launches"] == (nfr if ingest else 0)
            assert c1["frame_zero_copy_launches"] - c0["frame_zero_copy_launches"] == (nfr if ingest == 2 else 0)
            assert c1["scan_small_launches"] - c0["scan_small_launches"] == (nfr if fused else 0)
            assert c1["visual_persistent_timeouts"] == c0["visual_persistent_timeouts"]
        base = recs[(0, 0, 0)]
        one, _ = frames_mod.run_frames_sharded(ctx, livo2.State, seq + seq[::-1], cfg, vcfg, 0, 1)          # the four-call sequence (set_scan, lidar_update, set_frame, visual_update)
        assert np.array_equal(base, one)
        for key, r in recs.items():
            assert np.array_equal(r, base), key
    finally:
        ctx.close()


def test_frame_larger_than_the_zero_copy_limit_goes_through_the_arena(livo2):
    """a frame whose payload exceeds FRAME_ZERO_COPY_MAX (1 MiB) is staged by ONE H2D copy into the arena even with frame_ingest = 2; a scan above 16 384 points keeps
    the library sort — same records as the copy-per-array form"""
    frames_mod = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(2, n_raw=160000, n_patches=1500, max_points=60000, map_rays_factor=3)
    assert all(len(f["xyz"]) > 16384 for f in seq)
    cfg = cfgs.lidar_cfg(_Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    ctx = livo2.Context(0)
    try:
        ctx.upload_map(fmap)
        ctx.set_option("frame_ingest", 0); ctx.set_option("frame_publish", 0)
        a, _ = frames_mod.run_frames_pipelined(ctx, livo2.State, seq, cfg, vcfg)
        ctx.set_option("frame_ingest", 2); ctx.set_option("frame_publish", 1)
        z0, s0, i0 = ctx.counter("frame_zero_copy_launches"), ctx.counter("scan_small_launches"), ctx.counter("frame_ingest_launches")
        b, _ = frames_mod.run_frames_pipelined(ctx, livo2.State, seq, cfg, vcfg)
        assert ctx.counter("frame_zero_copy_launches") == z0 and ctx.counter("scan_small_launches") == s0 and ctx.counter("frame_ingest_launches") == i0 + len(seq)
        assert np.array_equal(a, b)
    finally:
        ctx.close()
