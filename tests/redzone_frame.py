"""Run under LIVO2_REDZONE=<mode> (tests/test_redzone_gpu.py starts it as a subprocess: the mode is read once per process).
mode 1: the smoke sequence (LiDAR update, visual update as one resident grid and per step, batches, plane fits, retrieval chain, device-resident map build +
        update) + a 3-context C5 pass with every guard of every device allocation checked; then the checker is shown to work: a store 0 / 300 bytes behind and
        4 bytes in front of the control block must be reported with the allocation's source line.
With LIVO2_POISON set as well every new allocation is filled with that byte first: the results (checked against the oracle by smoke(), and 1 context == 3 contexts)
must not change.  (mode 2 / 3 run the same sequence on hipMemMap'ed allocations; not part of the suite, see tests/test_redzone_gpu.py.)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = int(os.environ.get("LIVO2_REDZONE", "0"))
    assert mode in (1, 2, 3)
    import __graft_entry__ as G
    G.smoke()
    livo2 = importlib.import_module("fast-livo2_amd")
    frames = importlib.import_module("fast-livo2_amd.frames")
    cfgs = importlib.import_module("fast-livo2_amd.configs")
    from scenarios import synth
    import bench
    fmap, lio_cfg, extR, extT, seq = synth.frame_sequence(6)
    cfg = cfgs.lidar_cfg(bench._Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(seq[0]["vs"], mp_proc_num=4)
    ctxs = [livo2.Context(0) for _ in range(3)]
    for c in ctxs:
        c.upload_map(fmap)
    recs1, _ = frames.run_frames_sharded(ctxs[0], livo2.State, seq, cfg, vcfg, 0, 1)
    recs3, _ = frames.run_frames_sharded(ctxs, livo2.State, seq, cfg, vcfg, 0, 1)
    assert (recs1 == recs3).all()
    for c in ctxs:
        c.synchronize()                                   # mode 1: scans all guards
    m, bad = ctxs[0].redzone_check()
    assert m == mode and bad == 0, (m, bad)
    # small scans whose size sits at the end of the scan buffers' capacity (advisor, round 5): the small-scan path pads the key array to a multiple of 16 words and
    # reads it with 16-byte loads — a first scan of 10 801 points used to give a capacity of 16 201, and scans of 16 193..16 201 points then touched up to 7 words past it
    c = livo2.Context(0)
    c.upload_map(fmap)
    rng = __import__("numpy").random.default_rng(3)
    for n in (10801, 16193, 16199, 16201, 16208):
        xyz = (rng.uniform(-8, 8, (n, 3))).astype("float32")
        c.set_scan(xyz, cfg)
        c.synchronize()
        assert c.redzone_check()[1] == 0, n
    c.close()
    print("REDZONE mode %d: smoke + 6 frames on 1 and 3 contexts + scans at the capacity edge clean" % mode, flush=True)
    if mode == 1:
        c = ctxs[0]
        for off, words in ((0, "BEHIND its end"), (300, "BEHIND its end"), (-4, "IN FRONT of its start")):
            assert c.lib.livo2_debug_redzone_poke(c.h, off) == 0
            try:
                c.redzone_check()
                raise SystemExit("redzone checker missed a store at offset %d" % off)
            except livo2.Livo2Error as exc:
                msg = str(exc)
                assert words in msg and "livo2_api.hip:" in msg, msg
                print("REDZONE poke %+d reported: %s" % (off, msg), flush=True)
            break                                             # (one damaged guard stays damaged: the later offsets would be reported as the first)
    for c in ctxs:
        c.close()
    print("REDZONE DONE", flush=True)


if __name__ == "__main__":
    main()
