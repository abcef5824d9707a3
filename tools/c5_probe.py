"""bench.py's C5 leg alone (torch only for pinned buffers): frames/s of the C1- and C4-shaped frames through the four-call path and through livo2_frame_update.
python tools/c5_probe.py [c1|c4] [frames]   (tools/frame_probe.py: the host / device split of one frame)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    shape = sys.argv[1] if len(sys.argv) > 1 else "c1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if shape == "c1" else 12)
    livo2 = importlib.import_module("fast-livo2_amd")
    H = importlib.import_module("fast-livo2_amd.configs")
    frames = importlib.import_module("fast-livo2_amd.frames")
    ctx = livo2.Context(0)

    def barrier():
        ctx.synchronize(); torch.cuda.synchronize()
    r = bench.c5_leg(ctx, livo2, frames, H, None, "cuda", 0, 1, min(n, 64 if shape == "c1" else 32), n, shape, barrier, torch)
    keep = ("shape", "frames", "frames_per_s", "frames_per_s_one_context", "frames_per_s_frame_api", "frames_per_s_frame_api_one_context", "points_per_frame_mean", "patches_per_frame", "gathered_copy_check")
    print(json.dumps({k: r[k] for k in keep}))
    ctx.close()


if __name__ == "__main__":
    main()
