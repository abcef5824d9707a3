#!/usr/bin/env python
"""Launch-by-launch durations of k_lidar_solve from a rocprofv3 rocpd sqlite of `tools/lidar_ab.py` (is the bimodal solve of the slower kind of box tied to a place in
the frame?): for every k_lidar_solve launch its position in its run of (residual, solve) pairs, its duration, the gap to the kernel before it, and that kernel's name.
Usage (GPU box): rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python tools/lidar_ab.py --rounds 1 --variants order=1 ; python tools/solve_sequence.py $(find /tmp/kt -name '*results.db')"""
import sqlite3
import sys
import collections


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    pos, out = 0, []
    by_pos = collections.defaultdict(list)
    res_pos = collections.defaultdict(list)
    prev = None
    for name, s, e in rows:
        short = "visual" if "visual_update_persistent" in name else ("residual" if "k_lidar_residual" in name else ("solve" if "k_lidar_solve" in name else name[:24]))
        if short == "visual":
            pos = 0
        if short == "residual":
            res_pos[pos % 5 + 1].append((e - s) / 1e3)
        if short == "solve":
            pos = pos % 5 + 1
            gap = (s - prev[2]) / 1e3 if prev else 0.0
            by_pos[pos].append((e - s) / 1e3)
            out.append((pos, (e - s) / 1e3, gap, prev[0] if prev else ""))
        prev = (short, s, e)
    print("# k_lidar_solve by position after the visual update (1 = first LiDAR iteration of a frame): n, mean, p10, p50, p90 (us)")
    for p in sorted(by_pos):
        v = sorted(by_pos[p]); n = len(v)
        print(f"  pos {p}: n {n}  mean {sum(v) / n:6.2f}  p10 {v[n // 10]:6.2f}  p50 {v[n // 2]:6.2f}  p90 {v[(9 * n) // 10]:6.2f}")
    print("# k_lidar_residual by the same position: " + "  ".join(f"pos {p}: mean {sum(v) / len(v):.2f} p50 {sorted(v)[len(v) // 2]:.2f}" for p, v in sorted(res_pos.items())))
    print("# a stretch of 60 consecutive solves: position, duration, gap before (us), kernel before")
    mid = len(out) // 2
    print("  " + "  ".join(f"{p}:{d:.1f}({g:.1f})" for p, d, g, _ in out[mid:mid + 30]))
    res = [(e - s) / 1e3 for name, s, e in rows if "k_lidar_residual" in name]
    vis = [(e - s) / 1e3 for name, s, e in rows if "visual_update_persistent" in name]
    if res: print(f"# k_lidar_residual mean {sum(res) / len(res):.2f} us over {len(res)}; visual mean {sum(vis) / max(len(vis), 1):.2f} us over {len(vis)}")


if __name__ == "__main__":
    main(sys.argv[1])
