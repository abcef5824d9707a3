"""Synthetic voxel point groups for the plane-fit tests: noisy planar patches, blobs, degenerate and large groups."""
import numpy as np


def random_spd(rng, n, scale=1e-4):
    A = rng.normal(size=(n, 3, 3))
    return scale * (A @ A.transpose(0, 2, 1) + 0.1 * np.eye(3))


def make_groups(seed=0, n_groups=300, big=(700, 5000)):
    rng = np.random.default_rng(seed)
    pts, var, off = [], [], [0]
    kinds = []
    for g in range(n_groups):
        kind = ("plane", "plane", "plane", "thick", "blob", "line")[g % 6]
        n = int(rng.integers(6, 60))
        if g < len(big):
            n, kind = big[g], "plane"
        centre = rng.uniform(-40, 40, 3)
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        ext = {"plane": (0.15, 0.12, 0.01), "thick": (0.15, 0.12, 0.055), "blob": (0.12, 0.11, 0.1), "line": (0.2, 0.004, 0.003)}[kind]
        p = (rng.normal(size=(n, 3)) * np.array(ext)) @ Q.T + centre
        p = p.astype(np.float32).astype(np.float64)          # point_w comes from a float32 cloud (voxel_map.cpp:524-526)
        pts.append(p); var.append(random_spd(rng, n)); off.append(off[-1] + n); kinds.append(kind)
    return np.concatenate(pts), np.concatenate(var).reshape(-1, 9), np.array(off, np.int32), kinds
