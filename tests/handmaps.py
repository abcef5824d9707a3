"""Hand-crafted VoxelMap snapshots for known-answer / edge-case tests (both the oracle on CPU and the HIP path on the GPU use them)."""
import numpy as np

from scenarios.synth import FlatMap

VS = 0.5


def plane_record(normal, center, radius=1.0, var_scale=1e-6):
    n = np.asarray(normal, float); n = n / np.linalg.norm(n)
    c = np.asarray(center, float)
    pv = np.eye(6) * var_scale
    return dict(normal=n, center=c, var=pv.ravel(), d=np.float32(-(n @ c)), radius=np.float32(radius))


class MapBuilder:
    def __init__(self, voxel_size=VS, max_layer=2):
        self.vs, self.max_layer = voxel_size, max_layer
        self.roots, self.nodes, self.planes = [], [], []

    def _node(self, plane=None):
        pi = -1
        if plane is not None:
            pi = len(self.planes); self.planes.append(plane)
        self.nodes.append(dict(plane=pi, child=[-1] * 8))
        return len(self.nodes) - 1

    def add_root(self, key, plane=None, children=None):
        """children: {leaf_index: plane | {leaf_index: plane}} (two levels at most)"""
        r = self._node(plane)
        vsf = np.float64(np.float32(self.vs))
        self.roots.append(dict(key=np.array(key, np.int64), node=r, center=(0.5 + np.array(key, float)) * vsf, quarter=np.float32(self.vs) / np.float32(4)))
        for li, sub in (children or {}).items():
            if isinstance(sub, dict) and "normal" not in sub:
                c = self._node(None)
                for lj, pl in sub.items():
                    self.nodes[c]["child"][lj] = self._node(pl)
            else:
                c = self._node(sub)
            self.nodes[r]["child"][li] = c
        return r

    def build(self):
        R, P = len(self.roots), len(self.planes)
        return FlatMap(
            self.vs, self.max_layer, np.array([r["key"] for r in self.roots], np.int64).reshape(R, 3), np.array([r["node"] for r in self.roots], np.int32),
            np.array([r["center"] for r in self.roots], float).reshape(R, 3), np.array([r["quarter"] for r in self.roots], np.float32),
            np.array([n["plane"] for n in self.nodes], np.int32), np.array([n["child"] for n in self.nodes], np.int32).reshape(-1, 8),
            np.array([p["normal"] for p in self.planes], float).reshape(P, 3), np.array([p["center"] for p in self.planes], float).reshape(P, 3),
            np.array([p["var"] for p in self.planes], float).reshape(P, 36), np.array([p["d"] for p in self.planes], np.float32),
            np.array([p["radius"] for p in self.planes], np.float32))


class HandScene:
    """identity pose / identity extrinsics so that body points == world points (up to float32)"""
    def __init__(self, fmap, xyz, sigma_num=3.0, max_layer=2):
        self.fmap = fmap
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.R_prior = self.R_true = np.eye(3)
        self.t_prior = self.t_true = np.zeros(3)
        P = np.eye(19) * 1e-6
        self.P = P
        self.extR, self.extT = np.eye(3), np.zeros(3)
        self.cfg = dict(max_iterations=3, dept_err=0.02, beam_err=0.05, voxel_size=fmap.voxel_size, max_layer=max_layer, sigma_num=sigma_num)
