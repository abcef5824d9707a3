"""Seeded synthetic inputs for the ESIKF measurement-update path: piecewise-planar rooms, LiDAR scans with
range/bearing noise, a VoxelMap snapshot built by a numpy restatement of BuildVoxelMap/init_plane, gray images and visual
sub-maps.  Pure numpy/scipy — this package contains neither the oracle nor the product; both are fed from it.

Flat map = index-based mirror of `unordered_map<VOXEL_LOCATION, VoxelOctoTree*>` (reference include/voxel_map.h:194), the
interchange format of include/livo2_hip.h (`livo2_map_view`) and of oracle/orc_api.cpp.
"""
from dataclasses import dataclass, field

import numpy as np

# ---- configuration values of the reference's config/avia.yaml ---------------------------------------------------------
AVIA = dict(
    extrinsic_T=np.array([0.04165, 0.02326, -0.0284]), extrinsic_R=np.eye(3),
    Rcl=np.array([[0.00610193, -0.999863, -0.0154172], [-0.00615449, 0.0153796, -0.999863], [0.999962, 0.00619598, -0.0060598]]),
    Pcl=np.array([0.0194384, 0.104689, -0.0251952]),
    lio=dict(max_iterations=5, dept_err=0.02, beam_err=0.05, min_eigen_value=0.0025, voxel_size=0.5, max_layer=2, max_points_num=50,
             layer_init_num=[5, 5, 5, 5, 5], sigma_num=3.0),
    vio=dict(max_iterations=5, img_point_cov=100.0, patch_size=8, patch_pyrimid_level=4, exposure_estimate_en=True),
    # config/camera_pinhole.yaml values x scale 0.5 (vio.cpp:45-54), distortion dropped for the synthetic benchmark
    cam=dict(fx=1293.56944 * 0.5, fy=1293.3155 * 0.5, cx=626.91359 * 0.5, cy=522.799224 * 0.5, width=640, height=512),
    filter_size_surf=0.1, blind=0.8,
)
PCL_DEG2RAD = 0.017453293     # pcl/pcl_macros.h DEG2RAD factor


@dataclass
class FlatMap:
    voxel_size: float
    max_layer: int
    root_key: np.ndarray      # int64 [R,3]
    root_node: np.ndarray     # int32 [R]
    root_center: np.ndarray   # f64 [R,3]
    root_quarter: np.ndarray  # f32 [R]
    node_plane: np.ndarray    # int32 [Nn]
    node_child: np.ndarray    # int32 [Nn,8]
    plane_normal: np.ndarray  # f64 [P,3]
    plane_center: np.ndarray  # f64 [P,3]
    plane_var: np.ndarray     # f64 [P,36]
    plane_d: np.ndarray       # f32 [P]
    plane_radius: np.ndarray  # f32 [P]

    @property
    def n_planes(self):
        return len(self.plane_d)


def rot_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def so3_exp(v):
    v = np.asarray(v, float)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


# ---- scene ---------------------------------------------------------------------------------------------------------------
@dataclass
class Scene:
    """Axis-aligned rectangles in a scene frame, mapped to the world by T_ws = (R_ws, t_ws) so that no plane is aligned with
    the voxel grid."""
    faces: list = field(default_factory=list)     # (axis, coord, lo(2), hi(2)) with the two other axes in increasing order
    R_ws: np.ndarray = field(default_factory=lambda: np.eye(3))
    t_ws: np.ndarray = field(default_factory=lambda: np.zeros(3))


def make_room(rng, size=(20.0, 20.0, 6.0), n_boxes=8):
    sx, sy, sz = size
    lo = np.array([-sx / 2 + 0.137, -sy / 2 + 0.211, -0.173])
    hi = lo + np.array(size)
    sc = Scene(R_ws=rot_from_rpy(0.03, -0.02, 0.31), t_ws=np.array([0.4, -0.3, 0.2]))

    def add_box(blo, bhi):
        for ax in range(3):
            o = [a for a in range(3) if a != ax]
            for c in (blo[ax], bhi[ax]):
                sc.faces.append((ax, float(c), np.array([blo[o[0]], blo[o[1]]]), np.array([bhi[o[0]], bhi[o[1]]])))

    add_box(lo, hi)
    for _ in range(n_boxes):
        w = rng.uniform(0.8, 0.15 * min(sx, sy), 3)
        w[2] = rng.uniform(0.5, 0.5 * sz)
        c = np.array([rng.uniform(lo[0] + 2, hi[0] - 2), rng.uniform(lo[1] + 2, hi[1] - 2), lo[2]])
        if np.hypot(c[0], c[1]) < 3.0:       # keep the sensor's neighbourhood free
            c[:2] += 3.0 * np.sign(c[:2] + 1e-9)
        add_box(np.array([c[0] - w[0] / 2, c[1] - w[1] / 2, lo[2]]), np.array([c[0] + w[0] / 2, c[1] + w[1] / 2, lo[2] + w[2]]))
    return sc


def cast(scene, origin_w, dirs_w):
    """First-hit distance of world rays against the scene (inf when nothing is hit)."""
    o = scene.R_ws.T @ (origin_w - scene.t_ws)
    d = dirs_w @ scene.R_ws            # rows: R_ws^T d
    best = np.full(len(d), np.inf)
    for ax, c, lo, hi in scene.faces:
        oth = [a for a in range(3) if a != ax]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (c - o[ax]) / d[:, ax]
        h0 = o[oth[0]] + t * d[:, oth[0]]
        h1 = o[oth[1]] + t * d[:, oth[1]]
        ok = (t > 1e-6) & (h0 >= lo[0]) & (h0 <= hi[0]) & (h1 >= lo[1]) & (h1 <= hi[1]) & (t < best)
        best = np.where(ok, t, best)
    return best


def sample_dirs(rng, n, fov_h_deg=70.4, fov_v_deg=77.2, full_sphere=False):
    """Unit directions in the LiDAR frame (+x forward).  Avia FoV by default."""
    if full_sphere:
        v = rng.normal(size=(n, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    az = np.deg2rad(rng.uniform(-fov_h_deg / 2, fov_h_deg / 2, n))
    el = np.deg2rad(rng.uniform(-fov_v_deg / 2, fov_v_deg / 2, n))
    return np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)


def boresight_towards(axis_l):
    """rotation B (LiDAR frame) with B e_x = axis_l: turns the +x-forward field of view of sample_dirs towards `axis_l` (a spinning LiDAR sees all around; the
    synthetic scan covers the part of it a camera mounted along `axis_l` looks at)"""
    a = np.asarray(axis_l, np.float64); a = a / np.linalg.norm(a)
    h = np.array([0.0, 0.0, 1.0]) if abs(a[2]) < 0.9 else np.array([0.0, 1.0, 0.0])
    b = np.cross(h, a); b /= np.linalg.norm(b)
    return np.stack([a, b, np.cross(a, b)], 1)


def lidar_scan(rng, scene, R_wi, t_wi, extR, extT, n_rays, dept_err, beam_err_deg, blind=0.8, full_sphere=False, max_range=80.0, boresight=None):
    """Noisy scan in the LiDAR body frame (float32 xyz), rays cast from the true pose."""
    d_l = sample_dirs(rng, n_rays, full_sphere=full_sphere)
    if boresight is not None:
        d_l = d_l @ np.asarray(boresight, np.float64).T
    o_w = R_wi @ extT + t_wi
    d_w = d_l @ (R_wi @ extR).T
    r = cast(scene, o_w, d_w)
    keep = np.isfinite(r) & (r > blind) & (r < max_range)
    d_l, r = d_l[keep], r[keep]
    r = r + rng.normal(0, dept_err, len(r))
    # bearing noise: rotate by a small random vector orthogonal-ish to the ray
    w = rng.normal(0, np.deg2rad(beam_err_deg), (len(r), 3))
    d_n = d_l + np.cross(w, d_l)
    d_n /= np.linalg.norm(d_n, axis=1, keepdims=True)
    return (d_n * r[:, None]).astype(np.float32)


def voxel_grid_downsample(xyz, leaf):
    """Centroid voxel-grid filter in the spirit of pcl::VoxelGrid (reference src/LIVMapper.cpp:351-352): one centroid per occupied
    leaf, output ordered by leaf index (z-major, then y, then x) like PCL's sorted-index pass."""
    xyz = np.asarray(xyz, np.float32)
    ijk = np.floor(xyz / np.float32(leaf)).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    idx = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    order = np.argsort(idx, kind="stable")
    idx_s = idx[order]
    first = np.concatenate([[0], np.nonzero(np.diff(idx_s))[0] + 1])
    counts = np.diff(np.concatenate([first, [len(idx_s)]]))
    sums = np.add.reduceat(xyz[order].astype(np.float64), first, axis=0)
    return (sums / counts[:, None]).astype(np.float32)


# ---- numpy restatement of calcBodyCov (reference src/voxel_map.cpp:15-34), vectorised ----------------------------------
def body_cov(p_l, dept_err, beam_err, deg2rad=PCL_DEG2RAD):
    p = np.array(p_l, np.float64, copy=True)
    p[p[:, 2] == 0, 2] = 0.001
    rng_f = np.sqrt((p * p).sum(1)).astype(np.float32).astype(np.float64)
    range_var = np.float64(np.float32(dept_err) * np.float32(dept_err))
    dv = np.sin(np.float64(np.float32(beam_err)) * deg2rad) ** 2
    d = p / np.linalg.norm(p, axis=1, keepdims=True)
    b1 = np.stack([np.ones(len(p)), np.ones(len(p)), -(d[:, 0] + d[:, 1]) / d[:, 2]], 1)
    b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    b2 = np.cross(b1, d)
    b2 /= np.linalg.norm(b2, axis=1, keepdims=True)
    hat = np.zeros((len(p), 3, 3))
    hat[:, 0, 1], hat[:, 0, 2], hat[:, 1, 0], hat[:, 1, 2], hat[:, 2, 0], hat[:, 2, 1] = -d[:, 2], d[:, 1], d[:, 2], -d[:, 0], -d[:, 1], d[:, 0]
    N = np.stack([b1, b2], 2)
    A = rng_f[:, None, None] * hat @ N
    return range_var * d[:, :, None] * d[:, None, :] + dv * A @ A.transpose(0, 2, 1)


def skew(v):
    v = np.atleast_2d(v)
    S = np.zeros((len(v), 3, 3))
    S[:, 0, 1], S[:, 0, 2], S[:, 1, 0], S[:, 1, 2], S[:, 2, 0], S[:, 2, 1] = -v[:, 2], v[:, 1], v[:, 2], -v[:, 0], -v[:, 1], v[:, 0]
    return S


def world_points_and_var(xyz_l, R, t, extR, extT, P, dept_err, beam_err):
    """World points and covariances as BuildVoxelMap forms them (reference src/voxel_map.cpp:542-555, LIVMapper.cpp:358)."""
    pl = np.asarray(xyz_l, np.float64)
    pw = (pl @ extR.T + extT) @ R.T + t
    pw = pw.astype(np.float32).astype(np.float64)          # feats_down_world_ is a float32 cloud
    cb = body_cov(pl, dept_err, beam_err)
    RE = R @ extR
    X = skew(pl)                                           # voxel_map.cpp:550 uses the LiDAR-frame point here
    var = RE @ cb @ RE.T + X @ P[0:3, 0:3] @ X.transpose(0, 2, 1) + P[3:6, 3:6]
    return pw, var


# ---- numpy restatement of BuildVoxelMap / init_plane / cut_octo_tree (reference src/voxel_map.cpp:55-217, 532-591) --------
def _fit_planes(pw, var, gid, G):
    """Per-group plane fit: returns (is_plane-independent) centre, eigen-decomposition and plane covariance."""
    cnt = np.bincount(gid, minlength=G).astype(np.float64)
    ctr = np.stack([np.bincount(gid, pw[:, k], G) for k in range(3)], 1) / cnt[:, None]
    outer = pw[:, :, None] * pw[:, None, :]
    cov = np.stack([np.bincount(gid, outer[:, i, j], G) for i in range(3) for j in range(3)], 1).reshape(G, 3, 3) / cnt[:, None, None]
    cov -= ctr[:, :, None] * ctr[:, None, :]
    ev, evec = np.linalg.eigh(cov)                         # ascending: min = 0, mid = 1, max = 2
    vmin = evec[:, :, 0]
    n_pts = cnt[gid]
    F = np.zeros((len(pw), 3, 3))
    dp = pw - ctr[gid]
    for m in (1, 2):
        vm = evec[gid][:, :, m]
        sym = vm[:, :, None] * vmin[gid][:, None, :] + vmin[gid][:, :, None] * vm[:, None, :]
        denom = n_pts * (ev[gid, 0] - ev[gid, m])
        with np.errstate(divide="ignore", invalid="ignore"):
            F[:, m, :] = np.einsum("ni,nij->nj", dp, sym) / denom[:, None]
    J = np.zeros((len(pw), 6, 3))
    J[:, 0:3, :] = evec[gid] @ F
    J[:, 3, 0] = J[:, 4, 1] = J[:, 5, 2] = 1.0 / n_pts
    JVJ = J @ var @ J.transpose(0, 2, 1)
    pv = np.stack([np.bincount(gid, JVJ[:, i, j], G) for i in range(6) for j in range(6)], 1)
    return cnt, ctr, ev, evec, pv


def build_voxel_map(pw, var, voxel_size=0.5, max_layer=2, layer_init_num=(5, 5, 5, 5, 5), planer_threshold=0.0025):
    pw = np.asarray(pw, np.float64)
    vs_f = np.float64(np.float32(voxel_size))              # `float voxel_size` local (voxel_map.cpp:534)
    loc = (pw / vs_f).astype(np.float32)
    loc = np.where(loc < 0, (loc.astype(np.float64) - 1.0).astype(np.float32), loc)
    key = loc.astype(np.int64)                             # truncation toward zero
    ukey, gid = np.unique(key, axis=0, return_inverse=True)
    gid = gid.reshape(-1)
    R = len(ukey)
    root_center = (0.5 + ukey.astype(np.float64)) * vs_f
    root_quarter = np.full(R, np.float32(voxel_size) / np.float32(4), np.float32)

    node_plane, node_child = [-1] * R, [[-1] * 8 for _ in range(R)]
    planes = dict(normal=[], center=[], var=[], d=[], radius=[])
    # active set: point index array, group id per point (0..G-1), per group node id / centre / quarter
    act_pts = np.arange(len(pw))
    act_gid = gid
    act_node = np.arange(R)
    act_ctr, act_q = root_center, root_quarter
    for layer in range(max_layer + 1):
        G = len(act_node)
        if G == 0 or len(act_pts) == 0:
            break
        cnt = np.bincount(act_gid, minlength=G)
        elig = cnt > layer_init_num[layer]
        sel = elig[act_gid]
        sub_pts, sub_gid_old = act_pts[sel], act_gid[sel]
        remap = -np.ones(G, np.int64)
        remap[elig] = np.arange(elig.sum())
        sg = remap[sub_gid_old]
        Ge = int(elig.sum())
        if Ge == 0:
            break
        _, ctr, ev, evec, pvar = _fit_planes(pw[sub_pts], var[sub_pts], sg, Ge)
        is_plane = ev[:, 0] < np.float64(np.float32(planer_threshold))
        e_nodes = act_node[elig]
        for g in np.nonzero(is_plane)[0]:
            n = evec[g][:, 0]
            node_plane[e_nodes[g]] = len(planes["d"])
            planes["normal"].append(n)
            planes["center"].append(ctr[g])
            planes["var"].append(pvar[g])
            planes["d"].append(np.float32(-(n[0] * ctr[g][0] + n[1] * ctr[g][1] + n[2] * ctr[g][2])))
            planes["radius"].append(np.float32(np.sqrt(max(ev[g, 2], 0.0))))
        if layer == max_layer:
            break
        # subdivide non-planar eligible groups (cut_octo_tree)
        np_mask = ~is_plane
        selp = np_mask[sg]
        c_pts, c_g = sub_pts[selp], sg[selp]
        if len(c_pts) == 0:
            break
        e_ctr, e_q = act_ctr[elig], act_q[elig]
        bits = (pw[c_pts] > e_ctr[c_g]).astype(np.int64)
        code = 4 * bits[:, 0] + 2 * bits[:, 1] + bits[:, 2]
        pair = c_g * 8 + code
        upair, new_gid = np.unique(pair, return_inverse=True)
        new_gid = new_gid.reshape(-1)
        par_g, par_code = upair // 8, upair % 8
        new_nodes = np.arange(len(node_plane), len(node_plane) + len(upair))
        node_plane.extend([-1] * len(upair))
        node_child.extend([[-1] * 8 for _ in range(len(upair))])
        for k in range(len(upair)):
            node_child[e_nodes[par_g[k]]][par_code[k]] = int(new_nodes[k])
        xyzb = np.stack([(par_code >> 2) & 1, (par_code >> 1) & 1, par_code & 1], 1)
        q_par = e_q[par_g]
        new_ctr = e_ctr[par_g] + ((2 * xyzb - 1).astype(np.float32) * q_par[:, None]).astype(np.float64)
        new_q = (q_par / np.float32(2)).astype(np.float32)
        act_pts, act_gid, act_node, act_ctr, act_q = c_pts, new_gid, new_nodes, new_ctr, new_q
    P = len(planes["d"])
    return FlatMap(
        voxel_size=float(voxel_size), max_layer=int(max_layer), root_key=ukey.astype(np.int64), root_node=np.arange(R, dtype=np.int32),
        root_center=root_center, root_quarter=root_quarter, node_plane=np.array(node_plane, np.int32),
        node_child=np.array(node_child, np.int32).reshape(-1, 8),
        plane_normal=np.array(planes["normal"], np.float64).reshape(P, 3), plane_center=np.array(planes["center"], np.float64).reshape(P, 3),
        plane_var=np.array(planes["var"], np.float64).reshape(P, 36), plane_d=np.array(planes["d"], np.float32),
        plane_radius=np.array(planes["radius"], np.float32))


# ---- LiDAR scenario ------------------------------------------------------------------------------------------------------------
@dataclass
class LidarScenario:
    fmap: FlatMap
    xyz: np.ndarray            # float32 [N,3] down-sampled body-frame scan (feats_down_body_)
    R_true: np.ndarray
    t_true: np.ndarray
    R_prior: np.ndarray
    t_prior: np.ndarray
    P: np.ndarray              # 19x19 prior covariance
    extR: np.ndarray
    extT: np.ndarray
    cfg: dict


def default_cov():
    """StatesGroup() covariance (reference include/common_lib.h:137-139)."""
    P = np.eye(19) * 0.01
    P[6, 6] = 0.00001
    P[10:19, 10:19] = np.eye(9) * 0.00001
    return P


def prior_cov(rng):
    """A propagated-looking SPD prior: StatesGroup() diagonal scaled down + a small random symmetric coupling."""
    P = default_cov() * 0.02
    A = rng.normal(size=(19, 19)) * 1e-4
    P = P + A @ A.T
    return 0.5 * (P + P.T)


def lidar_scenario(seed=1, n_points=10000, room=(20.0, 20.0, 6.0), n_boxes=8, full_sphere=False, map_rays_factor=12, downsample=None,
                   rot_sigma_deg=0.5, pos_sigma=0.03, cfg=None, extR=None, extT=None, test_rays=None, thin="head"):
    """C1/C2-style scenario: map from a dense first sweep at the true pose, test scan with fresh noise, perturbed prior.
    extR / extT: LiDAR->IMU extrinsics (default: avia.yaml's identity rotation; HILTI22.yaml has a non-identity one, quirk Q6)."""
    rng = np.random.default_rng(seed)
    c = dict(AVIA["lio"])
    if cfg:
        c.update(cfg)
    extR = AVIA["extrinsic_R"].copy() if extR is None else np.asarray(extR, np.float64)
    extT = AVIA["extrinsic_T"].copy() if extT is None else np.asarray(extT, np.float64)
    scene = make_room(rng, room, n_boxes)
    R_true = scene.R_ws @ rot_from_rpy(0.01, -0.015, 0.4)
    t_true = scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = default_cov() * 1e-3
    # map sweep (dense), in the world frame with the true pose
    n_map = int(n_points * map_rays_factor)
    chunks = []
    for _ in range(max(1, n_map // 400000 + 1)):
        chunks.append(lidar_scan(rng, scene, R_true, t_true, extR, extT, min(n_map, 400000), c["dept_err"], c["beam_err"], AVIA["blind"], full_sphere))
    xyz_map = np.concatenate(chunks)[:n_map]
    pw, var = world_points_and_var(xyz_map, R_true, t_true, extR, extT, P0, c["dept_err"], c["beam_err"])
    fmap = build_voxel_map(pw, var, c["voxel_size"], c["max_layer"], c["layer_init_num"], c["min_eigen_value"])
    # test scan
    over = 1.6 if downsample else 1.25
    xyz = lidar_scan(rng, scene, R_true, t_true, extR, extT, int(test_rays) if test_rays else int(n_points * over) + 64, c["dept_err"], c["beam_err"], AVIA["blind"], full_sphere)
    if downsample:
        xyz = voxel_grid_downsample(xyz, downsample)
    if len(xyz) > n_points:
        # "head": the first n_points (of a voxel-grid cloud: the low leaf indices); "random": an order-preserving random subset of exactly n_points
        xyz = xyz[:n_points] if thin == "head" else xyz[np.sort(np.random.default_rng(seed + 7919).permutation(len(xyz))[:n_points])]
    dth = rng.normal(0, np.deg2rad(rot_sigma_deg), 3)
    dp = rng.normal(0, pos_sigma, 3)
    R_prior = R_true @ so3_exp(dth)
    t_prior = t_true + dp
    return LidarScenario(fmap, np.ascontiguousarray(xyz, np.float32), R_true, t_true, R_prior, t_prior, prior_cov(rng), extR, extT, c)


def frame_sequence(n_frames, seed0=1000, n_raw=24000, room=(20.0, 20.0, 6.0), n_boxes=8, full_sphere=False, map_rays_factor=12, downsample=0.1, n_patches=350,
                   max_points=None, map_points=None):
    """F distinct frames against one VoxelMap — frame f (seed seed0 + f) has its own sensor pose (a walk around the map pose), its own noisy scan of
    n_raw rays through the voxel-grid filter, its own perturbed prior, and its own image / visual sub-map.  ONE scene per frame: the image and the sub-map are
    generated at the frame's TRUE sensor pose (the photometric optimum is the pose the scan was taken from), so the visual update can start from the LiDAR
    posterior as in the reference (LIVMapper.cpp:135-136, 371; vio.cpp:1799-1810), and the frame's prior carries the exposure prior.  Returns
    (fmap, cfg dict, extR, extT, frames) with frames[f] = dict(xyz float32 [n][3], R_prior, t_prior, P, vs = VisualScenario, R_true, t_true)."""
    rng = np.random.default_rng(seed0 - 1)
    c = dict(AVIA["lio"])
    extR, extT = AVIA["extrinsic_R"].copy(), AVIA["extrinsic_T"].copy()
    scene = make_room(rng, room, n_boxes)
    R0 = scene.R_ws @ rot_from_rpy(0.01, -0.015, 0.4)
    t0 = scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = default_cov() * 1e-3
    n_map = int(map_points or n_raw * map_rays_factor)
    chunks = [lidar_scan(rng, scene, R0, t0, extR, extT, min(n_map, 400000), c["dept_err"], c["beam_err"], AVIA["blind"], full_sphere) for _ in range(max(1, n_map // 400000 + 1))]
    pw, var = world_points_and_var(np.concatenate(chunks)[:n_map], R0, t0, extR, extT, P0, c["dept_err"], c["beam_err"])
    fmap = build_voxel_map(pw, var, c["voxel_size"], c["max_layer"], c["layer_init_num"], c["min_eigen_value"])
    frames = []
    for f in range(n_frames):
        r = np.random.default_rng(seed0 + f)
        Rf = R0 @ so3_exp(r.normal(0, np.deg2rad(2.0), 3))
        tf = t0 + scene.R_ws @ np.array([r.normal(0, 0.3), r.normal(0, 0.3), r.normal(0, 0.05)])
        xyz = lidar_scan(r, scene, Rf, tf, extR, extT, n_raw, c["dept_err"], c["beam_err"], AVIA["blind"], full_sphere)
        if downsample:
            xyz = voxel_grid_downsample(xyz, downsample)
        if max_points and len(xyz) > max_points:
            xyz = xyz[np.sort(r.permutation(len(xyz))[:max_points])]
        vs = visual_scenario(seed=seed0 + f, n_patches=n_patches, R_true=Rf, t_true=tf)
        frames.append(dict(xyz=np.ascontiguousarray(xyz, np.float32), R_prior=Rf @ so3_exp(r.normal(0, np.deg2rad(0.5), 3)), t_prior=tf + r.normal(0, 0.03, 3), P=prior_cov(r), vs=vs,
                           R_true=Rf, t_true=tf))
    return fmap, c, extR, extT, frames


# ---- visual scenario -----------------------------------------------------------------------------------------------------------
def make_image(rng, width=640, height=512, sigma=2.0):
    from scipy.ndimage import gaussian_filter
    img = gaussian_filter(rng.uniform(0, 1, (height, width)), sigma)
    img += 0.5 * gaussian_filter(rng.uniform(0, 1, (height, width)), 4 * sigma)
    img = (img - img.min()) / (img.max() - img.min())
    return np.clip(np.round(img * 255.0), 0, 255).astype(np.uint8)


def vio_constants(extR, extT, Rcl, Pcl):
    """Rci, Pci of initializeVIO (reference src/vio.cpp:27-38, 57-58)."""
    Rli = extR.T
    Pli = -extR.T @ extT
    return Rcl @ Rli, Rcl @ Pli + Pcl


def sample_patch(img, pc, scale):
    """The reference's bilinear 8x8 sample at stride `scale` around pixel pc (reference src/vio.cpp:1580-1620), float32 math."""
    img = np.asarray(img)
    f32 = np.float32
    u_ref, v_ref = f32(pc[0]), f32(pc[1])
    u_i = int(np.floor(f32(pc[0] / scale)) * f32(scale))
    v_i = int(np.floor(f32(pc[1] / scale)) * f32(scale))
    su = f32((u_ref - f32(u_i)) / f32(scale))
    sv = f32((v_ref - f32(v_i)) / f32(scale))
    w_tl = f32((1.0 - np.float64(su)) * (1.0 - np.float64(sv)))
    w_tr = f32(np.float64(su) * (1.0 - np.float64(sv)))
    w_bl = f32((1.0 - np.float64(su)) * np.float64(sv))
    w_br = f32(su * sv)
    xs = v_i + (np.arange(8) - 4) * scale
    ys = u_i + (np.arange(8) - 4) * scale
    a = img[np.ix_(xs, ys)].astype(f32)
    b = img[np.ix_(xs, ys + scale)].astype(f32)
    c = img[np.ix_(xs + scale, ys)].astype(f32)
    d = img[np.ix_(xs + scale, ys + scale)].astype(f32)
    return ((w_tl * a + w_tr * b) + w_bl * c) + w_br * d


@dataclass
class VisualScenario:
    img: np.ndarray
    pos: np.ndarray            # f64 [M,3] world positions (VisualPoint::pos_)
    warp_patch: np.ndarray     # f32 [M,L,64]
    search_levels: np.ndarray  # int32 [M]
    inv_expo_list: np.ndarray  # f64 [M]
    R_true: np.ndarray
    t_true: np.ndarray
    tau_true: float
    R_prior: np.ndarray
    t_prior: np.ndarray
    tau_prior: float
    P: np.ndarray
    extR: np.ndarray
    extT: np.ndarray
    Rcl: np.ndarray
    Pcl: np.ndarray
    cam: dict
    cfg: dict


HILTI_EQUIDISTANT = (-0.03696737352869157, -0.008917880497032812, 0.008912969593422046, -0.0037685977496087313)   # config/camera_fisheye_HILTI22.yaml:9-12 k1..k4 (vk::EquidistantCamera)
AVIA_RADTAN = (-0.076160, 0.123001, -0.00113, 0.000251, 0.0)      # config/camera_pinhole.yaml:9-12 cam_d0..cam_d3 (+ d4 = 0); the coefficients act on
                                                                   # normalised coordinates, so the yaml's `scale: 0.5` leaves them unchanged


def radtan_project(cam, d, pf):
    """vk::PinholeCamera::world2cam with distortion (rpg_vikit): pf [.., 3] camera-frame points -> pixels [.., 2]"""
    x, y = pf[..., 0] / pf[..., 2], pf[..., 1] / pf[..., 2]
    r2 = x * x + y * y; r4 = r2 * r2; r6 = r4 * r2
    a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
    cd = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6
    xd, yd = x * cd + d[2] * a1 + d[3] * a2, y * cd + d[2] * a3 + d[3] * a1
    return np.stack([xd * cam["fx"] + cam["cx"], yd * cam["fy"] + cam["cy"]], -1)


def visual_scenario(seed=3, n_patches=2000, L=4, rot_sigma_deg=0.05, pos_sigma=0.004, noise_sigma=1.0, R_true=None, t_true=None, distortion=None):
    rng = np.random.default_rng(seed)
    cam = dict(AVIA["cam"])
    cfg = dict(AVIA["vio"])
    cfg["patch_pyrimid_level"] = L
    extR, extT, Rcl, Pcl = AVIA["extrinsic_R"].copy(), AVIA["extrinsic_T"].copy(), AVIA["Rcl"].copy(), AVIA["Pcl"].copy()
    img = make_image(rng, cam["width"], cam["height"])
    if R_true is None:
        R_true = rot_from_rpy(0.02, -0.01, 0.7)
    if t_true is None:
        t_true = np.array([1.0, -0.5, 1.2])
    tau_true = 1.0
    Rci, Pci = vio_constants(extR, extT, Rcl, Pcl)
    Rcw = Rci @ R_true.T
    Pcw = -Rci @ R_true.T @ t_true + Pci
    search = rng.choice([0, 1, 2], size=n_patches, p=[0.8, 0.15, 0.05]).astype(np.int32)
    margin = 5 * (1 << (L - 1 + search)) + 24
    margin = np.minimum(margin, min(cam["width"], cam["height"]) // 2 - 8)
    u = rng.uniform(margin, cam["width"] - 1 - margin)
    v = rng.uniform(margin, cam["height"] - 1 - margin)
    depth = rng.uniform(2.0, 15.0, n_patches)
    p_c = np.stack([(u - cam["cx"]) / cam["fx"] * depth, (v - cam["cy"]) / cam["fy"] * depth, depth], 1)
    pos = (p_c - Pcw) @ Rcw                                   # Rcw^T (p_c - Pcw)
    inv_ref = rng.uniform(0.9, 1.1, n_patches)
    warp = np.zeros((n_patches, L, 64), np.float32)
    for i in range(n_patches):
        pf = Rcw @ pos[i] + Pcw
        pc = np.array([cam["fx"] * pf[0] / pf[2] + cam["cx"], cam["fy"] * pf[1] / pf[2] + cam["cy"]]) if distortion is None else radtan_project(cam, distortion, pf)
        for lvl in range(L):
            cur = sample_patch(img, pc, 1 << (lvl + int(search[i]))).astype(np.float64)
            warp[i, lvl] = (cur * (tau_true / inv_ref[i]) + rng.normal(0, noise_sigma, (8, 8))).astype(np.float32).ravel()
    R_prior = R_true @ so3_exp(rng.normal(0, np.deg2rad(rot_sigma_deg), 3))
    t_prior = t_true + rng.normal(0, pos_sigma, 3)
    tau_prior = tau_true * (1.0 + rng.normal(0, 0.01))
    return VisualScenario(img, pos, warp, search, inv_ref, R_true, t_true, tau_true, R_prior, t_prior, tau_prior, prior_cov(rng), extR, extT, Rcl, Pcl,
                          cam, cfg)


def visual_inverse_scenario(seed=5, n_patches=300, L=4, n_ref=2, **kw):
    """Inverse-compositional variant (reference src/vio.cpp:1327-1518): every point carries a reference patch = (reference image,
    px_, bearing f_, R_ref_w, camera centre) taken at a reference pose; ref patches hold the reference-frame intensities (no exposure,
    no search level).  Here the reference frames see the same synthetic texture from the TRUE pose, so the true pose is the optimum."""
    vs = visual_scenario(seed=seed, n_patches=n_patches, L=L, **kw)
    rng = np.random.default_rng(seed + 1000)
    Rci, Pci = vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    Rcw = Rci @ vs.R_true.T
    Pcw = -Rci @ vs.R_true.T @ vs.t_true + Pci
    M = len(vs.pos)
    vs.search_levels = np.zeros(M, np.int32)
    vs.ref_imgs = np.stack([vs.img] * n_ref)
    vs.ref_img_idx = rng.integers(0, n_ref, M).astype(np.int32)
    p_c = vs.pos @ Rcw.T + Pcw
    vs.ref_px = np.stack([vs.cam["fx"] * p_c[:, 0] / p_c[:, 2] + vs.cam["cx"], vs.cam["fy"] * p_c[:, 1] / p_c[:, 2] + vs.cam["cy"]], 1)
    vs.ref_f = p_c / np.linalg.norm(p_c, axis=1, keepdims=True)
    vs.ref_R = np.repeat(Rcw.reshape(1, 9), M, 0)
    vs.ref_pos = np.repeat((-Rcw.T @ Pcw).reshape(1, 3), M, 0)
    for i in range(M):
        for lvl in range(L):
            cur = sample_patch(vs.img, vs.ref_px[i], 1 << lvl).astype(np.float64)
            vs.warp_patch[i, lvl] = (cur + rng.normal(0, 1.0, (8, 8))).astype(np.float32).ravel()
    vs.tau_prior = 1.0
    return vs


# ---- candidates of retrieveFromVisualSparseMap's per-point tail (reference src/vio.cpp:598-767) -----------------------
@dataclass
class RetrieveScenario:
    img: np.ndarray            # current image u8 [H,W]
    ref_imgs: np.ndarray       # u8 [n_ref,H,W]
    pos: np.ndarray            # [n,3] pt->pos_
    normal: np.ndarray         # [n,3] pt->normal_
    ref_img_idx: np.ndarray    # [n] int32
    ref_px: np.ndarray         # [n,2] ref_ftr->px_
    ref_f: np.ndarray          # [n,3] ref_ftr->f_
    ref_R: np.ndarray          # [n,9] ref_ftr->T_f_w_ rotation
    ref_t: np.ndarray          # [n,3] ref_ftr->T_f_w_ translation
    ref_level: np.ndarray      # [n] int32
    ref_inv_expo: np.ndarray   # [n]
    R_cur: np.ndarray          # new_frame_->T_f_w_
    t_cur: np.ndarray
    inv_expo_cur: float
    cam: dict
    cfg: dict


def retrieve_scenario(seed=21, n_cand=2000, n_ref=3, L=4, normal_en=True, ncc_en=False, outlier_threshold=1000.0, ncc_thre=0.5, margin=40):
    """Reference frames look at the same synthetic texture from poses near the current one (the same image is reused as their picture),
    so most warps are near-identity and pass the photometric gate; a share of candidates gets a much closer reference camera
    (det A > 3: search level 1-2), a strongly mis-registered reference (rejected by the gate) or a reference pixel near the border
    (out-of-image samples = 0)."""
    rng = np.random.default_rng(seed)
    cam = dict(AVIA["cam"])
    cfg = dict(AVIA["vio"])
    cfg.update(patch_pyrimid_level=L, normal_en=int(normal_en), ncc_en=int(ncc_en), outlier_threshold=float(outlier_threshold), ncc_thre=float(ncc_thre))
    img = make_image(rng, cam["width"], cam["height"], sigma=7.0)      # smooth texture: a 1-2 px mis-registration changes a patch only mildly
    ref_imgs = np.stack([img] + [make_image(rng, cam["width"], cam["height"], sigma=7.0) for _ in range(n_ref - 1)])
    R_cw = rot_from_rpy(0.03, -0.02, 0.4)
    t_cw = np.array([0.3, -0.2, 0.1])
    # points seen by the current camera, comfortably inside the image
    u = rng.uniform(margin, cam["width"] - 1 - margin, n_cand)
    v = rng.uniform(margin, cam["height"] - 1 - margin, n_cand)
    depth = rng.uniform(2.0, 12.0, n_cand)
    p_c = np.stack([(u - cam["cx"]) / cam["fx"] * depth, (v - cam["cy"]) / cam["fy"] * depth, depth], 1)
    pos = (p_c - t_cw) @ R_cw                                  # R^T (p_c - t)
    normal_c = -p_c / np.linalg.norm(p_c, axis=1, keepdims=True) + rng.normal(0, 0.2, (n_cand, 3))     # roughly facing the camera
    normal_c /= np.linalg.norm(normal_c, axis=1, keepdims=True)
    normal = normal_c @ R_cw
    kind = rng.choice(4, n_cand, p=[0.7, 0.12, 0.1, 0.08])    # 0 near-identity, 1 close-up reference, 2 mis-registered, 3 border pixel
    ref_R = np.zeros((n_cand, 9)); ref_t = np.zeros((n_cand, 3)); ref_px = np.zeros((n_cand, 2)); ref_f = np.zeros((n_cand, 3))
    for i in range(n_cand):
        rot_s, tr_s = (0.05, 0.004) if kind[i] != 2 else (3.0, 0.4)
        dR = so3_exp(rng.normal(0, np.deg2rad(rot_s), 3))
        dt = rng.normal(0, tr_s, 3)
        if kind[i] == 1:
            dt = dt + np.array([0, 0, -0.55 * depth[i]])       # reference camera at less than half the distance
        R_rw = dR @ R_cw
        t_rw = dR @ t_cw + dt
        pr = R_rw @ pos[i] + t_rw
        px = np.array([cam["fx"] * pr[0] / pr[2] + cam["cx"], cam["fy"] * pr[1] / pr[2] + cam["cy"]])
        if kind[i] == 3:
            px = np.array([rng.choice([3.2, cam["width"] - 4.7]), rng.uniform(20, cam["height"] - 20)])
        ref_R[i] = R_rw.ravel(); ref_t[i] = t_rw; ref_px[i] = px
        f = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
        ref_f[i] = f / np.linalg.norm(f)
    idx = np.where(kind == 2, rng.integers(1, n_ref, n_cand), 0).astype(np.int32) if n_ref > 1 else np.zeros(n_cand, np.int32)
    return RetrieveScenario(img, ref_imgs, pos, normal, idx, ref_px, ref_f, ref_R, ref_t, rng.integers(0, 3, n_cand).astype(np.int32),
                            rng.uniform(0.9, 1.1, n_cand), R_cw, t_cw, 1.02, cam, cfg)


# ---- raw scan + IMU poses for the pre-stage (reference src/IMU_Processing.cpp:494-539, src/LIVMapper.cpp:351-352) -----------
@dataclass
class RawScanScenario:
    xyz: np.ndarray            # f32 [n,3] raw points, LiDAR frame
    curvature: np.ndarray      # f32 [n] ms from the scan start, ascending
    poses: np.ndarray          # f64 [K,22] Pose6D rows: offset_time, acc3, gyr3, vel3, pos3, rot9
    rot_end: np.ndarray
    pos_end: np.ndarray
    extR: np.ndarray
    extT: np.ndarray
    leaf: float
    cfg: dict


def raw_scan_scenario(seed=51, n_raw=24000, scan_ms=100.0, imu_hz=200.0, extR=None, extT=None, leaf=None):
    """A Livox-Avia-like raw scan (24 000 points / 100 ms, src/preprocess.cpp:185) taken while the IMU moves: per-point time stamps and the
    IMUpose list the forward propagation would have left (one Pose6D per IMU sample, piecewise-constant angular rate and acceleration)."""
    rng = np.random.default_rng(seed)
    c = dict(AVIA["lio"])
    extR = AVIA["extrinsic_R"].copy() if extR is None else np.asarray(extR, np.float64)
    extT = AVIA["extrinsic_T"].copy() if extT is None else np.asarray(extT, np.float64)
    scene = make_room(rng, (20.0, 20.0, 6.0), 8)
    R0 = scene.R_ws @ rot_from_rpy(0.01, -0.015, 0.4)
    t0 = scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    xyz = lidar_scan(rng, scene, R0, t0, extR, extT, n_raw, c["dept_err"], c["beam_err"], AVIA["blind"], False)
    cur = np.sort(rng.uniform(0.0, scan_ms, len(xyz))).astype(np.float32)
    cur[0] = np.float32(0.0)
    K = int(round(scan_ms / 1000.0 * imu_hz)) + 1
    poses = np.zeros((K, 22))
    R, p, v = R0.copy(), t0.copy(), np.array([0.8, -0.3, 0.05])
    dt = 1.0 / imu_hz
    for k in range(K):
        gyr = np.array([0.2, -0.35, 0.5]) + rng.normal(0, 0.05, 3)
        acc = np.array([0.3, 0.2, -0.1]) + rng.normal(0, 0.1, 3)
        poses[k, 0] = k * dt
        poses[k, 1:4], poses[k, 4:7], poses[k, 7:10], poses[k, 10:13], poses[k, 13:22] = acc, gyr, v, p, R.ravel()
        R = R @ so3_exp(gyr * dt)
        p = p + v * dt + 0.5 * acc * dt * dt
        v = v + acc * dt
    return RawScanScenario(np.ascontiguousarray(xyz, np.float32), cur, poses, R, p, extR, extT, float(AVIA["filter_size_surf"] if leaf is None else leaf), c)


# ---- inputs of retrieveFromVisualSparseMap's selection half (reference src/vio.cpp:352-486, 598-635) ---------------------------
@dataclass
class SelectScenario:
    pg: np.ndarray             # [n_pg,3] pv_list_ point_w of the current scan
    pos: np.ndarray            # [n,3] visual map points (VisualPoint::pos_)
    keys: np.ndarray           # [n,3] int64 feat_map voxel each point is filed under (insertPointIntoVoxelMap)
    active: np.ndarray         # [n] uint8: pt != nullptr && obs_.size() > 0
    R_cur: np.ndarray          # new_frame_->T_f_w_
    t_cur: np.ndarray
    cam: dict
    border: int
    grid_size: int
    grid_n_width: int
    grid_n_height: int


def feat_map_key_np(pos):
    """insertPointIntoVoxelMap's key (reference src/vio.cpp:227-236): float(p / 0.5), -1 for negatives, truncation."""
    loc = (np.asarray(pos, np.float64) / 0.5).astype(np.float32)
    loc = np.where(loc < 0, (loc.astype(np.float64) - 1.0).astype(np.float32), loc)
    return loc.astype(np.int64)


def select_scenario(seed=71, n_pg=10000, n_vis=6000, L=4, grid_n_height=17, scene=None, R0=None, t0=None, raycast=False, cam=None, extrinsics=None):
    """scene / R0 / t0: take the visual points and the scan from THIS room at THIS sensor pose (scenarios/live_inputs.py: the one-scene live chain) instead of a
    room of the scenario's own.  cam / extrinsics = (extR, extT, Rcl, Pcl): another shipped configuration's camera intrinsics + image size and sensor mounting
    (scenarios/shipped_configs.py); default: config/avia.yaml."""
    rng = np.random.default_rng(seed)
    cam = dict(AVIA["cam"]) if cam is None else dict(cam)
    extR, extT, Rcl, Pcl = (AVIA["extrinsic_R"].copy(), AVIA["extrinsic_T"].copy(), AVIA["Rcl"].copy(), AVIA["Pcl"].copy()) if extrinsics is None else [np.array(x, np.float64) for x in extrinsics]
    c = dict(AVIA["lio"])
    if scene is None:
        scene = make_room(rng, (20.0, 20.0, 6.0), 8)
        R0 = scene.R_ws @ rot_from_rpy(0.01, -0.015, 0.4)
        t0 = scene.R_ws @ np.array([0.3, -0.2, 1.0 if raycast else 1.4]) + scene.t_ws          # (raycast: low enough for the 2.9-m rays to reach the floor's planes)
    bore = None if extrinsics is None else boresight_towards(Rcl[2])             # the camera's optical axis in the LiDAR frame (third row of Rcl)
    xyz = lidar_scan(rng, scene, R0, t0, extR, extT, n_pg + 3 * n_vis, c["dept_err"], c["beam_err"], AVIA["blind"], False, boresight=bore).astype(np.float64)
    pw = (xyz @ extR.T + extT) @ R0.T + t0
    pg = pw[:n_pg].astype(np.float32).astype(np.float64)            # point_w comes from a float32 cloud
    pos = pw[n_pg:][rng.permutation(len(pw) - n_pg)[:n_vis]] + rng.normal(0, 0.01, (n_vis, 3))
    pos[: n_vis // 20] = -pos[: n_vis // 20] + 2 * t0               # a few points behind the camera
    # foreground clutter in the scan: points ~1 m closer on (almost) the same rays as some visual points -> their depth-image neighbourhood
    # disagrees with the visual point's depth by more than 0.5 m (the depth-continuity test rejects those)
    k0, k1 = n_vis // 20, n_vis // 20 + n_vis // 12
    ray = t0 - pos[k0:k1]
    fg = pos[k0:k1] + ray / np.linalg.norm(ray, axis=1, keepdims=True) * rng.uniform(0.8, 1.4, (k1 - k0, 1)) + rng.normal(0, 0.01, (k1 - k0, 3))
    pg = np.concatenate([pg, fg.astype(np.float32).astype(np.float64)])[rng.permutation(len(pg) + len(fg))]
    Rci, Pci = vio_constants(extR, extT, Rcl, Pcl)
    R_cur = Rci @ R0.T
    t_cur = -Rci @ R0.T @ t0 + Pci
    map_pw = None
    if raycast:
        # A scene for the RayCasting module (vio.cpp:487-591; rays are sampled to 2.9 m only): the LiDAR map keeps every scan point (`map_pw`), but most scan points
        # within 3.5 m of the camera are dropped from `pg`, so that the near voxels are NOT in sub_feat_map and the rays of cells without a map point get to work;
        # a few dozen visual points are re-drawn inside the right half of the near field (1.2 .. 2.9 m along random pixels), where only a ray can find them.
        map_pw = pg.copy()
        cam_c = -R_cur.T @ t_cur
        near = np.linalg.norm(pg - cam_c, axis=1) < 3.5
        pg = pg[~near | (pg[:, 1] > cam_c[1] + 1.2)]                  # (the near field keeps its scan points on one side only: rays into it end at a sub_feat_map voxel)
        k = 48                                                            # few and not closer than 1.2 m: a voxel holding one of them stops EVERY ray that crosses it
        bd = (4 + 1) * (1 << L)
        u = rng.uniform(cam["cx"], cam["width"] - bd - 2, k); v = rng.uniform(bd + 2, cam["height"] - bd - 2, k); dd = rng.uniform(1.2, 2.9, k)
        p_c = np.stack([(u - cam["cx"]) / cam["fx"] * dd, (v - cam["cy"]) / cam["fy"] * dd, dd], 1)
        pos[-k:] = (p_c - t_cur) @ R_cur
        # ... and the left half of the near field has NO visual points at all: there a ray finds nothing in feat_map and goes on to the LiDAR map's planes
        empty = (np.linalg.norm(pos - cam_c, axis=1) < 3.6) & (((pos - cam_c) @ R_cur.T)[:, 0] < 0.0)
        far = np.flatnonzero(np.linalg.norm(pos - cam_c, axis=1) >= 3.6)
        pos[empty] = pos[rng.choice(far, int(empty.sum()))] + rng.normal(0, 0.02, (int(empty.sum()), 3))
    grid_size = int(cam["height"] / grid_n_height)                   # vio.cpp:74-76
    gh = int(np.ceil(float(int(cam["height"] / grid_size))))
    gw = int(np.ceil(float(int(cam["width"] / grid_size))))
    border = (4 + 1) * (1 << L)                                      # vio.cpp:154
    active = (rng.uniform(size=n_vis) > 0.05).astype(np.uint8)
    ss = SelectScenario(pg, pos, feat_map_key_np(pos), active, R_cur, t_cur, cam, border, grid_size, gw, gh)
    if raycast:
        ss.map_pw = map_pw
    return ss


# ---- the whole retrieveFromVisualSparseMap: selection -> reference-patch choice -> warp/gate tail (reference src/vio.cpp:352-780) -----------
@dataclass
class RetrieveChainScenario:
    sel: SelectScenario        # scan points, visual points, current pose, grid
    img: np.ndarray            # current image u8 [H,W]
    ref_imgs: np.ndarray       # u8 [n_ref,H,W]: Feature::img_ of every frame that observed something
    normal: np.ndarray         # [n,3] VisualPoint::normal_
    normal_initialized: np.ndarray   # [n] uint8 is_normal_initialized_
    ref_patch: np.ndarray      # [n] int32: global observation index of pt->ref_patch, -1 = !has_ref_patch_
    obs_offset: np.ndarray     # [n+1] int32: CSR over pt->obs_ (list order)
    obs_id: np.ndarray         # per observation: Feature::id_ (= id of the frame it was made in)
    obs_img_idx: np.ndarray    # index into ref_imgs
    obs_px: np.ndarray         # [m,2] px_
    obs_f: np.ndarray          # [m,3] f_
    obs_R: np.ndarray          # [m,9] T_f_w_ rotation
    obs_t: np.ndarray          # [m,3] T_f_w_ translation
    obs_level: np.ndarray      # [m] int32 level_
    obs_inv_expo: np.ndarray   # [m] inv_expo_time_
    obs_patch: np.ndarray      # [m,64] float32 patch_
    inv_expo_cur: float
    cfg: dict                  # patch_pyrimid_level, normal_en, ncc_en, ncc_thre, outlier_threshold


def retrieve_chain_scenario(seed=81, n_pg=10000, n_vis=20000, L=4, grid_n_height=17, normal_en=True, ncc_en=False, ncc_thre=0.5, outlier_threshold=1000.0,
                            max_obs=6, scene=None, R0=None, t0=None, raycast=False, cam=None, extrinsics=None):
    """Visual points of select_scenario, each observed by 1..max_obs features made in a handful of earlier frames: frames 0-3 stand close to the
    current pose and show the current texture (their patches warp almost identically and pass the gates), frame 4 is a close-up with another
    texture, frame 5 looks at the scene from the side (more than 60 degrees off: getCloseViewObs rejects it).  A share of points carries two
    observations of ONE frame (same id_), only same-id observations, a preset ref_patch, or an uninitialised normal."""
    base = select_scenario(seed=seed, n_pg=n_pg, n_vis=n_vis, L=L, grid_n_height=grid_n_height, scene=scene, R0=R0, t0=t0, raycast=raycast, cam=cam, extrinsics=extrinsics)
    rng = np.random.default_rng(seed + 1000)
    cam = base.cam
    W, H = cam["width"], cam["height"]
    img = make_image(rng, W, H, sigma=7.0)
    n_frames = 6
    ref_imgs = np.stack([img, img, img, img, make_image(rng, W, H, sigma=7.0), make_image(rng, W, H, sigma=7.0)])
    cam_pos = -base.R_cur.T @ base.t_cur
    fr_R, fr_t = [], []
    for k in range(n_frames):
        rot_s, tr_s = [(0.02, 0.002), (0.05, 0.004), (0.1, 0.01), (0.3, 0.03), (0.1, 0.01), (0.1, 0.01)][k]
        dR = so3_exp(rng.normal(0, np.deg2rad(rot_s), 3))
        R_fw = dR @ base.R_cur
        c_w = cam_pos + rng.normal(0, tr_s, 3)
        if k == 4:
            c_w = c_w + base.R_cur.T @ np.array([0.0, 0.0, 1.2])          # 1.2 m closer along the optical axis
        if k == 5:
            c_w = c_w + base.R_cur.T @ np.array([25.0, 0.0, 2.0])         # far to the side
        fr_R.append(R_fw); fr_t.append(-R_fw @ c_w)
    fr_ie = rng.uniform(0.9, 1.1, n_frames)
    n = len(base.pos)
    n_obs = rng.integers(1, max_obs + 1, n)
    kind = rng.choice(4, n, p=[0.8, 0.08, 0.06, 0.06])          # 1: two observations of one frame, 2: only one id, 3: side view among them
    offs = np.zeros(n + 1, np.int32)
    ids, iidx, px_l, f_l, R_l, t_l, lvl_l, ie_l, patch_l = [], [], [], [], [], [], [], [], []
    for i in range(n):
        m = int(n_obs[i])
        frames = rng.choice(5, m, p=[0.3, 0.25, 0.2, 0.15, 0.1])
        if kind[i] == 1 and m >= 2:
            frames[1] = frames[0]
        elif kind[i] == 2:
            frames[:] = frames[0]
        elif kind[i] == 3:
            frames[rng.integers(0, m)] = 5
            if rng.uniform() < 0.5:
                frames[:] = 5
        for k in frames:
            pr = fr_R[k] @ base.pos[i] + fr_t[k]
            z = pr[2] if abs(pr[2]) > 1e-3 else 1e-3
            px = np.array([cam["fx"] * pr[0] / z + cam["cx"], cam["fy"] * pr[1] / z + cam["cy"]]) + rng.normal(0, 0.3, 2)
            px = np.clip(px, [6.0, 6.0], [W - 7.0, H - 7.0])
            f = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
            xi, yi = int(px[0]), int(px[1])
            patch = ref_imgs[k][yi - 4:yi + 4, xi - 4:xi + 4].astype(np.float32).ravel() + rng.normal(0, 2.0, 64).astype(np.float32)
            ids.append(100 + k); iidx.append(k); px_l.append(px); f_l.append(f / np.linalg.norm(f)); R_l.append(fr_R[k].ravel()); t_l.append(fr_t[k])
            lvl_l.append(rng.integers(0, 3)); ie_l.append(fr_ie[k]); patch_l.append(patch)
        offs[i + 1] = offs[i] + m
    to_cam = cam_pos - base.pos
    normal = to_cam / np.linalg.norm(to_cam, axis=1, keepdims=True) + rng.normal(0, 0.2, (n, 3))
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    ninit = (rng.uniform(size=n) > 0.07).astype(np.uint8)
    ref_patch = np.full(n, -1, np.int32)
    preset = (rng.uniform(size=n) < 0.4) & (n_obs >= 2)
    ref_patch[preset] = offs[:-1][preset] + (rng.integers(0, 1 << 30, preset.sum()) % n_obs[preset])
    cfg = dict(patch_pyrimid_level=L, normal_en=int(normal_en), ncc_en=int(ncc_en), ncc_thre=float(ncc_thre), outlier_threshold=float(outlier_threshold))
    return RetrieveChainScenario(base, img, ref_imgs, normal, ninit, ref_patch, offs, np.array(ids, np.int32), np.array(iidx, np.int32), np.array(px_l),
                                 np.array(f_l), np.array(R_l), np.array(t_l), np.array(lvl_l, np.int32), np.array(ie_l), np.array(patch_l, np.float32),
                                 1.02, cfg)


# ---- one LiDAR-inertial frame: IMU steps + raw scan registered to a map (reference src/LIVMapper.cpp:342-377) --------------------------------
@dataclass
class LioFrameScenario:
    sc: LidarScenario          # map, extrinsics, prior pose / covariance; sc.xyz = the RAW scan (LiDAR frame, not down-sampled)
    curvature: np.ndarray      # f32 [n] ms from the scan start, ascending
    steps: np.ndarray          # [n_steps,8] gyr3 acc3 dt offs_t
    first_pose: np.ndarray     # [22] IMUpose[0] = set_pose6d(0, acc_s_last, angvel_last, vel, pos, rot)
    vel: np.ndarray
    bg: np.ndarray
    ba: np.ndarray
    grav: np.ndarray
    inv_expo: float
    imu: dict                  # covariances, G_m_s2, mean_acc_norm, flags


def lio_frame_scenario(seed=61, n_raw=20000, n_steps=20):
    """A sensor almost at rest at the LiDAR scenario's prior pose: raw scan with per-point times over 100 ms, IMU steps that measure gravity (+ bias, + noise),
    so that the propagated state stays within a few mm of the pose the scan was taken from and the update matches most points."""
    sc = lidar_scenario(seed=seed, n_points=n_raw, downsample=None)
    rng = np.random.default_rng(seed + 7)
    imu = dict(cov_gyr=[0.1, 0.1, 0.1], cov_acc=[0.1, 0.12, 0.09], cov_bias_gyr=[1e-4, 1.2e-4, 0.9e-4], cov_bias_acc=[1e-4, 1e-4, 2e-4], cov_inv_expo=0.2,
               G_m_s2=9.81, mean_acc_norm=9.79, ba_bg_est_en=1, gravity_est_en=1, exposure_estimate_en=1)
    curv = np.sort(rng.uniform(0.0, 100.0, len(sc.xyz))).astype(np.float32)
    grav, vel = np.array([0.0, 0.0, -9.81]), np.array([0.02, -0.01, 0.005])
    bg, ba = np.array([1e-3, -2e-3, 5e-4]), np.array([0.01, -0.02, 0.015])
    dt = 0.1 / max(n_steps, 1)
    rest = sc.R_prior.T @ (-grav) * imu["mean_acc_norm"] / imu["G_m_s2"] + ba          # specific force in sensor units + bias (IMU_Processing.cpp:375-379)
    steps = np.c_[rng.normal(0, 0.01, (n_steps, 3)) + bg, rng.normal(0, 0.02, (n_steps, 3)) + rest, np.full(n_steps, dt), np.arange(1, n_steps + 1) * dt]
    first = np.concatenate([[0.0], np.zeros(3), np.zeros(3), vel, sc.t_prior, sc.R_prior.ravel()])
    return LioFrameScenario(sc, curv, steps, first, vel, bg, ba, grav, 0.97, imu)
