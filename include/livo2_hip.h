/*
 * livo2_hip.h — C ABI of the MI355X (gfx950) ESIKF measurement-update library `liblivo2_hip.so`.
 *
 * This is the drop-in boundary for ONE hot path of hku-mars/FAST-LIVO2: the per-frame ESIKF measurement update
 *   - LiDAR  : VoxelMapManager::StateEstimation(StatesGroup&)            (reference src/voxel_map.cpp:338-511,
 *              called from LIVMapper::handleLIO, src/LIVMapper.cpp:370)
 *   - visual : VIOManager::computeJacobianAndUpdateEKF(cv::Mat)           (reference src/vio.cpp:784-802,
 *              called from VIOManager::processFrame, src/vio.cpp:1810)
 * Plain pointers and sizes only; no C++/torch/Eigen types cross this boundary.  All matrices are ROW-MAJOR doubles.
 * Every function returns LIVO2_OK (0) or a negative error code and never throws; `livo2_last_error(ctx)` gives text.
 * One ctx = one GPU + one HIP stream; a ctx is not thread-safe; use one ctx per GPU / per host thread
 * (reference threading: single caller thread, src/LIVMapper.cpp:534-552).
 * The caller owns every host buffer for the duration of a call; the ctx owns device memory and pinned staging.
 * There is NO CPU fallback: without a usable gfx950 device `livo2_ctx_create` fails with LIVO2_ERR_NO_DEVICE.
 */
#ifndef LIVO2_HIP_H
#define LIVO2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIVO2_DIM_STATE 19      /* reference include/common_lib.h:30  (DIM_STATE) */
#define LIVO2_MAX_BATCH 64      /* frames per livo2_lidar_batch_* call */
#define LIVO2_MAX_ITERS 16      /* upper bound accepted for lio/vio max_iterations (reference default 5) */
#define LIVO2_MAX_LEVELS 8      /* upper bound for vio/patch_pyrimid_level (reference default 4) */
#define LIVO2_MAX_POINTS_NUM 1000 /* upper bound for lio/max_points_num of the device-resident tree (reference configs: 50, HILTI22: 100) */
#define LIVO2_MAX_LAYER 4       /* upper bound for lio/max_layer (layer_init_num has 5 entries, voxel_map.cpp:46) */
#define LIVO2_PATCH 8           /* vio/patch_size; the kernels are specialised for 8x8 (config/avia.yaml:33) */

enum {
  LIVO2_OK = 0,
  LIVO2_ERR_INVALID = -1,       /* bad argument (NULL, size, range) */
  LIVO2_ERR_NO_DEVICE = -2,     /* no gfx950 device / HIP runtime unusable */
  LIVO2_ERR_HIP = -3,           /* a HIP call failed */
  LIVO2_ERR_NO_MAP = -4,        /* LiDAR call before livo2_map_upload */
  LIVO2_ERR_NO_SCAN = -5,       /* LiDAR call before livo2_lidar_set_scan */
  LIVO2_ERR_NO_FRAME = -6,      /* visual call before livo2_visual_set_frame */
  LIVO2_ERR_RANGE = -7          /* voxel key outside int32 / value outside supported range */
};

typedef struct livo2_ctx livo2_ctx;

/* Mirror of StatesGroup (reference include/common_lib.h:126-223).  Error-state order:
 * [0:3) rot, [3:6) pos, [6] inv_expo_time, [7:10) vel, [10:13) bias_g, [13:16) bias_a, [16:19) gravity. */
typedef struct livo2_state {
  double rot[9];                /* rot_end, row-major */
  double pos[3];                /* pos_end */
  double inv_expo;              /* inv_expo_time */
  double vel[3];                /* vel_end */
  double bg[3];                 /* bias_g */
  double ba[3];                 /* bias_a */
  double grav[3];               /* gravity */
  double cov[LIVO2_DIM_STATE * LIVO2_DIM_STATE];
} livo2_state;

/* ---- context ------------------------------------------------------------------------------------------------- */
int livo2_ctx_create(int device, livo2_ctx **out);
/* Same, but all work is enqueued on a caller-provided hipStream_t (e.g. torch's current stream) instead of a private one. */
int livo2_ctx_create_on_stream(int device, void *hip_stream, livo2_ctx **out);
void livo2_ctx_destroy(livo2_ctx *ctx);
const char *livo2_last_error(const livo2_ctx *ctx);
void *livo2_ctx_stream(livo2_ctx *ctx);                 /* the hipStream_t this ctx launches on */
int livo2_ctx_synchronize(livo2_ctx *ctx);
/* Page-locked host memory for the caller's buffers (scans, images, per-point outputs): transfers from / to it run at PCIe speed and asynchronously; from pageable
 * memory the runtime stages every copy through its own bounce buffers.  (No reference counterpart: the reference never leaves the host.) */
int livo2_host_alloc_pinned(size_t bytes, void **out);
void livo2_host_free_pinned(void *p);
const char *livo2_version(void);
/* sizeof() of a struct of this header as the library was compiled ("livo2_state", "livo2_lidar_cfg", ...), 0 for an unknown name.  MANDATORY for a binding:
 * compare every struct it mirrors at load time and refuse to run on a mismatch — structs grow at their end between versions (0.2: livo2_lidar_points.pinned) and
 * former padding becomes meaningful (0.2: livo2_select_cfg.raycast_en, validated to be 0 or 1), so a caller built against an older header must zero-initialise
 * every struct it passes and must not pass a shorter one.  livo2_version() names the version ("livo2_hip 0.2 (gfx950)"). */
int32_t livo2_abi_sizeof(const char *struct_name);

/* Per-kernel timing with HIP events on the ctx stream (off by default; adds an event pair per launch).
 * which: 0 = LiDAR residual kernel, 1 = visual residual kernel, 2 = LiDAR solve kernel (and livo2_esikf_solve), 3 = visual solve kernel. */
int livo2_ctx_kernel_timing(livo2_ctx *ctx, int enable);
/* Execution options of a context (results do not depend on them; they select between equivalent launch sequences):
 *   "visual_persistent" (default 1): livo2_visual_update runs computeJacobianAndUpdateEKF (vio.cpp:784-802) as ONE resident grid with the level / iteration
 *                        loops on the device; 0: one residual + one solve launch per (level, iteration).  The persistent grid is used only when it fits on
 *                        the device next to the persistent grids of this process that are still in flight; otherwise the per-step sequence runs.
 *   "lidar_block_order" (default 1): a LiDAR launch that needs more than one round of blocks starts the chunks of the scan in the order of their block lifetimes in
 *                        the previous launch (longest first; recorded by every launch, sorted on the device by the solve of the previous iteration, reset by
 *                        livo2_lidar_set_scan); 0: scan order.
 *                        Admission counts the resident grids of THIS process only; if another process or a long foreign kernel keeps a block of the grid off the
 *                        device, every block gives up after the watchdog time, nothing is committed, and livo2_visual_update_fetch re-runs the update as the
 *                        per-step sequence (counter "visual_persistent_timeouts"; "visual_persistent_debug_timeout" = 1 provokes exactly that in 2 ms, for the tests).
 *                        After a time-out the ctx stays on the per-step sequence for the next 8 updates (16, 32, ... up to 1024 while time-outs repeat; counter
 *                        "visual_persistent_backoff_skips") before it tries a resident grid again.
 *   "visual_persistent_inverse" (default 1, round 6): with cfg->inverse_composition_en the resident grid also runs precomputeReferencePatches + updateStateInverse
 *                        (vio.cpp:1327-1518; 0: that form stays on the launch-per-step sequence — same bits, tests/test_visual_inverse_gpu.py).
 *   "visual_error_waves" (default 1, round 6): the frame error — each OpenMP thread's serial float sum over its block of patch errors (vio.cpp:1554, 1634), whose exact
 *                        bits decide accept / revert — is formed by a group of 32 lanes per thread (fast-livo2_amd/csrc/float_chain.hpp: start -> end tables on consecutive floats, composed in a scan) when the blocks hold >= 768
 *                        patches and the threads fit five waves; 0: always one lane per thread (a dependent chain of adds).  Same bits either way
 *                        (tests/test_float_chain_gpu.py, tests/test_visual_gpu.py).
 *   "map_update_spread" (default 0, round 6): lanes per touched root voxel in the octree update = 8 x this (1, 2, 4, 8; of which 8 work, or all 64 in the plane re-fit when
 *                        this is 8 and "map_update_wide_fit" = 1, the default); 0: chosen per update from the number of roots the previous one touched.  Every choice gives the
 *                        same tree (plane parameters to the rounding of the re-fit's summation order: tests/test_map_tree_gpu.py).
 *   "visual_persistent_timeout_us" (default 20000; environment LIVO2_VP_TIMEOUT_US): that watchdog, in microseconds, [100, 10000000].  A C4-sized update takes
 *                        0.25 ms; the default keeps a 10 Hz pipeline inside its frame budget when a grid loses a compute unit (2 s until round 4).
 *   "lidar_fused_iteration" (default 0; environment LIVO2_LIDAR_FUSED): one launch per ESIKF iteration (k_lidar_iteration: the last block of the residual grid to
 *                        publish its partial row reduces and solves) instead of k_lidar_residual + k_lidar_solve.  Bit-identical results; measured slower at C4
 *                        (profiles/r05_lidar_fused_iteration_ab.txt), kept for grids whose solve launch is the larger share.
 *   "scan_small_fused" (default 1; environment LIVO2_SCAN_SMALL_FUSED): a scan of up to 16 384 points (the reference's operating point: ~10 k points after the
 *                        0.1-m filter, preprocess.cpp:185) is prepared by two launches — Morton keys (in the launch that scatters the frame's inputs), then the
 *                        position of every point in the stable key order COUNTED instead of sorted, with the SoA gather and calcBodyCov (voxel_map.cpp:15-34,
 *                        349-360) at the final positions — instead of k_morton_keys + the library's device-wide sort (5-8 launches) + k_gather_xyz + k_body_cov.
 *                        Same permutation, same bits downstream (tests/test_frame_ingest_gpu.py).  0: the launch sequence.
 *   "frame_ingest" (default 2; LIVO2_FRAME_INGEST): how the inputs of livo2_frame_update_* (and the scan of livo2_lidar_set_scan) reach the device.  0: one copy
 *                        command per array (prior, scan, image, four sub-map arrays).  1: one H2D copy of the pinned staging block into a device arena + one
 *                        launch that scatters it (and prepares the scan).  2: that launch reads the pinned block itself over the link — no copy command — for
 *                        payloads up to 1 MiB; larger frames take form 1.
 *   "frame_publish" (default 1; LIVO2_FRAME_PUBLISH): the two result blocks and the watchdog flag of a frame are written into the pinned result slot by one
 *                        launch instead of three D2H copies.
 * Counters: "visual_persistent_launches", "visual_persistent_fallbacks", "visual_persistent_timeouts", "visual_persistent_backoff_skips", "map_tree_grow_events",
 *           "lidar_fused_launches", "scan_small_launches", "frame_ingest_launches", "frame_zero_copy_launches", "frame_publish_launches". */
int livo2_ctx_set_option(livo2_ctx *ctx, const char *name, int32_t value);
int livo2_ctx_get_counter(livo2_ctx *ctx, const char *name, int64_t *value);
int livo2_ctx_kernel_timing_read(livo2_ctx *ctx, int which, double *total_ms, int64_t *launches, int reset);
/* Debug allocator (no reference counterpart; fast-livo2_amd/csrc/dev_alloc.hpp).  The environment variable LIVO2_REDZONE, read at the first device allocation
 * of the process, puts every device allocation of the library behind a checker: 1 = poisoned guard regions in front of and behind each allocation, scanned by
 * livo2_ctx_synchronize, every *_fetch and this call (a damaged guard => LIVO2_ERR_HIP, livo2_last_error names the allocation's source line, size, side and
 * offset); 2 / 3 = EXPERIMENTAL: each allocation is its own virtual-memory mapping that ends (2) or starts (3) at unmapped address space, so that an
 * out-of-bounds READ would fault deterministically — on ROCm 7.2 / gfx950 purely in-bounds traffic inside hipMemMap'ed ranges already miscomputes
 * (tools/fence_selftest.hip, profiles/r04_memory_fault_hunt.txt), so these two modes are left out of the test suite and must not be used to judge the library.
 * *mode receives the active mode, *damaged_words the number of overwritten guard words (mode 1). */
int livo2_debug_redzone_check(livo2_ctx *ctx, int32_t *mode, int64_t *damaged_words);
/* Self-test of the checker: one 4-byte device store `byte_offset` bytes behind the END of the ctx's control block (negative: in front of its start).
 * Refused (LIVO2_ERR_INVALID) unless LIVO2_REDZONE is set. */
int livo2_debug_redzone_poke(livo2_ctx *ctx, int64_t byte_offset);
/* Self-test of the frame error's accumulation (reference src/vio.cpp:1554, 1634, 1636: each OpenMP thread adds the patch errors of its static block into a float,
 * in index order).  errors[n] (n <= 8192) are split over `threads` blocks as libgomp splits them; sums_wave[c] is thread c's partial sum as the visual solve forms it
 * for long blocks (a group of `lanes_per_chain` = 16 / 32 / 64 lanes per chain, fast-livo2_amd/csrc/float_chain.hpp; threads <= 7 * 64 / lanes_per_chain),
 * sums_serial[c] as one lane adds it.  Both equal the serial float loop bit for bit (tests/test_float_chain_gpu.py). */
int livo2_debug_float_chain(livo2_ctx *ctx, const float *errors, int32_t n, int32_t threads, int32_t lanes_per_chain, float *sums_wave, float *sums_serial);

/* ---- VoxelMap snapshot ("flat map") -------------------------------------------------------------------------- */
/* Index-based mirror of `std::unordered_map<VOXEL_LOCATION, VoxelOctoTree*> voxel_map_` (reference include/voxel_map.h:194)
 * restricted to what StateEstimation reads: per root the key, voxel_center_, quater_length_ (voxel_map.h:139,141); per octree
 * node plane_ptr_->is_plane_ and leaves_[8] (voxel_map.h:135,138); per plane normal_, center_, plane_var_, d_, radius_
 * (voxel_map.h:69-94). */
typedef struct livo2_map_view {
  int32_t n_roots, n_nodes, n_planes;
  const int64_t *root_key;      /* [n_roots][3]  VOXEL_LOCATION x,y,z (must fit int32, else LIVO2_ERR_RANGE) */
  const int32_t *root_node;     /* [n_roots]     node index of the root VoxelOctoTree */
  const double *root_center;    /* [n_roots][3]  voxel_center_ */
  const float *root_quarter;    /* [n_roots]     quater_length_ */
  const int32_t *node_plane;    /* [n_nodes]     plane index if plane_ptr_->is_plane_, else -1 */
  const int32_t *node_child;    /* [n_nodes][8]  leaves_[k] node index or -1 */
  const double *plane_normal;   /* [n_planes][3] */
  const double *plane_center;   /* [n_planes][3] */
  const double *plane_var;      /* [n_planes][36] plane_var_ row-major 6x6 */
  const float *plane_d;         /* [n_planes] */
  const float *plane_radius;    /* [n_planes] */
} livo2_map_view;

int livo2_map_upload(livo2_ctx *ctx, const livo2_map_view *map);
/* Refresh `n` plane records in place after UpdateVoxelMap touched them (VoxelPlane::is_update_, voxel_map.h:86). */
int livo2_map_update_planes(livo2_ctx *ctx, const int32_t *plane_idx, int32_t n, const double *normal, const double *center,
                            const double *plane_var, const float *d, const float *radius);

/* ---- IMU forward propagation ----------------------------------------------------------------------------------------
 * The forward loop of ImuProcess::UndistortPcl (src/IMU_Processing.cpp:298-445): state_in = state_inout at prop_beg_time; per IMU sample
 * pair a step carries the averaged raw measurements 0.5 * (head + tail) (335-341), dt and offs_t — the time-stamp logic that yields them
 * (332, 355-372) stays with the caller.  state_out = state_propagat (rot_end, pos_end, vel_end, cov; biases, gravity, inv_expo_time
 * unchanged); poses[n_steps] = the Pose6D pushed per sample (the entry at offset 0, IMU_Processing.cpp:281, is the caller's).
 * cfg: cov_gyr / cov_acc / cov_bias_gyr / cov_bias_acc / cov_inv_expo (IMU_Processing.cpp:19-23, 92-100), G_m_s2 (common_lib.h:29),
 * mean_acc_norm = mean_acc.norm() (353), and the three estimation switches (386-390, 395). */
typedef struct livo2_imu_step { double gyr[3], acc[3], dt, offs_t; } livo2_imu_step;
typedef struct livo2_imu_cfg {
  double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_inv_expo, G_m_s2, mean_acc_norm;
  int32_t ba_bg_est_en, gravity_est_en, exposure_estimate_en;
  int32_t first_call;           /* !imu_time_init: the first UndistortPcl call forces state_inout.inv_expo_time = tau = 1.0 (IMU_Processing.cpp:305-317, 444);
                                 * every later call passes the state's own inv_expo_time through */
} livo2_imu_cfg;
struct livo2_imu_pose;
int livo2_imu_propagate(livo2_ctx *ctx, const livo2_state *state_in, const livo2_imu_step *steps, int32_t n_steps, const livo2_imu_cfg *cfg,
                        livo2_state *state_out, struct livo2_imu_pose *poses);
double livo2_imu_propagate_last_kernel_us(const livo2_ctx *ctx);

/* ---- map maintenance: batched plane fit ----------------------------------------------------------------
 * VoxelOctoTree::init_plane (src/voxel_map.cpp:55-135) for n_groups voxels at once: group g owns points
 * [offsets[g], offsets[g+1]) of point_w / var (pointWithVar::point_w, ::var of its temp_points_).  out[g] receives the VoxelPlane
 * members init_plane writes (include/voxel_map.h:69-94).  If plane_idx != NULL and plane_idx[g] >= 0 and the fit is a plane, the
 * record of that plane in the resident map is refreshed in place (as livo2_map_update_planes would) — the octree bookkeeping
 * (UpdateOctoTree / cut_octo_tree, voxel_map.cpp:163-290) stays with the caller, who re-uploads the map when a fit changes the
 * tree's shape (a plane lost, a voxel subdivided).  Eigen-decomposition: cyclic Jacobi (the reference's Eigen::EigenSolver is a
 * third-party dependency); eigenvector signs are solver-dependent and cancel in everything downstream. */
typedef struct livo2_plane_fit {
  double center[3], normal[3], y_normal[3], x_normal[3];
  double covariance[9];         /* covariance_, row-major */
  double plane_var[36];         /* plane_var_, row-major 6x6 (zero when not a plane) */
  float radius, min_eigen_value, mid_eigen_value, max_eigen_value, d;   /* float members of VoxelPlane */
  int32_t points_size, is_plane, pad;
} livo2_plane_fit;
int livo2_plane_fit_batch(livo2_ctx *ctx, const double *point_w, const double *var, const int32_t *offsets, int32_t n_groups,
                          float planer_threshold, const int32_t *plane_idx, livo2_plane_fit *out);
/* average kernel time of the last call's k_plane_fit launch in microseconds (HIP events on the ctx stream) */
double livo2_plane_fit_last_kernel_us(const livo2_ctx *ctx);

/* ---- LiDAR point-to-plane update ------------------------------------------------------------------------------ */
/* Knobs of VoxelMapConfig the path reads (reference include/voxel_map.h:35-52, src/voxel_map.cpp:36-53) + extrinsics
 * extR_/extT_ (voxel_map.h:200-201).  deg2rad is PCL's DEG2RAD factor used by calcBodyCov (voxel_map.cpp:21); pass 0 for the
 * PCL default 0.017453293. */
typedef struct livo2_lidar_cfg {
  int32_t max_iterations;       /* lio/max_iterations */
  int32_t max_layer;            /* lio/max_layer */
  double sigma_num;             /* lio/sigma_num, in (0, 30]: above ~38 the reference's first-plane probability underflows to 0 and it records an
                                   uninitialised match (voxel_map.cpp:739-753); such values are refused with LIVO2_ERR_INVALID */
  double dept_err;              /* lio/dept_err  (narrowed to float like calcBodyCov's parameter) */
  double beam_err;              /* lio/beam_err  (narrowed to float) */
  double voxel_size;            /* lio/voxel_size */
  double deg2rad;
  double extR[9];               /* extrinsic_R (LiDAR -> IMU) */
  double extT[3];               /* extrinsic_T */
} livo2_lidar_cfg;

/* Upload one down-sampled scan `feats_down_body_` (xyz float32, AoS [n][3]) and run the once-per-scan precompute
 * (calcBodyCov per point, voxel_map.cpp:349-360).  The scan stays resident until the next set_scan. */
int livo2_lidar_set_scan(livo2_ctx *ctx, const float *xyz, int32_t n, const livo2_lidar_cfg *cfg);

/* ---- device-resident VoxelMap (second half) ------------------------------------------------------------------------------
 * VoxelMapManager::BuildVoxelMap / UpdateVoxelMap (src/voxel_map.cpp:532-591, 609-641) with the octree itself — VoxelOctoTree::UpdateOctoTree /
 * init_octo_tree / cut_octo_tree (137-290), the temp_points_ of every node, the re-fit every update_size_threshold_ points, subdivision, freezing at
 * max_points_num_ — kept in device memory: the residual kernel reads the structure these calls maintain, so no snapshot is flattened or uploaded
 * between frames.  Points of one root voxel are processed in input order like the reference's loop; root voxels are independent and run in
 * parallel.  livo2_map_tree_create makes the (empty) tree the resident map of the ctx, replacing any snapshot of livo2_map_upload (and vice versa);
 * with a tree resident, the plane indices of livo2_lidar_points (match_plane / normal_plane) are rows of the device plane table — livo2_map_tree_export
 * returns the planes in that numbering.  The capacities given at creation (0 = defaults derived from max_roots) are STARTING sizes: after every update the
 * pools of nodes, points, plane rows and candidate records are checked and, when more than 60 % full, doubled (device-to-device copy; candidate ranges are first
 * re-packed) before the next frame, and what mapSliding / freezing nodes release is reused first.  An update can therefore only fail with LIVO2_ERR_RANGE if a
 * SINGLE frame needs more than the free 40 % of a pool (the failed frame's points are then partly dropped; allocation counters are rolled back, so the tree
 * stays usable for later frames), or if the number of root voxels outgrows the hash table (8 buckets per max_roots; not grown).  max_points_num up to
 * LIVO2_MAX_POINTS_NUM (the point regions are sized max_points_num + 2 at creation). */
typedef struct livo2_map_tree_cfg {
  double voxel_size;            /* lio/voxel_size (narrowed to float like BuildVoxelMap's local, voxel_map.cpp:534) */
  double planer_threshold;      /* lio/min_eigen_value (narrowed to float: VoxelOctoTree::planer_threshold_) */
  int32_t max_layer;            /* lio/max_layer */
  int32_t max_points_num;       /* lio/max_points_num */
  int32_t layer_init_num[5];    /* lio/layer_init_num */
  int32_t max_roots;            /* root voxels the cuckoo table is sized for (load factor <= 1/8) */
  int32_t max_nodes, max_planes, max_points, max_cand;   /* pool capacities; 0 = 3 x / 2 x / 80 x / 1 x max_roots */
} livo2_map_tree_cfg;
int livo2_map_tree_create(livo2_ctx *ctx, const livo2_map_tree_cfg *cfg);
/* input_points as the reference passes them: pointWithVar::point_w ([n][3]) and ::var ([n][9] row-major), in pv_list_ order.  build != 0:
 * BuildVoxelMap (all points of a voxel first, then init_octo_tree); else UpdateVoxelMap. */
int livo2_map_tree_update(livo2_ctx *ctx, const double *point_w, const double *var, int32_t n, int32_t build);
/* The same with pv_list_ formed on the device from the resident scan and the given (posterior) state — what LIVMapper::handleLIO does between
 * StateEstimation and UpdateVoxelMap (src/LIVMapper.cpp:413-423): point_w = float32(R (extR p + extT) + t), var = (R extR) body_cov (R extR)^T +
 * [p_i]x P_rr [p_i]x^T + P_tt.  Nothing crosses PCIe but the state. */
int livo2_map_tree_update_from_scan(livo2_ctx *ctx, const livo2_state *state, const livo2_lidar_cfg *cfg, int32_t build);
/* The same (UpdateVoxelMap, build = 0) OFF the critical path of the frame: the reference runs UpdateVoxelMap (src/LIVMapper.cpp:413-424) before handleVIO
 * (:281-334), but the visual half of a frame does not read the LiDAR voxel map (unless vio/raycast_en).  pv_list_ is formed on the context's stream — the retrieval
 * may read it there (LIVO2_PG_FROM_MAP_UPDATE) — and the octree update itself runs on a second stream of the context, concurrently with whatever is enqueued next
 * (livo2_visual_retrieve_from_map, livo2_visual_update).  state == NULL: the posterior the last livo2_lidar_update of this context left on the device.  Returns at
 * once.  livo2_map_tree_update_join waits for the update, reads its pool counters, grows the pools and reports its errors exactly as the synchronous call does
 * (livo2_map_tree_last_kernel_us is valid after it); every entry point that touches the tree, the scan or the LiDAR update joins a pending update first. */
int livo2_map_tree_update_from_scan_async(livo2_ctx *ctx, const livo2_state *state, const livo2_lidar_cfg *cfg);
int livo2_map_tree_update_join(livo2_ctx *ctx);
/* pv_list_[i].point_w ([n][3]) / .var ([n][9]) as the last livo2_map_tree_update[_from_scan] consumed them (the reference keeps them: `_pv_list =
 * voxelmap_manager->pv_list_`, LIVMapper.cpp:426, read by handleVIO :306 and the publishers); either pointer may be NULL.  *n receives the point count. */
int livo2_map_tree_read_pv(livo2_ctx *ctx, double *point_w, double *var, int32_t capacity, int32_t *n);
/* counts[8]: nodes, temp points reserved, plane rows, candidate records reserved, (0), error bits of the last update, roots touched by it, root voxels */
int livo2_map_tree_stats(livo2_ctx *ctx, int32_t *counts);
/* Export as a flat map (same arrays as livo2_map_view; caller buffers sized from livo2_map_tree_stats: roots = counts[7], nodes = counts[0],
 * planes = counts[2]); node_plane[i] = row of the node's plane if is_plane_, else -1.  node_temp (may be NULL) = temp_points_.size() per node. */
int livo2_map_tree_export(livo2_ctx *ctx, int64_t *root_key, int32_t *root_node, double *root_center, float *root_quarter, int32_t *node_plane, int32_t *node_child,
                          double *plane_normal, double *plane_center, double *plane_var, float *plane_d, float *plane_radius, int32_t *node_temp);
/* The VoxelPlane members a match carries into ptpl_list_ (PointToPlane: normal_, center_, plane_var_, d_, layer_; src/voxel_map.cpp:744-755) for `n` rows of the
 * device plane table (the values of livo2_lidar_points::match_plane / normal_plane while a tree is resident).  Any output may be NULL. */
int livo2_map_tree_read_planes(livo2_ctx *ctx, const int32_t *rows, int32_t n, double *normal, double *center, double *plane_var, float *d, float *radius, int32_t *layer);
/* VoxelMapManager::mapSliding + clearMemOutOfMap (src/voxel_map.cpp:924-972; local_map/map_sliding_en, sliding_thresh, half_map_size): when position_last
 * (voxel_map.cpp:492: the posterior position) has moved at least sliding_thresh from the position of the last slide (initially the origin, voxel_map.h:209),
 * every root voxel whose key lies outside [loc - half_map_size, loc + half_map_size] on some axis (loc = the float voxel index of position_last, one lower for
 * negatives, truncated: voxel_map.cpp:936-940) is deleted with its subtree.  *removed = root voxels deleted, -1 when the threshold was not reached (nothing
 * changes).  The node ids, plane rows and 52-point regions of the deleted subtrees are handed out again by later updates before fresh pool memory is
 * (free_counts[3], may be NULL: what the three free stacks hold after the call); regions of other sizes and candidate ranges are not recycled.  (A node
 * that freezes — update_enable_ = false, temp_points_ released — returns its 52-point region the same way, with or without sliding.)  Updates take from the
 * free stacks only once some pool is more than half used (the allocation-by-bump kernels are ~40 % faster; env LIVO2_MAP_RECYCLE=1 forces recycling). */
int livo2_map_tree_slide(livo2_ctx *ctx, const double *position_last, double sliding_thresh, int32_t half_map_size, int32_t *removed, int32_t *free_counts);
/* kernel time of the last livo2_map_tree_update* call in microseconds (HIP events on the ctx stream; sort + segmentation + octree + emit) */
double livo2_map_tree_last_kernel_us(const livo2_ctx *ctx);

/* Raw scan -> feats_down_body on the device: ImuProcess::UndistortPcl's backward propagation of every point to the
 * scan-end pose (src/IMU_Processing.cpp:494-539: xyz [n][3] and curvature [n] = PointType x,y,z,curvature of pcl_wait_proc, sorted by
 * curvature like the reference sorts it, IMU_Processing.cpp:154-156; poses = IMUpose, rot_end / pos_end = state_inout after the forward
 * propagation; Lid_rot_to_IMU / Lid_offset_to_IMU = cfg->extR / extT) followed by downSizeFilterSurf (pcl::VoxelGrid centroid filter,
 * leaf_size = filter_size_surf, src/LIVMapper.cpp:351-352), and then the per-scan work of livo2_lidar_set_scan on the result — the
 * filtered cloud becomes the scan of the next livo2_lidar_update without a host round trip.  *n_down = feats_down_size_.
 * feats_undistort ([n][3]) and feats_down_body (capacity [n][3]) receive copies when not NULL (the caller needs feats_down_body for
 * pv_list_ / UpdateVoxelMap).  n_poses < 2: no undistortion.  pcl::VoxelGrid is third-party (parity unpinned); the order of
 * summation inside a leaf is the input order here (unspecified in PCL). */
typedef struct livo2_imu_pose {   /* Pose6D (msg/Pose6D.msg, include/common_lib.h:225-242) */
  double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9];
} livo2_imu_pose;
int livo2_lidar_preprocess_scan(livo2_ctx *ctx, const float *xyz, const float *curvature, int32_t n, const livo2_imu_pose *poses, int32_t n_poses,
                                const double *rot_end, const double *pos_end, double leaf_size, const livo2_lidar_cfg *cfg, int32_t *n_down,
                                float *feats_undistort, float *feats_down_body);
/* kernel time (undistort + voxel grid, without the Morton/body-cov stage) of the last call in microseconds */
double livo2_lidar_preprocess_last_kernel_us(const livo2_ctx *ctx);

/* Result of ONE residual+Jacobian+reduction pass (voxel_map.cpp:374-466 without the solve). */
typedef struct livo2_lidar_sums {
  double HtH[36];               /* Hsub_T_R_inv * Hsub, row-major 6x6 (voxel_map.cpp:466) */
  double Htz[6];                /* Hsub_T_R_inv * meas_vec (voxel_map.cpp:464) */
  double total_residual;        /* sum |dis_to_plane_| (voxel_map.cpp:399-402) */
  int32_t n_eff;                /* effct_feat_num_ */
  int32_t pad;
} livo2_lidar_sums;

/* Optional per-point outputs (any pointer may be NULL).  They describe the LAST executed iteration, i.e. what the
 * reference leaves in pv_list_ / ptpl_list_ / body_cov_list_ when StateEstimation returns. */
typedef struct livo2_lidar_points {
  int32_t *match_plane;         /* [n]    plane index of the matched plane (-1: no residual)  -> ptpl_list_ membership */
  float *dis_to_plane;          /* [n]    PointToPlane::dis_to_plane_ (signed, float32), 0 when unmatched */
  float *point_w;               /* [n][3] pv.point_w (float32-rounded world point, voxel_map.cpp:524-526) */
  int32_t *normal_plane;        /* [n]    plane whose normal_ pv.normal holds (persists across iterations), -1 = zero */
  double *var;                  /* [n][9] pv.var (voxel_map.cpp:387) */
  double *body_cov;             /* [n][9] body_cov_list_[i] */
  double *r_inv;                /* [n]    R_inv(i) of matched points (debug / parity), 0 when unmatched */
  double *h_row;                /* [n][6] Hsub.row(i) of matched points (debug / parity) */
  int64_t pinned;               /* != 0: every array above lies in page-locked memory (livo2_host_alloc_pinned / hipHostMalloc): the results are copied straight into
                                 * them instead of through the context's staging block (one host copy less) */
} livo2_lidar_points;

/* One pass for the given current iterate `cur` and prior `prop` (state_propagat); no state update. */
int livo2_lidar_iterate(livo2_ctx *ctx, const livo2_state *cur, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                        livo2_lidar_sums *sums, const livo2_lidar_points *points);

typedef struct livo2_lidar_result {
  livo2_state state;            /* posterior state_ (cov updated, voxel_map.cpp:489-490) */
  int32_t n_iters;              /* iterations executed */
  int32_t converged;            /* flg_EKF_converged of the last iteration */
  livo2_lidar_sums iter_sums[LIVO2_MAX_ITERS];
  double iter_solution[LIVO2_MAX_ITERS][LIVO2_DIM_STATE];
  double position_last[3];      /* position_last_ (voxel_map.cpp:492) */
} livo2_lidar_result;

/* The whole StateEstimation loop (voxel_map.cpp:365-500) with every iteration, the 19x19 solves, boxplus/boxminus, the
 * convergence / rematch logic and the final covariance update resident on the GPU (one result read-back at the end).
 * state_in = state_ on entry (LIVMapper.cpp:257), prop = state_propagat. */
int livo2_lidar_update(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                       livo2_lidar_result *result, const livo2_lidar_points *points);
/* Asynchronous form for pipelined callers / benchmarks: enqueue only; results stay on the device until _fetch.
 * `want` (may be NULL) only selects WHICH per-point arrays the kernels produce (non-NULL members); nothing is written
 * through it.  _fetch synchronises the stream and copies the result and the selected per-point arrays out. */
int livo2_lidar_update_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                             const livo2_lidar_points *want);
int livo2_lidar_update_fetch(livo2_ctx *ctx, livo2_lidar_result *result, const livo2_lidar_points *points);
/* Enqueue `iters` consecutive ESIKF iterations (residual kernel + solve kernel each) without convergence stopping and
 * without host interaction; used by bench.py to time the per-iteration cost. */
int livo2_lidar_iterations_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                                 int32_t iters);

/* ---- batch of frames (offline / replay: BASELINE config "batched frames") ------------------------------------------
 * n_frames independent VoxelMapManager::StateEstimation problems (src/voxel_map.cpp:338-511; call site LIVMapper.cpp:370) — each
 * with its own feats_down_body scan, state_ and state_propagat — solved against the ONE resident map snapshot.  Every ESIKF
 * iteration is a single residual grid over all frames plus one solve block per frame, so the GPU is kept full where a lone
 * 100k-point scan leaves half of it idle; frames stop individually (convergence / rematch logic per frame).  Every frame makes the same
 * discrete decisions as its own livo2_lidar_update call and agrees with it to rounding (the batched grid groups the partial sums in 64-point blocks,
 * the single-scan grid in 256-point blocks); repeated batched runs are bit-identical.  Per-point outputs are not produced in batch mode.
 * xyz: the scans concatenated, [sum(counts)][3] float32 (sensor frame); counts[f] = points of frame f (0 allowed). */
int livo2_lidar_batch_set_scans(livo2_ctx *ctx, int32_t n_frames, const float *xyz, const int32_t *counts, const livo2_lidar_cfg *cfg);
/* state_in[f], prop[f]: state_ / state_propagat of frame f; results[f] as livo2_lidar_update's. */
int livo2_lidar_batch_update(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                             livo2_lidar_result *results);
int livo2_lidar_batch_update_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg);
int livo2_lidar_batch_update_fetch(livo2_ctx *ctx, int32_t n_frames, livo2_lidar_result *results);
/* fixed iteration count, no stopping (bench.py) */
int livo2_lidar_batch_iterations_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop,
                                       const livo2_lidar_cfg *cfg, int32_t iters);

/* ---- visual photometric update ---------------------------------------------------------------------------------- */
/* vk::AbstractCamera as used on the path: cam->fx()/fy()/cx()/cy()/width()/height() already scaled (vio.cpp:45-54) and
 * cam->world2cam() (vio.cpp:1574).  distortion selects the rpg_vikit model (cam_model of the camera yaml):
 *   0  Pinhole without distortion
 *   1  Pinhole with the radial-tangential coefficients cam_d0..cam_d3 (+ d4 = r^6 term) in d[0..4]   (config/camera_pinhole.yaml)
 *   2  EquidistantCamera with k1..k4 in d[0..3]: theta_d = theta (1 + k1 theta^2 + ... + k4 theta^8)   (config/camera_fisheye_HILTI22.yaml)
 * computeProjectionJacobian (vio.cpp:189-201) ignores the distortion in every model, as in the reference. */
typedef struct livo2_cam {
  double fx, fy, cx, cy;
  double d[5];
  int32_t distortion;
  int32_t width, height;
  int32_t pad;
} livo2_cam;

typedef struct livo2_visual_cfg {
  livo2_cam cam;
  double Rcl[9], Pcl[3];        /* extrin_calib Rcl / Pcl (vio.cpp:33-37) */
  double extR[9], extT[3];      /* extrinsic_R / extrinsic_T (vio.cpp:27-31) */
  double img_point_cov;         /* vio/img_point_cov */
  int32_t patch_pyrimid_level;  /* vio/patch_pyrimid_level (L) */
  int32_t max_iterations;       /* vio/max_iterations */
  int32_t exposure_estimate_en; /* vio/exposure_estimate_en */
  int32_t inverse_composition_en; /* vio/inverse_composition_en: 1 = updateStateInverse (needs livo2_visual_set_reference) */
  int32_t mp_proc_num;          /* MP_PROC_NUM of the reference build (CMakeLists.txt:44-55; 4 on any host with more than 4 cores): the frame error of updateState
                                 * is a FLOAT OpenMP reduction (vio.cpp:1546-1554, 1634) — thread t adds the per-patch errors of its static block of patches in
                                 * index order, the per-thread sums are joined (here: in thread order; the reference: in completion order).  0 or 1 = the serial
                                 * loop (a build without MP_EN, and always updateStateInverse, which has no OpenMP loop).  Decides `error <= last_error` at ties. */
  int32_t pad;
} livo2_visual_cfg;

/* Upload the current gray image (CV_8UC1, row stride `stride` bytes) and the visual sub-map arrays the update reads
 * (SubSparseMap, reference include/vio.h:26-57): voxel_points[i]->pos_ ([M][3]), warp_patch ([M][L][64] float32, ragged
 * vector<vector<float>> flattened), search_levels ([M]), inv_expo_list ([M]).  M = total_points. */
int livo2_visual_set_frame(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const double *pos,
                           const float *warp_patch, const int32_t *search_levels, const double *inv_expo_list, int32_t M, int32_t L);

/* Inverse-compositional variant only (reference src/vio.cpp:1327-1518): the reference patch of every point, i.e. what
 * precomputeReferencePatches reads through VisualPoint::ref_patch (include/feature.h:19-54, include/frame.h): the reference gray
 * image (n_ref images of the same width/height/stride as the current frame; ref_img_idx[i] selects ref_patch->img_), ref_patch->px_
 * ([M][2]), ref_patch->f_ ([M][3]), ref_patch->T_f_w_.rotation_matrix() ([M][9] row-major) and ref_patch->pos() ([M][3]).
 * Call after livo2_visual_set_frame (same M); stays valid until the next set_frame. */
int livo2_visual_set_reference(livo2_ctx *ctx, const uint8_t *ref_imgs, int32_t n_ref, const int32_t *ref_img_idx, const double *ref_px,
                               const double *ref_f, const double *ref_R, const double *ref_pos);

/* ---- visual sub-map retrieval, selection half -----------------------------------------------------------------------
 * VIOManager::retrieveFromVisualSparseMap, src/vio.cpp:352-486 and 598-635 (raycast_en = false): the points of the current scan
 * (pv_list_.point_w) mark the voxels to look into and fill the depth image; among the visual map points filed under those voxels the
 * nearest one per image grid cell is selected; the depth-continuity test then vets each selected point.  The visual map is mirrored
 * on the device as flat arrays: pos = VisualPoint::pos_, voxel_key = the VOXEL_LOCATION the point is filed under in feat_map
 * (NULL: recomputed with insertPointIntoVoxelMap's formula, src/vio.cpp:227-236), active = (pt != nullptr && pt->obs_.size() > 0)
 * (NULL: all active).  Keys must fit 21 bits per axis (LIVO2_ERR_RANGE otherwise).
 * livo2_visual_select outputs, per grid cell c in [0, grid_n_width * grid_n_height): cell_point[c] = index of retrieve_voxel_points[c]
 * in the uploaded arrays or -1 (grid_num[c] != TYPE_MAP), cell_dist[c] = map_dist[c], cell_discontinuous[c] = 1 if the loop at
 * vio.cpp:612-635 would skip the point; per visual point: point_in_fov[i] = it passed isInFrame (voxel_in_fov of its voxel = any of its
 * points).  The host then picks ref_ftr for the surviving cells (vio.cpp:640-695) and hands them to livo2_visual_retrieve_warp — or
 * livo2_visual_retrieve_from_map (below) runs selection, choice and tail as one chain on the device.
 * Exactly equidistant points of one cell: the lowest index wins (the reference: the last one in unordered_map iteration order). */
int livo2_visual_map_upload(livo2_ctx *ctx, int32_t n_points, const double *pos, const int64_t *voxel_key, const uint8_t *active);
typedef struct livo2_select_cfg {
  livo2_cam cam;
  double R_cur[9], t_cur[3];    /* new_frame_->T_f_w_ */
  int32_t border;               /* (patch_size_half + 1) * (1 << patch_pyrimid_level), src/vio.cpp:154 */
  int32_t grid_size, grid_n_width, grid_n_height;   /* src/vio.cpp:67-78 */
  int32_t patch_size_half;
  int32_t raycast_en;           /* vio/raycast_en (LIVMapper.cpp:63, 141; off in the shipped configs): the RayCasting module of retrieveFromVisualSparseMap (vio.cpp:487-591,
                                 * rays of initializeVIO vio.cpp:80-118) runs between the nearest-point selection and the depth-continuity test.  plane_map = the
                                 * device-resident VoxelMap (livo2_map_tree_*); with a snapshot map resident the call fails (LIVO2_ERR_NO_MAP), with no LiDAR map the rays
                                 * see no planes.  At most 32768 grid cells. */
} livo2_select_cfg;
/* visual_submap->add_from_voxel_map of the last selection / retrieval that ran with raycast_en: [n][6] = plane center_, normal_ in push (= grid cell) order
 * (vio.cpp:578-583; its consumer generateVisualMapPoints is out of scope).  *n receives the number of entries, at most `capacity` are copied. */
int livo2_visual_raycast_fetch(livo2_ctx *ctx, double *center_normal, int32_t capacity, int32_t *n);
int livo2_visual_select(livo2_ctx *ctx, const double *pg_point_w, int32_t n_pg, const livo2_select_cfg *cfg, int32_t *cell_point, float *cell_dist,
                        uint8_t *cell_discontinuous, uint8_t *point_in_fov);
double livo2_visual_select_last_kernel_us(const livo2_ctx *ctx);

/* ---- visual sub-map retrieval, per-point tail ----------------------------------------------------------------------
 * VIOManager::retrieveFromVisualSparseMap keeps its grid / voxel / depth-continuity selection and reference-patch choice
 * (src/vio.cpp:352-698); for the n points it selected, this call does what the loop body does next (src/vio.cpp:698-767):
 * affine warp matrix (getWarpMatrixAffineHomography if normal_en, else getWarpMatrixAffine), getBestSearchLevel, warpAffine for
 * patch_pyrimid_level levels, getImagePatch of the current image, photometric error + optional NCC gates — and leaves the SURVIVORS,
 * in candidate order, resident as the frame of the next livo2_visual_update (it replaces livo2_visual_set_frame: pos, warp_patch,
 * search_levels, inv_expo_list are produced on the device).  Per candidate: pos = pt->pos_, normal = pt->normal_, and of its ref_ftr
 * (include/feature.h:19-54): ref_img_idx (index into ref_imgs, images of the current image's size), ref_px = px_, ref_f = f_,
 * ref_R / ref_t = T_f_w_ rotation (row-major) / translation, ref_level = level_, ref_inv_expo = inv_expo_time_, ref_id = id_ (may be
 * NULL).  warp_map (src/vio.cpp:369, 716-734, !normal_en only): the reference caches the warp under ref_ftr->id_, and id_ is the id of the
 * FRAME the feature was made in (vio.cpp:882, 961), so every candidate reuses A_cur_ref and search_level of the FIRST candidate of this call
 * whose ref_ftr has the same id_; with ref_id given that is reproduced, with NULL every candidate computes its own.
 * Outputs (each may be NULL): accepted[n] (1 = appended to visual_submap), search_level[n], error[n] (the float photometric error),
 * ncc[n], A_cur_ref[n][4] row-major, patch_wrap[n][L][64] (all candidates, for inspection).  *n_accepted = survivors.
 * cam.distortion = 1: rpg_vikit's radial-tangential model; its cam2world is OpenCV's undistortPoints (float32 point in, five fixed-point iterations in double,
 * float32 point out), restated from the published source (third party, parity unpinned).  A candidate whose 9x9 current-image window
 * leaves the image is rejected with error = +inf (the reference reads out of bounds there). */
typedef struct livo2_retrieve_cfg {
  livo2_cam cam;
  double R_cur[9], t_cur[3];    /* new_frame_->T_f_w_ (frame from world), row-major rotation */
  double inv_expo_cur;          /* state->inv_expo_time */
  int32_t patch_pyrimid_level, normal_en, ncc_en, pad;
  double ncc_thre, outlier_threshold;
} livo2_retrieve_cfg;
typedef struct livo2_retrieve_candidates {
  int32_t n, pad;
  const double *pos, *normal;
  const int32_t *ref_img_idx;
  const double *ref_px, *ref_f, *ref_R, *ref_t;
  const int32_t *ref_level;
  const double *ref_inv_expo;
  const int32_t *ref_id;        /* may be NULL; values must differ from INT32_MAX */
} livo2_retrieve_candidates;
typedef struct livo2_retrieve_out {
  int32_t *accepted, *search_level;
  float *error;
  double *ncc, *A_cur_ref;
  float *patch_wrap;
} livo2_retrieve_out;
int livo2_visual_retrieve_warp(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const uint8_t *ref_imgs, int32_t n_ref,
                               const livo2_retrieve_candidates *cand, const livo2_retrieve_cfg *cfg, livo2_retrieve_out *out, int32_t *n_accepted);
/* kernel time (k_warp_candidates + scan + gather) of the last call in microseconds (HIP events on the ctx stream) */
double livo2_visual_retrieve_last_kernel_us(const livo2_ctx *ctx);

/* ---- visual sub-map retrieval, the whole function ---------------------------------------------------------------------
 * VIOManager::retrieveFromVisualSparseMap, src/vio.cpp:352-780 (raycast_en = false), as ONE chain of launches: selection (above) ->
 * reference-patch choice (vio.cpp:644-696: is_normal_initialized_ gate; normal_en: the observation whose stored patch_ differs least from
 * the patches of the point's observations made in OTHER frames, computed once and kept in ref_patch / has_ref_patch_; !normal_en:
 * VisualPoint::getCloseViewObs, src/visual_point.cpp:57-95) -> per-point tail (above), the survivors left resident as the frame of the next
 * livo2_visual_update.  No host round trip between the stages.
 * The observations of the visual map are mirrored next to the points of livo2_visual_map_upload (call that first; same point order):
 * a CSR table over VisualPoint::obs_ in list order.  Per observation (include/feature.h:19-54): id = id_, img_idx = which of ref_imgs is
 * img_, px = px_, f = f_, R / t = T_f_w_, level = level_, inv_expo = inv_expo_time_, patch = patch_ (64 floats).  Per point
 * (include/visual_point.h): normal = normal_, normal_initialized = is_normal_initialized_, ref_patch = GLOBAL index of pt->ref_patch in the
 * observation arrays, -1 = !has_ref_patch_.  ref_imgs: n_ref gray images of the current image's width / height / stride.
 * A point all of whose >= 2 observations carry one id_ has no valid score (0/0; the reference then reads an uninitialised pointer): it is
 * skipped and keeps ref_patch = -1. */
typedef struct livo2_visual_obs {
  int32_t n_obs, n_ref;
  const int32_t *point_offset;  /* [n_points + 1] */
  const int32_t *id, *img_idx;
  const double *px, *f, *R, *t;
  const int32_t *level;
  const double *inv_expo;
  const float *patch;
  const double *normal;
  const uint8_t *normal_initialized;
  const int32_t *ref_patch;
  const uint8_t *ref_imgs;
  int32_t width, height, stride, pad;
} livo2_visual_obs;
int livo2_visual_obs_upload(livo2_ctx *ctx, const livo2_visual_obs *obs);

/* ---- incremental maintenance of that mirror (round 5) ----------------------------------------------------------------------------------
 * The reference changes its visual map by a few points per frame: generateVisualMapPoints appends new VisualPoints with one Feature each
 * (src/vio.cpp:804-895, insertPointIntoVoxelMap 227-246), updateVisualMapPoints pushes one new Feature to the FRONT of obs_ of sub-map points and drops one where
 * a list has reached 30 (vio.cpp:908-967, visual_point.cpp:35-55), updateReferencePatch rewrites normal_ / ref_patch / has_ref_patch_ (vio.cpp:969-1100) — and every
 * new Feature points at the image of the frame it was made in.  Re-flattening and re-uploading the whole map (and all reference images) for that costs tens of
 * milliseconds per frame; this call applies ONE frame's changes in O(changes): one pinned staging copy + two small launches, asynchronous on the ctx stream.
 *   new points       appended behind the resident ones: their indices are n_points_before + k (pos, voxel_key as in livo2_visual_map_upload, active); a new point
 *                    gets its obs_ list, normal_ ... through a `touched` row like any other point
 *   new observations appended behind the resident ones: global indices n_obs_before + k, members as in livo2_visual_obs; img_idx = slot of the reference image
 *   touched points   every point whose obs_ list, normal_, is_normal_initialized_, active flag or ref_patch changed: its WHOLE new state — obs_ as global
 *                    observation indices in list order (at most the stride the last livo2_visual_obs_upload chose: max(32, longest list rounded up to a power of
 *                    two); LIVO2_ERR_RANGE beyond it: re-upload), ref_patch = global index of an observation OF THAT LIST or -1.  An observation that is in no
 *                    list any more (deleteFeatureRef) simply stays unreferenced; a point that left the map is touched with active = 0 and an empty list.
 *   new image        at most one per call: stored in slot img_slot — == the current number of reference images: appended; smaller: replaces an image no
 *                    observation refers to any more (the caller's bookkeeping).  Copied through the ctx's staging: the caller's buffer is free on return.
 * ref_patch of points that are NOT touched keeps the value the device remembered (livo2_visual_retrieve_from_map writes it, and reports it in out->ref_patch).
 * Results are identical to a full livo2_visual_map_upload + livo2_visual_obs_upload of the same map (tests/test_visual_map_delta_gpu.py). */
typedef struct livo2_visual_map_delta {
  int32_t n_new_points, n_new_obs, n_touched, img_slot;
  const double *new_pos; const int64_t *new_voxel_key; const uint8_t *new_active;                                  /* [n_new_points] */
  const int32_t *obs_id, *obs_img_idx; const double *obs_px, *obs_f, *obs_R, *obs_t; const int32_t *obs_level;     /* [n_new_obs] */
  const double *obs_inv_expo; const float *obs_patch;
  const int32_t *touched_point;       /* [n_touched] point index, resident or new; each point at most once */
  const int32_t *touched_offset;      /* [n_touched + 1] CSR into touched_obs */
  const int32_t *touched_obs;         /* global observation indices, obs_ list order */
  const double *touched_normal;       /* [n_touched][3] */
  const uint8_t *touched_normal_initialized, *touched_active;
  const int32_t *touched_ref_patch;   /* [n_touched] */
  const uint8_t *img;                 /* NULL: no new reference image; else width x height x stride as in the last livo2_visual_obs_upload */
} livo2_visual_map_delta;
int livo2_visual_map_apply(livo2_ctx *ctx, const livo2_visual_map_delta *delta);
/* (n_points, n_obs, n_ref, obs stride) of the resident mirror */
int livo2_visual_map_counts(livo2_ctx *ctx, int32_t *counts4);
/* Outputs, each may be NULL.  length = grid_n_width * grid_n_height.  Per cell [length]: cell_point / cell_dist / cell_discontinuous as
 * livo2_visual_select, cell_obs = global index of the chosen ref_ftr or -1 (no candidate from this cell).  ref_patch [n_points]: pt->ref_patch
 * after the call (the device copy is updated too, so the next call sees it).  Per candidate, in grid-cell order, arrays of CAPACITY length:
 * cand_cell, and the members of `tail` as in livo2_visual_retrieve_warp (patch_wrap: [length][L][64]).  Per survivor (capacity length):
 * sub_point / sub_obs = visual_submap->voxel_points[k] (index into the uploaded points) and its ref_ftr. */
typedef struct livo2_retrieve_chain_out {
  int32_t *cell_point; float *cell_dist; uint8_t *cell_discontinuous; int32_t *cell_obs;
  int32_t *ref_patch;
  int32_t *cand_cell;
  livo2_retrieve_out tail;
  int32_t *sub_point, *sub_obs;
} livo2_retrieve_chain_out;
/* pg_point_w = NULL with n_pg = LIVO2_PG_FROM_MAP_UPDATE: `pg` is the pv_list_ the last livo2_map_tree_update[_from_scan] of this scan consumed — the scan's world
 * points at the LIO posterior, which is what the reference passes (LIVMapper.cpp:413-426 `_pv_list`, handed to processFrame at :306); they never leave the device. */
#define LIVO2_PG_FROM_MAP_UPDATE (-1)
int livo2_visual_retrieve_from_map(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const double *pg_point_w, int32_t n_pg,
                                   const livo2_select_cfg *sel, const livo2_retrieve_cfg *cfg, livo2_retrieve_chain_out *out, int32_t *n_candidates,
                                   int32_t *n_accepted);
/* kernel time of the whole chain of the last call in microseconds (HIP events on the ctx stream) */
double livo2_visual_retrieve_from_map_last_kernel_us(const livo2_ctx *ctx);

typedef struct livo2_visual_sums {
  double HtH[49];               /* H_sub^T H_sub, row-major 7x7 (vio.cpp:1660); row/col 6 zero if !exposure_estimate_en */
  double Htz[7];                /* H_sub^T z (vio.cpp:1662) */
  double err_sum;               /* sum of patch errors before the division (double accumulation) */
  float error;                  /* error / n_meas as float (vio.cpp:1636) */
  int32_t n_meas;
} livo2_visual_sums;

/* One evaluation at pyramid `level` for iterate `cur` (vio.cpp:1538-1636 + 1657-1662); no state update.
 * Optional outputs (NULL to skip): errors[M] = visual_submap->errors; z[M*64]; H_sub[M*64][7].
 * With cfg->inverse_composition_en the pass is precomputeReferencePatches(level) + one updateStateInverse evaluation (column 6 of H_sub is 0). */
int livo2_visual_iterate(livo2_ctx *ctx, int32_t level, const livo2_state *cur, const livo2_visual_cfg *cfg, livo2_visual_sums *sums,
                         float *errors, double *z, double *H_sub);

typedef struct livo2_visual_step {
  int32_t level, iteration, accepted, n_meas;
  float error;
  int32_t pad;
  double HtH[49], Htz[7], solution[LIVO2_DIM_STATE];
} livo2_visual_step;

typedef struct livo2_visual_result {
  livo2_state state;            /* posterior *state incl. cov -= G*cov (vio.cpp:800) */
  double G[LIVO2_DIM_STATE * LIVO2_DIM_STATE];   /* last G */
  double Rcw[9], Pcw[3];        /* new_frame_->T_f_w_ (vio.cpp:1690-1697) */
  int32_t n_steps;
  int32_t pad;
  livo2_visual_step steps[LIVO2_MAX_LEVELS * LIVO2_MAX_ITERS];
} livo2_visual_result;

/* The whole computeJacobianAndUpdateEKF (coarse-to-fine levels x iterations, accept/revert, solves, final covariance update)
 * resident on the GPU.  errors[M] (optional) = visual_submap->errors after the last evaluation. */
int livo2_visual_update(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg,
                        livo2_visual_result *result, float *errors);
int livo2_visual_update_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg);
int livo2_visual_update_fetch(livo2_ctx *ctx, livo2_visual_result *result, float *errors);
int livo2_visual_iterations_async(livo2_ctx *ctx, int32_t level, const livo2_state *state_in, const livo2_state *prop,
                                  const livo2_visual_cfg *cfg, int32_t iters);

/* ---- one LIO + VIO frame as ONE call (round 5) ------------------------------------------------------------------------------------------------------
 * LIVMapper::handleLIO + handleVIO for a frame whose visual sub-map is known (reference src/LIVMapper.cpp:336-482, 281-334): scan upload + per-scan precompute
 * (livo2_lidar_set_scan), StateEstimation from `prior` (state_ = state_propagat = prior, LIVMapper.cpp:256-257, 370), image + sub-map upload
 * (livo2_visual_set_frame), computeJacobianAndUpdateEKF on the SHARED state — `state` and `state_propagat` are the LiDAR posterior (LIVMapper.cpp:135-136, 256,
 * 371), handed over ON THE DEVICE — and both result blocks back in one copy.  Same kernels, same bits as the four separate calls (tests/test_c5_gpu.py); what it
 * saves is host work: one library call, one stream synchronisation and no host round trip of the state per frame instead of six calls, four synchronisations and a
 * D2H + H2D of the posterior — a 10 k-point frame is ~0.3 ms of kernels and was bound by exactly that (missing 5).
 * _async only enqueues (the caller's arrays are free on return: everything is staged through pinned blocks of the ctx); up to two frames may be in flight per ctx:
 * _fetch returns them in order.  A third _async without a _fetch fails with LIVO2_ERR_INVALID.  The map must be resident (livo2_map_upload / livo2_map_tree_*). */
typedef struct livo2_visual_reference {   /* the arguments of livo2_visual_set_reference, for the M points of the frame's sub-map */
  const uint8_t *ref_imgs;      /* [n_ref][height][stride] */
  int32_t n_ref, pad;
  const int32_t *ref_img_idx;   /* [M] */
  const double *ref_px, *ref_f, *ref_R, *ref_pos;   /* [M][2], [M][3], [M][9], [M][3] */
} livo2_visual_reference;
typedef struct livo2_frame_in {
  const float *xyz;             /* [n_points][3] down-sampled body-frame scan */
  int32_t n_points, M, L, width, height, stride;
  const livo2_state *prior;
  const livo2_lidar_cfg *lidar_cfg;
  const livo2_visual_cfg *visual_cfg;
  const uint8_t *img;           /* [height][stride] */
  const double *pos;            /* the visual sub-map, as livo2_visual_set_frame takes it */
  const float *warp_patch;
  const int32_t *search_levels;
  const double *inv_expo_list;
  /* round 6: vio/inverse_composition_en in the frame call too (updateStateInverse, src/vio.cpp:1327-1518) — required iff visual_cfg->inverse_composition_en, else
   * NULL.  The reference patches are copied before the call returns (this form waits for the previous frame of the context: one frame in flight). */
  const livo2_visual_reference *reference;
} livo2_frame_in;
int livo2_frame_update_async(livo2_ctx *ctx, const livo2_frame_in *frame);
/* (visual->steps: entries [0, n_steps) are written; the 128 - n_steps entries behind them are left as the caller's struct holds them — a frame fills ~15 of them,
 * and the slot crosses the link once per frame) */
int livo2_frame_update_fetch(livo2_ctx *ctx, livo2_lidar_result *lidar, livo2_visual_result *visual);
int livo2_frame_update(livo2_ctx *ctx, const livo2_frame_in *frame, livo2_lidar_result *lidar, livo2_visual_result *visual);


/* ---- batch of frames, visual (offline / replay: BASELINE config "batched frames") ------------------------------------------------------
 * n_frames independent VIOManager::computeJacobianAndUpdateEKF problems (src/vio.cpp:784-802; call site vio.cpp:1810) — each with its own gray image,
 * visual sub-map (SubSparseMap arrays as in livo2_visual_set_frame), *state and *state_propagat — advanced in lockstep: every (level, iteration) is ONE residual
 * grid over the patches of all frames plus one solve block per frame; a frame whose level has ended (EKF_end, vio.cpp:1675-1681) drops out of the level's later
 * grids.  Every frame makes the same decisions and produces the same bits as its own livo2_visual_update call (same kernels, same per-frame reduction order).
 * imgs: n_frames images of width x height with row stride `stride`, back to back; pos / warp_patch / search_levels / inv_expo_list: the frames' arrays
 * concatenated; counts[f] = total_points of frame f (0 allowed).
 * cfg->inverse_composition_en (round 6): precomputeReferencePatches + updateStateInverse per frame (src/vio.cpp:1327-1518) in the same lockstep; the frames' reference
 * patches come from livo2_visual_batch_set_references (after _set_frames, which forgets them): the arguments of livo2_visual_set_reference concatenated like the sub-map
 * arrays; ref_imgs = the frames' reference images back to back (n_ref[f] of them for frame f, size and stride of the batch's images), ref_img_idx[i] indexes the
 * images of point i's own frame. */
int livo2_visual_batch_set_frames(livo2_ctx *ctx, int32_t n_frames, const uint8_t *imgs, int32_t width, int32_t height, int32_t stride, const double *pos,
                                  const float *warp_patch, const int32_t *search_levels, const double *inv_expo_list, const int32_t *counts, int32_t L);
int livo2_visual_batch_set_references(livo2_ctx *ctx, int32_t n_frames, const uint8_t *ref_imgs, const int32_t *n_ref, const int32_t *ref_img_idx, const double *ref_px,
                                      const double *ref_f, const double *ref_R, const double *ref_pos);
int livo2_visual_batch_update(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg,
                              livo2_visual_result *results);
int livo2_visual_batch_update_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg);
int livo2_visual_batch_update_fetch(livo2_ctx *ctx, int32_t n_frames, livo2_visual_result *results);
/* fixed iteration count at one level, always accepted, no stopping (bench.py) */
int livo2_visual_batch_iterations_async(livo2_ctx *ctx, int32_t n_frames, int32_t level, const livo2_state *state_in, const livo2_state *prop,
                                        const livo2_visual_cfg *cfg, int32_t iters);

/* ---- one LiDAR-inertial frame on the device (N4 -> N3 -> the update) ------------------------------------------------------
 * What LIVMapper::handleLIO does with a synchronised (scan, IMU) package before the map update (src/LIVMapper.cpp:342-377):
 * ImuProcess::UndistortPcl forward loop (src/IMU_Processing.cpp:322-445) -> state_propagat and IMUpose; its backward loop (494-539) and
 * downSizeFilterSurf (LIVMapper.cpp:351-352) -> feats_down_body; voxelmap_manager->StateEstimation(state_propagat) with state_ = state_propagat
 * (LIVMapper.cpp:366-370).  One call: the previous state, the IMU steps and the raw scan go in, the posterior comes out; state_propagat,
 * IMUpose and feats_down_body stay on the device (state_propagat / poses [n_steps] are copied out only if non-NULL), one host round trip
 * (the number of voxel-grid leaves) instead of the three calls' six.  first_pose = the Pose6D the reference pushes before the loop
 * (set_pose6d(0, acc_s_last, angvel_last, vel, pos, rot), IMU_Processing.cpp:312-313), so that IMUpose = {first_pose, one pose per step}.
 * The result equals, bit for bit, livo2_imu_propagate + livo2_lidar_preprocess_scan(poses = IMUpose, rot_end / pos_end of state_propagat) +
 * livo2_lidar_update(state_propagat, state_propagat) called in sequence; afterwards the scan is resident like after livo2_lidar_set_scan. */
int livo2_lio_frame(livo2_ctx *ctx, const livo2_state *state_in, const livo2_imu_step *steps, int32_t n_steps, const livo2_imu_cfg *imu_cfg,
                    const livo2_imu_pose *first_pose, const float *xyz, const float *curvature, int32_t n, double leaf_size, const livo2_lidar_cfg *cfg,
                    livo2_state *state_propagat, livo2_imu_pose *poses, int32_t *n_down, livo2_lidar_result *result);

/* ---- ESIKF solve alone (voxel_map.cpp:468-474 with sign=+1,k=6,meas_cov_scale=1; vio.cpp:1661-1669 with sign=-1,k=7,
 * meas_cov_scale=img_point_cov) — runs the same device kernel the update loops use. ------------------------------ */
int livo2_esikf_solve(livo2_ctx *ctx, const double *HtH /*k*k*/, const double *Htz /*k*/, int32_t k, double meas_cov_scale, int32_t sign,
                      const livo2_state *cur, const livo2_state *prop, livo2_state *out_state, double *solution /*19*/,
                      double *G /*361*/);

#ifdef __cplusplus
}
#endif
#endif /* LIVO2_HIP_H */
