// C++ host shim: the reference's operator surface for the ESIKF measurement update, implemented on top of the C ABI
// (include/livo2_hip.h).  Class / member names mirror the reference so that call sites read the same:
//   VoxelMapManager::StateEstimation(StatesGroup&)            reference include/voxel_map.h:229, src/voxel_map.cpp:338-511
//   VIOManager::computeJacobianAndUpdateEKF(img)              reference include/vio.h:153,     src/vio.cpp:784-802
//   StatesGroup, pointWithVar, PointToPlane, VoxelPlane, VOXEL_LOCATION, VoxelOctoTree, VoxelMapConfig, SubSparseMap, VisualPoint
// Eigen / PCL / OpenCV are not available in this image, so vectors and matrices are plain row-major arrays (V3D = double[3],
// M3D = double[9]); INTEGRATION.md shows the same shim against the reference's Eigen types.
// No arithmetic of the update happens here: the shim only flattens the pointer-based VoxelMap into the index-based snapshot the
// ABI takes, moves data across, and scatters the results back into the members the rest of LIVMapper reads.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/livo2_hip.h"

namespace livo2 {

typedef std::array<double, 3> V3D;
typedef std::array<double, 9> M3D;          // row-major

struct StatesGroup {                        // reference include/common_lib.h:126-223
  M3D rot_end{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  V3D pos_end{}, vel_end{}, bias_g{}, bias_a{}, gravity{};
  double inv_expo_time = 1.0;
  std::array<double, LIVO2_DIM_STATE * LIVO2_DIM_STATE> cov{};
  StatesGroup() {
    for (int i = 0; i < LIVO2_DIM_STATE; i++) cov[i * LIVO2_DIM_STATE + i] = 0.01;
    cov[6 * LIVO2_DIM_STATE + 6] = 0.00001;
    for (int i = 10; i < 19; i++) cov[i * LIVO2_DIM_STATE + i] = 0.00001;
  }
  void to_abi(livo2_state &s) const {
    std::memcpy(s.rot, rot_end.data(), 72); std::memcpy(s.pos, pos_end.data(), 24); s.inv_expo = inv_expo_time;
    std::memcpy(s.vel, vel_end.data(), 24); std::memcpy(s.bg, bias_g.data(), 24); std::memcpy(s.ba, bias_a.data(), 24);
    std::memcpy(s.grav, gravity.data(), 24); std::memcpy(s.cov, cov.data(), sizeof(s.cov));
  }
  void from_abi(const livo2_state &s) {
    std::memcpy(rot_end.data(), s.rot, 72); std::memcpy(pos_end.data(), s.pos, 24); inv_expo_time = s.inv_expo;
    std::memcpy(vel_end.data(), s.vel, 24); std::memcpy(bias_g.data(), s.bg, 24); std::memcpy(bias_a.data(), s.ba, 24);
    std::memcpy(gravity.data(), s.grav, 24); std::memcpy(cov.data(), s.cov, sizeof(s.cov));
  }
};

struct PointXYZ { float x, y, z; };         // the fields of pcl::PointXYZINormal the path reads

struct pointWithVar {                       // reference include/common_lib.h:102-123
  V3D point_b{}, point_i{}, point_w{}, normal{};
  M3D var_nostate{}, body_var{}, var{}, point_crossmat{};
};

struct PointToPlane {                       // reference include/voxel_map.h:54-67
  V3D point_b_{}, point_w_{}, normal_{}, center_{};
  std::array<double, 36> plane_var_{};
  M3D body_cov_{};
  int layer_ = 0;
  double d_ = 0, eigen_value_ = 0;
  bool is_valid_ = false;
  float dis_to_plane_ = 0;
};

struct VoxelPlane {                         // reference include/voxel_map.h:69-94 (fields the update reads and init_plane writes)
  V3D center_{}, normal_{}, y_normal_{}, x_normal_{};
  M3D covariance_{};
  std::array<double, 36> plane_var_{};
  float radius_ = 0, d_ = 0, min_eigen_value_ = 1, mid_eigen_value_ = 1, max_eigen_value_ = 1;
  int points_size_ = 0;
  bool is_plane_ = false, is_update_ = false;
};

struct VOXEL_LOCATION {                     // reference include/voxel_map.h:96-104
  int64_t x, y, z;
  VOXEL_LOCATION(int64_t vx = 0, int64_t vy = 0, int64_t vz = 0) : x(vx), y(vy), z(vz) {}
  bool operator==(const VOXEL_LOCATION &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelLocationHash {                  // reference include/voxel_map.h:109-117
  size_t operator()(const VOXEL_LOCATION &s) const {
    return (size_t)(((((s.z) * 116101) % 10000000000LL + (s.y)) * 116101) % 10000000000LL + (s.x));
  }
};

struct VoxelOctoTree {                      // reference include/voxel_map.h:129-183
  std::vector<pointWithVar> temp_points_;
  float planer_threshold_ = 0.0025f;
  VoxelPlane *plane_ptr_ = new VoxelPlane;
  int layer_ = 0;
  VoxelOctoTree *leaves_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double voxel_center_[3] = {0, 0, 0};
  float quater_length_ = 0;
  // map-maintenance state (UpdateOctoTree / init_octo_tree / cut_octo_tree, reference src/voxel_map.cpp:137-290)
  int octo_state_ = 0, new_points_ = 0, points_size_threshold_ = 5, update_size_threshold_ = 5, max_points_num_ = 50, max_layer_ = 2;
  bool init_octo_ = false, update_enable_ = true;
  std::vector<int> layer_init_num_;
  VoxelOctoTree() = default;
  VoxelOctoTree(int max_layer, int layer, int points_size_threshold, int max_points_num, float planer_threshold)    // voxel_map.h:151-163
      : planer_threshold_(planer_threshold), layer_(layer), points_size_threshold_(points_size_threshold), max_points_num_(max_points_num), max_layer_(max_layer) {}
  VoxelOctoTree *clone() const;             // deep copy of the subtree
  ~VoxelOctoTree() { for (auto *l : leaves_) delete l; delete plane_ptr_; }
  VoxelOctoTree(const VoxelOctoTree &) = delete;
  VoxelOctoTree &operator=(const VoxelOctoTree &) = delete;
};

struct VoxelMapConfig {                     // reference include/voxel_map.h:35-52
  double max_voxel_size_ = 0.5;
  int max_layer_ = 2, max_iterations_ = 5;
  double beam_err_ = 0.05, dept_err_ = 0.02, sigma_num_ = 3;
  std::vector<int> layer_init_num_{5, 5, 5, 5, 5};
  int max_points_num_ = 50;
  double planner_threshold_ = 0.0025;       // local_map / min_eigen_value
  bool map_sliding_en = false;              // local_map/map_sliding_en, half_map_size, sliding_thresh (reference src/voxel_map.cpp:50-52)
  int half_map_size = 100;
  double sliding_thresh = 8;
};

class Device {                              // one GPU + stream, shared by the two managers of a LIVMapper
public:
  explicit Device(int device = 0) { if (livo2_ctx_create(device, &ctx_) != LIVO2_OK) throw std::runtime_error("livo2_ctx_create failed: no gfx950 device"); }
  ~Device() { livo2_ctx_destroy(ctx_); }
  Device(const Device &) = delete;
  Device &operator=(const Device &) = delete;
  livo2_ctx *ctx() const { return ctx_; }
  void check(int rc) const { if (rc != LIVO2_OK) throw std::runtime_error(std::string("livo2: ") + livo2_last_error(ctx_)); }
private:
  livo2_ctx *ctx_ = nullptr;
};

class VoxelMapManager {                     // reference include/voxel_map.h:187-256
public:
  VoxelMapConfig config_setting_;
  std::unordered_map<VOXEL_LOCATION, VoxelOctoTree *, VoxelLocationHash> voxel_map_;
  std::vector<PointXYZ> feats_down_body_;
  int feats_down_size_ = 0, effct_feat_num_ = 0;
  M3D extR_{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  V3D extT_{};
  StatesGroup state_;
  V3D position_last_{};
  V3D last_slide_position{};                // reference include/voxel_map.h:209
  std::array<double, 4> geoQuat_{{0, 0, 0, 1}};   // geometry_msgs::Quaternion x, y, z, w (reference include/voxel_map.h:212, src/voxel_map.cpp:493)
  std::vector<M3D> cross_mat_list_, body_cov_list_;
  std::vector<pointWithVar> pv_list_;
  std::vector<PointToPlane> ptpl_list_;

  explicit VoxelMapManager(Device &dev) : dev_(dev) {}
  ~VoxelMapManager() { for (auto &kv : voxel_map_) delete kv.second; livo2_host_free_pinned(pin_); }

  // Call after BuildVoxelMap / UpdateVoxelMap changed the tree structure (new voxels / nodes).  Plane-only refreshes can go
  // through RefreshPlanes instead.
  void MarkMapDirty() { map_dirty_ = true; }
  // VoxelPlane::is_update_ planes (reference include/voxel_map.h:86): refresh their records in place.
  void RefreshPlanes(const std::vector<const VoxelPlane *> &planes);

  void StateEstimation(StatesGroup &state_propagat);   // reference src/voxel_map.cpp:338-511

  // ImuProcess::UndistortPcl's backward loop + downSizeFilterSurf (reference src/IMU_Processing.cpp:494-539, src/LIVMapper.cpp:351-352) on
  // the device: raw scan (x, y, z, curvature[ms], sorted by curvature) + IMUpose + the scan-end pose in, feats_down_body_ out — and the
  // filtered cloud stays resident, so the next StateEstimation uploads no scan.
  void UndistortAndDownsample(const std::vector<PointXYZ> &pcl_wait_proc, const std::vector<float> &curvature, const std::vector<livo2_imu_pose> &IMUpose,
                              const StatesGroup &state_end, double filter_size_surf);

  // VoxelOctoTree::init_plane(temp_points_, plane_ptr_) (reference src/voxel_map.cpp:55-135) for many voxels in one device call; planes
  // that are in the uploaded snapshot and stay planes are refreshed in place on the device, a changed is_plane_ marks the map dirty.
  void FitPlanes(const std::vector<VoxelOctoTree *> &voxels);

  // BuildVoxelMap / UpdateVoxelMap (reference src/voxel_map.cpp:532-591, 609-641) with the reference's own schedule — every point goes
  // through UpdateOctoTree in input order, re-fits every update_size_threshold_ points, subdivision of non-planar voxels — but with every
  // init_plane evaluated on the device: root voxels are independent, so each touched root is replayed on a scratch copy until all the
  // fits it asks for are known; the fits requested in one sweep over the roots go to the device as ONE livo2_plane_fit_batch call.
  // input_points carry point_w and var (BuildVoxelMap's own var computation, voxel_map.cpp:546-553, is the caller's here).
  void BuildVoxelMap(const std::vector<pointWithVar> &input_points);
  void UpdateVoxelMap(const std::vector<pointWithVar> &input_points);
  int last_fit_rounds_ = 0, last_fit_count_ = 0;      // device batches / plane fits of the last Build / Update call
  // Device-resident map (livo2_map_tree_*): with device_map_ set BEFORE BuildVoxelMap, voxel_map_ stays empty on the host — the octree, its temp_points_ and the
  // plane table live on the GPU, BuildVoxelMap / UpdateVoxelMap feed them, StateEstimation reads them, and ptpl_list_ / pv.normal are filled from the device
  // plane rows of the matches.  UpdateVoxelMapFromPosterior() is LIVMapper.cpp:413-424 in one call: pv_list_[i].point_w / .var of the posterior state_ are
  // formed on the device from the scan that StateEstimation left resident, and fed to UpdateVoxelMap there (nothing crosses PCIe but the state).
  bool device_map_ = false;
  // per-point members StateEstimation leaves behind (pv_list_, ptpl_list_, body_cov_list_, cross_mat_list_: 168 B / point D2H + ~900 B / point of host structs).
  // With device_map_ their consumers UpdateVoxelMap (LIVMapper.cpp:413-424 -> UpdateVoxelMapFromPosterior) run on the device; a caller that neither publishes
  // (publish_effect_world, LIVMapper.cpp:1308) nor feeds generateVisualMapPoints (vio.cpp:811) can switch them off: only state_, effct_feat_num_, position_last_, geoQuat_ come back.
  bool host_point_lists_ = true;
  int fill_threads_ = 16;                   // host threads that write the per-point lists (one per 2048 points, at most this many)
  int device_map_max_roots_ = 300000;
  void UpdateVoxelMapFromPosterior();
  // async_map_update_: UpdateVoxelMapFromPosterior() only ENQUEUES the octree update (with host_point_lists_ it still reads pv_list_ at the posterior back, from the
  // context's stream, while the octree update runs) — pv_list_ at the posterior on the context's
  // stream, the octree update on its second stream (livo2_map_tree_update_from_scan_async) — so that handleVIO's retrieval and update (which do not read the LiDAR
  // voxel map unless raycast_en) run beside it.  JoinMapUpdate() collects it (pool counters, growth, errors); the next StateEstimation does so implicitly.
  bool async_map_update_ = false;
  void JoinMapUpdate();
  // reference src/voxel_map.cpp:924-972 (LIVMapper.cpp:430-433 calls it when config_setting_.map_sliding_en): root voxels outside the box around
  // position_last_ are deleted — on the device tree with device_map_ (livo2_map_tree_slide), in voxel_map_ otherwise.  Returns the count, -1 below sliding_thresh.
  int mapSliding();
  double last_map_kernel_us_ = 0;

private:
  void FlattenAndUpload();
  void maintain(const std::vector<pointWithVar> &input_points, bool build);
  Device &dev_;
  bool map_dirty_ = true;
  bool scan_resident_ = false;              // set by UndistortAndDownsample, consumed by StateEstimation
  std::unordered_map<const VoxelPlane *, int32_t> plane_index_;
  std::vector<const VoxelPlane *> plane_by_index_;
  std::vector<int> plane_layer_;
  void *pin_ = nullptr; size_t pin_bytes_ = 0;       // page-locked receive buffer of the per-point outputs
  livo2_state posterior_{}; bool posterior_valid_ = false;      // state_ as StateEstimation left it (= what is still on the device)
};

// ---- IMU ------------------------------------------------------------------------------------------------------------------
class ImuProcess {                          // reference include/IMU_Processing.h (members the forward propagation reads / writes)
public:
  V3D cov_acc{{0.1, 0.1, 0.1}}, cov_gyr{{0.1, 0.1, 0.1}}, cov_bias_gyr{{0.1, 0.1, 0.1}}, cov_bias_acc{{0.1, 0.1, 0.1}}, mean_acc{{0, 0, -1.0}};   // IMU_Processing.cpp:19-24
  double cov_inv_expo = 0.2;
  bool ba_bg_est_en = true, gravity_est_en = true, exposure_estimate_en = true;
  bool imu_time_init = false;               // IMU_Processing.cpp:28, 305-310
  std::vector<livo2_imu_pose> IMUpose;      // Pose6D list (msg/Pose6D.msg)
  explicit ImuProcess(Device &dev) : dev_(dev) {}
  // The forward loop of UndistortPcl (reference src/IMU_Processing.cpp:322-445) for the steps its time-stamp logic produced: one step per
  // processed IMU pair = averaged raw gyro / accelerometer sample, dt, offs_t.  Appends one Pose6D per step to IMUpose and advances state_inout.
  void ForwardPropagate(StatesGroup &state_inout, const std::vector<livo2_imu_step> &steps);
private:
  Device &dev_;
};

// ---- visual ---------------------------------------------------------------------------------------------------------------
struct Feature {                            // reference include/feature.h:19-54 — what precomputeReferencePatches (vio.cpp:1346-1356) and the retrieval loop (vio.cpp:644-767) read
  int id_ = 0;                              // id of the frame the feature was made in (vio.cpp:882, 961)
  const float *patch_ = nullptr;            // patch_size_total floats, level 0
  const uint8_t *img_ = nullptr;            // reference gray image (same size / stride as the current frame)
  std::array<double, 2> px_{};
  V3D f_{};
  M3D R_f_w{};                              // T_f_w_.rotation_matrix()
  V3D t_f_w{};                              // T_f_w_.translation()
  V3D pos{};                                // pos(): camera centre of the reference frame
  int level_ = 0;
  double inv_expo_time_ = 1.0;
  int32_t mirror_index_ = -1;               // shim: global observation index on the device, -1 = not mirrored yet
};
struct VisualPoint {                        // reference include/visual_point.h:23-46
  V3D pos_{}, normal_{};
  Feature *ref_patch = nullptr;
  std::vector<Feature *> obs_;              // (std::list in the reference; the order is what matters: addFrameRef pushes to the FRONT, visual_point.cpp:35-38)
  bool is_normal_initialized_ = true, has_ref_patch_ = false;
  int32_t mirror_index_ = -1;               // shim: index of the point in the device mirror, -1 = not mirrored yet
  bool mirror_dirty_ = false;               // shim: waiting in the next delta (VIOManager::markPointDirty)
};

struct VOXEL_POINTS {                       // reference include/vio.h:59-69
  std::vector<VisualPoint *> voxel_points;
  int count = 0;
};

struct SubSparseMap {                       // reference include/vio.h:26-57
  std::vector<float> errors;
  std::vector<std::vector<float>> warp_patch;
  std::vector<int> search_levels;
  std::vector<VisualPoint *> voxel_points;
  std::vector<double> inv_expo_list;
  std::vector<pointWithVar> add_from_voxel_map;   // raycast_en: planes of the LiDAR map the rays ended at (point_w = center_, normal = normal_; reference include/vio.h:34, src/vio.cpp:578-583)
};

struct GrayImage { const uint8_t *data = nullptr; int cols = 0, rows = 0, step = 0; };   // cv::Mat CV_8UC1 view

class VIOManager {                          // reference include/vio.h:89-186 (members the update reads / writes)
public:
  StatesGroup *state = nullptr, *state_propagat = nullptr;
  M3D Rcl{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, extR{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, Rcw{};
  V3D Pcl{}, extT{}, Pcw{};
  double fx = 0, fy = 0, cx = 0, cy = 0;
  // cam_d0..cam_d4 of config/camera_pinhole.yaml (vk::PinholeCamera's radial-tangential model); distortion_en = vikit's `distortion_` (any coefficient non-zero)
  double cam_d[5] = {0, 0, 0, 0, 0};
  bool distortion_en = false;
  // cam_model "EquidistantCamera" of config/camera_fisheye_HILTI22.yaml (vk::EquidistantCamera: k1..k4 in cam_d[0..3]); takes precedence over distortion_en
  bool equidistant_en = false;
  int cam_distortion_model() const { return equidistant_en ? 2 : (distortion_en ? 1 : 0); }
  int width = 0, height = 0;
  int patch_pyrimid_level = 4, patch_size = 8, max_iterations = 5, total_points = 0;
  double img_point_cov = 100;
  bool exposure_estimate_en = true, inverse_composition_en = false, normal_en = true, ncc_en = false;
  bool raycast_en = false;                      // vio/raycast_en (reference include/vio.h:100): the RayCasting module runs inside retrieveFromVisualSparseMap / selectFromVisualSparseMap;
                                                // plane_map = the device-resident VoxelMap of the same Device (VoxelMapManager::device_map_)
  int mp_proc_num = 4;                          // MP_PROC_NUM of the reference build (CMakeLists.txt:44-55): partition of the float error reduction (vio.cpp:1554)
  double compute_jacobian_time = 0, update_ekf_time = 0;   // reference include/vio.h:114: seconds inside the residual / solve kernels of the last update when
  bool kernel_times_en = false;                 // kernel_times_en (HIP event pair per launch, off by default: the events cost what they measure); else 0 (vio.cpp:788)
  double ncc_thre = 0, outlier_threshold = 1000;
  M3D R_f_w_new{{1, 0, 0, 0, 1, 0, 0, 0, 1}};   // new_frame_->T_f_w_ (reference include/frame.h:34)
  V3D t_f_w_new{};
  SubSparseMap *visual_submap = nullptr;
  std::array<double, LIVO2_DIM_STATE * LIVO2_DIM_STATE> G{}, H_T_H{};

  explicit VIOManager(Device &dev) : dev_(dev) {}
  void setImuToLidarExtrinsic(const V3D &transl, const M3D &rot) { extT = transl; extR = rot; }      // reference src/vio.cpp:27-31
  void setLidarToCameraExtrinsic(const M3D &R, const V3D &P) { Rcl = R; Pcl = P; }                     // reference src/vio.cpp:33-37
  void computeJacobianAndUpdateEKF(const GrayImage &img);                                              // reference src/vio.cpp:784-802
  // Rcw / Pcw and new_frame_->T_f_w_ from a state (processFrame calls it with *state before the retrieval, vio.cpp:1799-1800)  reference src/vio.cpp:1690-1697
  void updateFrameState(const StatesGroup &s);

  // Tail of retrieveFromVisualSparseMap (reference src/vio.cpp:698-767) for the points the host-side selection kept: warp, search level,
  // patches, gates on the device; fills visual_submap (voxel_points, search_levels, errors, inv_expo_list) and total_points, and leaves
  // the survivors resident as the frame of the next computeJacobianAndUpdateEKF (which then uploads nothing but the states).
  struct Candidate { VisualPoint *pt; Feature *ref_ftr; };
  void warpAndGateCandidates(const GrayImage &img, const std::vector<Candidate> &cands);

  // Selection half of retrieveFromVisualSparseMap (reference src/vio.cpp:385-486, 598-635): mirrors feat_map on the device (whenever
  // feat_map_dirty_ is set), runs the scan-voxel / depth-image / nearest-point-per-cell / depth-continuity stages there and returns, in grid
  // order, the points the loop at vio.cpp:598 would go on with (retrieve_voxel_points[i] of the TYPE_MAP cells that pass the depth test).
  std::unordered_map<VOXEL_LOCATION, VOXEL_POINTS *, VoxelLocationHash> feat_map;
  int grid_size = 5, grid_n_width = 0, grid_n_height = 17, border = 80;     // reference src/vio.cpp:67-78, 154
  bool feat_map_dirty_ = true;
  std::vector<float> map_dist;              // per grid cell, as the reference keeps it
  std::vector<VisualPoint *> selectFromVisualSparseMap(const std::vector<pointWithVar> &pg);

  // The whole retrieveFromVisualSparseMap (reference src/vio.cpp:352-780, raycast_en = false; call site src/vio.cpp:1808) as one chain on the
  // device: selection, reference-patch choice (pt->ref_patch / has_ref_patch_ are written back for the points it went through), warp and gates.
  // Fills visual_submap (voxel_points, search_levels, errors, inv_expo_list), map_dist and total_points; the survivors stay resident as the frame
  // of the next computeJacobianAndUpdateEKF.  feat_map_dirty_ must be set whenever feat_map, an obs_ list, a normal or a ref_patch changed on the host.
  void retrieveFromVisualSparseMap(const GrayImage &img, const std::vector<pointWithVar> &pg);
  // retrieveFromVisualSparseMap(img, pg); computeJacobianAndUpdateEKF(img); — the two calls VIOManager::processFrame makes back to back (reference src/vio.cpp:1808,
  // 1810) — as ONE member with the same results in the same members (tests/test_live_chain_gpu.py): the update is enqueued as soon as the retrieval chain has returned
  // its counts, visual_submap's host lists are built while it runs on the GPU (round 6: ~0.06 ms of pointer chasing per avia frame leaves the critical path).
  void retrieveAndUpdate(const GrayImage &img, const std::vector<pointWithVar> &pg);
  std::chrono::steady_clock::time_point update_enqueued_at_{};      // when the last update went onto the stream (a caller that times the two stages splits retrieveAndUpdate here)
  // pg_from_map_update_: `pg` is ignored and the scan's posterior world points are read where VoxelMapManager::UpdateVoxelMap[FromPosterior] (same Device) left
  // them on the GPU (LIVMapper.cpp:413-426 `_pv_list`, :306) — the lean form: pv_list_ never exists on the host.
  bool pg_from_map_update_ = false;
  // mirror feat_map (+ observations, reference images) on the device now instead of inside the next retrieveFromVisualSparseMap (the cost of a visual-map change,
  // which the reference's map maintenance — out of scope — causes once per frame)
  void syncFeatMap(const GrayImage &img);
  // ---- incremental mirror (round 5).  The reference changes feat_map by a few points per frame (generateVisualMapPoints vio.cpp:804-895, updateVisualMapPoints
  // 908-967, updateReferencePatch 969-1100 — the maintenance itself is out of scope); three hooks in those functions keep the device mirror in step in O(changes)
  // (livo2_visual_map_apply) instead of the re-flatten + re-upload that feat_map_dirty_ = true asks for:
  void insertPointIntoVoxelMap(VisualPoint *pt_new);     // reference src/vio.cpp:227-246: files the point under its voxel in feat_map AND queues it for the next sync
  void markPointDirty(VisualPoint *pt);                  // after addFrameRef / deleteFeatureRef / a change of normal_, is_normal_initialized_, ref_patch, has_ref_patch_
  void erasePointFromVoxelMap(VisualPoint *pt);          // the point leaves feat_map (and the mirror: its slot stays, inactive); the caller may delete it afterwards
  size_t mirroredPoints() const { return mirror_.size(); }
  size_t mirroredObservations() const { return obs_mirror_.size(); }
  int delta_syncs_ = 0, full_syncs_ = 0;                 // how the mirror was brought up to date so far

private:
  Device &dev_;
  bool frame_resident_ = false;             // set by warpAndGateCandidates, consumed by computeJacobianAndUpdateEKF
  std::vector<VisualPoint *> mirror_;       // device index -> VisualPoint*
  std::vector<Feature *> obs_mirror_;       // device observation index -> Feature*
  bool obs_resident_ = false;
  std::vector<VisualPoint *> pending_new_, pending_dirty_;       // queued by the hooks above, consumed by applyPendingDelta
  std::vector<VOXEL_LOCATION> pending_new_keys_;
  std::vector<int32_t> pending_removed_;
  // device image slot -> Feature::img_.  LIFETIME (advisor, round 5): a slot is identified by the POINTER, and slots are only ever appended — an image buffer must stay
  // alive and keep its address for as long as a Feature that names it is in the map (the reference keeps every reference frame's cv::Mat alive through its Features'
  // shared pixels, feature.h:19-54); a caller that frees one and gets a new frame at the same address must mark the map dirty (feat_map_dirty_ = true: full upload).
  std::vector<const uint8_t *> img_slots_;
  // stages of retrieveFromVisualSparseMap / computeJacobianAndUpdateEKF (livo2_host.cpp) and the chain's outputs they hand on
  bool retrieveChain(const GrayImage &img, const std::vector<pointWithVar> &pg);
  void retrieveLists();
  void retrieveRaycast();
  void updateEnqueue(const GrayImage &img);
  void updateFetch();
  std::vector<int32_t> r_cell_, r_cobs_, r_acc_, r_sl_, r_cand_cell_;
  std::vector<float> r_err_;
  int32_t r_n_cand_ = 0;
  std::chrono::steady_clock::time_point u_t0_{};
  void applyPendingDelta(const GrayImage &img);
  void mirrorFeatMap(bool with_obs, const GrayImage *img);
  void gridSetup();
  livo2_select_cfg selectCfg() const;
};

} // namespace livo2
