// Device-resident VoxelMap (second half): the octree bookkeeping of map maintenance on the GPU.
//   VoxelMapManager::BuildVoxelMap / UpdateVoxelMap        reference src/voxel_map.cpp:532-591, 609-641
//   VoxelOctoTree::init_octo_tree / cut_octo_tree / UpdateOctoTree   reference src/voxel_map.cpp:137-290
//   VoxelOctoTree::init_plane                               reference src/voxel_map.cpp:55-135 (plane_fit_core, map_kernels.hpp)
//   pv_list_ of the posterior (point_w, var)                reference src/LIVMapper.cpp:413-423
// The map lives in five pools in HBM — octree nodes, the temp_points_ of every node, the 256-B plane records, the candidate lists of
// the non-plane roots, the 2-choice cuckoo table of root voxels — and k_lidar_residual reads the last three directly: no host octree, no
// re-flatten, no snapshot upload between frames.
//
// Schedule: the reference feeds the points of a scan ONE BY ONE, in input order, through UpdateOctoTree; what a point does depends on what
// the earlier points of the same root voxel did (counters, re-fit every 5 points, subdivision, freezing at max_points_num_).  Root voxels
// are independent of each other, so the scan is sorted by root voxel (stable: input order inside a voxel) and every touched root is
// walked serially by ONE group of 8 lanes — the reference's state machine, statement by statement — while the init_plane calls inside
// are evaluated by the 8 lanes together.  Tree shape and decisions are those of the serial loop; the fitted planes agree with the CPU
// evaluation to the tolerance of plane_fit_core (summation order).
#pragma once
#include "map_kernels.hpp"

#define MT_LPG 8                 // lanes per root voxel
#define MT_SLAB_MIN 52           // smallest region size: max_points_num_ 50 (every shipped config but HILTI22) + the point that trips the limit + 1.  The size in use is
                                 // MapTreeArgs::slab = max(MT_SLAB_MIN, max_points_num + 2), fixed at livo2_map_tree_create (config/HILTI22.yaml:66 has max_points_num 100)
#define MT_NO_KEY INT32_MIN     // key of an empty root bucket (voxel keys are 21-bit per axis)
#define MT_STACK (LIVO2_MAX_LAYER + 2)
enum { MTC_NODES = 0, MTC_POINTS = 1, MTC_PLANES = 2, MTC_CAND = 3, MTC_OVERFLOW = 4, MTC_ERROR = 5, MTC_DIRTY = 6, MTC_ROOTS = 7, MTC_COUNT = 8,
       // beyond what livo2_map_tree_stats reports: tops of the free stacks that mapSliding fills (k_mt_slide) and the allocators drain, root voxels it removed
       // (4 KB away from the bump counters: those take an atomic per allocation from thousands of groups, and any access that shares their memory channel queues
       //  behind them — three loads of a NEIGHBOURING line per new root made k_mt_roots 25x slower)
       MTC_FREE_NODES = 1024, MTC_FREE_PLANES = 1025, MTC_FREE_SLABS = 1026, MTC_REMOVED = 1027, MTC_PENDING_SLABS = 1028, MTC_TOTAL = 1056 };
enum { MTE_NODES = 1, MTE_POINTS = 2, MTE_PLANES = 4, MTE_CAND = 8, MTE_TABLE = 16, MTE_RANGE = 32, MTE_REGION = 64 };

struct __attribute__((aligned(128))) DevNode {      // VoxelOctoTree (reference include/voxel_map.h:129-183), 128 B
  double center[3];              // voxel_center_
  float quarter;                 // quater_length_
  int32_t layer;                 // layer_
  int32_t init_octo, is_plane, update_enable, octo_state;     // init_octo_, plane_ptr_->is_plane_, update_enable_, octo_state_
  int32_t new_points, n_temp, pts_off, pts_cap;                // new_points_, temp_points_.size(), its region in the point pool
  int32_t plane, root, cand_begin, cand_cap;                   // row of its plane record (-1: never was a plane); root node; (roots) candidate-list region
  int32_t child[8];              // leaves_
  int32_t key[3];                // (roots) VOXEL_LOCATION
  int32_t dirty;                 // (roots) queued for k_mt_emit in this update
};
static_assert(sizeof(DevNode) == 128, "DevNode is two cache lines of 64 B");

struct MapTreeArgs {
  DevNode *nodes; double *pool_pw, *pool_var;        // point pool: [cap][3], [cap][9]
  double *planes, *planes_hot, *cand; PlaneAux *plane_aux, *cand_aux; RootSlot *slots;     // master records [.][32]; hot words [.][16] by plane row / candidate position
  int32_t *counters;                                 // [MTC_COUNT]
  int32_t *dirty_list, *overflow_list;
  int32_t cap_nodes, cap_points, cap_planes, cap_cand, cap_overflow;
  uint32_t mask, seed1, seed2;
  double voxel_size_d; float voxel_size_f, planer_threshold;
  int32_t max_layer, max_points_num, update_size_threshold;
  int32_t layer_init_num[LIVO2_MAX_LAYER + 1];
  // the frame being fed
  const double *in_pw, *in_var;                      // [n][3], [n][9] in input order
  const int32_t *order;                              // [n] input index of the k-th point in (root voxel, input order) order
  const unsigned long long *skeys;                   // [n] sorted packed keys
  const int32_t *seg_head, *seg_slot;                // [n] head flags / exclusive scan (segment number at heads)
  int32_t *seg_begin, *seg_root;                     // [n_seg + 1], [n_seg]
  int32_t n, build;
  int32_t slab;                                      // points per default temp_points_ region (see MT_SLAB_MIN)
  int32_t may_pop;                                   // host-side note: bit 0 / 1 / 2 = the free stack of node ids / plane rows / 52-point regions is non-empty
  int32_t spread;                                    // k_mt_update / k_mt_emit: lanes per root = MT_LPG * spread, the first MT_LPG work (set per launch)
};

// The free stacks live behind the counters in the SAME allocation — [MTC_TOTAL counters][node ids: cap_nodes][plane rows: cap_planes][regions: cap_points /
// slab + 1] — so that they need no kernel arguments of their own: k_mt_update sits at the register limit, and four more pointers
// in its argument block cost a 70 % longer build (measured).
__device__ __forceinline__ int32_t *mt_free_nodes(const MapTreeArgs &a) { return a.counters + MTC_TOTAL; }
__device__ __forceinline__ int32_t *mt_free_planes(const MapTreeArgs &a) { return a.counters + MTC_TOTAL + a.cap_nodes; }
__device__ __forceinline__ int32_t *mt_free_slabs(const MapTreeArgs &a) { return a.counters + MTC_TOTAL + a.cap_nodes + a.cap_planes; }

// point regions come in whole multiples of `slab` points, so that a freed region of any size goes back as `slab`-sized pieces
__device__ __forceinline__ int mt_round_slab(const MapTreeArgs &a, int points) { return ((points + a.slab - 1) / a.slab) * a.slab; }
__device__ __forceinline__ int grp_first(int v) { return __shfl(v, (threadIdx.x & 63) & ~(MT_LPG - 1), 64); }
template <int LPG> __device__ __forceinline__ int grp_first_n(int v) { return __shfl(v, (threadIdx.x & 63) & ~(LPG - 1), 64); }
__device__ __forceinline__ void mt_error(const MapTreeArgs &a, int bit) { atomicOr(&a.counters[MTC_ERROR], bit); }

// packed 3 x 21-bit voxel key (offset 2^20) <-> VOXEL_LOCATION
__device__ __forceinline__ void mt_unpack(unsigned long long k, int32_t key[3]) {
  key[0] = (int32_t)((k >> 42) & 0x1fffff) - (1 << 20); key[1] = (int32_t)((k >> 21) & 0x1fffff) - (1 << 20); key[2] = (int32_t)(k & 0x1fffff) - (1 << 20);
}

// ---- pv_list_ of the posterior, root-voxel key of every point -----------------------------------------------------------------------------
// LIVMapper.cpp:413-423: point_w = float32(R p_i + t) widened; var = (R extR) C_b (R extR)^T + (-X) P_rr (-X)^T + P_tt, X = cross_mat_list_[i]
// (the z-patched IMU-frame point's skew matrix, voxel_map.cpp:352-358).  The scan is resident in Morton order; outputs go to INPUT order.
struct MapPvArgs {
  const float *x, *y, *z; const double *cb; const int32_t *perm; int32_t n;
  double ER[9], Et[3];
  double *out_pw, *out_var;
};
__global__ void __launch_bounds__(256) k_mt_pv_from_scan(MapPvArgs a, const livo2_state *__restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double *R = st->rot, *t = st->pos, *cov = st->cov;
  const double plx = a.x[i], ply = a.y[i], plz = a.z[i];
  const int o = a.perm[i];
  double pi[3], pc[3];
#pragma unroll
  for (int j = 0; j < 3; j++) pi[j] = ((a.ER[j * 3] * plx + a.ER[j * 3 + 1] * ply) + a.ER[j * 3 + 2] * plz) + a.Et[j];
#pragma unroll
  for (int j = 0; j < 3; j++) pc[j] = pi[j];
  if (plz == 0) {
#pragma unroll
    for (int j = 0; j < 3; j++) pc[j] = ((a.ER[j * 3] * plx + a.ER[j * 3 + 1] * ply) + a.ER[j * 3 + 2] * 0.001) + a.Et[j];
  }
#pragma unroll
  for (int j = 0; j < 3; j++) a.out_pw[(size_t)o * 3 + j] = (double)(float)(((R[j * 3] * pi[0] + R[j * 3 + 1] * pi[1]) + R[j * 3 + 2] * pi[2]) + t[j]);
  double RE[9];
  mat3_mul(R, a.ER, RE);
  const double Cbf[9] = {a.cb[i], a.cb[(size_t)a.n + i], a.cb[(size_t)2 * a.n + i], a.cb[(size_t)a.n + i], a.cb[(size_t)3 * a.n + i], a.cb[(size_t)4 * a.n + i],
                         a.cb[(size_t)2 * a.n + i], a.cb[(size_t)4 * a.n + i], a.cb[(size_t)5 * a.n + i]};
  double T[9], RC[9], XP[9], XPX[9];
  mat3_mul(RE, Cbf, T); mat3_mul_Bt(T, RE, RC);
  const double nX[9] = {-0.0, pc[2], -pc[1], -pc[2], -0.0, pc[0], pc[1], -pc[0], -0.0};          // -point_crossmat
  const double Prr[9] = {cov[0], cov[1], cov[2], cov[DS], cov[DS + 1], cov[DS + 2], cov[2 * DS], cov[2 * DS + 1], cov[2 * DS + 2]};
  mat3_mul(nX, Prr, XP); mat3_mul_Bt(XP, nX, XPX);
#pragma unroll
  for (int e = 0; e < 9; e++) { const int r = e / 3, c = e % 3; a.out_var[(size_t)o * 9 + e] = (RC[e] + XPX[e]) + cov[(3 + r) * DS + 3 + c]; }
}

// root voxel of every point (voxel_map.cpp:559-567 / 617-625: float division, -1 for negatives, truncation), packed for the sort
__global__ void __launch_bounds__(256) k_mt_keys(const double *__restrict__ pw, int n, float voxel_size, unsigned long long *__restrict__ keys, int32_t *__restrict__ idx, int32_t *__restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long k = 0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float l = (float)(pw[(size_t)i * 3 + j] / (double)voxel_size);      // double / float -> double, narrowed to the float loc_xyz
    if (l < 0) l = (float)((double)l - 1.0);
    if (!(l > -1048000.f && l < 1048000.f)) { bad = true; l = 0.f; }
    const long long v = (long long)l + (1 << 20);
    k = (k << 21) | (unsigned long long)(v & 0x1fffff);
  }
  if (bad) atomicOr(&counters[MTC_ERROR], MTE_RANGE);
  keys[i] = k; idx[i] = i;
}
__global__ void __launch_bounds__(256) k_mt_heads(const unsigned long long *__restrict__ skeys, int n, int32_t *__restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_mt_segments(const int32_t *__restrict__ head, const int32_t *__restrict__ slot, int n, int32_t *__restrict__ seg_begin, int32_t *__restrict__ n_seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (head[i]) seg_begin[slot[i]] = i;
  if (i == n - 1) { const int ns = slot[i] + (head[i] ? 1 : 0); seg_begin[ns] = n; *n_seg = ns; }
}

// ---- allocation ----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mt_alloc(const MapTreeArgs &a, int which, int count, int cap, int errbit) {      // one lane calls; -1 when the pool is exhausted
  const int at = atomicAdd(&a.counters[which], count);
  if (at + count > cap) { atomicSub(&a.counters[which], count); mt_error(a, errbit); return -1; }      // rolled back: the pool stays usable for requests that still fit
  return at;
}
// what mapSliding released comes back first (the stacks are only pushed by k_mt_slide, never while an update runs: a pop is one atomic)
__device__ __forceinline__ int mt_pop(const MapTreeArgs &a, int which, const int32_t *stack) {
  if (a.counters[which] <= 0) return -1;
  const int top = atomicSub(&a.counters[which], 1);
  if (top > 0) return stack[top - 1];
  atomicAdd(&a.counters[which], 1);
  return -1;
}
// Recycling is a property of the LAUNCH (template parameter): k_mt_update sits at the register limit and any code on its allocation paths — a flag test, a call,
// an inlined pop — costs it ~60 % (measured: 14.5 -> 23-27 ms for a 1.2 M-point build), so the plain kernels allocate by bumping only and the host starts the
// <true> variants when the free stacks hold something AND a pool is more than half used (livo2_api.hip, map_tree_run).
template <bool RECYCLE> __device__ __forceinline__ int mt_take(const MapTreeArgs &a, int which_free, const int32_t *stack, int which, int count, int cap, int errbit) {
  if (RECYCLE) { const int id = mt_pop(a, which_free, stack); if (id >= 0) return id; }
  return mt_alloc(a, which, count, cap, errbit);
}
template <bool RECYCLE> __device__ __forceinline__ int mt_alloc_node(const MapTreeArgs &a) { return mt_take<RECYCLE>(a, MTC_FREE_NODES, mt_free_nodes(a), MTC_NODES, 1, a.cap_nodes, MTE_NODES); }
template <bool RECYCLE> __device__ __forceinline__ int mt_alloc_plane(const MapTreeArgs &a) { return mt_take<RECYCLE>(a, MTC_FREE_PLANES, mt_free_planes(a), MTC_PLANES, 1, a.cap_planes, MTE_PLANES); }
template <bool RECYCLE> __device__ __forceinline__ int mt_alloc_region(const MapTreeArgs &a, int cap) {
  if (RECYCLE && cap == a.slab) { const int off = mt_pop(a, MTC_FREE_SLABS, mt_free_slabs(a)); if (off >= 0) return off; }
  return mt_alloc(a, MTC_POINTS, cap, a.cap_points, MTE_POINTS);
}
__device__ __forceinline__ void mt_init_node(DevNode &nd, const double c[3], float quarter, int layer, int root, int pts_off, int pts_cap) {
  nd.center[0] = c[0]; nd.center[1] = c[1]; nd.center[2] = c[2]; nd.quarter = quarter; nd.layer = layer;
  nd.init_octo = 0; nd.is_plane = 0; nd.update_enable = 1; nd.octo_state = 0;
  nd.new_points = 0; nd.n_temp = 0; nd.pts_off = pts_off; nd.pts_cap = pts_cap;
  nd.plane = -1; nd.root = root; nd.cand_begin = 0; nd.cand_cap = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) nd.child[k] = -1;
  nd.key[0] = nd.key[1] = nd.key[2] = 0; nd.dirty = 0;
}

// ---- roots: lookup or creation --------------------------------------------------------------------------------------------------------
template <bool RECYCLE> __global__ void __launch_bounds__(256) k_mt_roots(MapTreeArgs a, const int32_t *__restrict__ n_seg_p) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= *n_seg_p) return;
  const int b = a.seg_begin[g], e = a.seg_begin[g + 1];
  int32_t key[3];
  mt_unpack(a.skeys[b], key);
  const uint32_t h1 = voxel_hash(key[0], key[1], key[2], a.seed1) & a.mask, h2 = voxel_hash(key[0], key[1], key[2], a.seed2) & a.mask;
  const uint32_t hh[2] = {h1, h2};
  // Another thread of this launch may be half-way through claiming one of the two buckets (val == -3, key / pad not yet written): such a bucket holds either the
  // empty-slot sentinel key (MT_NO_KEY: set at creation and by k_mt_slide, no voxel has it) or the claimer's key, which is not ours (one thread per distinct key) —
  // it can never be mistaken for our root.  A finished bucket (val >= -2) was published behind a fence; the value word is read first, with acquire semantics.
  for (int t = 0; t < 2; t++) {
    const RootSlot &s = a.slots[hh[t]];
    const int32_t val = __hip_atomic_load(&s.val, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (val != -1 && val != -3 && s.kx == key[0] && s.ky == key[1] && s.kz == key[2]) { a.seg_root[g] = s.pad; return; }
  }
  // new root voxel (voxel_map.cpp:574-583 / 630-637)
  const int cap = a.build ? mt_round_slab(a, e - b + 1) : a.slab;
  const int id = mt_alloc_node<RECYCLE>(a);
  const int off = mt_alloc_region<RECYCLE>(a, cap);
  if (id < 0 || off < 0) { a.seg_root[g] = -1; return; }
  atomicAdd(&a.counters[MTC_ROOTS], 1);
  DevNode nd;
  const double c[3] = {(0.5 + (double)key[0]) * (double)a.voxel_size_f, (0.5 + (double)key[1]) * (double)a.voxel_size_f, (0.5 + (double)key[2]) * (double)a.voxel_size_f};
  mt_init_node(nd, c, a.voxel_size_f / 4, 0, id, off, cap);
  nd.key[0] = key[0]; nd.key[1] = key[1]; nd.key[2] = key[2];
  a.nodes[id] = nd;
  a.seg_root[g] = id;
  RootSlot ns{};
  ns.kx = key[0]; ns.ky = key[1]; ns.kz = key[2]; ns.val = -2; ns.center[0] = c[0]; ns.center[1] = c[1]; ns.center[2] = c[2]; ns.quarter = nd.quarter;
  ns.cand_begin = 0; ns.cand_count = 0; ns.pad = id;
  for (int t = 0; t < 2; t++) {
    const uint32_t h = hh[t];
    if (atomicCAS(&a.slots[h].val, -1, -3) == -1) {            // claimed (no lookup runs concurrently with an update)
      RootSlot *d = &a.slots[h];
      d->kx = ns.kx; d->ky = ns.ky; d->kz = ns.kz; d->center[0] = c[0]; d->center[1] = c[1]; d->center[2] = c[2]; d->quarter = ns.quarter;
      d->cand_begin = 0; d->cand_count = 0; d->pad = id;
      __threadfence();
      d->val = -2;
      return;
    }
  }
  const int at = atomicAdd(&a.counters[MTC_OVERFLOW], 1);       // both buckets taken: the serial cuckoo pass places it
  if (at < a.cap_overflow) a.overflow_list[at] = id; else mt_error(a, MTE_TABLE);
}
// both buckets of a new root were occupied: standard cuckoo eviction, one thread (a few per cent of the new roots at load factor <= 1/8)
__global__ void __launch_bounds__(64) k_mt_overflow(MapTreeArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = min(a.counters[MTC_OVERFLOW], a.cap_overflow);
  for (int k = 0; k < n; k++) {
    const DevNode &nd = a.nodes[a.overflow_list[k]];
    RootSlot cur{};
    cur.kx = nd.key[0]; cur.ky = nd.key[1]; cur.kz = nd.key[2]; cur.val = -2; cur.center[0] = nd.center[0]; cur.center[1] = nd.center[1]; cur.center[2] = nd.center[2];
    cur.quarter = nd.quarter; cur.cand_begin = 0; cur.cand_count = 0; cur.pad = a.overflow_list[k];
    uint32_t h = voxel_hash(cur.kx, cur.ky, cur.kz, a.seed1) & a.mask;
    bool placed = false;
    for (int kick = 0; kick < 512; kick++) {
      if (a.slots[h].val == -1) { a.slots[h] = cur; placed = true; break; }
      const RootSlot t = a.slots[h]; a.slots[h] = cur; cur = t;      // evict the resident, move it to its other bucket
      const uint32_t ha = voxel_hash(cur.kx, cur.ky, cur.kz, a.seed1) & a.mask, hb = voxel_hash(cur.kx, cur.ky, cur.kz, a.seed2) & a.mask;
      h = (h == ha) ? hb : ha;
    }
    if (!placed) mt_error(a, MTE_TABLE);
  }
  a.counters[MTC_OVERFLOW] = 0;
}

// ---- the octree state machine, one group of LPG (MT_LPG, or a whole wave) lanes per touched root --------------------------------------------------------------
template <bool RECYCLE, int LPG = MT_LPG> struct MtGroup {
  const MapTreeArgs &a; int lane;
  __device__ MtGroup(const MapTreeArgs &a_, int l) : a(a_), lane(l) {}
  // (round 6: the previous point's path kept per depth in LDS — a root's consecutive points mostly end in the same leaf, and every node on the way is a dependent 128-B
  //  load — changed nothing measurable once a wave ran one root only: 328 / 351 / 316 us without, 335 / 345 / 362 us with, whole chain at 15 k points.  Not kept.)
  __device__ DevNode get_node(int, int id) { return a.nodes[id]; }
  __device__ void put_node(int, int id, const DevNode &n) { if (lane == 0) a.nodes[id] = n; }

  __device__ int thr(int layer) const { return a.layer_init_num[layer <= LIVO2_MAX_LAYER ? layer : LIVO2_MAX_LAYER]; }
  // temp_points_.push_back(pv)
  __device__ bool push(DevNode &n, const double *pw, const double *var) {
    if (n.n_temp >= n.pts_cap) { mt_error(a, MTE_REGION); return false; }
    const size_t at = (size_t)n.pts_off + n.n_temp;
    for (int q = lane; q < 12; q += LPG) { if (q < 3) a.pool_pw[at * 3 + q] = pw[q]; else a.pool_var[at * 9 + q - 3] = var[q - 3]; }
    n.n_temp++;
    return true;
  }
  // std::vector<pointWithVar>().swap(temp_points_); update_enable_ = false
  // A frozen node never stores a point again: its slab-sized region goes back to the pool.  Here it is only MARKED (pts_cap = -slab): an atomic per freeze on one
  // counter would serialise behind the allocators' (a build freezes most of its planes); k_mt_collect gathers the marks of the touched roots after the update.
  __device__ void freeze(DevNode &n) { n.update_enable = 0; n.n_temp = 0; if (n.pts_cap > 0) n.pts_cap = -n.pts_cap; }
  // init_plane(temp_points_, plane_ptr_): fit + the packed record the residual kernel reads
  __device__ void fit(DevNode &n) {
    wave_sync();                                              // the pushes of this group are visible to its 8 lanes
    FitRes R;
    plane_fit_core<LPG, true>(a.pool_pw, a.pool_var, n.pts_off, n.pts_off + n.n_temp, lane, a.planer_threshold, R);      // upper triangle of plane_var_ only
    n.is_plane = R.is_plane ? 1 : 0;
    if (!R.is_plane) return;
    if (n.plane < 0) { int row = 0; if (lane == 0) row = mt_alloc_plane<RECYCLE>(a); n.plane = grp_first_n<LPG>(row); if (n.plane < 0) { n.is_plane = 0; return; } }
    double *rec = a.planes + (size_t)n.plane * PLANE_REC_DOUBLES;
    const double nrm[3] = {R.vmin[0], R.vmin[1], R.vmin[2]};
    const float radius = (float)sqrt(R.ev_max);
    const float dd = (float)(-((nrm[0] * R.c[0] + nrm[1] * R.c[1]) + nrm[2] * R.c[2]));
    auto put = [&](int k, double v) { if (lane == k % LPG) rec[k] = v; };
#pragma unroll
    for (int k = 0; k < 3; k++) { put(k, nrm[k]); put(3 + k, R.c[k]); }
    double S[21];
    {
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int u = r; u < 6; u++) { S[q] = R.pv[r * 6 + u]; put(6 + q, S[q]); q++; }
    }
    put(27, __builtin_bit_cast(double, make_float2(dd, radius)));
    double hot[PLANE_HOT_DOUBLES];
    plane_hot_words(nrm, R.c, S, hot);
#pragma unroll
    for (int k = 0; k < PLANE_HOT_DOUBLES; k++) if (lane == k % LPG) a.planes_hot[(size_t)n.plane * PLANE_HOT_DOUBLES + k] = hot[k];
    if (lane == 0) { PlaneAux x; x.d = dd; x.radius = radius; x.meta = n.plane | (n.layer << CAND_LAYER_SHIFT); x.pad = 0; a.plane_aux[n.plane] = x; }
    put(28, __builtin_bit_cast(double, make_int2(n.layer, 0)));       // PointToPlane::layer_ of a match (livo2_map_tree_read_planes)
#pragma unroll
    for (int k = 29; k < 32; k++) put(k, 0.0);
  }
  __device__ int octant(const DevNode &n, const double *pw) const { return 4 * (pw[0] > n.center[0] ? 1 : 0) + 2 * (pw[1] > n.center[1] ? 1 : 0) + (pw[2] > n.center[2] ? 1 : 0); }
  // leaves_[leafnum] = new VoxelOctoTree(...) (voxel_map.cpp:179-186 / 255-262); returns its id (-1: pool exhausted)
  __device__ int new_leaf(const DevNode &n, int leafnum, int cap) {
    int id = -1, off = -1;
    if (lane == 0) { id = mt_alloc_node<RECYCLE>(a); off = mt_alloc_region<RECYCLE>(a, cap); }
    id = grp_first_n<LPG>(id); off = grp_first_n<LPG>(off);
    if (id < 0 || off < 0) return -1;
    DevNode l;
    const int xyz[3] = {(leafnum >> 2) & 1, (leafnum >> 1) & 1, leafnum & 1};
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = n.center[k] + (double)((2 * xyz[k] - 1) * n.quarter);       // int * float -> float, widened for the add
    mt_init_node(l, c, n.quarter / 2, n.layer + 1, n.root, off, cap);
    if (lane == 0) a.nodes[id] = l;
    return id;
  }

  // init_octo_tree body after the size test / the per-leaf body of cut_octo_tree (voxel_map.cpp:141-159, 194-214): fit, then plane bookkeeping or subdivision.
  // `parent_new_points` is the counter the reference zeroes when it freezes a LEAF inside cut_octo_tree (it writes the PARENT's new_points_, voxel_map.cpp:204).
  __device__ void init_node(int id, DevNode &n, int *parent_new_points) {
    // explicit stack instead of the recursion init -> cut -> (leaf) init -> cut ...
    struct Frame { int id; DevNode n; int next; int *parent_np; };
    // depth <= max_layer + 1; frames live in scratch memory (rare path)
    Frame st[MT_STACK];
    int sp = 0;
    st[0].id = id; st[0].n = n; st[0].next = -1; st[0].parent_np = parent_new_points;
    while (sp >= 0) {
      Frame &f = st[sp];
      if (f.next < 0) {                                       // entering: init_plane + decision
        fit(f.n);
        if (f.n.is_plane) {
          f.n.octo_state = 0;
          if (f.n.n_temp > a.max_points_num) { freeze(f.n); if (f.parent_np) *f.parent_np = 0; else f.n.new_points = 0; }
          f.n.init_octo = 1; f.n.new_points = 0;
          if (lane == 0) a.nodes[f.id] = f.n;
          if (sp == 0) n = f.n;
          sp--;
          continue;
        }
        f.n.octo_state = 1;
        // cut_octo_tree (voxel_map.cpp:163-217)
        if (f.n.layer >= a.max_layer) { f.n.octo_state = 0; f.n.init_octo = 1; f.n.new_points = 0; if (lane == 0) a.nodes[f.id] = f.n; if (sp == 0) n = f.n; sp--; continue; }
        wave_sync();
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < f.n.n_temp; i++) {
          const double *pw = a.pool_pw + ((size_t)f.n.pts_off + i) * 3;
          const double p[3] = {pw[0], pw[1], pw[2]};
          const int o = octant(f.n, p);
#pragma unroll
          for (int k = 0; k < 8; k++) cnt[k] += (o == k) ? 1 : 0;
        }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (cnt[k] > 0 && f.n.child[k] < 0) { f.n.child[k] = new_leaf(f.n, k, mt_round_slab(a, cnt[k] + 1)); if (f.n.child[k] < 0) ok = false; }
        if (!ok) { if (lane == 0) a.nodes[f.id] = f.n; if (sp == 0) n = f.n; return; }
        wave_sync();                                          // the new leaves are in memory for every lane
        // copy the points to the leaves in list order (temp_points_.push_back, new_points_++)
        int cur[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = (f.n.child[k] >= 0) ? a.nodes[f.n.child[k]].n_temp : 0;
        for (int i = 0; i < f.n.n_temp; i++) {
          const size_t src = (size_t)f.n.pts_off + i;
          const double p[3] = {a.pool_pw[src * 3], a.pool_pw[src * 3 + 1], a.pool_pw[src * 3 + 2]};
          const int o = octant(f.n, p);
          int at = 0, cid = 0;
#pragma unroll
          for (int k = 0; k < 8; k++) if (o == k) { at = cur[k]; cur[k]++; cid = f.n.child[k]; }
          const DevNode &cn = a.nodes[cid];
          if (at >= cn.pts_cap) { mt_error(a, MTE_REGION); continue; }
          const size_t dst = (size_t)cn.pts_off + at;
          for (int q = lane; q < 12; q += LPG) { if (q < 3) a.pool_pw[dst * 3 + q] = a.pool_pw[src * 3 + q]; else a.pool_var[dst * 9 + q - 3] = a.pool_var[src * 9 + q - 3]; }
        }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 8; k++) if (f.n.child[k] >= 0 && cnt[k] > 0) { DevNode *cn = &a.nodes[f.n.child[k]]; cn->n_temp = cur[k]; cn->new_points += cnt[k]; }
        }
        wave_sync();
        f.next = 0;
      }
      // the loop over the eight leaves (voxel_map.cpp:190-216)
      bool descended = false;
      while (f.next < 8) {
        const int k = f.next++;
        int cid = -1;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q == k) cid = f.n.child[q];
        if (cid < 0) continue;
        DevNode cn = a.nodes[cid];
        if (cn.n_temp > thr(cn.layer)) {
          st[sp + 1].id = cid; st[sp + 1].n = cn; st[sp + 1].next = -1; st[sp + 1].parent_np = &f.n.new_points;
          sp++; descended = true;
          break;
        }
      }
      if (descended) continue;
      f.n.init_octo = 1; f.n.new_points = 0;
      if (lane == 0) a.nodes[f.id] = f.n;
      if (sp == 0) n = f.n;
      sp--;
    }
  }

  // init_octo_tree (voxel_map.cpp:137-161)
  __device__ void init_octo_tree(int id, DevNode &n) {
    if (n.n_temp > thr(n.layer)) init_node(id, n, nullptr);
  }

  // UpdateOctoTree(pv) from the root (voxel_map.cpp:219-290); the recursion only ever descends, so it is a loop
  __device__ void update(int root, const double *pw, const double *var) {
    int id = root;
    for (int depth = 0; depth <= LIVO2_MAX_LAYER + 1; depth++) {
      DevNode n = get_node(depth, id);
      if (!n.init_octo) {
        n.new_points++;
        if (!push(n, pw, var)) { put_node(depth, id, n); return; }
        if (n.n_temp > thr(n.layer)) init_octo_tree(id, n);
        put_node(depth, id, n);
        return;
      }
      if (!n.is_plane && n.layer < a.max_layer) {
        const int leafnum = octant(n, pw);
        int cid = -1;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q == leafnum) cid = n.child[q];
        if (cid < 0) {
          cid = new_leaf(n, leafnum, a.slab);
          if (cid < 0) return;
#pragma unroll
          for (int q = 0; q < 8; q++) if (q == leafnum) n.child[q] = cid;
          put_node(depth, id, n);
          wave_sync();
        }
        id = cid;
        continue;
      }
      // a plane (voxel_map.cpp:233-251), or a non-plane node at max_layer_ that keeps collecting (:268-286): the same statements but for the freeze test (>= / >).  ONE
      // copy of them, so that the groups of a wave that re-fit in this iteration do it together whichever of the two they are (the fit is the long part)
      if (n.update_enable) {
        const int limit = n.is_plane ? a.max_points_num : a.max_points_num + 1;
        n.new_points++;
        push(n, pw, var);
        if (n.new_points > a.update_size_threshold) { fit(n); n.new_points = 0; }
        if (n.n_temp >= limit) { freeze(n); n.new_points = 0; }
        put_node(depth, id, n);
      }
      return;
    }
  }
};

// Two waves per SIMD: the state machine with the plane fit inlined wants every register there is (256 VGPRs + as many accumulation registers as spill space, 2.8 KB of
// scratch per lane on top: ONE wave per SIMD); held to 256 it runs two, and the chain at 94 k points takes 446-487 us instead of 511-541 (15 k: unchanged).  Three
// or four (128 VGPRs) do not compile with this toolchain ("Subtarget requires even aligned vector registers" on a scratch reload).  round 6
#ifndef MT_UPDATE_WPE
#define MT_UPDATE_WPE 2
#endif
template <bool RECYCLE, int LPG> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT_UPDATE_WPE))) k_mt_update(MapTreeArgs a, const int32_t *__restrict__ n_seg_p) {
  // Lanes per root.  LPG = MT_LPG (8): a.spread (1, 2 or 4) groups' worth of lanes per root, of which the first 8 work.  The eight state machines of a wave DIVERGE
  // (one descends, one re-fits a plane, one cuts a node ...) and a wave runs them one after the other — every root stops at every re-fit of its seven neighbours.
  // LPG = 64: ONE root per wave, and the re-fit (init_plane: one Jacobi eigen-decomposition + a 6x3 Jacobian product per point, the long part) spread over all 64
  // lanes.  Round 6, whole update chain at 15 k points (1 900 touched roots): 503-579 us with eight roots to a wave, 320-367 us with one (8 of 64 lanes working); at
  // 94 k points (12 000 roots) eight times the waves no longer fit the device and the chain gets slower (517 -> 926 us), so the host picks the form from the number of
  // roots the previous update touched (map_tree_run; profiles/r06_map_update_spread.txt).
  // (Also tried: a wave each only for the roots with >= 12 / 24 / 48 points of the frame, the others eight to a wave — two lists from one more small kernel.  At 94 k
  //  points 537-585 us against 511-533 us; at 15 k only the 12-point threshold, i.e. nearly every root alone, reached the one-root-per-wave time: it is not the ONE
  //  busiest root that the neighbours' re-fits hold up, it is every root.)
  const int width = LPG * (LPG == MT_LPG ? max(a.spread, 1) : 1);
  if ((int)(threadIdx.x & (width - 1)) >= LPG) return;
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) / width, lane = threadIdx.x & (LPG - 1);
  if (g >= *n_seg_p) return;
  const int root = a.seg_root[g];
  if (root < 0) return;
  const int b = a.seg_begin[g], e = a.seg_begin[g + 1];
  MtGroup<RECYCLE, LPG> G(a, lane);
  if (a.build) {                                              // BuildVoxelMap: every point of the voxel first, then init_octo_tree (voxel_map.cpp:568-590)
    DevNode n = a.nodes[root];
    for (int k = b; k < e; k++) {
      const int i = a.order[k];
      const double pw[3] = {a.in_pw[(size_t)i * 3], a.in_pw[(size_t)i * 3 + 1], a.in_pw[(size_t)i * 3 + 2]};
      double var[9];
#pragma unroll
      for (int q = 0; q < 9; q++) var[q] = a.in_var[(size_t)i * 9 + q];
      G.push(n, pw, var); n.new_points++;
    }
    if (lane == 0) a.nodes[root] = n;
    wave_sync();
    G.init_octo_tree(root, n);
    if (lane == 0) a.nodes[root] = n;
  } else {
    for (int k = b; k < e; k++) {
      const int i = a.order[k];
      const double pw[3] = {a.in_pw[(size_t)i * 3], a.in_pw[(size_t)i * 3 + 1], a.in_pw[(size_t)i * 3 + 2]};
      double var[9];
#pragma unroll
      for (int q = 0; q < 9; q++) var[q] = a.in_var[(size_t)i * 9 + q];
      G.update(root, pw, var);
      wave_sync();
    }
  }
  if (lane == 0) { const int at = atomicAdd(&a.counters[MTC_DIRTY], 1); a.dirty_list[at] = root; }      // (one segment per root: no duplicates)
}

// The regions of the nodes that froze during the update just finished (marked pts_cap = -MT_SLAB) go onto the free stack: one thread per touched root walks its
// subtree; the pushes of a wave share one atomic.  Runs after k_mt_update / before anything pops.
__global__ void __launch_bounds__(256) k_mt_collect(MapTreeArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_dirty = a.counters[MTC_DIRTY];
  int st[8 * (LIVO2_MAX_LAYER + 1) + 1];
  int sp = -1;
  if (t < n_dirty) { sp = 0; st[0] = a.dirty_list[t]; }
  while (__any(sp >= 0)) {
    int off = -1, pieces = 0;
    if (sp >= 0) {
      const int id = st[sp--];
      DevNode &nd = a.nodes[id];
      for (int k = 0; k < 8; k++) if (nd.child[k] >= 0 && sp + 1 < (int)(sizeof(st) / sizeof(st[0]))) st[++sp] = nd.child[k];
      if (nd.pts_cap < 0) { off = nd.pts_off; pieces = -nd.pts_cap / a.slab; nd.pts_cap = 0; nd.pts_off = 0; }
    }
    while (true) {                                               // one slab-sized piece per lane and turn (a build-time region is a few pieces)
      const unsigned long long m = __ballot(pieces > 0);
      if (!m) break;
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&a.counters[MTC_FREE_SLABS], __popcll(m));
      base = __shfl(base, leader, 64);
      if (pieces > 0) { mt_free_slabs(a)[base + __popcll(m & ((1ull << lane) - 1))] = off; off += a.slab; pieces--; }
    }
  }
}

// ---- VoxelMapManager::mapSliding / clearMemOutOfMap (voxel_map.cpp:924-972): every root voxel whose key lies outside the box is deleted with its subtree ----------
// One thread per bucket.  The bucket becomes empty (a 2-choice table needs no tombstone: a lookup reads both buckets of a key whatever else they hold); the
// subtree's node ids, plane rows and slab-point regions go onto the free stacks, from where the next updates take them before they touch fresh pool memory
// (regions of another size — the build sizes them by count — and candidate ranges are not recycled).
struct SlideBox { int32_t x_max, x_min, y_max, y_min, z_max, z_min; };
__global__ void __launch_bounds__(256) k_mt_slide(MapTreeArgs a, SlideBox b) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h > a.mask) return;
  RootSlot *s = &a.slots[h];
  if (s->val == -1) return;
  const long long x = s->kx, y = s->ky, z = s->kz;
  const bool should_remove = x > b.x_max || x < b.x_min || y > b.y_max || y < b.y_min || z > b.z_max || z < b.z_min;
  if (!should_remove) return;
  int st[8 * (LIVO2_MAX_LAYER + 1) + 1];
  int sp = 0;
  st[0] = s->pad;                                               // the root's node id
  while (sp >= 0) {
    const int id = st[sp--];
    DevNode &nd = a.nodes[id];
    for (int k = 0; k < 8; k++) if (nd.child[k] >= 0 && sp + 1 < (int)(sizeof(st) / sizeof(st[0]))) st[++sp] = nd.child[k];
    if (nd.plane >= 0) mt_free_planes(a)[atomicAdd(&a.counters[MTC_FREE_PLANES], 1)] = nd.plane;
    for (int q = 0; q < nd.pts_cap; q += a.slab) mt_free_slabs(a)[atomicAdd(&a.counters[MTC_FREE_SLABS], 1)] = nd.pts_off + q;      // (a frozen node has pts_cap 0)
    nd.root = -1; nd.layer = -1; nd.plane = -1; nd.is_plane = 0;          // (livo2_map_tree_export recognises roots by layer == 0 && root == id)
    mt_free_nodes(a)[atomicAdd(&a.counters[MTC_FREE_NODES], 1)] = id;
  }
  s->kx = s->ky = s->kz = MT_NO_KEY; s->pad = -1;              // an empty bucket matches no voxel, whatever val a concurrent claim puts there later
  s->val = -1;
  atomicSub(&a.counters[MTC_ROOTS], 1);
  atomicAdd(&a.counters[MTC_REMOVED], 1);
}

// rows of the plane table -> a packed array (livo2_map_tree_read_planes)
__global__ void __launch_bounds__(256) k_mt_gather_planes(const double *__restrict__ planes, const int32_t *__restrict__ rows, int n, double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, p = t >> 5, k = t & 31;
  if (p < n) out[(size_t)p * PLANE_REC_DOUBLES + k] = planes[(size_t)rows[p] * PLANE_REC_DOUBLES + k];
}

// ---- candidate-range compaction: every root becomes "dirty" with an empty range, the bump counter restarts, k_mt_emit rewrites all lists back to back ----
// (a root whose list outgrew its range took a new one and left the old behind: on a long run the ranges are re-packed when the pool is more than half used)
__global__ void __launch_bounds__(256) k_mt_all_roots_dirty(MapTreeArgs a) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h > a.mask) return;
  const RootSlot &s = a.slots[h];
  if (s.val == -1) return;
  DevNode &r = a.nodes[s.pad];
  r.cand_begin = 0; r.cand_cap = 0;
  a.dirty_list[atomicAdd(&a.counters[MTC_DIRTY], 1)] = s.pad;
}

// ---- what k_lidar_residual reads: the root's slot and, for a non-plane root, the depth-first list of its descendant planes (record copies) -----------
// build_single_residual (voxel_map.cpp:713-786) evaluates a node's plane if is_plane_, otherwise recurses into all eight leaves while layer < max_layer.
__global__ void __launch_bounds__(256) k_mt_emit(MapTreeArgs a) {
  const int width = MT_LPG * max(a.spread, 1);                // (lanes per root as in k_mt_update: the subtree walks of a wave's roots diverge too)
  if ((int)(threadIdx.x & (width - 1)) >= MT_LPG) return;
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) / width, lane = threadIdx.x & (MT_LPG - 1);
  if (g >= a.counters[MTC_DIRTY]) return;
  const int rid = a.dirty_list[g];
  DevNode r = a.nodes[rid];
  int val = -2, count = 0;
  if (r.is_plane) val = r.plane;
  else {
    // ONE walk that writes while the list fits the root's region (the usual case: a region is allocated at twice the length that first needed it); only a list that
    // has outgrown it gets a new region and a second walk.  (Rounds 3-5 walked every subtree twice, once to count and once to write; whole chain at 15 k points 264-284 -> 253-258 us.)
    for (int attempt = 0; attempt < 2; attempt++) {
      int n_out = 0;
      int st_id[MT_STACK], st_next[MT_STACK];
      int sp = 0; st_id[0] = rid; st_next[0] = 0;
      while (sp >= 0) {
        const DevNode &nd = a.nodes[st_id[sp]];
        if (st_next[sp] >= 8 || nd.layer >= LIVO2_MAX_LAYER) { sp--; continue; }
        const int cid = nd.child[st_next[sp]++];
        if (cid < 0) continue;
        const DevNode &c = a.nodes[cid];
        if (c.is_plane) {
          if (n_out < r.cand_cap) {
            const size_t at = (size_t)(r.cand_begin + n_out);
            double *dst = a.cand + at * PLANE_HOT_DOUBLES;
            const double *src = a.planes_hot + (size_t)c.plane * PLANE_HOT_DOUBLES;
            for (int q = lane; q < PLANE_HOT_DOUBLES; q += MT_LPG) dst[q] = src[q];
            if (lane == 0) { PlaneAux x = a.plane_aux[c.plane]; x.meta = c.plane | (c.layer << CAND_LAYER_SHIFT); a.cand_aux[at] = x; }
          }
          n_out++;
        } else if (sp + 1 < MT_STACK) { sp++; st_id[sp] = cid; st_next[sp] = 0; }
      }
      count = n_out;
      if (count <= r.cand_cap) break;
      const int cap = max(16, 2 * count);
      int at = 0;
      if (lane == 0) at = mt_alloc(a, MTC_CAND, cap, a.cap_cand, MTE_CAND);
      at = grp_first(at);
      if (at < 0) { count = 0; break; }
      r.cand_begin = at; r.cand_cap = cap;
    }
  }
  if (lane == 0) {
    r.dirty = 0;
    a.nodes[rid] = r;
    const uint32_t h1 = voxel_hash(r.key[0], r.key[1], r.key[2], a.seed1) & a.mask, h2 = voxel_hash(r.key[0], r.key[1], r.key[2], a.seed2) & a.mask;
    const uint32_t hh[2] = {h1, h2};
    for (int t = 0; t < 2; t++) {
      RootSlot *s = &a.slots[hh[t]];
      if (s->val != -1 && s->kx == r.key[0] && s->ky == r.key[1] && s->kz == r.key[2]) { s->val = val; s->cand_begin = r.cand_begin; s->cand_count = count; break; }
    }
  }
}
