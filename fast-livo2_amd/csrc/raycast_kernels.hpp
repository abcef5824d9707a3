// RayCasting module of VIOManager::retrieveFromVisualSparseMap (reference src/vio.cpp:487-591, rays of initializeVIO src/vio.cpp:80-118; vio/raycast_en, off in the
// shipped configs) between stage B (nearest visual point per grid cell, select_kernels.hpp) and stage C (depth continuity).
//
// The reference walks the grid cells IN ORDER: a cell that is not TYPE_MAP yet (and not on the grid's border) follows its ray — 15 samples through the cell centre at
// depths 0.1, 0.3, ... 2.9 — until a sample's voxel (a) is already in sub_feat_map: the ray ends; (b) holds visual points (feat_map): they are projected like stage B's
// (grid_num = TYPE_MAP, nearest point per cell), the voxel joins sub_feat_map if one of them is in view, the ray ends; (c) is a voxel of the LiDAR map whose leaf at the
// sample (find_correspond) is a plane: (center_, normal_) goes to visual_submap->add_from_voxel_map, the ray ends.  The order matters in ONE way: a ray can turn a LATER
// cell into TYPE_MAP, and that cell then shoots no ray.  (That sub_feat_map grows does not: a voxel an earlier ray filed there is a feat_map voxel, so a later ray ends at
// it either way, and visiting its points twice changes nothing.)  Hence three launches:
//   k_ray_find     one thread per cell: what its ray WOULD do (nothing / visit voxel k / plane row p) — a function of the scan-voxel set, the visual map's voxel set and the
//                  LiDAR tree only, not of the other rays; the voxels wanted go into a hash set;
//   k_ray_points   one thread per visual point: if its voxel is wanted: in-frame test, target cell, distance -> one hit record;
//   k_ray_resolve  ONE block walks the cells in order with the cell types in LDS: a cell still not TYPE_MAP at its turn executes its action — its voxel's hits mark their
//                  target cells and compete for them (the same 64-bit atomicMin as stage B), or its plane is appended to add_from_voxel_map (in cell order).
// plane_map is the device-resident VoxelMap (livo2_map_tree_*: DevNode tree, map_tree_kernels.hpp); without one the rays see no planes.
#pragma once
#include "map_tree_kernels.hpp"
#include "select_kernels.hpp"

#define RAY_SAMPLES_MAX 16
#define RAY_MAX_CELLS 32768            // cell types of the resolve pass live in LDS (one byte each)

struct RayArgs {
  SelectArgs s;                          // camera, pose, grid, scan-voxel set, per-cell state of stage B
  const unsigned long long *vmset;       // voxel set of the visual map (livo2_visual_map_upload), capacity vm_mask + 1
  uint32_t vm_mask, ray_mask;
  unsigned long long *rayset;            // voxels some ray wants visited
  int32_t *action;                       // [length] 0 nothing, 1 visit voxel key[c], 2 plane row (int32) key[c]
  unsigned long long *key;               // [length]
  unsigned long long *hit_key, *hit_best; int32_t *hit_cell;   // [n_pts] one record per in-view point of a wanted voxel
  int32_t *counters;                     // [0] hits, [1] entries of add_from_voxel_map
  double *add6;                          // [length][6] center_, normal_
  // LiDAR tree (null nodes: no plane_map)
  const DevNode *nodes; const RootSlot *slots; const double *planes;
  uint32_t lmask, lseed1, lseed2; int32_t max_layer;
  double Rt[9], tinv[3];                 // T_f_w_.inverse()
};

__device__ __forceinline__ bool ray_set_has(const unsigned long long *set, uint32_t mask, unsigned long long key) {
  uint32_t h = sel_hash(key) & mask;
  for (;;) { const unsigned long long v = set[h]; if (v == key) return true; if (v == SEL_EMPTY) return false; h = (h + 1) & mask; }
}
__device__ __forceinline__ void ray_set_put(unsigned long long *set, uint32_t mask, unsigned long long key) {
  uint32_t h = sel_hash(key) & mask;
  for (;;) { const unsigned long long old = atomicCAS(&set[h], SEL_EMPTY, key); if (old == SEL_EMPTY || old == key) return; h = (h + 1) & mask; }
}

// voxel set of the visual map, once per livo2_visual_map_upload
__global__ void __launch_bounds__(256) k_vm_voxel_set(const unsigned long long *__restrict__ pkey, int n, unsigned long long *set, uint32_t mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && pkey[i] != SEL_EMPTY) ray_set_put(set, mask, pkey[i]);
}

__global__ void __launch_bounds__(256) k_ray_find(RayArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.s.length) return;
  int act = 0; unsigned long long out = 0ull;
  const int W = a.s.grid_n_width, H = a.s.length / a.s.grid_n_width;
  const int row = c / W, col = c % W;                                        // grid_row - 1, grid_col - 1 of vio.cpp:93-101
  const bool border = row == 0 || col == 0 || row == H - 1 || col == W - 1;
  if (!border) {
    const int u = a.s.grid_size / 2 + col * a.s.grid_size, v = a.s.grid_size / 2 + row * a.s.grid_size;
    double x, y;
    cam_unproject(a.s.distortion, a.s.d, a.s.fx, a.s.fy, a.s.cx, a.s.cy, (double)u, (double)v, x, y);      // cam->cam2world(u, v): normalize(x, y, 1)
    const double nrm = sqrt((x * x + y * y) + 1.0 * 1.0);
    const double f[3] = {x / nrm, y / nrm, 1.0 / nrm};
    float d_temp = 0.1f;
    for (int k = 0; k < RAY_SAMPLES_MAX && d_temp <= 3.0f; k++, d_temp += 0.2f) {
      const double sc = (double)d_temp / f[2];
      const double it[3] = {f[0] * sc, f[1] * sc, f[2] * sc};
      double pw[3]; long long loc[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        pw[j] = ((a.Rt[j * 3] * it[0] + a.Rt[j * 3 + 1] * it[1]) + a.Rt[j * 3 + 2] * it[2]) + a.tinv[j];
        int l = (int)floor(pw[j] / (double)0.5f);
        if (l < 0) l = (int)((double)l - 1.0);
        loc[j] = l;
      }
      unsigned long long key;
      if (!sel_pack(loc[0], loc[1], loc[2], key)) continue;                 // outside 21 bits per axis: no voxel of any map is there
      if (ray_set_has(a.s.set, a.s.mask, key)) break;                       // sub_feat_map (scan voxels): the ray ends
      if (ray_set_has(a.vmset, a.vm_mask, key)) { act = 1; out = key; ray_set_put(a.rayset, a.ray_mask, key); break; }
      if (a.nodes) {                                                         // plane_map.find(sample_pos) -> find_correspond(sample_point_w) -> is_plane_
        const int32_t kx = (int32_t)loc[0], ky = (int32_t)loc[1], kz = (int32_t)loc[2];
        const uint32_t hh[2] = {voxel_hash(kx, ky, kz, a.lseed1) & a.lmask, voxel_hash(kx, ky, kz, a.lseed2) & a.lmask};
        int node = -1;
        for (int t = 0; t < 2 && node < 0; t++) {
          const RootSlot &sl = a.slots[hh[t]];
          if (sl.val != -1 && sl.val != -3 && sl.kx == kx && sl.ky == ky && sl.kz == kz) node = sl.pad;
        }
        if (node >= 0) {
          for (;;) {
            const DevNode &n = a.nodes[node];
            if (!n.init_octo || n.is_plane || n.layer >= a.max_layer) break;
            const int leaf = 4 * (pw[0] > n.center[0] ? 1 : 0) + 2 * (pw[1] > n.center[1] ? 1 : 0) + (pw[2] > n.center[2] ? 1 : 0);
            if (n.child[leaf] < 0) break;
            node = n.child[leaf];
          }
          if (a.nodes[node].is_plane) { act = 2; out = (unsigned long long)(uint32_t)a.nodes[node].plane; break; }
        }
      }
    }
  }
  a.action[c] = act; a.key[c] = out;
}

__global__ void __launch_bounds__(256) k_ray_points(RayArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.s.n_pts || !a.s.active[i]) return;
  const unsigned long long key = a.s.pkey[i];
  if (key == SEL_EMPTY || !ray_set_has(a.rayset, a.ray_mask, key)) return;
  const double p[3] = {a.s.pos[(size_t)i * 3], a.s.pos[(size_t)i * 3 + 1], a.s.pos[(size_t)i * 3 + 2]};
  double pc3[3], px[2];
  sel_project(a.s, p, pc3, px);
  if (pc3[2] < 0) return;
  const int col = (int)px[0], row = (int)px[1];
  if (!sel_in_frame(a.s, col, row)) return;
  const int index = (int)(px[1] / a.s.grid_size) * a.s.grid_n_width + (int)(px[0] / a.s.grid_size);
  if (index < 0 || index >= a.s.length) return;
  const double o0 = a.s.cam_pos[0] - p[0], o1 = a.s.cam_pos[1] - p[1], o2 = a.s.cam_pos[2] - p[2];
  const float cur_dist = (float)sqrt((o0 * o0 + o1 * o1) + o2 * o2);
  const int h = atomicAdd(&a.counters[0], 1);
  a.hit_key[h] = key; a.hit_cell[h] = index;
  a.hit_best[h] = ((unsigned long long)__builtin_bit_cast(uint32_t, cur_dist) << 32) | (unsigned long long)(uint32_t)i;
}

__global__ void __launch_bounds__(256) k_ray_resolve(RayArgs a) {
  __shared__ uint8_t type[RAY_MAX_CELLS];
  __shared__ int n_add;
  const int tid = threadIdx.x, length = a.s.length;
  for (int c = tid; c < length; c += 256) type[c] = a.s.cell_type[c] == 1 ? 1 : 0;
  if (tid == 0) n_add = 0;
  const int n_hits = a.counters[0];
  __syncthreads();
  for (int c = 0; c < length; c++) {
    const int act = a.action[c];                                             // (block-uniform: every thread reads the same word)
    if (act == 0) continue;
    __syncthreads();                                                         // marks of the cells before this one
    const bool is_map = type[c] != 0;                                        // grid_num[i] == TYPE_MAP: no ray (vio.cpp:491)
    __syncthreads();                                                         // (everybody has read it before a hit of this very voxel can mark cell c itself)
    if (is_map) continue;
    if (act == 1) {
      const unsigned long long key = a.key[c];
      for (int h = tid; h < n_hits; h += 256)
        if (a.hit_key[h] == key) {
          const int t = a.hit_cell[h];
          type[t] = 1;
          atomicMin(&a.s.cell_best[t], a.hit_best[h]);
          a.s.in_fov[(uint32_t)a.hit_best[h]] = 1;
        }
    } else if (tid < 6) {
      const int row = (int)(uint32_t)a.key[c];
      a.add6[(size_t)n_add * 6 + tid] = a.planes[(size_t)row * PLANE_REC_DOUBLES + (tid < 3 ? 3 + tid : tid - 3)];      // center_ (words 3..5), normal_ (0..2)
    }
    if (act == 2) { __syncthreads(); if (tid == 0) n_add++; }
  }
  __syncthreads();
  for (int c = tid; c < length; c += 256) if (type[c]) a.s.cell_type[c] = 1;
  if (tid == 0) a.counters[1] = n_add;
}
