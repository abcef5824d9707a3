// Visual sub-map retrieval, selection half (reference src/vio.cpp:352-486 and 598-635; the RayCasting module of raycast_en = true, vio.cpp:487-591,
// runs between B and C: raycast_kernels.hpp):
//   A  k_sel_scan    every point of the current scan (pv_list_.point_w): its lookup voxel goes into a hash set (sub_feat_map), its
//                    projection writes the depth image — the LAST scan point landing on a pixel wins, as in the serial loop
//                    (64-bit atomicMax keyed by the point index);
//   B  k_sel_points  every visual map point whose feat_map voxel is in that set: behind-camera / in-frame tests, grid cell, and the
//                    nearest point per cell (64-bit atomicMin on {float distance bits, point index}; the reference's `<=` lets the last
//                    VISITED of exactly equidistant points win, here the lowest index does — float ties only);
//   C  k_sel_cells   per grid cell (one wave): the depth-continuity test of the selected point against the 9x9 depth-image window.
// The visual map lives on the device as flat arrays (position, the voxel it is filed under, active flag), uploaded by
// livo2_visual_map_upload.  Both voxel-key formulas are the reference's own, mismatch for negative coordinates included
// (vio.cpp:392-396 vs 232-236).  vikit's world2cam (pinhole, optional radial-tangential distortion) / isInFrame restated: parity unpinned at that boundary.
#pragma once
#include "livo2_device.hpp"

#define SEL_EMPTY 0xFFFFFFFFFFFFFFFFull

struct SelectArgs {
  double fx, fy, cx, cy, R[9], t[3], cam_pos[3];
  double d[5];                             // vk::PinholeCamera radial-tangential coefficients, used when distortion != 0
  int32_t distortion, pad2;
  int32_t width, height, border, grid_size, grid_n_width, length, patch_size_half, n_pg, n_pts, pad;
  const double *pg;                        // [n_pg][3]
  const double *pos;                       // [n_pts][3]
  const unsigned long long *pkey;          // [n_pts] packed feat_map key
  const uint8_t *active;                   // [n_pts]
  unsigned long long *set;                 // scan-voxel hash set, capacity mask + 1
  uint32_t mask;
  unsigned long long *depth;               // [height*width] {point index, depth bits}
  unsigned long long *cell_best;           // [length]  {distance bits, point index}
  int32_t *cell_type;                      // [length]
  uint8_t *in_fov;                         // [n_pts]
  int32_t *range_flag;                     // set when a voxel key does not fit 21 bits per axis
  int32_t *cell_point; float *cell_dist; uint8_t *cell_discont;
};

__device__ __forceinline__ bool sel_pack(long long x, long long y, long long z, unsigned long long &k) {
  const long long B = 1ll << 20;
  if (x < -B || x >= B || y < -B || y >= B || z < -B || z >= B) return false;
  k = ((unsigned long long)(x + B) << 42) | ((unsigned long long)(y + B) << 21) | (unsigned long long)(z + B);
  return true;
}
__device__ __forceinline__ uint32_t sel_hash(unsigned long long k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return (uint32_t)k; }

__device__ __forceinline__ bool sel_project(const SelectArgs &a, const double *p, double *pc3, double *px) {
#pragma unroll
  for (int r = 0; r < 3; r++) pc3[r] = ((a.R[r * 3] * p[0] + a.R[r * 3 + 1] * p[1]) + a.R[r * 3 + 2] * p[2]) + a.t[r];
  const double u = pc3[0] / pc3[2], v = pc3[1] / pc3[2];
  cam_project(a.distortion, a.d, a.fx, a.fy, a.cx, a.cy, u, v, px[0], px[1]);
  return true;
}
__device__ __forceinline__ bool sel_in_frame(const SelectArgs &a, int x, int y) { return x >= a.border && x < a.width - a.border && y >= a.border && y < a.height - a.border; }

// packed feat_map key of a visual point from its position (insertPointIntoVoxelMap, vio.cpp:227-236) — used when the caller passes no keys
__global__ void __launch_bounds__(256) k_sel_point_keys(const double *__restrict__ pos, int n, unsigned long long *__restrict__ pkey, int32_t *__restrict__ range_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long k[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float loc = (float)(pos[(size_t)i * 3 + j] / 0.5);
    if (loc < 0) loc = (float)((double)loc - 1.0);
    k[j] = (long long)loc;
  }
  unsigned long long pk = SEL_EMPTY;
  if (!sel_pack(k[0], k[1], k[2], pk)) range_flag[0] = 1;
  pkey[i] = pk;
}

// per-call reset of the selection state in ONE launch (five fill launches of ~4.4 us each before): empty voxel set, zero depth image, empty cells
__global__ void __launch_bounds__(256) k_sel_reset(SelectArgs a, uint32_t set_cap, uint32_t pixels) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < set_cap) a.set[i] = SEL_EMPTY;
  if (i < pixels) a.depth[i] = 0ull;
  if (i < (uint32_t)a.length) { a.cell_best[i] = SEL_EMPTY; a.cell_type[i] = 0; }
  if (i < 16) a.range_flag[i] = 0;
}

__global__ void __launch_bounds__(256) k_sel_scan(SelectArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pg) return;
  const double p[3] = {a.pg[(size_t)i * 3], a.pg[(size_t)i * 3 + 1], a.pg[(size_t)i * 3 + 2]};
  long long loc[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {                                      // vio.cpp:392-396: floor, then another -1 for negatives
    int l = (int)floor(p[j] / (double)0.5f);
    if (l < 0) l = (int)((double)l - 1.0);
    loc[j] = l;
  }
  unsigned long long key;
  if (sel_pack(loc[0], loc[1], loc[2], key)) {
    uint32_t h = sel_hash(key) & a.mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&a.set[h], SEL_EMPTY, key);
      if (old == SEL_EMPTY || old == key) break;
      h = (h + 1) & a.mask;
    }
  } else a.range_flag[0] = 1;
  double pc3[3], px[2];
  sel_project(a, p, pc3, px);
  if (pc3[2] > 0) {
    const int col = (int)px[0], row = (int)px[1];
    if (sel_in_frame(a, col, row)) {
      const float depth = (float)pc3[2];
      atomicMax(&a.depth[(size_t)a.width * row + col], ((unsigned long long)(uint32_t)i << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, depth));
    }
  }
}

__global__ void __launch_bounds__(256) k_sel_points(SelectArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pts) return;
  uint8_t fov = 0;
  if (a.active[i]) {
    const unsigned long long key = a.pkey[i];
    bool found = false;
    if (key != SEL_EMPTY) {
      uint32_t h = sel_hash(key) & a.mask;
      for (;;) { const unsigned long long s = a.set[h]; if (s == key) { found = true; break; } if (s == SEL_EMPTY) break; h = (h + 1) & a.mask; }
    }
    if (found) {
      const double p[3] = {a.pos[(size_t)i * 3], a.pos[(size_t)i * 3 + 1], a.pos[(size_t)i * 3 + 2]};
      double pc3[3], px[2];
      sel_project(a, p, pc3, px);
      if (!(pc3[2] < 0)) {
        const int col = (int)px[0], row = (int)px[1];
        if (sel_in_frame(a, col, row)) {
          fov = 1;
          const int index = (int)(px[1] / a.grid_size) * a.grid_n_width + (int)(px[0] / a.grid_size);
          if (index >= 0 && index < a.length) {
            a.cell_type[index] = 1;
            const double o0 = a.cam_pos[0] - p[0], o1 = a.cam_pos[1] - p[1], o2 = a.cam_pos[2] - p[2];
            const float cur_dist = (float)sqrt((o0 * o0 + o1 * o1) + o2 * o2);
            atomicMin(&a.cell_best[index], ((unsigned long long)__builtin_bit_cast(uint32_t, cur_dist) << 32) | (unsigned long long)(uint32_t)i);
          }
        }
      }
    }
  }
  a.in_fov[i] = fov;
}

// one wave per grid cell: the lanes share the (2 patch_size_half + 1)^2 depth-image window of the selected point
__global__ void __launch_bounds__(256) k_sel_cells(SelectArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.length) return;
  int32_t point = -1; float dist = 10000.0f; bool discont = false;
  const unsigned long long best = a.cell_best[c];
  if (a.cell_type[c] == 1 && best != SEL_EMPTY) {
    const float d = __builtin_bit_cast(float, (uint32_t)(best >> 32));
    if (d <= 10000.0f) {                                               // map_dist starts at 10000 (vio.cpp:166): farther points never replace it
      point = (int32_t)(uint32_t)best; dist = d;
      const double p[3] = {a.pos[(size_t)point * 3], a.pos[(size_t)point * 3 + 1], a.pos[(size_t)point * 3 + 2]};
      double pc3[3], px[2];
      sel_project(a, p, pc3, px);
      const int u0 = (int)px[0], v0 = (int)px[1];
      const int side = 2 * a.patch_size_half + 1;
      for (int k = lane; k < side * side; k += 64) {
        const int u = k / side - a.patch_size_half, v = k % side - a.patch_size_half;
        if (u == 0 && v == 0) continue;
        const float depth = __builtin_bit_cast(float, (uint32_t)a.depth[(size_t)a.width * (v + v0) + u + u0]);
        if (depth == 0.f) continue;
        if (fabs(pc3[2] - (double)depth) > 0.5) discont = true;
      }
    }
  }
  const bool any = __ballot(discont) != 0ull;
  if (lane == 0) { a.cell_point[c] = point; a.cell_dist[c] = dist; a.cell_discont[c] = any ? 1 : 0; }
}
