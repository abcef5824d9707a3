// Per-scan pre-stage: what happens to a raw LiDAR scan between the driver and VoxelMapManager::StateEstimation —
//   ImuProcess::UndistortPcl, backward propagation of every point to the scan-end pose     reference src/IMU_Processing.cpp:494-539
//   pcl::VoxelGrid centroid filter (downSizeFilterSurf, leaf = filter_size_surf)            reference src/LIVMapper.cpp:351-352
// so that a raw scan is uploaded once and feats_down_body never leaves the device before the update.  Both are restated with the
// oracle's operation order (oracle/orc_preprocess.hpp); pcl::VoxelGrid is third-party (PCL, unpinned): PARITY UNPINNED there.
#pragma once
#include "livo2_device.hpp"

struct UndistortArgs {
  float *xyz;                    // [n][3] in place
  const float *curvature;        // [n] ms from the scan start, ascending
  const double *poses;           // [n_poses][22] Pose6D: offset_time, acc3, gyr3, vel3, pos3, rot9
  int32_t n, n_poses;
  double extR_Ri[9], exrR_extT[3], ER[9], Et[3], pos_end[3];
  const livo2_state *end_state;  // scan-end state still on the device (livo2_lio_frame): extR_Ri / pos_end are then derived from it here; NULL: the values above
};

__device__ __forceinline__ void undistort_one(const UndistortArgs &a, const double *extR_Ri, const double *pos_end, const double *head, double t, float *p) {
  const double dt = t - head[0];
  const double *acc = head + 1, *gyr = head + 4, *vel = head + 7, *pos = head + 10, *Rimu = head + 13;
  double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double nrm = sqrt((gyr[0] * gyr[0] + gyr[1] * gyr[1]) + gyr[2] * gyr[2]);
  if (nrm > 0.0000001) {                                          // Exp(ang_vel, dt), so3_math.h:24-43
    const double r[3] = {gyr[0] / nrm, gyr[1] / nrm, gyr[2] / nrm};
    const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
    const double ang = nrm * dt, s = sin(ang), c1 = 1.0 - cos(ang);
    double Kc[9], KK[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Kc[k] = K[k] * c1;
    mat3_mul(Kc, K, KK);
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] = (E[k] + K[k] * s) + KK[k];
  }
  double Ri[9];
  mat3_mul(Rimu, E, Ri);
  double T[3], q[3], w[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; k++) T[k] = ((pos[k] + vel[k] * dt) + ((acc[k] * 0.5) * dt) * dt) - pos_end[k];
  const double P[3] = {(double)p[0], (double)p[1], (double)p[2]};
#pragma unroll
  for (int k = 0; k < 3; k++) q[k] = ((a.ER[k * 3] * P[0] + a.ER[k * 3 + 1] * P[1]) + a.ER[k * 3 + 2] * P[2]) + a.Et[k];
#pragma unroll
  for (int k = 0; k < 3; k++) w[k] = ((Ri[k * 3] * q[0] + Ri[k * 3 + 1] * q[1]) + Ri[k * 3 + 2] * q[2]) + T[k];
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = ((extR_Ri[k * 3] * w[0] + extR_Ri[k * 3 + 1] * w[1]) + extR_Ri[k * 3 + 2] * w[2]) - a.exrR_extT[k];
  p[0] = (float)o[0]; p[1] = (float)o[1]; p[2] = (float)o[2];
}

// One thread per point.  The reference walks the time-sorted cloud backwards, segment by segment; for a sorted cloud that is: a point
// is compensated with the LAST pose whose offset_time is < its time (none: untouched) — except the first point of the cloud, which the
// backward walk re-enters for every earlier segment as well (the `if (it_pcl == begin) break` only leaves the inner loop).
__global__ void __launch_bounds__(256) k_undistort(UndistortArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double t = (double)a.curvature[i] / double(1000);
  int lo = 0, hi = a.n_poses - 1;                                 // heads are poses [0, n_poses-2]; find the count of heads with offset_time < t
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (t > a.poses[(size_t)mid * 22]) lo = mid + 1; else hi = mid; }
  int h = lo - 1;
  if (h < 0) return;
  float p[3] = {a.xyz[(size_t)i * 3], a.xyz[(size_t)i * 3 + 1], a.xyz[(size_t)i * 3 + 2]};
  double eRi[9], pe[3];
  if (a.end_state) {                                                // the same expressions the host side of livo2_lidar_preprocess_scan evaluates
    const double *re = a.end_state->rot, *pz = a.end_state->pos;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) eRi[r * 3 + c] = (a.ER[0 * 3 + r] * re[c * 3 + 0] + a.ER[1 * 3 + r] * re[c * 3 + 1]) + a.ER[2 * 3 + r] * re[c * 3 + 2];
      pe[r] = pz[r];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) eRi[k] = a.extR_Ri[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pe[k] = a.pos_end[k];
  }
  const int h_stop = (i == 0) ? 0 : h;
  for (; h >= h_stop; h--) undistort_one(a, eRi, pe, a.poses + (size_t)h * 22, t, p);
  a.xyz[(size_t)i * 3] = p[0]; a.xyz[(size_t)i * 3 + 1] = p[1]; a.xyz[(size_t)i * 3 + 2] = p[2];
}

// ---- pcl::VoxelGrid ----------------------------------------------------------------------------------------------------------------
// bounds[0..2] = min, [3..5] = max of the cloud as order-preserving uint32 codes of the floats (atomicMax across blocks; min / max are exact, so the result does
// not depend on the order).  The minima are kept as the COMPLEMENT of their code, so that ONE fill with zeros presets all six words (round 6: one launch less).
__device__ __forceinline__ uint32_t f32_code(float f) { const uint32_t b = __builtin_bit_cast(uint32_t, f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float f32_decode(uint32_t c) { return __builtin_bit_cast(float, (c & 0x80000000u) ? (c & 0x7fffffffu) : ~c); }
__global__ void __launch_bounds__(1024) k_vg_minmax(const float *__restrict__ xyz, int n, uint32_t *__restrict__ bounds) {
  __shared__ float s_mn[16][3], s_mx[16][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * 1024 + tid; i < n; i += gridDim.x * 1024)
#pragma unroll
    for (int k = 0; k < 3; k++) { const float v = xyz[(size_t)i * 3 + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64)); }
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 3; k++) { s_mn[wave][k] = mn[k]; s_mx[wave][k] = mx[k]; }
  __syncthreads();
  if (tid < 3) {                                                  // one pair of atomics per block and axis
    float a = s_mn[0][tid], b = s_mx[0][tid];
    for (int w = 1; w < 16; w++) { a = fminf(a, s_mn[w][tid]); b = fmaxf(b, s_mx[w][tid]); }
    atomicMax(&bounds[tid], ~f32_code(a)); atomicMax(&bounds[3 + tid], f32_code(b));
  }
}

// leaf index of every point (applyFilter's first pass); flag[0] = 1 if the grid overflows int32
__global__ void __launch_bounds__(256) k_vg_keys(const float *__restrict__ xyz, int n, float inv_leaf, const uint32_t *__restrict__ bounds, uint32_t *__restrict__ keys,
                                                 int32_t *__restrict__ idx, int32_t *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  long long min_b[3], div_b[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { min_b[k] = (long long)floorf(f32_decode(~bounds[k]) * inv_leaf); div_b[k] = (long long)floorf(f32_decode(bounds[3 + k]) * inv_leaf) - min_b[k] + 1; }
  if (div_b[0] * div_b[1] * div_b[2] > 2147483647ll) { if (i == 0) flag[0] = 1; return; }
  if (i >= n) return;
  const int mul[3] = {1, (int)div_b[0], (int)(div_b[0] * div_b[1])};
  int key = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) key += (int)(floorf(xyz[(size_t)i * 3 + k] * inv_leaf) - (float)min_b[k]) * mul[k];
  keys[i] = (uint32_t)key; idx[i] = i;
}

// leaves per 256-point block of the sorted keys (a leaf belongs to the block of its first point): what k_vg_centroid needs to number the leaves without a device-wide scan
__global__ void __launch_bounds__(256) k_vg_heads(const uint32_t *__restrict__ keys_sorted, int n, int32_t *__restrict__ blk_count) {
  __shared__ int s_cnt[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool head = i < n && (i == 0 || keys_sorted[i] != keys_sorted[i - 1]);
  const unsigned long long m = __ballot(head);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blk_count[blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

// one thread per leaf head: float centroid of the leaf's points in sorted (= input) order.  The leaf's number (its row in `out`: leaves in ascending key order, as
// pcl::VoxelGrid emits them) = leaves in the blocks before this one (k_vg_heads' counts, summed by the block: <= 4 per thread for a million points) + heads before
// it in the block — the library's device-wide exclusive scan (three launches) is gone (round 6).
__global__ void __launch_bounds__(256) k_vg_centroid(const float *__restrict__ xyz, const uint32_t *__restrict__ keys_sorted, const int32_t *__restrict__ perm,
                                                     const int32_t *__restrict__ blk_count, int n, float *__restrict__ out, int32_t *__restrict__ count) {
  __shared__ int s_part[256], s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * blockDim.x + tid;
  const uint32_t key = i < n ? keys_sorted[i] : 0u;
  const bool head = i < n && (i == 0 || key != keys_sorted[i - 1]);
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += 256) before += blk_count[b];
  s_part[tid] = before;
  const unsigned long long m = __ballot(head);
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if (tid < d) s_part[tid] += s_part[tid + d]; __syncthreads(); }
  int s = s_part[0] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) s += s_wave[w];
  if (i == n - 1) count[0] = s + (head ? 1 : 0);
  if (!head) return;
  // The sums are float and in sorted (= input) order, as pcl::VoxelGrid accumulates them, so the ADDS are a chain; the loads need not be.  One point per turn made every
  // turn two dependent round trips (perm[j], then the point): the busiest leaf of a 240 k-point scan set the kernel's 61 us.  Sixteen points per turn now: their keys and
  // permutation entries in one batch, their coordinates in a second, then the adds in order, stopping at the first key that differs (round 6: 61 -> 22 us).
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  int j = i;
  bool more = true;
  while (more) {
    uint32_t kk[16]; int pp[16]; float q[16][3];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int jj = min(j + u, n - 1); kk[u] = keys_sorted[jj]; pp[u] = perm[jj]; }
#pragma unroll
    for (int u = 0; u < 16; u++) { q[u][0] = xyz[(size_t)pp[u] * 3]; q[u][1] = xyz[(size_t)pp[u] * 3 + 1]; q[u][2] = xyz[(size_t)pp[u] * 3 + 2]; }
    int taken = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (more && j + u < n && kk[u] == key) { c0 += q[u][0]; c1 += q[u][1]; c2 += q[u][2]; taken++; }
      else more = false;
    }
    j += taken;
  }
  const float cnt = (float)(j - i);
  out[(size_t)s * 3] = c0 / cnt; out[(size_t)s * 3 + 1] = c1 / cnt; out[(size_t)s * 3 + 2] = c2 / cnt;
}
