// Timing probe for the small-scan preparation of frame_kernels.hpp (round 5): block-wide radix sorts of ~10 k (key, index) pairs on ONE compute unit (what was tried
// first) against the order counted directly on the whole chip (k_frame_ingest keys + k_scan_rank: what is shipped), and the gather + calcBodyCov part on a grid.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ifast-livo2_amd/csrc -Iinclude tools/sort_probe.hip -o /tmp/sort_probe
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstring>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <rocprim/block/block_radix_sort.hpp>
#include "frame_kernels.hpp"

template <int IPT, int BITS, rocprim::block_radix_rank_algorithm ALG>
__global__ void __launch_bounds__(1024) k_sort_only(const float *__restrict__ xyz, int n, float inv_cell, int32_t *__restrict__ perm) {
  using Sort = rocprim::block_radix_sort<uint32_t, 1024, IPT, int32_t, 1, 1, BITS, ALG>;
  __shared__ typename Sort::storage_type storage;
  uint32_t keys[IPT]; int32_t idx[IPT];
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const int g = t * IPT + i;
    if (g < n) { keys[i] = morton_key_of(xyz + (size_t)g * 3, inv_cell); idx[i] = g; } else { keys[i] = 0x40000000u; idx[i] = -1; }
  }
  Sort().sort_to_striped(keys, idx, storage, 0, 31);
#pragma unroll
  for (int i = 0; i < IPT; i++) { const int pos = i * 1024 + t; if (pos < n) perm[pos] = idx[i]; }
}

struct ScanSmallOut { const float *xyz; float *x, *y, *z; int32_t *perm; double *cb; double deg2rad; float range_inc, degree_inc; int32_t n; };
__global__ void __launch_bounds__(256) k_gather_cov(ScanSmallOut s) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= s.n) return;
  const int o = s.perm[pos];
  const float px = s.xyz[(size_t)o * 3], py = s.xyz[(size_t)o * 3 + 1], pz = s.xyz[(size_t)o * 3 + 2];
  s.x[pos] = px; s.y[pos] = py; s.z[pos] = pz;
  double c6[6];
  body_cov_point(px, py, pz, s.range_inc, s.degree_inc, s.deg2rad, c6);
#pragma unroll
  for (int e = 0; e < 6; e++) s.cb[(size_t)e * s.n + pos] = c6[e];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> float time_us(F f, int reps = 50) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  hipEventRecord(a, 0); for (int i = 0; i < reps; i++) f(); hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10800;
  std::mt19937 rng(3); std::uniform_real_distribution<float> U(-25.f, 25.f);
  std::vector<float> h((size_t)n * 3); for (auto &v : h) v = U(rng);
  float *xyz, *x, *y, *z; int32_t *perm, *perm2; double *cb; uint32_t *z0; int32_t *z1;
  CK(hipMalloc(&xyz, n * 12)); CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&z, n * 4)); CK(hipMalloc(&perm, n * 4)); CK(hipMalloc(&perm2, n * 4)); CK(hipMalloc(&cb, n * 48));
  CK(hipMalloc(&z0, 4096)); CK(hipMalloc(&z1, 4096));
  CK(hipMemcpy(xyz, h.data(), n * 12, hipMemcpyHostToDevice));
  uint32_t *keys; CK(hipMalloc(&keys, (n + 16) * 4));
  FrameIngestArgs a{}; a.n_seg = 0; a.n_keys = n; a.xyz = xyz; a.keys = keys; a.inv_cell = 2.f; a.zero0 = z0; a.zero0_words = 64; a.zero1 = z1; a.zero1_words = 18;
  ScanRankArgs r{}; r.keys = keys; r.xyz = xyz; r.x = x; r.y = y; r.z = z; r.perm = perm; r.cb = cb; r.deg2rad = 0.017453293; r.range_inc = 0.02f; r.degree_inc = 0.05f; r.n = n;
  const float inv = 2.f;
  const int kb = (((n + 15) & ~15) + FRAME_THREADS - 1) / FRAME_THREADS;
  printf("n = %d\n", n);
  printf("k_frame_ingest (keys only, %d blocks): %.1f us\n", kb, time_us([&] { hipLaunchKernelGGL(k_frame_ingest, dim3(kb), dim3(FRAME_THREADS), 0, 0, a); }));
  printf("k_scan_rank (order by counting + gather + cov, %d blocks of %d): %.1f us\n", (n + 63) / 64, RANK_THREADS, time_us([&] { hipLaunchKernelGGL(k_scan_rank, dim3((n + 63) / 64), dim3(RANK_THREADS), 0, 0, r); }));
  printf("both, back to back: %.1f us\n", time_us([&] { hipLaunchKernelGGL(k_frame_ingest, dim3(kb), dim3(FRAME_THREADS), 0, 0, a); hipLaunchKernelGGL(k_scan_rank, dim3((n + 63) / 64), dim3(RANK_THREADS), 0, 0, r); }));
  using A = rocprim::block_radix_rank_algorithm;
#define SORT(IPT, BITS, ALG, NAME) if (n <= IPT * 1024) printf("sort only IPT=%d bits=%d %s: %.1f us\n", IPT, BITS, NAME, time_us([&] { hipLaunchKernelGGL((k_sort_only<IPT, BITS, ALG>), dim3(1), dim3(1024), 0, 0, xyz, n, inv, perm2); }))
  SORT(16, 8, A::match, "match"); SORT(16, 4, A::basic_memoize, "basic_memoize"); SORT(16, 4, A::match, "match"); SORT(16, 6, A::match, "match"); SORT(16, 5, A::basic_memoize, "basic_memoize"); SORT(16, 6, A::basic_memoize, "basic_memoize");
  SORT(12, 8, A::match, "match"); SORT(12, 4, A::basic_memoize, "basic_memoize"); SORT(12, 6, A::basic_memoize, "basic_memoize"); SORT(11, 6, A::basic_memoize, "basic_memoize");
  SORT(8, 8, A::match, "match"); SORT(8, 4, A::basic_memoize, "basic_memoize"); SORT(4, 8, A::match, "match"); SORT(4, 4, A::basic_memoize, "basic_memoize");
  const ScanSmallOut so = {xyz, x, y, z, perm, cb, r.deg2rad, r.range_inc, r.degree_inc, n};
  printf("gather + cov, %d blocks of 256: %.1f us\n", (n + 255) / 256, time_us([&] { hipLaunchKernelGGL(k_gather_cov, dim3((n + 255) / 256), dim3(256), 0, 0, so); }));
  // the counted order against a stable sort on the host, and against the last block-sort variant that fits
  std::vector<int32_t> p1(n), p2(n); CK(hipMemcpy(p1.data(), perm, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p2.data(), perm2, n * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> hk(n); CK(hipMemcpy(hk.data(), keys, n * 4, hipMemcpyDeviceToHost));
  std::vector<int32_t> ref(n); for (int i = 0; i < n; i++) ref[i] = i;
  std::stable_sort(ref.begin(), ref.end(), [&](int32_t u, int32_t v) { return hk[u] < hk[v]; });
  printf("perm of k_scan_rank == std::stable_sort by key: %s ; == last block-sort variant: %s\n", p1 == ref ? "yes" : "NO", p1 == p2 ? "yes" : "NO");
  return 0;
}
