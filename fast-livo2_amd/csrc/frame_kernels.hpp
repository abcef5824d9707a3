// Head and tail of a frame on the device (round 5): ONE launch takes a frame's inputs in and prepares its scan, ONE launch hands its results out.
//
// A C1-shaped frame (~10 k points, ~350 patches: the reference's real operating point, preprocess.cpp:185 / LIVMapper.cpp:351-352) is a chain of DEPENDENT small
// commands on one stream; before this file the chain held ~33 of them (H2D of the scan, two memsets, Morton keys, 5-8 launches of the library's merge sort, gather,
// body covariance, H2D of the states, 5 x (residual, solve), five H2D of the image + sub-map, hand-over, the resident visual grid, three D2H) and lasted 0.485 ms of which
// well under half was kernel time: every command boundary costs 5-15 us on this stack, copies more than kernels.
//
//  k_frame_ingest<IPT>: block 0 prepares the scan (IPT > 0: n <= 1024 * IPT points) — Morton keys, a STABLE block-wide radix sort of (key, index) in LDS
//    (rocprim::block_radix_sort: the same permutation as the device-wide stable radix sort it replaces, so every downstream sum keeps its order), the SoA gather
//    and calcBodyCov (voxel_map.cpp:15-34, 349-360) — and zeroes the per-scan words (block lifetimes, arrival tickets).  Blocks 1.. copy the frame's segments
//    (states + header, scan, image, sub-map arrays) from ONE staging block to their device buffers.  The source is either a device arena filled by one H2D copy or the
//    pinned staging block itself, read over the link by the kernel (no copy command at all).
//  k_frame_publish: both result blocks and the watchdog flag of a frame into its pinned result slot (one launch instead of three D2H copies).
#pragma once
#include <rocprim/block/block_radix_sort.hpp>
#include "lidar_kernels.hpp"

#define SCAN_SMALL_THREADS 1024
#define SCAN_SMALL_MAX_IPT 16
#define SCAN_SMALL_MAX (SCAN_SMALL_THREADS * SCAN_SMALL_MAX_IPT)
#define FRAME_MAX_SEGS 8
#define FRAME_COPY_BLOCKS 64

struct FrameSeg { const void *src; void *dst; unsigned long long bytes; };      // src and dst 16-byte aligned
struct FrameIngestArgs {
  FrameSeg seg[FRAME_MAX_SEGS];
  int32_t n_seg, n;                       // segments; points of the scan
  uint32_t *zero0; int32_t *zero1;        // words to clear: block lifetimes of the last scan, arrival tickets of k_lidar_iteration
  int32_t zero0_words, zero1_words;
  const float *xyz;                       // [n][3] the scan as block 0 reads it (the staging block / the arena / d_xyz_aos)
  float *x, *y, *z; int32_t *perm; double *cb;
  float inv_cell, range_inc, degree_inc, pad;
  double deg2rad;
};

__device__ __forceinline__ void frame_copy_segments(const FrameIngestArgs &a, int first_block, int n_blocks) {
  const int b = (int)blockIdx.x - first_block;
  if (b < 0) return;
  const size_t stride = (size_t)n_blocks * blockDim.x;
  for (int s = 0; s < a.n_seg; s++) {
    const FrameSeg sg = a.seg[s];
    const size_t units = sg.bytes >> 4;
    const uint4 *src = static_cast<const uint4 *>(sg.src);
    uint4 *dst = static_cast<uint4 *>(sg.dst);
    size_t u = (size_t)b * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per thread before the first store: over the link every load is a ~2-us round trip
    for (; u + 3 * stride < units; u += 4 * stride) {
      const uint4 v0 = src[u], v1 = src[u + stride], v2 = src[u + 2 * stride], v3 = src[u + 3 * stride];
      dst[u] = v0; dst[u + stride] = v1; dst[u + 2 * stride] = v2; dst[u + 3 * stride] = v3;
    }
    for (; u < units; u += stride) dst[u] = src[u];
    const size_t tail = sg.bytes & 15;
    if (tail && b == 0 && threadIdx.x < tail) static_cast<unsigned char *>(sg.dst)[(units << 4) + threadIdx.x] = static_cast<const unsigned char *>(sg.src)[(units << 4) + threadIdx.x];
  }
}

// gather + calcBodyCov of one point of the sorted scan (out of line: sixteen inlined copies of the covariance would be the whole kernel)
__device__ __attribute__((noinline)) void scan_small_point(const FrameIngestArgs &a, int pos, int o) {
  const float px = a.xyz[(size_t)o * 3], py = a.xyz[(size_t)o * 3 + 1], pz = a.xyz[(size_t)o * 3 + 2];
  a.perm[pos] = o; a.x[pos] = px; a.y[pos] = py; a.z[pos] = pz;
  double c6[6];
  body_cov_point(px, py, pz, a.range_inc, a.degree_inc, a.deg2rad, c6);
#pragma unroll
  for (int e = 0; e < 6; e++) a.cb[(size_t)e * a.n + pos] = c6[e];
}

template <int IPT> struct ScanSmallSort {
  using type = rocprim::block_radix_sort<uint32_t, SCAN_SMALL_THREADS, IPT, int32_t>;
};

template <int IPT>
__global__ void __launch_bounds__(SCAN_SMALL_THREADS) k_frame_ingest(FrameIngestArgs a) {
  if (blockIdx.x == 0) {
    for (int w = threadIdx.x; w < a.zero0_words; w += SCAN_SMALL_THREADS) a.zero0[w] = 0u;
    for (int w = threadIdx.x; w < a.zero1_words; w += SCAN_SMALL_THREADS) a.zero1[w] = 0;
  }
  if constexpr (IPT > 0) {
    if (blockIdx.x == 0) {
      using Sort = typename ScanSmallSort<IPT>::type;
      __shared__ typename Sort::storage_type storage;
      uint32_t keys[IPT]; int32_t idx[IPT];
      const int t = threadIdx.x;
#pragma unroll
      for (int i = 0; i < IPT; i++) {            // blocked arrangement: item t * IPT + i; items past the scan sort behind every real key (30-bit keys, bit 30 set)
        const int g = t * IPT + i;
        if (g < a.n) { keys[i] = morton_key_of(a.xyz + (size_t)g * 3, a.inv_cell); idx[i] = g; }
        else { keys[i] = 0x40000000u; idx[i] = -1; }
      }
      Sort().sort_to_striped(keys, idx, storage, 0, 31);
#pragma unroll
      for (int i = 0; i < IPT; i++) {            // striped arrangement: position i * THREADS + t
        const int pos = i * SCAN_SMALL_THREADS + t;
        if (pos < a.n) scan_small_point(a, pos, idx[i]);
      }
      return;
    }
    frame_copy_segments(a, 1, (int)gridDim.x - 1);
  } else {
    frame_copy_segments(a, 0, (int)gridDim.x);
  }
}

// FrameRes of api_visual.inc: {livo2_lidar_result, livo2_visual_result, int32 timed_out, int32 pad}
__global__ void __launch_bounds__(256) k_frame_publish(const DevCtl *__restrict__ ctl, double *__restrict__ out) {
  constexpr int NL = (int)(sizeof(livo2_lidar_result) / 8), NV = (int)(sizeof(livo2_visual_result) / 8);
  const double *l = reinterpret_cast<const double *>(&ctl->lidar), *v = reinterpret_cast<const double *>(&ctl->visual);
  for (int k = threadIdx.x; k < NL; k += 256) out[k] = l[k];
  for (int k = threadIdx.x; k < NV; k += 256) out[NL + k] = v[k];
  if (threadIdx.x == 0) { const int2 f = make_int2(ctl->hdr.pad[0], 0); out[NL + NV] = __builtin_bit_cast(double, f); }
}
