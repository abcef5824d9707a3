// Head and tail of a frame on the device (round 5): ONE launch takes a frame's inputs in, ONE more prepares a small scan, ONE hands the results out.
//
// A C1-shaped frame (~10 k points, ~350 patches: the reference's real operating point, preprocess.cpp:185 / LIVMapper.cpp:351-352) is a chain of DEPENDENT small
// commands on one stream; before this file the chain held ~33 of them (H2D of the scan, two memsets, Morton keys, 5-8 launches of the library's merge sort, gather,
// body covariance, H2D of the states, 5 x (residual, solve), five H2D of the image + sub-map, hand-over, the resident visual grid, three D2H) and lasted 0.485 ms of which
// well under half was kernel time: every command boundary costs 5-15 us on this stack, copies more than kernels.
//
//  k_frame_ingest: the frame's segments (states + header, scan, image, sub-map arrays) from ONE staging block to their device buffers — the source is either a device
//    arena filled by one H2D copy or the pinned staging block itself, read over the link by the kernel (no copy command at all) — plus, for a small scan, its Morton
//    keys, and the per-scan words to clear (block lifetimes, arrival tickets).
//  k_scan_rank: the scan's spatial order WITHOUT a sort.  The order wanted is the one a stable sort by Morton key gives, i.e. position(i) = #{j : (key_j, j) < (key_i, i)}.
//    For n <= 16 384 that count is cheaper to take directly than to run a block-wide radix sort (measured, tools/sort_probe.hip: rocprim::block_radix_sort of 10 800
//    pairs on 1 024 threads 45-55 us, its four digit passes serialised on ONE compute unit) or the library's device-wide merge sort (5-8 dependent launches): a block
//    owns 64 points (one per lane), its 16 waves each count over one sixteenth of the keys — the keys are wave-uniform, so they arrive by scalar loads, sixteen per
//    instruction, and a pair costs one compare and one add-with-carry — and wave 0 sums the sixteen counts, writes perm[] and does the SoA gather and calcBodyCov
//    (voxel_map.cpp:15-34, 349-360) of its 64 points at their final positions.  n^2 / 64 wave-pairs spread over every compute unit: ~10 us at 10 800 points.  Exactly the
//    permutation of the stable sort (tests/test_frame_ingest_gpu.py: every downstream sum keeps its bits).
//  k_frame_publish: both result blocks and the watchdog flag of a frame into its pinned result slot (one launch instead of three D2H copies).
#pragma once
#include "lidar_kernels.hpp"

#define SCAN_SMALL_MAX 16384
#define FRAME_MAX_SEGS 8
#define FRAME_COPY_BLOCKS 64
#define FRAME_THREADS 256

struct FrameSeg { const void *src; void *dst; unsigned long long bytes; };      // src and dst 16-byte aligned
struct FrameIngestArgs {
  FrameSeg seg[FRAME_MAX_SEGS];
  int32_t n_seg, n_keys;                  // segments; points of a small scan whose keys are wanted (0: none)
  uint32_t *zero0; int32_t *zero1;        // words to clear: block lifetimes of the last scan, arrival tickets of k_lidar_iteration
  int32_t zero0_words, zero1_words;
  const float *xyz;                       // [n_keys][3] the scan as the key blocks read it (the staging block / the arena / d_xyz_aos)
  uint32_t *keys;                         // [round_up(n_keys, 16)]: Morton keys, padded with 0xffffffff
  float inv_cell, pad;
};

// grid: FRAME_COPY_BLOCKS copy blocks (if n_seg > 0) followed by the key blocks (if n_keys > 0: one thread per padded key); block 0 also clears the per-scan words
__global__ void __launch_bounds__(FRAME_THREADS) k_frame_ingest(FrameIngestArgs a) {
  if (blockIdx.x == 0) {
    for (int w = threadIdx.x; w < a.zero0_words; w += FRAME_THREADS) a.zero0[w] = 0u;
    for (int w = threadIdx.x; w < a.zero1_words; w += FRAME_THREADS) a.zero1[w] = 0;
  }
  const int copy_blocks = a.n_seg > 0 ? FRAME_COPY_BLOCKS : 0;
  if ((int)blockIdx.x >= copy_blocks) {
    const int g = ((int)blockIdx.x - copy_blocks) * FRAME_THREADS + threadIdx.x;
    const int padded = (a.n_keys + 15) & ~15;
    if (g < a.n_keys) a.keys[g] = morton_key_of(a.xyz + (size_t)g * 3, a.inv_cell);
    else if (g < padded) a.keys[g] = 0xffffffffu;
    return;
  }
  const size_t stride = (size_t)copy_blocks * FRAME_THREADS;
  for (int s = 0; s < a.n_seg; s++) {
    const FrameSeg sg = a.seg[s];
    const size_t units = sg.bytes >> 4;
    const uint4 *src = static_cast<const uint4 *>(sg.src);
    uint4 *dst = static_cast<uint4 *>(sg.dst);
    size_t u = (size_t)blockIdx.x * FRAME_THREADS + threadIdx.x;
    // four independent 16-byte loads in flight per thread before the first store: over the link every load is a ~2-us round trip
    for (; u + 3 * stride < units; u += 4 * stride) {
      const uint4 v0 = src[u], v1 = src[u + stride], v2 = src[u + 2 * stride], v3 = src[u + 3 * stride];
      dst[u] = v0; dst[u + stride] = v1; dst[u + 2 * stride] = v2; dst[u + 3 * stride] = v3;
    }
    for (; u < units; u += stride) dst[u] = src[u];
    const size_t tail = sg.bytes & 15;
    if (tail && blockIdx.x == 0 && threadIdx.x < tail) static_cast<unsigned char *>(sg.dst)[(units << 4) + threadIdx.x] = static_cast<const unsigned char *>(sg.src)[(units << 4) + threadIdx.x];
  }
}

#define RANK_WAVES 16
#define RANK_THREADS (RANK_WAVES * LIVO2_WAVE)
struct ScanRankArgs {
  const uint32_t *keys;                   // [round_up(n, 16)], padded with 0xffffffff (k_frame_ingest)
  const float *xyz;                       // [n][3] the scan in the caller's order
  float *x, *y, *z; int32_t *perm; double *cb;
  double deg2rad; float range_inc, degree_inc; int32_t n, pad;
};
// #{j in [jb, je) : key_j < t}  (+ the ties before i when `exact`): keys wave-uniform (scalar loads), t per lane
template <bool EXACT> __device__ __forceinline__ uint32_t rank_count(const uint32_t *__restrict__ keys, int jb, int je, uint32_t t, uint32_t ki, int i) {
  uint32_t cnt = 0;
  for (int j = jb; j < je; j += 16) {
    const uint4 *p = reinterpret_cast<const uint4 *>(keys + j);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const uint32_t kk[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (EXACT) cnt += ((kk[u] < ki) || (kk[u] == ki && j + u < i)) ? 1u : 0u;
      else cnt += (kk[u] < t) ? 1u : 0u;
    }
  }
  return cnt;
}
__global__ void __launch_bounds__(RANK_THREADS) k_scan_rank(ScanRankArgs a) {
  __shared__ uint32_t part[RANK_WAVES][LIVO2_WAVE];
  const int lane = threadIdx.x & (LIVO2_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int i0 = (int)blockIdx.x * LIVO2_WAVE, i = i0 + lane;
  const int padded = (a.n + 15) & ~15;
  const uint32_t ki = (i < a.n) ? a.keys[i] : 0u;                       // (lanes past the scan count for nothing and write nothing)
  // wave w counts over the keys [w * P, (w + 1) * P), P a multiple of 16.  Keys j < i tie-break in favour of j: for a stretch that lies entirely before (behind) the
  // block's 64 points every lane compares with key_i + 1 (key_i) — one compare per pair; only the stretch around the block's own points needs the index test.
  const int P = (((padded + RANK_WAVES - 1) / RANK_WAVES) + 15) & ~15;
  const int jb = min(wave * P, padded), je = min(jb + P, padded);
  const int lo_end = min(max((i0 & ~15), jb), je);                      // [jb, lo_end): every j < i0 <= i
  const int hi_beg = min(max(((i0 + LIVO2_WAVE + 15) & ~15), jb), je);     // [hi_beg, je): every j >= i0 + 64 > i
  uint32_t cnt = rank_count<false>(a.keys, jb, lo_end, ki + 1u, ki, i);
  cnt += rank_count<true>(a.keys, lo_end, hi_beg, 0u, ki, i);
  cnt += rank_count<false>(a.keys, hi_beg, je, ki, ki, i);
  part[wave][lane] = cnt;
  __syncthreads();
  if (wave != 0 || i >= a.n) return;
  uint32_t pos = 0;
#pragma unroll
  for (int w = 0; w < RANK_WAVES; w++) pos += part[w][lane];
  const float px = a.xyz[(size_t)i * 3], py = a.xyz[(size_t)i * 3 + 1], pz = a.xyz[(size_t)i * 3 + 2];
  a.perm[pos] = i; a.x[pos] = px; a.y[pos] = py; a.z[pos] = pz;
  double c6[6];
  body_cov_point(px, py, pz, a.range_inc, a.degree_inc, a.deg2rad, c6);
#pragma unroll
  for (int e = 0; e < 6; e++) a.cb[(size_t)e * a.n + pos] = c6[e];
}

// FrameRes of api_visual.inc: {livo2_lidar_result, livo2_visual_result, int32 timed_out, int32 pad}
__global__ void __launch_bounds__(256) k_frame_publish(const DevCtl *__restrict__ ctl, double *__restrict__ out) {
  constexpr int NL = (int)(sizeof(livo2_lidar_result) / 8), NV = (int)(sizeof(livo2_visual_result) / 8);
  const double *l = reinterpret_cast<const double *>(&ctl->lidar), *v = reinterpret_cast<const double *>(&ctl->visual);
  for (int k = threadIdx.x; k < NL; k += 256) out[k] = l[k];
  // the step records are written up to n_steps only (a slot holds LIVO2_MAX_LEVELS x LIVO2_MAX_ITERS = 128 of them, 83 KB across the link, of which a frame fills ~15)
  constexpr int NVH = (int)(offsetof(livo2_visual_result, steps) / 8), NVS = (int)(sizeof(livo2_visual_step) / 8);
  const int ns = min(max(ctl->visual.n_steps, 0), LIVO2_MAX_LEVELS * LIVO2_MAX_ITERS);
  for (int k = threadIdx.x; k < NVH + ns * NVS && k < NV; k += 256) out[NL + k] = v[k];
  if (threadIdx.x == 0) { const int2 f = make_int2(ctl->hdr.pad[0], 0); out[NL + NV] = __builtin_bit_cast(double, f); }
}
