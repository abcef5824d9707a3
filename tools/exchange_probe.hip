// All-to-all exchange of tagged rows between the blocks of ONE resident grid — the hand-off of k_visual_update_persistent (fast-livo2_amd/csrc/visual_kernels.hpp) in
// isolation, to price its variants on gfx950 before touching the kernel ("pin down gfx950's L2-scope visibility with a litmus test first").
// G blocks x 512 threads; per step every block publishes a row of 40 doubles as 80 tagged 8-byte words {tag << 32 | 32 payload bits} (one write-through store per
// word) plus 16 tagged error words, then collects ALL rows and errors, sums them, and goes on when every word carries the step's tag.  Correctness never depends on
// the cache behaviour of a load: a word whose tag is not the step's is simply loaded again.  Variants of the COLLECT:
//   0  agent-scope (sc1, cache-bypassing) loads, retried as such                       <- what the kernel ships
//   1  first attempt with ordinary cached loads (hit the XCD's L2 if another block of the XCD already pulled the line), retries agent-scope
//   2  first attempt cached after an agent-scope acquire fence (buffer_inv sc1), retries agent-scope
//   3  two-level: row r is summed by leader block (r % 12) from the 21 rows r' = r (mod 12) (the slices of the kernel's reduction order: same bits), leaders publish
//      12 partial rows, everybody collects those 12 + the errors (agent-scope loads)
// NBUF buffers rotate with the step (the kernel uses 2).  Prints microseconds per step (block 0's s_memrealtime) and checks every sum.
// hipcc --offload-arch=gfx950 -O2 tools/exchange_probe.hip -o fast-livo2_amd/lib/exchange_probe ; exchange_probe [G=250] [steps=200] [M=4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long word;
#define ROW 40
#define NSUM 37
#define RPT 21
__device__ __forceinline__ word ld_agent(const word *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ word ld_cached(const word *p) { return *(const volatile word *)p; }
__device__ __forceinline__ void st_agent(word *p, word v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Args { word *rows, *errs, *part; int G, M, steps, variant, nbuf; unsigned long long *ticks; double *sums; int *bad; };

__device__ __forceinline__ double row_value(int r, int k, int step) { return (double)((r * 131 + k * 7 + step * 3) % 1000) * 0.125; }

template <int VARIANT> __device__ void collect_rows(const Args &a, const word *rows, int n_rows, uint32_t tag, double *scratch, int tid) {
  const int kidx = tid % ROW, slice = tid / ROW;                  // 12 slices x 40
  double acc = 0.0;
  word lo[RPT], hi[RPT];
  uint32_t need = 0, have = 0;
#pragma unroll
  for (int u = 0; u < RPT; u++) if (tid < 480 && kidx < NSUM && slice + 12 * u < n_rows) need |= 1u << u;
  bool first = true;
  if (VARIANT == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  while (have != need) {
    const uint32_t todo = need & ~have;
#pragma unroll
    for (int u = 0; u < RPT; u++)
      if (todo >> u & 1u) {
        const word *src = rows + ((size_t)(slice + 12 * u) * ROW + kidx) * 2;
        if ((VARIANT == 1 || VARIANT == 2) && first) { lo[u] = ld_cached(src); hi[u] = ld_cached(src + 1); }
        else { lo[u] = ld_agent(src); hi[u] = ld_agent(src + 1); }
      }
#pragma unroll
    for (int u = 0; u < RPT; u++) if ((todo >> u & 1u) && (uint32_t)(lo[u] >> 32) == tag && (uint32_t)(hi[u] >> 32) == tag) have |= 1u << u;
    first = false;
    if (have != need) __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int u = 0; u < RPT; u++) if (need >> u & 1u) acc += __longlong_as_double((long long)((lo[u] & 0xffffffffull) | (hi[u] << 32)));
  if (tid < 480) scratch[slice * 41 + kidx] = acc;
}

template <int VARIANT> __global__ void __launch_bounds__(512) k_exchange(Args a) {
  __shared__ double scratch[12 * 41];
  __shared__ double sums[ROW];
  __shared__ float errsum;
  const int tid = threadIdx.x, b = blockIdx.x, G = a.G, M = a.M;
  unsigned long long t_begin = 0;
  for (int step = 0; step < a.steps; step++) {
    if (step == 8 && tid == 0) t_begin = __builtin_amdgcn_s_memrealtime();
    const int buf = step % a.nbuf;
    const uint32_t tag = (uint32_t)(step + 1);
    word *rows = a.rows + (size_t)buf * G * ROW * 2, *errs = a.errs + (size_t)buf * M, *part = a.part + (size_t)buf * 12 * ROW * 2;
    // publish
    if (tid < NSUM) {
      const word bits = (word)__double_as_longlong(row_value(b, tid, step)), hi = (word)tag << 32;
      word *dst = rows + ((size_t)b * ROW + tid) * 2;
      st_agent(dst, hi | (bits & 0xffffffffull)); st_agent(dst + 1, hi | (bits >> 32));
    }
    for (int p = b * 16 + tid; p < M && tid < 16; p += G * 16) st_agent(errs + p, ((word)tag << 32) | (word)__float_as_uint((float)(p % 97)));
    // collect
    if (VARIANT != 3) collect_rows<VARIANT>(a, rows, G, tag, scratch, tid);
    else {
      if (b < 12) {                                                  // leader of slice b: rows b, b + 12, ... summed in order (thread k: value k); all loads of a round in flight together
        if (tid < NSUM) {
          word lo[RPT], hi[RPT];
          uint32_t need = 0, have = 0;
#pragma unroll
          for (int u = 0; u < RPT; u++) if (b + 12 * u < G) need |= 1u << u;
          while (have != need) {
            const uint32_t todo = need & ~have;
#pragma unroll
            for (int u = 0; u < RPT; u++) if (todo >> u & 1u) { const word *src = rows + ((size_t)(b + 12 * u) * ROW + tid) * 2; lo[u] = ld_agent(src); hi[u] = ld_agent(src + 1); }
#pragma unroll
            for (int u = 0; u < RPT; u++) if ((todo >> u & 1u) && (uint32_t)(lo[u] >> 32) == tag && (uint32_t)(hi[u] >> 32) == tag) have |= 1u << u;
            if (have != need) __builtin_amdgcn_s_sleep(1);
          }
          double acc = 0.0;
#pragma unroll
          for (int u = 0; u < RPT; u++) if (need >> u & 1u) acc += __longlong_as_double((long long)((lo[u] & 0xffffffffull) | (hi[u] << 32)));
          const word bits = (word)__double_as_longlong(acc), hiw = (word)tag << 32;
          word *dst = part + ((size_t)b * ROW + tid) * 2;
          st_agent(dst, hiw | (bits & 0xffffffffull)); st_agent(dst + 1, hiw | (bits >> 32));
        }
      }
      if (tid < 480) {                                               // everybody: the 12 partial rows (slice = row)
        const int kidx = tid % ROW, slice = tid / ROW;
        double v = 0.0;
        if (kidx < NSUM) {
          const word *src = part + ((size_t)slice * ROW + kidx) * 2;
          word lo = ld_agent(src), hi = ld_agent(src + 1);
          while ((uint32_t)(lo >> 32) != tag || (uint32_t)(hi >> 32) != tag) { __builtin_amdgcn_s_sleep(1); lo = ld_agent(src); hi = ld_agent(src + 1); }
          v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
        }
        scratch[slice * 41 + kidx] = v;
      }
    }
    // errors: every block needs all of them (the float error chain of the reference is sequential)
    float e = 0.0f;
    {
      word w[8];                                                    // (M <= 4096) one batch of loads, then only the late words again
#pragma unroll
      for (int u = 0; u < 8; u++) { const int p = tid + 512 * u; w[u] = p < M ? ((VARIANT == 1 || VARIANT == 2) ? ld_cached(errs + p) : ld_agent(errs + p)) : ((word)tag << 32); }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int p = tid + 512 * u; while ((uint32_t)(w[u] >> 32) != tag) { __builtin_amdgcn_s_sleep(1); w[u] = ld_agent(errs + p); } e += __uint_as_float((uint32_t)w[u]); }
    }
    __syncthreads();
    if (tid == 0) errsum = 0.0f;
    __syncthreads();
    atomicAdd(&errsum, e);
    if (tid < ROW) { double r = scratch[tid]; for (int s = 1; s < 12; s++) r += scratch[s * 41 + tid]; sums[tid] = r; }
    __syncthreads();
    // check (every block holds the same sums)
    if (tid < NSUM && (step == a.steps - 1 || step == 11)) {
      double want[12]; for (int s = 0; s < 12; s++) want[s] = 0.0;
      for (int r = 0; r < G; r++) want[r % 12] += row_value(r, tid, step);
      double w = want[0]; for (int s = 1; s < 12; s++) w += want[s];
      if (w != sums[tid]) atomicAdd(a.bad, 1);
    }
    __syncthreads();
  }
  if (tid == 0 && b == 0) a.ticks[0] = __builtin_amdgcn_s_memrealtime() - t_begin;
  if (tid < NSUM && b == 0) a.sums[tid] = sums[tid];
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 250, steps = argc > 2 ? atoi(argv[2]) : 200, M = argc > 3 ? atoi(argv[3]) : 4000;
  for (int nbuf : {2, 8}) for (int variant = 0; variant < 4; variant++) {
    Args a{}; a.G = G; a.M = M; a.steps = steps; a.variant = variant; a.nbuf = nbuf;
    const size_t rb = (size_t)nbuf * G * ROW * 2 * 8, eb = (size_t)nbuf * M * 8, pb = (size_t)nbuf * 12 * ROW * 2 * 8;
    if (hipMalloc(&a.rows, rb) || hipMalloc(&a.errs, eb) || hipMalloc(&a.part, pb) || hipMalloc(&a.ticks, 8) || hipMalloc(&a.sums, 8 * ROW) || hipMalloc(&a.bad, 4)) return 2;
    hipError_t e = hipMemset(a.rows, 0, rb); e = hipMemset(a.errs, 0, eb); e = hipMemset(a.part, 0, pb); e = hipMemset(a.bad, 0, 4); (void)e;
    double best = 1e30; int bad = 0;
    for (int rep = 0; rep < 3; rep++) {
      e = hipMemset(a.rows, 0, rb); e = hipMemset(a.errs, 0, eb); e = hipMemset(a.part, 0, pb);
      switch (variant) {
        case 0: k_exchange<0><<<G, 512>>>(a); break;
        case 1: k_exchange<1><<<G, 512>>>(a); break;
        case 2: k_exchange<2><<<G, 512>>>(a); break;
        default: k_exchange<3><<<G, 512>>>(a); break;
      }
      if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d failed\n", variant); return 1; }
      unsigned long long t = 0; e = hipMemcpy(&t, a.ticks, 8, hipMemcpyDeviceToHost); e = hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost);
      best = std::min(best, (double)t * 0.01 / (steps - 8));
    }
    printf("G=%d M=%d nbuf=%d variant %d: %.2f us per step, %d wrong sums\n", G, M, nbuf, variant, best, bad);
    e = hipFree(a.rows); e = hipFree(a.errs); e = hipFree(a.part); e = hipFree(a.ticks); e = hipFree(a.sums); e = hipFree(a.bad);
  }
  return 0;
}
