// What does an all-to-all hand-off between the blocks of a SMALL resident grid cost on gfx950, by placement and by the scope bits of its loads and stores?
// (the reference's real sub-map is <= 357 patches = 23 blocks of k_visual_update_persistent; its exchange crosses the XCDs with sc1 write-through
// stores and sc1 loads and costs the same ~3.5 us per step as with 250 blocks.)
// G logical blocks x 256 threads; per step every block publishes ROW tagged 8-byte words {tag << 32 | payload} and collects all G rows; a word whose tag is not the
// step's is simply loaded again, so no variant can return a wrong sum — a variant whose loads never see the stores times out instead (reported).
//   placement  spread : grid = G, the dispatcher deals the blocks round-robin over the 8 XCDs
//              packed : grid = 8 G, only blocks with blockIdx % 8 == 0 take part (all on one XCD: checked with HW_REG_XCC_ID)
//   flavour    0  store sc1 (agent scope, write-through)   load sc1 (agent scope)             <- the shipped idiom
//              1  store plain                               load sc0
//              2  store plain                               load sc1
//              3  store sc1                                 load sc0
//              4  store sc0 sc1 (system)                    load sc0 sc1
//              5  store plain                               global_atomic_or_x2 (v |= 0, pre-op value returned): a read performed BY the L2
//              6  store plain                               buffer_inv sc0, then plain loads (the L1 is emptied before every round of loads)
//              7  store sc1                                 global_atomic_or_x2 sc1
// (second edition: all loads of a round are issued before the first wait, as the kernel's collect does; the first edition waited per load and measured 4 serial trips)
// hipcc --offload-arch=gfx950 -O2 tools/xcd_exchange_probe.hip -o fast-livo2_amd/lib/xcd_exchange_probe ; xcd_exchange_probe [G=22] [steps=400]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long word;
#define ROW 40

template <int F> __device__ __forceinline__ void st(word *p, word v) {
  if (F == 0 || F == 3 || F == 7) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (F == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int F> __device__ __forceinline__ void ld_issue(word &v, const word *p) {
  if (F == 0 || F == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if (F == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else if (F == 5) { v = 0; asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=v"(v) : "v"(p), "v"((word)0) : "memory"); }
  else if (F == 7) { v = 0; asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0 sc1" : "=v"(v) : "v"(p), "v"((word)0) : "memory"); }
  else if (F == 6) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
}
template <int F> __device__ __forceinline__ word ld(const word *p) {
  word v;
  if (F == 0 || F == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Args { word *rows; int G, steps, packed; unsigned long long *ticks; int *bad, *xcc; };

template <int F> __global__ void __launch_bounds__(256) k_x(Args a) {
  if (a.packed && (blockIdx.x & 7) != 0) return;
  const int b = a.packed ? (int)blockIdx.x >> 3 : (int)blockIdx.x, tid = threadIdx.x, G = a.G;
  if (tid == 0) a.xcc[b] = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;      // HW_REG_XCC_ID, bits [3:0]
  __shared__ double red[256];
  __shared__ int timed_out;
  if (tid == 0) timed_out = 0;
  __syncthreads();
  unsigned long long t_begin = 0;
  for (int step = 0; step < a.steps && !timed_out; step++) {
    if (step == 8 && tid == 0) t_begin = __builtin_amdgcn_s_memrealtime();
    const uint32_t tag = (uint32_t)(step + 1);
    word *rows = a.rows + (size_t)(step & 3) * G * ROW;
    if (tid < ROW) st<F>(rows + (size_t)b * ROW + tid, ((word)tag << 32) | (uint32_t)(b * 131 + tid * 7 + step));
    // collect: thread t takes word t % ROW of rows t / ROW, t / ROW + 6, ...
    const int k = tid % ROW, r0 = tid / ROW;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tid < 6 * ROW) {
      word w[8]; uint32_t need = 0, have = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) if (r0 + 6 * u < G) need |= 1u << u;
      while (have != need && !timed_out) {
        if (F == 6) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) { const int r = (need >> u & 1u) ? r0 + 6 * u : r0; ld_issue<F>(w[u], rows + (size_t)r * ROW + k); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) { asm volatile("" : "+v"(w[u])); if ((need >> u & 1u) && (uint32_t)(w[u] >> 32) == tag) have |= 1u << u; }
        if (have != need) { if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { timed_out = 1; break; } __builtin_amdgcn_s_sleep(1); }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) if (need >> u & 1u) acc += (double)(uint32_t)w[u];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid == 0 && !timed_out && (step == a.steps - 1 || step == 5)) {      // (checked on two steps only: the serial check is ~8 us)
      double s = 0.0; for (int i = 0; i < 6 * ROW; i++) s += red[i];
      double want = 0.0; for (int r = 0; r < G; r++) for (int kk = 0; kk < ROW; kk++) want += (double)(uint32_t)(r * 131 + kk * 7 + step);
      if (s != want) atomicAdd(a.bad, 1);
    }
    __syncthreads();
  }
  if (tid == 0 && b == 0) { a.ticks[0] = __builtin_amdgcn_s_memrealtime() - t_begin; a.ticks[1] = (unsigned long long)timed_out; }
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 22, steps = argc > 2 ? atoi(argv[2]) : 400;
  printf("# G=%d blocks, %d steps: us per hand-off step (best of 3), wrong sums, time-out flag, XCC ids of the blocks\n", G, steps);
  for (int packed = 0; packed < 2; packed++) for (int f = 0; f < 8; f++) {
    if (f == 1 || f == 3) continue;      // sc0 loads never see another CU's store (first edition: they time out in either placement)
    Args a{}; a.G = G; a.steps = steps; a.packed = packed;
    const size_t rb = (size_t)4 * G * ROW * 8;
    if (hipMalloc(&a.rows, rb) || hipMalloc(&a.ticks, 16) || hipMalloc(&a.bad, 4) || hipMalloc(&a.xcc, 4 * G)) return 2;
    hipError_t e = hipMemset(a.bad, 0, 4); (void)e;
    double best = 1e30; int bad = 0; unsigned long long t[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
      e = hipMemset(a.rows, 0, rb);
      const dim3 grid(packed ? 8 * G : G);
      switch (f) {
        case 0: k_x<0><<<grid, 256>>>(a); break;
        case 1: k_x<1><<<grid, 256>>>(a); break;
        case 2: k_x<2><<<grid, 256>>>(a); break;
        case 3: k_x<3><<<grid, 256>>>(a); break;
        case 4: k_x<4><<<grid, 256>>>(a); break;
        case 5: k_x<5><<<grid, 256>>>(a); break;
        case 6: k_x<6><<<grid, 256>>>(a); break;
        default: k_x<7><<<grid, 256>>>(a); break;
      }
      if (hipDeviceSynchronize() != hipSuccess) { printf("flavour %d failed\n", f); return 1; }
      e = hipMemcpy(t, a.ticks, 16, hipMemcpyDeviceToHost); e = hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost);
      if (!t[1]) best = std::min(best, (double)t[0] * 0.01 / (steps - 8));
    }
    int xcc[64] = {0}; e = hipMemcpy(xcc, a.xcc, 4 * std::min(G, 64), hipMemcpyDeviceToHost);
    printf("%s flavour %d: %6.2f us per step, %d wrong, timeout %llu, xcc", packed ? "packed" : "spread", f, best, bad, t[1]);
    for (int i = 0; i < std::min(G, 24); i++) printf(" %d", xcc[i]);
    printf("\n");
    e = hipFree(a.rows); e = hipFree(a.ticks); e = hipFree(a.bad); e = hipFree(a.xcc);
  }
  return 0;
}
