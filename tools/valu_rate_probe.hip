// Issue cost of the VALU opcodes that dominate k_lidar_residual, measured on the device: cycles per instruction of ONE wave alone on its SIMD, and of two waves
// sharing a SIMD (512-thread block).  hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o fast-livo2_amd/lib/valu_rate_probe && fast-livo2_amd/lib/valu_rate_probe
// (round 6: decides whether the four voxel hashes of a point — 28 v_mul_lo_u32 — are worth replacing by 24-bit multiplies; profiles/r06_valu_rate_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64
#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND> __global__ void __launch_bounds__(1024) k_probe(unsigned long long *out, int iters) {
  unsigned a[8]; double d[8]; float f[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; d[i] = 1.0 + 1e-9 * (threadIdx.x + i); f[i] = 1.0f + 1e-6f * (threadIdx.x + i); }
  const unsigned c = 0x9e3779b1u; const double dc = 1.0000001; const float fc = 1.000001f;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (KIND == 0) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        OPS8(S)
#undef S
      } else if (KIND == 1) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        OPS8(S)
#undef S
      } else if (KIND == 2) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
        OPS8(S)
#undef S
      } else if (KIND == 3) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
        OPS8(S)
#undef S
      } else if (KIND == 4) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
        OPS8(S)
#undef S
      } else if (KIND == 5) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        OPS8(S)
#undef S
      } else if (KIND == 6) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(fc));
        OPS8(S)
#undef S
      } else if (KIND == 7) {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        OPS8(S)
#undef S
      } else if (KIND == 8) {
#define S(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
        OPS8(S)
#undef S
      } else if (KIND == 9) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        OPS8(S)
#undef S
      } else if (KIND == 10) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[i]));
        OPS8(S)
#undef S
      } else if (KIND == 11) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
        OPS8(S)
#undef S
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0; double dacc = 0; float facc = 0;
  for (int i = 0; i < 8; i++) { acc ^= a[i]; dacc += d[i]; facc += f[i]; }
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; out[4 + 4 * w] = t0; out[5 + 4 * w] = t1; out[6 + 4 * w] = r0; out[7 + 4 * w] = r1; }
  if (acc == 0x12345u && dacc == 3.0 && facc == 2.0f) out[blockIdx.x * 2 + 1] = 1;
}

static double g_ns_per_instr = 0;
template <int KIND> double run(int threads, unsigned long long *d_out, int iters) {
  hipLaunchKernelGGL(k_probe<KIND>, dim3(1), dim3(threads), 0, 0, d_out, iters);
  hipLaunchKernelGGL(k_probe<KIND>, dim3(1), dim3(threads), 0, 0, d_out, iters);
  hipDeviceSynchronize();
  unsigned long long h[4 + 64];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;                // the BLOCK: first wave's start to last wave's end (the oldest wave of a SIMD is served first)
  for (int w = 0; w < threads / 64; w++) { t0 = std::min(t0, h[4 + 4 * w]); t1 = std::max(t1, h[5 + 4 * w]); r0 = std::min(r0, h[6 + 4 * w]); r1 = std::max(r1, h[7 + 4 * w]); }
  g_ns_per_instr = (double)(r1 - r0) * 10.0 / ((double)iters * REP);          // s_memrealtime: 100 MHz
  return (double)(t1 - t0) / ((double)iters * REP);
}

int main() {
  unsigned long long *d_out; hipMalloc(&d_out, 1024);
  const int iters = 2000;
  const char *names[12] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_xor_b32", "v_fma_f32", "v_rcp_f64", "v_sqrt_f64", "v_mad_u32_u24", "v_alignbit_b32", "v_cvt_f32_f64"};
  printf("# cycles per instruction, 8 independent chains per lane: one wave (64 threads) | four waves, one per SIMD (256) | eight waves, two per SIMD (512: per-wave cost when the SIMD is shared)\n");
  double r[12][5], ns[12][5];
  const int T[5] = {64, 256, 512, 768, 1024};
#define RUN(K) for (int j = 0; j < 5; j++) { r[K][j] = run<K>(T[j], d_out, iters); ns[K][j] = g_ns_per_instr; }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  printf("# columns: 1 wave | 1 per SIMD | 2 per SIMD | 3 per SIMD | 4 per SIMD ; s_memtime ticks per instruction PER WAVE of the whole block (first start to last end), then ns (100 MHz clock); SIMD cost per wave-instruction = value / waves per SIMD\n");
  for (int k = 0; k < 12; k++) printf("%-16s %6.2f %6.2f %6.2f %6.2f %6.2f   ns %5.2f %5.2f %5.2f %5.2f %5.2f\n", names[k], r[k][0], r[k][1], r[k][2], r[k][3], r[k][4], ns[k][0], ns[k][1], ns[k][2], ns[k][3], ns[k][4]);
  hipFree(d_out);
  return 0;
}
