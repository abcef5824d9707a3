// Cross-queue dependency latency on one GPU: kernel A on stream 1, an event behind it, stream 2 waits for the event and runs kernel B — how long after A's last
// instruction does B's first one run, against the same pair back to back on ONE stream?  (Would a k_lidar_solve that polls the partial rows on a second stream beside
// the residual grid pay for itself?  It would have to hand the next residual launch over through exactly this dependency.)
//   hipcc --offload-arch=gfx950 -O2 -o fast-livo2_amd/lib/xq_probe tools/xq_probe.hip && fast-livo2_amd/lib/xq_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void k_stamp(unsigned long long *out, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)spin) t = __builtin_amdgcn_s_memrealtime();      // 100 MHz ticks
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = t; }
}
int main() {
  unsigned long long *d; CHECK(hipMalloc(&d, 4096 * 16));
  hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int N = 400;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto &e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int mode = 0; mode < 3; mode++) {               // 0: one stream; 1: A on s1, B on s2 behind an event; 2: ping-pong s1 -> s2 -> s1 (as res / solve / res would)
    for (int rep = 0; rep < 2; rep++) {
      for (int i = 0; i < N; i++) {
        if (mode == 0) { hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, s1, d + 4 * i, 500); hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s1, d + 4 * i + 2, 300); }
        else {
          if (mode == 2 && i > 0) CHECK(hipStreamWaitEvent(s1, ev[2 * i - 1], 0));
          hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, s1, d + 4 * i, 500);
          CHECK(hipEventRecord(ev[2 * i], s1)); CHECK(hipStreamWaitEvent(s2, ev[2 * i], 0));
          hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s2, d + 4 * i + 2, 300);
          CHECK(hipEventRecord(ev[2 * i + 1], s2));
        }
      }
      CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
    }
    std::vector<unsigned long long> h(4 * N); CHECK(hipMemcpy(h.data(), d, 4 * N * 8, hipMemcpyDeviceToHost));
    std::vector<double> ab, ba;
    for (int i = 1; i < N; i++) { ab.push_back(((double)h[4 * i + 2] - (double)h[4 * i + 1]) / 100.0); ba.push_back(((double)h[4 * i] - (double)h[4 * (i - 1) + 3]) / 100.0); }
    std::sort(ab.begin(), ab.end()); std::sort(ba.begin(), ba.end());
    std::printf("mode %d: end of A (256 blocks) -> start of B: median %.2f us (p10 %.2f, p90 %.2f) ; end of B -> start of the next A: median %.2f us (p10 %.2f, p90 %.2f)\n", mode,
                ab[ab.size() / 2], ab[ab.size() / 10], ab[ab.size() * 9 / 10], ba[ba.size() / 2], ba[ba.size() / 10], ba[ba.size() * 9 / 10]);
  }
  return 0;
}
