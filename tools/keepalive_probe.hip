// Does HBM traffic from otherwise idle compute units keep the chip out of the state in which k_lidar_solve runs in its slow mode after the visual update (one box in three,
// profiles/r05_solve_by_position.txt)?  A background PROCESS: `blocks` small blocks stream a 1-GiB buffer for `seconds`; run tools/lidar_ab.py beside it and compare.
// hipcc --offload-arch=gfx950 -O3 -o fast-livo2_amd/lib/keepalive_probe tools/keepalive_probe.hip ; ./keepalive_probe <seconds> <blocks> <sleep>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_keepalive(const uint4 *buf, size_t n16, unsigned long long ticks, int sleep, unsigned *sink) {
  // workgroup i goes to XCD i % 8; a resident visual grid of 250 blocks needs ALL 32 compute units of XCD 0 and 1 (32 + 32 + 6 x 31): stay off those two
  if ((blockIdx.x & 7) < 2) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0;
  size_t i = (size_t)blockIdx.x * 1000003 + threadIdx.x;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    const unsigned *q = reinterpret_cast<const unsigned *>(buf + (i % n16));
    acc += __builtin_nontemporal_load(q) ^ __builtin_nontemporal_load(q + 1) ^ __builtin_nontemporal_load(q + 2) ^ __builtin_nontemporal_load(q + 3);
    i += 4099 * 64;                        // a new 1-KiB row of the wave every time: nothing of this stays in a cache
    if (sleep) __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0x12345678u) *sink = acc;
}
int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int blocks = argc > 2 ? atoi(argv[2]) : 4, sleep = argc > 3 ? atoi(argv[3]) : 0;
  uint4 *buf; unsigned *sink; const size_t bytes = 1ull << 30;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, bytes);
  hipLaunchKernelGGL(k_keepalive, dim3(blocks), dim3(64), 0, 0, buf, bytes / 16, (unsigned long long)(seconds * 1e8), sleep, sink);
  (void)hipDeviceSynchronize();
  printf("keepalive done: %d blocks x 64 threads for %.1f s\n", blocks, seconds);
  return 0;
}
