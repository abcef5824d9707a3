// Does an XCD's L2 keep lines ACROSS kernel launches on gfx950?  (Question behind "can iterations 2..5 of a LiDAR update find the map records of iteration 1 in L2":
// k_lidar_residual re-reads the same ~34 MB — 4.2 MB per XCD slab — in every launch, rocprofv3 shows a 23 % L2 hit rate and no sign of reuse between launches.)
// A grid of 2048 blocks x 256 threads reads a buffer of S bytes with 16-B loads; block b reads from slab (b % 8) (the XCD it runs on) when `local`, from slab
// ((b / 8) % 8) otherwise.  The same launch is repeated back to back; the duration of a launch is max(end) - min(start) of its blocks' s_memrealtime stamps.
// If L2 contents survive the launch boundary the repeat of a <= 4 MiB-per-XCD slab runs at L2 speed and the curve has a knee at 32 MiB; if the boundary drops
// them the curve is smooth (Infinity Cache / HBM).  Variant `nt`: a second, streaming buffer of the same size is read with non-temporal loads between the
// resident one's passes (does a streaming read evict the resident set?).
// hipcc --offload-arch=gfx950 -O2 tools/l2_retention_probe.hip -o fast-livo2_amd/lib/l2_retention_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
#define GRID 2048
#define BLOCK 256

template <int NT> __device__ __forceinline__ v4f ld(const v4f *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <int NT> __global__ void __launch_bounds__(BLOCK) k_read(const v4f *__restrict__ buf, size_t slab_vec, int local, float *__restrict__ out,
                                                                  unsigned long long *__restrict__ t0, unsigned long long *__restrict__ t1, int rep) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned long long s = __builtin_amdgcn_s_memrealtime();
  const int slab = local ? (b & 7) : ((b >> 3) & 7);
  const int k = b >> 3, per = GRID / 8;                        // k-th of the 256 blocks that work on a slab (local) / some slab (not local: k's low bits pick it)
  const int part = local ? k : ((b & 7) * (per / 8) + (b >> 6));
  const size_t chunk = slab_vec / per;                          // 16-B vectors per block
  const v4f *p = buf + (size_t)slab * slab_vec + (size_t)part * chunk;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = tid; i < chunk; i += 4 * BLOCK) {
    v4f a = ld<NT>(p + i), c = i + BLOCK < chunk ? ld<NT>(p + i + BLOCK) : acc, d = i + 2 * BLOCK < chunk ? ld<NT>(p + i + 2 * BLOCK) : acc,
        e = i + 3 * BLOCK < chunk ? ld<NT>(p + i + 3 * BLOCK) : acc;
    acc += a; acc += c; acc += d; acc += e;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[b * BLOCK + tid] = acc.x;
  __syncthreads();
  if (tid == 0) { t0[rep * GRID + b] = s; t1[rep * GRID + b] = __builtin_amdgcn_s_memrealtime(); }
}

#define REPS 9
static void spans_us(unsigned long long *d0, unsigned long long *d1, std::vector<double> &us) {
  std::vector<unsigned long long> a(GRID * REPS), b(GRID * REPS);
  hipError_t e = hipMemcpy(a.data(), d0, GRID * REPS * 8, hipMemcpyDeviceToHost); e = hipMemcpy(b.data(), d1, GRID * REPS * 8, hipMemcpyDeviceToHost); (void)e;
  for (int r = 0; r < REPS; r++)
    us.push_back((double)(*std::max_element(b.begin() + r * GRID, b.begin() + (r + 1) * GRID) - *std::min_element(a.begin() + r * GRID, a.begin() + (r + 1) * GRID)) * 0.01);
}

int main() {
  const size_t max_bytes = 96ull << 20;
  v4f *res, *stream; float *out; unsigned long long *t0, *t1;
  if (hipMalloc(&res, max_bytes) || hipMalloc(&stream, max_bytes) || hipMalloc(&out, GRID * BLOCK * 4) || hipMalloc(&t0, (REPS + 1) * GRID * 8) || hipMalloc(&t1, (REPS + 1) * GRID * 8)) return 2;
  hipError_t e = hipMemset(res, 0, max_bytes); e = hipMemset(stream, 0, max_bytes); (void)e;
  printf("# MiB total (per XCD) : us of the 1st launch, median us of launches 2..9 -> GB/s   [local = slab of the block's own XCD]\n");
  for (int local = 1; local >= 0; local--)
    for (int mode = 0; mode < 3; mode++) {               // 0: resident buffer only; 1: + plain streaming buffer between passes; 2: + non-temporal streaming buffer
      for (size_t mib : {4, 8, 16, 24, 28, 32, 40, 48, 64, 96}) {
        const size_t bytes = mib << 20, slab_vec = bytes / 8 / 16;
        // evict: read the other buffer at full size first
        k_read<0><<<GRID, BLOCK>>>(stream, max_bytes / 8 / 16, 1, out, t0, t1, REPS);
        std::vector<double> us;
        for (int rep = 0; rep < REPS; rep++) {                // back to back on one stream, as the iterations of an update are: nothing but launches in between
          k_read<0><<<GRID, BLOCK>>>(res, slab_vec, local, out, t0, t1, rep);
          if (mode == 1) k_read<0><<<GRID, BLOCK>>>(stream, slab_vec / 4, local, out, t0, t1, REPS);
          if (mode == 2) k_read<1><<<GRID, BLOCK>>>(stream, slab_vec / 4, local, out, t0, t1, REPS);
        }
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        spans_us(t0, t1, us);
        const double first = us[0];
        std::sort(us.begin() + 1, us.end());
        const double med = us[1 + 4];
        printf("local %d mode %d  %3zu MiB (%5.2f) : first %7.2f us  repeat %7.2f us -> %7.0f GB/s\n", local, mode, mib, (double)mib / 8, first, med, (double)bytes / med * 1e-3);
      }
    }
  return 0;
}
