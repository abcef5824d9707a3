// Device allocations of liblivo2_hip.so: EVERY hipMalloc / hipFree of the library goes through dev_malloc / dev_free (DMALLOC / DFREE record the source line).
// Default: exact-size hipMalloc, nothing else (no slack: round 3's "+ 8192" behind ensure() is gone).
//
// Debug modes, selected by the environment variable LIVO2_REDZONE before the first allocation of the process (tests/test_redzone_gpu.py):
//   1  write redzones: RZ_GUARD poisoned bytes in front of and behind every allocation; livo2_debug_redzone_check (called by livo2_ctx_synchronize and every
//      *_fetch in this mode) scans all guards with k_rz_scan and names the allocation (source line, size, side, offset) whose guard changed.
//   2  electric fence behind: the allocation is its own HIP virtual-memory mapping whose END is the end of the mapped range — the next byte is reserved but
//      unmapped address space, so an out-of-bounds READ or write past the end faults at once and deterministically ("Memory access fault by GPU" + the abort
//      handler's table of allocations), not only when the neighbouring page happens to be unmapped on some box.  User pointers keep 16-byte alignment
//      (the widest vector access of the kernels), so up to 15 bytes behind an odd-sized array stay mapped.
//   3  electric fence in front: the allocation starts at the first byte of its mapping, below it reserved unmapped address space (index underflow).
// Modes 2 / 3 serialise nothing by themselves; run them with AMD_SERIALIZE_KERNEL=3 to have the fault reported against the launch that caused it.
#pragma once
#include <hip/hip_runtime.h>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace devalloc {

constexpr size_t RZ_GUARD = 65536;
constexpr uint32_t RZ_WORD = 0xCBCBCBCBu;

struct Rec { void *base; size_t bytes; size_t mapped; int line; int mode; hipMemGenericAllocationHandle_t handle; size_t reserved; };

inline const char *src_part(int tagged_line) {                   // LIVO2_HERE = part * 100000 + line
  static const char *names[] = {"livo2_api.hip", "api_map.inc", "api_imu.inc", "api_map_tree.inc", "api_lidar.inc", "api_retrieve.inc", "api_visual.inc"};
  const int k = tagged_line / 100000;
  return (k >= 0 && k < 7) ? names[k] : "?";
}
inline int mode() {
  static const int m = [] { const char *e = std::getenv("LIVO2_REDZONE"); return e ? std::atoi(e) : 0; }();
  return m;
}
inline std::mutex &mtx() { static std::mutex m; return m; }
inline std::unordered_map<void *, Rec> &table() { static std::unordered_map<void *, Rec> t; return t; }
struct Freed { void *user; size_t bytes; int line; };
inline std::vector<Freed> &graveyard() { static std::vector<Freed> g; return g; }      // freed allocations, oldest first (a fault inside one of them = use after free)

inline void dump_table(FILE *f) {
  // (called from the abort handler too: no locking, plain stdio — a debugging aid, not a service)
  fprintf(f, "livo2 device allocations (LIVO2_REDZONE=%d): user pointer .. end, bytes, source line (part * 100000 + line; parts: 0 livo2_api.hip, 1 api_map.inc, 2 api_imu.inc, 3 api_map_tree.inc, 4 api_lidar.inc, 5 api_retrieve.inc, 6 api_visual.inc)\n", mode());
  for (auto &kv : table())
    fprintf(f, "  %p .. %p  %zu B  line %d\n", kv.first, (void *)((char *)kv.first + kv.second.bytes), kv.second.bytes, kv.second.line);
  fprintf(f, "freed allocations (oldest first; an address inside one of them is a use after free):\n");
  for (auto &g : graveyard())
    fprintf(f, "  freed %p .. %p  %zu B  line %d\n", g.user, (void *)((char *)g.user + g.bytes), g.bytes, g.line);
  fflush(f);
}
inline void on_abort(int) { dump_table(stderr); std::signal(SIGABRT, SIG_DFL); std::abort(); }

__global__ void k_rz_fill(uint32_t *p, size_t words) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = RZ_WORD;
}
struct RzDesc { const uint32_t *guard; int id; int side; };
struct RzOut { unsigned long long bad_words; int first_id; int first_side; long long first_byte; int lock; int pad; long long last_byte; uint32_t sample[8]; };
__global__ void k_rz_scan(const RzDesc *d, RzOut *out) {
  const RzDesc g = d[blockIdx.x];
  unsigned long long bad = 0; long long first = -1;
  for (size_t i = threadIdx.x; i < RZ_GUARD / 4; i += blockDim.x)
    if (g.guard[i] != RZ_WORD) { bad++; if (first < 0) first = (long long)i * 4; }
  if (bad) {
    atomicAdd(&out->bad_words, bad);
    if (atomicCAS(&out->lock, 0, 1) == 0) {                       // (one thread of one block describes its guard: extent of the damage and the first damaged words)
      out->first_id = g.id; out->first_side = g.side;
      long long lo = -1, hi = -1; int ns = 0;
      for (size_t i = 0; i < RZ_GUARD / 4; i++) if (g.guard[i] != RZ_WORD) { if (lo < 0) lo = (long long)i * 4; hi = (long long)i * 4; if (ns < 8) out->sample[ns++] = g.guard[i]; }
      out->first_byte = lo; out->last_byte = hi;
    }
  }
}

inline hipError_t fence_alloc(void **p, size_t bytes, int line, int m) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0;
  static const bool recommended = [] { const char *g = std::getenv("LIVO2_FENCE_GRAN"); return g && g[0] == 'r'; }();      // "recommended": the runtime's preferred granule (2 MiB) instead of the minimum (4 KiB)
  if ((e = hipMemGetAllocationGranularity(&gran, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  const size_t mapped = (bytes + gran - 1) / gran * gran, reserved = mapped + 2 * gran;
  void *va = nullptr;
  if ((e = hipMemAddressReserve(&va, reserved, gran, nullptr, 0)) != hipSuccess) return e;
  hipMemGenericAllocationHandle_t h;
  // (every early return gives back what it has taken so far: advisor, round 4)
  if ((e = hipMemCreate(&h, mapped, &prop, 0)) != hipSuccess) { hipError_t e2 = hipMemAddressFree(va, reserved); (void)e2; return e; }
  char *map_at = (char *)va + gran;                         // one unmapped granule on either side of the mapping
  if ((e = hipMemMap(map_at, mapped, 0, h, 0)) != hipSuccess) { hipError_t e2 = hipMemRelease(h); e2 = hipMemAddressFree(va, reserved); (void)e2; return e; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(map_at, mapped, &acc, 1)) != hipSuccess) { hipError_t e2 = hipMemUnmap(map_at, mapped); e2 = hipMemRelease(h); e2 = hipMemAddressFree(va, reserved); (void)e2; return e; }
  char *user = (m == 2) ? map_at + ((mapped - bytes) & ~(size_t)15) : map_at;
  *p = user;
  std::lock_guard<std::mutex> lk(mtx());
  table()[user] = Rec{va, bytes, mapped, line, m, h, reserved};
  return hipSuccess;
}

// LIVO2_POISON=<byte, e.g. 0xCB>: every new allocation (any mode, also the plain one) is filled with that byte, so that a kernel which READS memory nobody wrote
// computes on garbage instead of on whatever the allocator recycled (hipMalloc mostly hands out zeros, which hides such reads).  LIVO2_POISON_LINES=<lo>:<hi>
// restricts the fill to the allocations made at those source lines (tools/poison_bisect.py walks the lines to name the buffer a failing test depends on).
inline int poison_byte() {
  static const int b = [] { const char *e = std::getenv("LIVO2_POISON"); return e ? (int)(std::strtol(e, nullptr, 0) & 0xff) : -1; }();
  return b;
}
inline bool poison_line(int line) {        // LIVO2_POISON_LINES=lo:hi restricts the fill to allocations made at source lines lo..hi
  static const std::pair<int, int> r = [] {
    const char *e = std::getenv("LIVO2_POISON_LINES");
    int lo = 0, hi = 1 << 30;
    if (e) { if (std::sscanf(e, "%d:%d", &lo, &hi) < 2) hi = lo; }
    return std::make_pair(lo, hi);
  }();
  return line >= r.first && line <= r.second;
}
inline hipError_t dev_malloc_raw(void **p, size_t bytes, int line);
inline hipError_t dev_malloc(void **p, size_t bytes, int line) {
  hipError_t e = dev_malloc_raw(p, bytes, line);
  if (e != hipSuccess || !*p || poison_byte() < 0) return e;
  if (!poison_line(line)) return e;
  if ((e = hipMemset(*p, poison_byte(), bytes)) != hipSuccess) return e;
  return hipDeviceSynchronize();
}
inline hipError_t dev_malloc_raw(void **p, size_t bytes, int line) {
  const int m = mode();
  if (bytes == 0) { *p = nullptr; return hipSuccess; }
  if (m == 0) return hipMalloc(p, bytes);
  static const bool hooked = [] { std::signal(SIGABRT, on_abort); return true; }();
  (void)hooked;
  if (m == 2 || m == 3) return fence_alloc(p, bytes, line, m);
  // mode 1: [guard | user, rounded up to 256 B | guard]
  const size_t body = (bytes + 255) & ~(size_t)255;
  char *base = nullptr;
  hipError_t e = hipMalloc((void **)&base, body + 2 * RZ_GUARD);
  if (e != hipSuccess) return e;
  k_rz_fill<<<64, 256>>>((uint32_t *)base, RZ_GUARD / 4);
  k_rz_fill<<<64, 256>>>((uint32_t *)(base + RZ_GUARD + bytes / 4 * 4 + ((bytes & 3) ? 4 : 0)), (body - (bytes + 3) / 4 * 4 + RZ_GUARD) / 4);   // the rounding tail is guard too
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  *p = base + RZ_GUARD;
  std::lock_guard<std::mutex> lk(mtx());
  table()[*p] = Rec{base, bytes, body + 2 * RZ_GUARD, line, 1, {}, 0};
  return hipSuccess;
}

inline hipError_t dev_free(void *p) {
  if (!p) return hipSuccess;
  if (mode() == 0) return hipFree(p);
  Rec r;
  // the lock is held until the memory is gone: check() (another host thread, another context) scans the guards of every allocation in the table and must never look
  // at one that is being freed — its address may already belong to a new, larger allocation whose user bytes would then be taken for a damaged guard
  std::lock_guard<std::mutex> lk(mtx());
  {
    auto it = table().find(p);
    if (it == table().end()) return hipFree(p);
    r = it->second; table().erase(it);
    if (graveyard().size() < 65536) graveyard().push_back(Freed{p, r.bytes, r.line});
  }
  hipError_t e = hipDeviceSynchronize();
  if (r.mode == 1) return hipFree(r.base);
  char *map_at = (char *)r.base + (r.reserved - r.mapped) / 2;
  if ((e = hipMemUnmap(map_at, r.mapped)) != hipSuccess) return e;
  if ((e = hipMemRelease(r.handle)) != hipSuccess) return e;
  return hipMemAddressFree(r.base, r.reserved);
}

// Mode 1: scan every guard of the process.  Returns the number of damaged 4-byte words (0 = intact) and describes the first damaged allocation in `msg`.
inline long long check(char *msg, size_t msg_len) {
  if (msg && msg_len) msg[0] = 0;
  if (mode() != 1) return 0;
  std::vector<RzDesc> desc; std::vector<Rec> recs; std::vector<void *> users;
  std::lock_guard<std::mutex> lk(mtx());                      // held to the end: no allocation of the table is freed (or its address re-used) while its guards are read
  {
    for (auto &kv : table()) {
      const Rec &r = kv.second;
      const int id = (int)recs.size();
      recs.push_back(r); users.push_back(kv.first);
      desc.push_back(RzDesc{(const uint32_t *)r.base, id, 0});
      desc.push_back(RzDesc{(const uint32_t *)((char *)r.base + r.mapped - RZ_GUARD), id, 1});
      // (the rounding tail between the last user byte and the rear guard is poisoned as well; it is checked through a third descriptor when it is a whole guard
      //  long only — the rear guard proper catches every overrun of more than 255 bytes, the tail word check below the shorter ones)
    }
  }
  if (desc.empty()) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  RzDesc *d_desc = nullptr; RzOut *d_out = nullptr;
  if (hipMalloc((void **)&d_desc, desc.size() * sizeof(RzDesc)) != hipSuccess || hipMalloc((void **)&d_out, sizeof(RzOut)) != hipSuccess) return -1;
  RzOut out = {}; out.first_id = -1;
  hipError_t e = hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(RzDesc), hipMemcpyHostToDevice);
  e = hipMemcpy(d_out, &out, sizeof(out), hipMemcpyHostToDevice);
  k_rz_scan<<<(int)desc.size(), 256>>>(d_desc, d_out);
  e = hipMemcpy(&out, d_out, sizeof(out), hipMemcpyDeviceToHost);
  // short overruns: the (< 256 B) rounding tail right behind the user bytes
  long long tail_bad = 0; int tail_id = -1; long long tail_off = 0;
  for (size_t i = 0; i < recs.size() && tail_id < 0; i++) {
    const Rec &r = recs[i];
    const size_t start = (r.bytes + 3) / 4 * 4, body = r.mapped - 2 * RZ_GUARD;
    if (body == start) continue;
    uint32_t w[64];
    if (hipMemcpy(w, (char *)users[i] + start, body - start, hipMemcpyDeviceToHost) != hipSuccess) continue;
    for (size_t k = 0; k < (body - start) / 4; k++) if (w[k] != RZ_WORD) { tail_bad++; if (tail_id < 0) { tail_id = (int)i; tail_off = (long long)(start + 4 * k - r.bytes); } }
  }
  e = hipFree(d_desc); e = hipFree(d_out);
  e = hipGetLastError(); (void)e;                              // leave no error of the checker's own calls behind: the caller's next library call (rocPRIM) reads the thread's last error
  const long long bad = (long long)out.bad_words + tail_bad;
  if (bad && msg && msg_len) {
    if (tail_id >= 0)
      snprintf(msg, msg_len, "redzone: allocation of %zu B made at %s:%d was written %lld B BEHIND its end", recs[tail_id].bytes, src_part(recs[tail_id].line), recs[tail_id].line % 100000, tail_off);
    else if (out.first_id >= 0) {
      const Rec &r = recs[out.first_id];
      if (out.first_side == 0) snprintf(msg, msg_len, "redzone: allocation of %zu B made at %s:%d was written %lld B IN FRONT of its start (%llu damaged words in all)",
                                        r.bytes, src_part(r.line), r.line % 100000, (long long)RZ_GUARD - out.first_byte, out.bad_words);
      else snprintf(msg, msg_len, "redzone: allocation of %zu B made at %s:%d was written %lld .. %lld B BEHIND its end (%llu damaged words in all; first words %08x %08x %08x %08x %08x %08x)", r.bytes, src_part(r.line), r.line % 100000,
                    (long long)(r.mapped - 2 * RZ_GUARD - r.bytes) + out.first_byte, (long long)(r.mapped - 2 * RZ_GUARD - r.bytes) + out.last_byte, out.bad_words,
                    out.sample[0], out.sample[1], out.sample[2], out.sample[3], out.sample[4], out.sample[5]);
    }
  }
  return bad;
}

// hipMemcpyAsync as the library calls it.  From pageable host memory the runtime stages a copy into ordinary device memory before it returns, so callers may free
// or reuse the host buffer at once — the library (and every HIP program) relies on that.  For destinations in hipMemMap'ed ranges (modes 2 / 3) that was observed
// NOT to hold on ROCm 7.2 (results changed with the fence allocator and came back under AMD_SERIALIZE_KERNEL=3, gpurun_out/r04c-e): in those modes every copy is
// completed before the call returns.
inline hipError_t memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
  hipError_t e = ::hipMemcpyAsync(dst, src, bytes, kind, stream);
  if (e != hipSuccess || mode() < 2) return e;
  return hipStreamSynchronize(stream);
}

}  // namespace devalloc

// The library is ONE translation unit in several files (livo2_api.hip + api_*.inc): an allocation is named by part * 100000 + line.  Every part sets LIVO2_SRC_ID.
#ifndef LIVO2_SRC_ID
#define LIVO2_SRC_ID 0
#endif
#define LIVO2_HERE (LIVO2_SRC_ID * 100000 + __LINE__)
#define DMALLOC(pp, bytes) devalloc::dev_malloc((void **)(pp), (bytes), LIVO2_HERE)
#define DMALLOC_AT(line, pp, bytes) devalloc::dev_malloc((void **)(pp), (bytes), (line))
#define DFREE(p) devalloc::dev_free((void *)(p))
