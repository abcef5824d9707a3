// gfx950 kernels for the visual (direct photometric, 8x8 patch) ESIKF update, forward-compositional form.
//   k_visual_residual : per (level, iteration) fused pass     reference src/vio.cpp:1540-1636 (+189-201)
//   k_visual_solve    : reduction + accept/revert + 19x19 solve reference src/vio.cpp:1636-1685
//   k_visual_finish   : cov -= G*cov, T_f_w                    reference src/vio.cpp:800-801, 1690-1697
//
//   k_visual_update_persistent : the WHOLE computeJacobianAndUpdateEKF as one launch (level / iteration loops on the device; per step one all-to-all exchange
//                                of tagged words between the resident blocks, no grid barrier)  reference src/vio.cpp:784-802 around 1520-1697
//
// Mapping: a wavefront owns FOUR patches (16 lanes each, 4 pixels per lane; pixel p = 8*x + y, x = patch row, y = patch column — the
// reference's loop order).  The (8+3)^2 strided u8 window is staged once in LDS as float, the 10x10 grid of bilinear samples B is
// built from it with the reference's float expression, and every lane then reads the 5 samples of each of its pixels (centre, +-u, +-v).
// All rows of one patch share the patch-constant 2x6 matrix M with  row = [g*M, cur]  (g = image gradient scaled by
// inv_expo/scale), so H^T H and H^T z are accumulated as 9 per-patch moment sums (registers, then an LDS transpose tile — no wave
// butterflies) and expanded with M once per patch:  sum J^T J = M^T (sum g g^T) M  etc. instead of a 64M x 7 dense product.
#pragma once
#include "esikf_solve.hpp"
#include <float.h>

#define VIS_BLOCK 256          // residual kernels: 4 waves = one per SIMD of a CU (round 3: 512-thread blocks put two waves of a latency chain on every SIMD of HALF the chip)
#define VIS_WAVES (VIS_BLOCK / LIVO2_WAVE)
#define VIS_NSUM 37          // 28 (sym 7x7) + 7 + err_sum + n_meas
#define VIS_PSTRIDE 40
#define VIS_NMOM 10          // 9 moment products + res^2 per pixel
#define VIS_TPITCH 65        // row pitch of the per-wave transpose tile in doubles (consecutive lanes -> consecutive 8-B words)

struct VisualKernelArgs {
  const uint8_t *img; int32_t width, height, stride;
  const double *pos; const float *warp; const int32_t *search_levels; const double *inv_expo;
  int32_t M, L, level, exposure_en;
  double fx, fy, cx, cy, d[5]; int32_t distortion; int32_t pad;
  double Rci[9], Pci[3], Jdp_dR[9];            // initializeVIO constants (vio.cpp:57-65), Jdphi_dR == Rci
  float *errors; double *z; double *H_sub;     // errors: always written; z / H_sub: optional debug outputs (device pointers or null)
};

__device__ __forceinline__ void vis_wave_sync() { wave_sync(); }
// value held by lane `src` (a compile-time constant) as a wave-uniform scalar: two v_readlane_b32, no LDS traffic
template <int SRC> __device__ __forceinline__ double lane_value(double v) {
  const int2 w = __builtin_bit_cast(int2, v);
  return __builtin_bit_cast(double, make_int2(__builtin_amdgcn_readlane(w.x, SRC), __builtin_amdgcn_readlane(w.y, SRC)));
}

// Jacobian row chain of the reference for an image-gradient row g (vio.cpp:1611-1617)
__device__ __forceinline__ void jac_row(double g0, double g1, const double Jpi[6], const double pf[3], const double *Rci, const double *JdpdR,
                                        const double *Rcw, double out[6]) {
  // a = Jimg * Jdpi (1x3)
  double a0 = g0 * Jpi[0] + g1 * Jpi[3], a1 = g0 * Jpi[1] + g1 * Jpi[4], a2 = g0 * Jpi[2] + g1 * Jpi[5];
  // Jdphi = a * p_hat, p_hat = skew(pf)
  double ph[9] = {0.0, -pf[2], pf[1], pf[2], 0.0, -pf[0], -pf[1], pf[0], 0.0};
  double f[3], na[3] = {-a0, -a1, -a2};
#pragma unroll
  for (int j = 0; j < 3; j++) f[j] = (a0 * ph[j] + a1 * ph[3 + j]) + a2 * ph[6 + j];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double t1 = (f[0] * Rci[j] + f[1] * Rci[3 + j]) + f[2] * Rci[6 + j];            // Jdphi * Jdphi_dR
    double t2 = (na[0] * JdpdR[j] + na[1] * JdpdR[3 + j]) + na[2] * JdpdR[6 + j];   // Jdp * Jdp_dR
    out[j] = t1 + t2;
    out[3 + j] = (na[0] * Rcw[j] + na[1] * Rcw[3 + j]) + na[2] * Rcw[6 + j];        // Jdp * Jdp_dt
  }
}

// pc = cam->world2cam(pf): projection to z = 1, then the camera model (livo2_device.hpp, cam_project: pinhole / radtan / equidistant)
__device__ __forceinline__ void vis_world2cam(const VisualKernelArgs &a, const double pf[3], double &pcx, double &pcy) {
  const double u0 = pf[0] / pf[2], u1 = pf[1] / pf[2];
  cam_project(a.distortion, a.d, a.fx, a.fy, a.cx, a.cy, u0, u1, pcx, pcy);
}

#ifdef LIVO2_PHASE_PROF
#define VIS_PROF_WAVES (1 << 14)
__device__ unsigned long long g_vis_prof[VIS_PROF_WAVES][8];
// drain every outstanding memory op, then stamp (slot 0 and 7: chip-wide 100 MHz clock, the others: this CU's cycle counter)
#define VPHASE(k)                                                                                                            \
  do {                                                                                                                       \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);                     \
    const unsigned gw = blockIdx.x * VIS_WAVES + (threadIdx.x >> 6);                                                         \
    if ((threadIdx.x & 63) == 0 && gw < VIS_PROF_WAVES && (threadIdx.x >> 6) < VIS_WAVES)                                    \
      g_vis_prof[gw][k] = ((k) == 0 || (k) == 7) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();          \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
  } while (0)
// solve kernel: wave w stamps row VIS_PROF_WAVES - 1 - w (block 0 only)
#define VSPHASE(k)                                                                                                           \
  do {                                                                                                                       \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0)                                                                          \
      g_vis_prof[VIS_PROF_WAVES - 1 - (threadIdx.x >> 6)][k] = ((k) == 0 || (k) == 7) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
  } while (0)
#else
#define VPHASE(k) do { } while (0)
#define VSPHASE(k) do { } while (0)
#endif

// Mapping: a wave owns VIS_PPW = 4 patches, 16 lanes each, every lane 4 pixels (pixel p = j + 16k of its patch, j = lane & 15: the reference's loop index
// p = 8x + y).  Why not a wave per patch (round 1): everything that is constant over a patch — Rcw, pf, the projection with its three f64 divisions, Jpi, the 2x6 M —
// was evaluated redundantly by 64 lanes, ~500 of the ~890 VALU instructions a wave issued; with 4 waves per SIMD the kernel was VALU-bound (PMC: 3.6 k VALU-busy
// cycles per wave, kernel 11 us for 4 000 patches although a wave's memory chain is ~3 us).  Now one instruction stream covers four patches.
#define VIS_PPW 4
#define VIS_LPP (LIVO2_WAVE / VIS_PPW)
#define VIS_PPB (VIS_WAVES * VIS_PPW)        // patches per block = rows of `partials` saved: 4 000 patches -> 125 rows

// Per-wave LDS: the staging buffers of the windows (Wf) and of the bilinear grids (Bf) alias the transpose tile T that the reduction uses afterwards.
struct __attribute__((aligned(16))) VisWaveLds {
  union {
    struct { float Wf[VIS_PPW][11 * 11 + 3]; float Bf[VIS_PPW][10 * 10 + 4]; } st;
    double T[9 * VIS_TPITCH];
  };
  double Rr[VIS_PPW][64];      // res^2 per pixel, pixel order
  double S[VIS_PPW][9];        // the 9 moment sums of each patch
  double Mx[VIS_PPW][12];      // patch-constant 2x6 M
  float nm[VIS_PPW];
};

// Stores / loads of data that another workgroup of the SAME launch consumes (formerly the fused step kernel; kept for experiments): relaxed atomics at agent scope = write-through stores and
// cache-bypassing loads (sc1), so the hand-off needs no L2 write-back / invalidate fence (HIP guide section 6, guideline 16, second recipe).
template <bool XB, typename T> __device__ __forceinline__ void xb_store(T *p, T v) {
  if (XB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool XB, typename T> __device__ __forceinline__ T xb_load(const T *p) {
  return XB ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// value q of a patch's 37-vector from its moment sums S and its M:  q<28: upper-tri (r<=c) of the 7x7 ; 28..34: Htz ; 35: err ; 36: n_meas.
// (r, c) of lane q are found once, before the first memory wait (vis_rc); the four M entries a lane needs are then four independent LDS reads per patch.
struct VisRC { int r, c; };
__device__ __forceinline__ VisRC vis_rc(int q) {
  VisRC o = {0, 0};
  if (q < 28) {
    int r = 0, t = q;
#pragma unroll
    for (int k = 0; k < 6; k++) if (t >= 7 - r) { t -= 7 - r; r++; }
    o.r = r; o.c = r + t;
  } else if (q < 35) { o.r = q - 28; o.c = 7; }
  return o;
}
__device__ __forceinline__ double vis_expand(int q, VisRC rc, const double *S, const double *Mx, float nm) {
  const int r = rc.r < 6 ? rc.r : 0, c = rc.c < 6 ? rc.c : 0;                 // clamped: the loads below are unconditional
  const double m0r = Mx[r], m1r = Mx[6 + r], m0c = Mx[c], m1c = Mx[6 + c];
  const double Sg00 = S[0], Sg01 = S[1], Sg11 = S[2], Sgc0 = S[3], Sgc1 = S[4], Scc = S[5], Sgr0 = S[6], Sgr1 = S[7], Scr = S[8];
  double v;
  if (q < 28) {
    if (rc.c < 6) v = (m0r * m0c) * Sg00 + (m0r * m1c + m1r * m0c) * Sg01 + (m1r * m1c) * Sg11;
    else if (rc.r < 6) v = m0r * Sgc0 + m1r * Sgc1;
    else v = Scc;
  } else if (q < 35) v = (rc.r < 6) ? (m0r * Sgr0 + m1r * Sgr1) : Scr;
  else if (q == 36) v = (double)nm;            // (q == 35, the error sum, is formed by the solve kernel from errors[]: 0 here)
  else v = 0.0;
  return v;
}

// ---- inverse-compositional form (vio/inverse_composition_en; reference src/vio.cpp:1327-1518): what precomputeReferencePatches leaves per level (visual_inverse_kernels.hpp)
struct VisualRefArgs {
  const uint8_t *ref_imgs;           // [n_ref][height][stride]
  const int32_t *ref_idx;            // [M] reference image of each point (ref_patch->img_)
  const double *ref_px;              // [M][2] ref_patch->px_
  const double *ref_f;               // [M][3] ref_patch->f_
  const double *ref_R;               // [M][9] ref_patch->T_f_w_.rotation_matrix()
  const double *ref_pos;             // [M][3] ref_patch->pos()
  double *gref;                      // [M][64][2]
  double *mref;                      // [M][16]: M_ref (12), sum g0g0, g0g1, g1g1, valid
  int32_t n_ref, pad;
};

// bilinear weights and integer anchor of the reference (vio.cpp:1359-1368 / 1449-1458), float/double mix reproduced
struct Anchor { int ui, vi; float w_tl, w_tr, w_bl, w_br; };
__device__ __forceinline__ Anchor make_anchor(double pcx, double pcy, int scale) {
  Anchor A;
  const float u_ref = (float)pcx, v_ref = (float)pcy;
  A.ui = (int)(floorf((float)(pcx / scale)) * (float)scale);
  A.vi = (int)(floorf((float)(pcy / scale)) * (float)scale);
  const float su = (u_ref - (float)A.ui) / (float)scale, sv = (v_ref - (float)A.vi) / (float)scale;
  A.w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv));
  A.w_tr = (float)((double)su * (1.0 - (double)sv));
  A.w_bl = (float)((1.0 - (double)su) * (double)sv);
  A.w_br = su * sv;
  return A;
}

// precomputeReferencePatches for ONE patch by one wave (lane = pixel 8x + y); Wf / Bf: 11 x 11 and 10 x 10 floats of the calling wave's LDS
__device__ __forceinline__ void ref_precompute_patch(const VisualKernelArgs &a, const VisualRefArgs &r, const int level, const int patch, const int lane, float *Wf, float *Bf) {
  const int scale = 1 << level;                              // no search level here (vio.cpp:1341)
  const double p[3] = {a.pos[(size_t)patch * 3], a.pos[(size_t)patch * 3 + 1], a.pos[(size_t)patch * 3 + 2]};
  const double *rp = r.ref_pos + (size_t)patch * 3, *rf = r.ref_f + (size_t)patch * 3, *RR = r.ref_R + (size_t)patch * 9;
  const double dx = p[0] - rp[0], dy = p[1] - rp[1], dz = p[2] - rp[2];
  const double depth = sqrt((dx * dx + dy * dy) + dz * dz);   // (pt->pos_ - ref_patch->pos()).norm()
  const double pf[3] = {rf[0] * depth, rf[1] * depth, rf[2] * depth};
  double Jpi[6];
  {
    const double z_inv = 1. / pf[2], z_inv_2 = z_inv * z_inv;
    Jpi[0] = a.fx * z_inv; Jpi[1] = 0.0; Jpi[2] = -a.fx * pf[0] * z_inv_2;
    Jpi[3] = 0.0; Jpi[4] = a.fy * z_inv; Jpi[5] = -a.fy * pf[1] * z_inv_2;
  }
  const Anchor A = make_anchor(r.ref_px[(size_t)patch * 2], r.ref_px[(size_t)patch * 2 + 1], scale);
  const int ridx = r.ref_idx[patch];
  const bool inside = (ridx >= 0) && (ridx < r.n_ref) && (A.ui - 5 * scale >= 0) && (A.ui + 5 * scale < a.width) && (A.vi - 5 * scale >= 0) && (A.vi + 5 * scale < a.height);
  double g0 = 0.0, g1 = 0.0;
  if (inside) {
    const uint8_t *img = r.ref_imgs + (size_t)ridx * a.height * a.stride;
    for (int e = lane; e < 121; e += LIVO2_WAVE) { const int wr = e / 11, wc = e - wr * 11; Wf[e] = (float)img[(size_t)(A.vi + (wr - 5) * scale) * a.stride + (A.ui + (wc - 5) * scale)]; }
    vis_wave_sync();
    for (int e = lane; e < 100; e += LIVO2_WAVE) { const int br = e / 10, bc = e - br * 10; const float *w = &Wf[br * 11 + bc]; Bf[e] = ((A.w_tl * w[0] + A.w_tr * w[1]) + A.w_bl * w[11]) + A.w_br * w[12]; }
    vis_wave_sync();
    const int x = lane >> 3, y = lane & 7;
    const float *b = &Bf[(x + 1) * 10 + (y + 1)];
    const float du = 0.5f * (b[1] - b[-1]), dv = 0.5f * (b[10] - b[-10]);
    const double isc = 1.0 / scale;                           // Jimg = Jimg * (1.0 / scale)  (vio.cpp:1385)
    g0 = (double)du * isc; g1 = (double)dv * isc;
  }
  r.gref[((size_t)patch * 64 + lane) * 2] = g0;
  r.gref[((size_t)patch * 64 + lane) * 2 + 1] = g1;
  if (lane < 2) {                                             // M_ref row `lane`: JdR = ((g Jpi) R_ref_w) [p]x ; Jdt = ((-g) Jpi) R_ref_w   (vio.cpp:1387-1388)
    const double e0 = lane == 0 ? 1.0 : 0.0, e1 = lane == 1 ? 1.0 : 0.0;
    const double a3[3] = {e0 * Jpi[0] + e1 * Jpi[3], e0 * Jpi[1] + e1 * Jpi[4], e0 * Jpi[2] + e1 * Jpi[5]};
    double aR[3], naR[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { aR[j] = (a3[0] * RR[j] + a3[1] * RR[3 + j]) + a3[2] * RR[6 + j]; naR[j] = ((-a3[0]) * RR[j] + (-a3[1]) * RR[3 + j]) + (-a3[2]) * RR[6 + j]; }
    const double ph[9] = {0.0, -p[2], p[1], p[2], 0.0, -p[0], -p[1], p[0], 0.0};
    double *m = r.mref + (size_t)patch * 16 + lane * 6;
#pragma unroll
    for (int j = 0; j < 3; j++) { m[j] = (aR[0] * ph[j] + aR[1] * ph[3 + j]) + aR[2] * ph[6 + j]; m[3 + j] = naR[j]; }
  }
  if (lane == 2) { double *m = r.mref + (size_t)patch * 16; m[12] = 0.0; m[13] = 0.0; m[14] = 0.0; m[15] = inside ? 1.0 : 0.0; }
}

// Four patches per wave (the body shared by the single-frame and the batched kernel).  patch0 = first patch of this wave.  Returns lane q's value q (q < 37) of the
// SUM of the wave's patch vectors (slot order).
// The iterate enters as three pointers (rot_end, pos_end, inv_expo_time): HBM (ctl->cur) in the per-step kernels, LDS in the persistent kernel.
// errors_out: float[M] (plain stores, or write-through stores with XB: the resident grid of k_visual_update_persistent reads them from other blocks).
// ROLE 0: one wave does everything.  ROLE 1 / 2 (k_visual_update_persistent with one row per block, round 5): TWO waves of the block share the four patches — the
// critical path of a patch is projection -> window -> B grid -> pixel loop -> moment sums -> expansion; the 2x6 M (needed only by the expansion) and the 64-step float
// chain of patch_error (needed only for errors[]) hang off it.  ROLE 1 (main) walks the critical path, ROLE 2 (partner, same SIMD, same VisWaveLds) repeats the
// projection, forms M while the main wave waits for its window, and runs the chain from the main wave's Rr while that one reduces and expands.  One block barrier in
// the middle (Rr and Mx complete), so ALL waves of the block must call the body the same number of times.  Same expressions in the same order: same bits.
// INV (round 6): the same body evaluates updateStateInverse (vio.cpp:1418-1477) — the residual is the current image's bilinear sample minus the reference patch (one
// float expression, no exposure factors), the per-pixel gradient is the REFERENCE image's (rp->gref, precomputed per level), the patch-constant 2x6 matrix is
// N = M_ref T(state) (vio.cpp:1470-1474) and the exposure column is zero; window 9 x 9 at scale 2^level (no search level).  Moment sums, expansion, error chain,
// roles and barriers are the forward form's, so the resident grid and the per-step kernel share one code for either form.
template <bool DEBUG_ROWS, bool XB = false, int ROLE = 0, bool INV = false>
__device__ __forceinline__ double visual_wave_body(const VisualKernelArgs &a, const int level, const double *Rwi, const double *Pwi, const double *tau_p, float *errors_out,
                                                   VisWaveLds &L, int patch0, int lane, const VisualRefArgs *rp = nullptr) {
  const int slot = lane / VIS_LPP, j = lane % VIS_LPP;
  const VisRC rc = vis_rc(lane);
  const int patch_raw = patch0 + slot;
  const bool valid = patch_raw < a.M;
  const int patch = valid ? patch_raw : a.M - 1;            // clamped: invalid slots compute on the last patch and contribute nothing
  // loads that do not depend on the state go first, so they travel together with the scalar loads of the state
  const double p0 = a.pos[(size_t)patch * 3], p1 = a.pos[(size_t)patch * 3 + 1], p2 = a.pos[(size_t)patch * 3 + 2];
  const int search_level = INV ? 0 : a.search_levels[patch];
  double inv_ref_expo = 0.0, tau = 0.0;
  float Pref[4] = {0.f, 0.f, 0.f, 0.f};
  double gr0[4] = {0.0, 0.0, 0.0, 0.0}, gr1[4] = {0.0, 0.0, 0.0, 0.0}, mr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool ref_ok = true;
  if (INV) {
    const double *m = rp->mref + (size_t)patch * 16;
    ref_ok = m[15] != 0.0;
    if (ROLE != 1) {
#pragma unroll
      for (int k = 0; k < 12; k++) mr[k] = m[k];
    }
  }
  if (ROLE != 2) {
    if (!INV) inv_ref_expo = a.inv_expo[patch];
#pragma unroll
    for (int k = 0; k < 4; k++) Pref[k] = a.warp[((size_t)patch * a.L + level) * 64 + j + VIS_LPP * k];
    if (!INV) tau = *tau_p;
    if (INV) {
#pragma unroll
      for (int k = 0; k < 4; k++) { const double2 g = *reinterpret_cast<const double2 *>(rp->gref + ((size_t)patch * 64 + j + VIS_LPP * k) * 2); gr0[k] = g.x; gr1[k] = g.y; }
    }
  }
  VPHASE(1);
  double Rcw[9], Pcw[3];
  mat3_mul_Bt(a.Rci, Rwi, Rcw);                            // Rcw = Rci * Rwi^T
#pragma unroll
  for (int i = 0; i < 3; i++) Pcw[i] = a.Pci[i] - ((Rcw[i * 3] * Pwi[0] + Rcw[i * 3 + 1] * Pwi[1]) + Rcw[i * 3 + 2] * Pwi[2]);
  double pf[3];
#pragma unroll
  for (int i = 0; i < 3; i++) pf[i] = ((Rcw[i * 3] * p0 + Rcw[i * 3 + 1] * p1) + Rcw[i * 3 + 2] * p2) + Pcw[i];
  double pcx, pcy;
  vis_world2cam(a, pf, pcx, pcy);
  const int scale = 1 << (level + search_level);
  const float inv_scale = 1.0f / (float)scale;
  const float u_ref = (float)pcx, v_ref = (float)pcy;
  const int u_ref_i = (int)(floorf((float)(pcx / scale)) * (float)scale);
  const int v_ref_i = (int)(floorf((float)(pcy / scale)) * (float)scale);
  // the reference reads this window unchecked (vio.cpp:1595-1609); a window leaving the image is skipped here
  constexpr int HW = INV ? 4 : 5, WN = 2 * HW + 1;          // half-width and side of the strided window (INV: vio.cpp:1463-1468 reads rows / columns -4..+4 unchecked)
  const bool inside = (u_ref_i - HW * scale >= 0) && (u_ref_i + HW * scale < a.width) && (v_ref_i - HW * scale >= 0) && (v_ref_i + HW * scale < a.height);
  const bool ok = valid && inside && ref_ok;
  // the 11x11 strided window of the lane's patch: 8 byte loads per lane, all issued before the projection Jacobian below
  uint8_t px[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ROLE != 2) {
    const size_t base = ok ? (size_t)(v_ref_i - HW * scale) * a.stride + (size_t)(u_ref_i - HW * scale) : 0;
    const int sc = ok ? scale : 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = j + VIS_LPP * k < WN * WN ? j + VIS_LPP * k : WN * WN - 1;
      const int wr = e / WN, wc = e - wr * WN;
      px[k] = a.img[base + (size_t)(wr * sc) * a.stride + (size_t)(wc * sc)];
    }
  }
  VPHASE(2);
  const float subpix_u = (u_ref - (float)u_ref_i) / (float)scale;
  const float subpix_v = (v_ref - (float)v_ref_i) / (float)scale;
  const float w_tl = (float)((1.0 - (double)subpix_u) * (1.0 - (double)subpix_v));
  const float w_tr = (float)((double)subpix_u * (1.0 - (double)subpix_v));
  const float w_bl = (float)((1.0 - (double)subpix_u) * (double)subpix_v);
  const float w_br = subpix_u * subpix_v;
  // computeProjectionJacobian (vio.cpp:189-201) and the patch-constant 2x6 M
  double Jpi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (ROLE != 1 || DEBUG_ROWS) {
    const double z_inv = 1. / pf[2], z_inv_2 = z_inv * z_inv;
    Jpi[0] = a.fx * z_inv; Jpi[1] = 0.0; Jpi[2] = -a.fx * pf[0] * z_inv_2;
    Jpi[3] = 0.0; Jpi[4] = a.fy * z_inv; Jpi[5] = -a.fy * pf[1] * z_inv_2;
  }
  double M0[6] = {0, 0, 0, 0, 0, 0}, M1[6] = {0, 0, 0, 0, 0, 0};
  if (ROLE != 1) {
    if (!INV) {
      jac_row(1.0, 0.0, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, M0);
      jac_row(0.0, 1.0, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, M1);
    } else {
      // N = M_ref * T :  JdR = J_dR Rwi + (J_dt [Pwi]x) Rwi ; Jdt = J_dt Rwi    (vio.cpp:1472-1473)
      const double Ph[9] = {0.0, -Pwi[2], Pwi[1], Pwi[2], 0.0, -Pwi[0], -Pwi[1], Pwi[0], 0.0};
#pragma unroll
      for (int row = 0; row < 2; row++) {
        const double *m = mr + row * 6;
        double tP[3];
#pragma unroll
        for (int q = 0; q < 3; q++) tP[q] = (m[3] * Ph[q] + m[4] * Ph[3 + q]) + m[5] * Ph[6 + q];
        double *N = row == 0 ? M0 : M1;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const double t1 = (m[0] * Rwi[q] + m[1] * Rwi[3 + q]) + m[2] * Rwi[6 + q];
          const double t2 = (tP[0] * Rwi[q] + tP[1] * Rwi[3 + q]) + tP[2] * Rwi[6 + q];
          N[q] = t1 + t2;
          N[3 + q] = (m[3] * Rwi[q] + m[4] * Rwi[3 + q]) + m[5] * Rwi[6 + q];
        }
      }
    }
    if (j == 0) {
#pragma unroll
      for (int k = 0; k < 6; k++) { L.Mx[slot][k] = M0[k]; L.Mx[slot][6 + k] = M1[k]; }
    }
  }
  double acc[9];
#pragma unroll
  for (int v = 0; v < 9; v++) acc[v] = 0.0;
  if (ROLE != 2) {
#pragma unroll
  for (int k = 0; k < 8; k++) { const int e = j + VIS_LPP * k; if (e < WN * WN) L.st.Wf[slot][e] = (float)px[k]; }
  wave_sync();
  if (INV) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = j + VIS_LPP * k, x = p >> 3, y = p & 7;
      const float *w = &L.st.Wf[slot][x * 9 + y];
      const double res = (double)((((w_tl * w[0] + w_tr * w[1]) + w_bl * w[9]) + w_br * w[10]) - Pref[k]);   // all-float expression (vio.cpp:1466-1467)
      const double g0 = gr0[k], g1 = gr1[k];
      if (DEBUG_ROWS && ok) {
        if (a.z) a.z[(size_t)patch * 64 + p] = res;
        if (a.H_sub) {
          double *h = a.H_sub + ((size_t)patch * 64 + p) * 7;
#pragma unroll
          for (int c = 0; c < 6; c++) h[c] = g0 * M0[c] + g1 * M1[c];
          h[6] = 0.0;
        }
      }
      acc[0] += g0 * g0; acc[1] += g0 * g1; acc[2] += g1 * g1;
      acc[6] += g0 * res; acc[7] += g1 * res;
      L.Rr[slot][p] = res * res;
    }
    wave_sync();
  } else {
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const int e = j + VIS_LPP * k;
    if (e < 100) {
      const int br = e / 10, bc = e - br * 10;
      const float *w = &L.st.Wf[slot][br * 11 + bc];
      L.st.Bf[slot][e] = ((w_tl * w[0] + w_tr * w[1]) + w_bl * w[11]) + w_br * w[12];      // the reference's float expression (vio.cpp:1600-1620)
    }
  }
  wave_sync();
  VPHASE(3);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p = j + VIS_LPP * k, x = p >> 3, y = p & 7;
    const float *b = &L.st.Bf[slot][(x + 1) * 10 + (y + 1)];
    const float du = 0.5f * (b[1] - b[-1]);
    const float dv = 0.5f * (b[10] - b[-10]);
    const double cur = (double)b[0];
    const double g0 = ((double)du * tau) * (double)inv_scale;
    const double g1 = ((double)dv * tau) * (double)inv_scale;
    const double res = tau * cur - inv_ref_expo * (double)Pref[k];
    const double cexp = a.exposure_en ? cur : 0.0;
    if (DEBUG_ROWS && ok) {
      double row[6];
      jac_row(g0, g1, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, row);
      if (a.z) a.z[(size_t)patch * 64 + p] = res;
      if (a.H_sub) {
        double *h = a.H_sub + ((size_t)patch * 64 + p) * 7;
#pragma unroll
        for (int c = 0; c < 6; c++) h[c] = row[c];
        h[6] = cexp;
      }
    }
    acc[0] += g0 * g0; acc[1] += g0 * g1; acc[2] += g1 * g1; acc[3] += g0 * cexp; acc[4] += g1 * cexp; acc[5] += cexp * cexp;
    acc[6] += g0 * res; acc[7] += g1 * res; acc[8] += cexp * res;
    L.Rr[slot][p] = res * res;
  }
  wave_sync();                                              // the staging buffers alias the tile written next; Rr is complete
  }
  }
  VPHASE(4);
  if (ROLE != 0) __syncthreads();                           // main: Rr is written; partner: Mx is written
  // patch error: the reference's accumulator is a FLOAT updated in pixel order, `patch_error += res * res` = float(double(patch_error) + res*res)
  // (vio.cpp:1563,1624); the 16 lanes of a slot run that 64-step chain from broadcast LDS reads, so errors[] equals the CPU loop bit for bit.  The chain is
  // 192 dependent operations (~1.2 us) that nothing else in the wave needs: it is cut into three pieces placed inside the three synchronisation regions
  // below, so that the scheduler fills its stalls with the reduction's LDS traffic and the expansion's arithmetic.
  float pe = 0.0f;
  const double *rr = L.Rr[slot];
#ifndef VIS_EXP_NOCHAIN
#define VIS_CHAIN(lo, hi) _Pragma("unroll") for (int i = (lo); i < (hi); i++) pe = (float)((double)pe + rr[i]);
#else
#define VIS_CHAIN(lo, hi)
#endif
  if (ROLE == 2) {
    VIS_CHAIN(0, 64)
    if (!ok) pe = 0.0f;
    if (XB && pe != pe) pe = __uint_as_float(0x7fc00000u);      // the exchange's "not published yet" word is the all-ones NaN: a NaN that came in with that payload (garbage patches) must not look like it (advisor, round 5)
    if (j == 0 && valid) xb_store<XB>(&errors_out[patch], pe);
    return 0.0;
  }
#pragma unroll
  for (int v = 0; v < 9; v++) L.T[v * VIS_TPITCH + lane] = ok ? acc[v] : 0.0;
  if (ROLE == 0) { VIS_CHAIN(0, 16) }
  wave_sync();
  {
    // lane 4v+q adds columns [16q, 16q+16) of row v = the 16 lanes of patch slot q (16 independent LDS reads)
    const int v = lane >> 2, q = lane & 3;
    if (v < 9) {
      const double *row = L.T + v * VIS_TPITCH + q * VIS_LPP;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int i = 0; i < VIS_LPP; i += 4) { s0 += row[i]; s1 += row[i + 1]; s2 += row[i + 2]; s3 += row[i + 3]; }
      L.S[q][v] = (s0 + s1) + (s2 + s3);
    }
  }
  if (ROLE == 0) { VIS_CHAIN(16, 40) }
  if (j == 0) L.nm[slot] = ok ? 64.0f : 0.0f;
  wave_sync();
  VPHASE(5);
  double out_val = 0.0;
  if (lane < VIS_NSUM) {
#pragma unroll
    for (int sl = 0; sl < VIS_PPW; sl++) out_val += vis_expand(lane, rc, L.S[sl], L.Mx[sl], L.nm[sl]);
  }
  if (ROLE == 0) {
    VIS_CHAIN(40, 64)
    if (!ok) pe = 0.0f;
    if (XB && pe != pe) pe = __uint_as_float(0x7fc00000u);
    if (j == 0 && valid) xb_store<XB>(&errors_out[patch], pe);
  }
#undef VIS_CHAIN
  return out_val;
}
template <bool DEBUG_ROWS, bool XB = false>
__device__ __forceinline__ double visual_wave_body(const VisualKernelArgs &a, const DevCtl *__restrict__ ctl, VisWaveLds &L, int patch0, int lane) {
  return visual_wave_body<DEBUG_ROWS, XB>(a, a.level, ctl->cur.rot, ctl->cur.pos, &ctl->cur.inv_expo, a.errors, L, patch0, lane);
}

// the block's 8 wave vectors -> one partial row (fixed order: deterministic)
template <bool XB = false>
__device__ __forceinline__ void vis_block_store(double (*red)[VIS_PSTRIDE], double out_val, double *__restrict__ partial_row) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (lane < VIS_PSTRIDE) red[wave][lane] = out_val;
  __syncthreads();
  if (tid < VIS_PSTRIDE) {
    double v = red[0][tid];
#pragma unroll
    for (int w = 1; w < VIS_WAVES; w++) v = v + red[w][tid];
    xb_store<XB>(&partial_row[tid], v);
  }
}

template <bool DEBUG_ROWS>
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_residual(VisualKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                               int check_stop) {
  VPHASE(0);
  if (check_stop && ctl->hdr.stop) return;
  __shared__ VisWaveLds lds[VIS_WAVES];
  __shared__ double red[VIS_WAVES][VIS_PSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch0 = (blockIdx.x * VIS_WAVES + wave) * VIS_PPW;   // wave-uniform
  double out_val = 0.0;                                      // lane q < VIS_NSUM holds value q of this wave's patches
  if (patch0 < a.M) out_val = visual_wave_body<DEBUG_ROWS>(a, ctl, lds[wave], patch0, lane);
  VPHASE(6);
  vis_block_store(red, out_val, partials + (size_t)blockIdx.x * VIS_PSTRIDE);
  VPHASE(7);
}

// Batched launch: several independent computeJacobianAndUpdateEKF problems (own image, own sub-map, own state) in ONE grid per (level, iteration) —
// the visual counterpart of k_lidar_residual_batch.  A frame whose level has ended (hdr.stop) drops out at its blocks' first instruction.
struct VisualBatchEntry { VisualKernelArgs a; DevCtl *ctl; double *partials; int32_t block_begin, nblocks; VisualRefArgs r; };      // r: the inverse-compositional form (livo2_visual_batch_set_references)
template <bool INV = false>
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_residual_batch(const VisualBatchEntry *__restrict__ entries, const int32_t *__restrict__ block_frame, int level, int check_stop) {
  const int f = block_frame[blockIdx.x];
  const VisualBatchEntry &e = entries[f];
  if (e.a.M == 0 || (check_stop && e.ctl->hdr.stop)) return;      // total_points == 0: computeJacobianAndUpdateEKF returns at once (vio.cpp:786)
  __shared__ VisWaveLds lds[VIS_WAVES];
  __shared__ double red[VIS_WAVES][VIS_PSTRIDE];
  VisualKernelArgs a = e.a;
  a.level = level;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pblock = (int)blockIdx.x - e.block_begin;
  const int patch0 = (pblock * VIS_WAVES + wave) * VIS_PPW;
  double out_val = 0.0;
  if (patch0 < a.M) out_val = INV ? visual_wave_body<false, false, 0, true>(a, a.level, e.ctl->cur.rot, e.ctl->cur.pos, &e.ctl->cur.inv_expo, a.errors, lds[wave], patch0, lane, &e.r)
                                  : visual_wave_body<false>(a, e.ctl, lds[wave], patch0, lane);
  vis_block_store(red, out_val, e.partials + (size_t)pblock * VIS_PSTRIDE);
}
// precomputeReferencePatches of every frame of a batch at one level (the grid of k_visual_residual_batch: a block = VIS_PPB patches of one frame, a wave takes four in turn)
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_ref_precompute_batch(const VisualBatchEntry *__restrict__ entries, const int32_t *__restrict__ block_frame, int level, int check_stop) {
  const int f = block_frame[blockIdx.x];
  const VisualBatchEntry &e = entries[f];
  if (e.a.M == 0 || (check_stop && e.ctl->hdr.stop)) return;
  __shared__ float Wf[VIS_WAVES][11 * 11 + 3];
  __shared__ float Bf[VIS_WAVES][10 * 10 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pblock = (int)blockIdx.x - e.block_begin;
  for (int q = 0; q < VIS_PPW; q++) {
    const int patch = (pblock * VIS_WAVES + wave) * VIS_PPW + q;
    if (patch < e.a.M) ref_precompute_patch(e.a, e.r, level, patch, lane, Wf[wave], Bf[wave]);
    vis_wave_sync();
  }
}

// ---- reduction + accept / revert + solve --------------------------------------------------------------------------------------------
// mode 0: bare evaluation -> ctl->sums_v ; mode 1: full ESIKF step (accept / revert / solve) ; mode 2: benchmark (always accept, never stop)
// Roles inside the 512-thread block (every wave stays alive to the last barrier):
//   all waves : issue their partial-row loads (12 slices x 40 values, up to 48 rows per thread in flight at once = 576 rows per pass) and stage errors[] in LDS
//   wave 0    : prefetch of P / states (before the rows arrive), then the 19-dim algebra, SPECULATIVELY (nothing is written to HBM yet)
//   wave 1    : Log(cur^T prop) while the rows are in flight
//   wave 2    : the frame error with the reference's float accumulation order — lane t adds the per-patch errors of OpenMP thread t's static block in
//               index order (vio.cpp:1554, 1634), the T partial sums are joined in thread order (error_threads = MP_PROC_NUM of the reference build)
//   then      : one barrier; wave 0 compares `error <= last_error` (vio.cpp:1648) and commits either the update or the revert.
#include "float_chain.hpp"
#define VIS_SOLVE_THREADS 480         // 12 slices x 40 values
#define VIS_ERR_STAGE 8192            // per-patch errors staged per pass (floats)
struct VisualSolveArgs { const float *errors; int32_t M, error_threads; };      // error_threads: MP_PROC_NUM of the reference build; NEGATIVE: that many threads, and the frame error on one lane per thread only (option "visual_error_waves" = 0)

// acc + e[lo] + e[lo+1] + ... in this order, one float rounding per add (the serial CPU loop).  The adds are a dependent chain; the LDS reads are not, but
// the compiler sinks them next to their first use (measured: 14.5 cycles per element, one exposed LDS round trip per 16 adds).  The reads are therefore issued
// through asm statements in program order — 16 values ahead of the adds — and joined with an s_waitcnt that names the registers it releases.
typedef float v4f __attribute__((ext_vector_type(4)));
struct F16 { v4f a, b, c, d; };
__device__ __forceinline__ void lds_issue16(F16 &v, const float *p, float &acc) {
  const uint32_t addr = (uint32_t)(uintptr_t)p;             // LDS addresses are 32-bit
  // `acc` is tied to the statement so that the adds that follow in program order stay behind the issue (plain ALU code may otherwise cross an asm volatile)
  asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48"
               : "=&v"(v.a), "=&v"(v.b), "=&v"(v.c), "=&v"(v.d), "+v"(acc) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_land16(F16 &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.a), "+v"(v.b), "+v"(v.c), "+v"(v.d)); }
// the oldest of three batches in flight has landed (LDS returns in order: 8 newer ds_read_b128 may still be outstanding)
__device__ __forceinline__ void lds_land16_of3(F16 &v) { asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v.a), "+v"(v.b), "+v"(v.c), "+v"(v.d)); }
__device__ __forceinline__ float add16(float acc, const F16 &v) {
  acc += v.a.x; acc += v.a.y; acc += v.a.z; acc += v.a.w; acc += v.b.x; acc += v.b.y; acc += v.b.z; acc += v.b.w;
  acc += v.c.x; acc += v.c.y; acc += v.c.z; acc += v.c.w; acc += v.d.x; acc += v.d.y; acc += v.d.z; acc += v.d.w;
  return acc;
}
// Three batches of 16 values are kept in flight: with one batch ahead the chain stalled at every hand-over (measured 12.5 cycles per element: the LDS round trip of
// four ds_read_b128 is longer than the 16 dependent adds it was supposed to hide behind).
__device__ __forceinline__ float float_chain(const float *e, int lo, int hi, float acc) {
  int i = lo;
  for (; i < hi && (i & 3); i++) acc += e[i];
  if (i + 48 <= hi) {
    F16 A, B, C;
    lds_issue16(A, e + i, acc); lds_issue16(B, e + i + 16, acc); lds_issue16(C, e + i + 32, acc);
    for (; i + 96 <= hi; i += 48) {
      lds_land16_of3(A); acc = add16(acc, A); lds_issue16(A, e + i + 48, acc);
      lds_land16_of3(B); acc = add16(acc, B); lds_issue16(B, e + i + 64, acc);
      lds_land16_of3(C); acc = add16(acc, C); lds_issue16(C, e + i + 80, acc);
    }
    lds_land16(C);                                           // everything has landed
    acc = add16(acc, A); acc = add16(acc, B); acc = add16(acc, C);
    i += 48;
  }
  if (i + 16 <= hi) {
    F16 A;
    for (; i + 16 <= hi; i += 16) { lds_issue16(A, e + i, acc); lds_land16(A); acc = add16(acc, A); }
  }
  for (; i < hi; i++) acc += e[i];
  return acc;
}

// ---- the frame error on whole waves (float_chain.hpp): a chain per group of FC_W lanes instead of a chain per lane, same bits ----------------------------------
// Used when every OpenMP thread's block of patches is long enough to pay for the rounds (C4: 1 000 per thread, 4.1-5.4 us as one dependent chain of adds — longer than the
// 19-dim solve beside it, 3.2-4.2 us — against ~2 us here) and the chains fit the waves set aside for them; otherwise lane t of one wave adds thread t's block serially.
#define VS_FC_WAVES 5            // chain waves of k_visual_solve and of the resident grid's solve phase
#define FC_W 32
#define FC_LMAX 35
#define FC_CPW (LIVO2_WAVE / FC_W)
// chain waves of an 8-wave block whose wave 0 runs the solve: SIMDs 2, 3 first, never the solve's own SIMD (waves 0 and 4)
__device__ __forceinline__ int fc_wave_index(int wave, int nwaves) {
  const int idx = wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : wave == 7 ? 3 : wave == 5 ? 4 : -1;
  return idx < nwaves ? idx : -1;
}
__device__ __forceinline__ int fc_threads(int error_threads) { const int t = error_threads < 0 ? -error_threads : error_threads; return t < 1 ? 1 : (t > LIVO2_WAVE ? LIVO2_WAVE : t); }
// a single OpenMP thread (updateStateInverse has no parallel loop; MP_PROC_NUM = 1): its one chain gets all 64 lanes of the first chain wave and twice the elements per pass
template <typename E> __device__ __forceinline__ float fc_one_chain(const E *errs, int n, int lane) { return float_chain_wave<64, FC_LMAX>(errs, 0, n, 0.0f, lane); }
__device__ __forceinline__ bool fc_use_waves(int M, int T, int nwaves) { return T <= nwaves * FC_CPW && M / T >= FC_MIN_N; }
// OpenMP static partition of M patches over T threads (libgomp: the first M % T threads get one patch more)
__device__ __forceinline__ void fc_partition(int M, int T, int c, int &b, int &e) {
  const int q = M / T, r = M % T;
  b = c < r ? c * (q + 1) : c * q + r; e = b + (c < r ? q + 1 : q);
  if (c >= T) { b = 0; e = 0; }
}

struct __attribute__((aligned(16))) VisSolveLds {
  float errs[VIS_ERR_STAGE];
  SolveLds s;
  double sums[64];
  double scratch[12 * 41];
  float err_chunk[LIVO2_WAVE];
  float err_total;
  double err_sum_d;
};
template <bool XB = false>
__device__ __forceinline__ void visual_solve_body(VisSolveLds &SL, DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int level, int iter,
                                                  double img_point_cov, VisualSolveArgs va, const double *craw_in = nullptr) {
  SolveLds &s = SL.s;
  double *sums = SL.sums, *scratch = SL.scratch;
  float *errs = SL.errs, *err_chunk = SL.err_chunk;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  VSPHASE(0);
  // every global read of this kernel is issued here, in one batch: loop-control words, covariance + states, the partial rows
  const int hdr_stop = ctl->hdr.stop, hdr_steps = ctl->hdr.n_steps;
  const float hdr_last_error = ctl->hdr.last_error;
  double craw[6];
  if (craw_in) {                                             // k_visual_step: P / states were fetched before the ticket
#pragma unroll
    for (int qq = 0; qq < 6; qq++) craw[qq] = craw_in[qq];
  } else if (mode != 0 && wave == 0) esikf_prefetch_wave(ctl, s, img_point_cov, lane, craw);
  if (mode != 0 && t == LIVO2_WAVE) esikf_log_lane(ctl, s);                   // second wave: overlaps the partial rows
  {
    const int kidx = t % VIS_PSTRIDE, slice = t / VIS_PSTRIDE;     // 12 slices (threads >= 480 idle here)
    double acc = 0.0;
    if (t < VIS_SOLVE_THREADS) {
      for (int base = 0; base < nblocks; base += 12 * 48) {
        double v[48];
#pragma unroll
        for (int u = 0; u < 48; u++) { const int b = base + slice + 12 * u; v[u] = (b < nblocks) ? xb_load<XB>(&partials[(size_t)b * VIS_PSTRIDE + kidx]) : 0.0; }
#pragma unroll
        for (int u = 0; u < 48; u++) acc += v[u];
      }
      scratch[slice * 41 + kidx] = acc;
    }
  }
  // frame error, part 1: stage the per-patch errors in LDS (the first VIS_ERR_STAGE of them; a larger sub-map reads the rest from HBM)
  const int n_stage = min(VIS_ERR_STAGE, va.M);
  {
    float ev[VIS_ERR_STAGE / 512];                           // all loads of a thread in flight at once (a load-store loop costs one HBM round trip per turn)
#pragma unroll
    for (int u = 0; u < VIS_ERR_STAGE / 512; u++) { const int i = t + 512 * u; ev[u] = (i < n_stage) ? xb_load<XB>(&va.errors[i]) : 0.0f; }
#pragma unroll
    for (int u = 0; u < VIS_ERR_STAGE / 512; u++) { const int i = t + 512 * u; if (i < n_stage) errs[i] = ev[u]; }
  }
  VSPHASE(1);
  __syncthreads();                                           // scratch, s.vec[0..2] (Log), s.P / s.cur / s.prop, errs visible
  VSPHASE(2);
  if (mode == 1 && iter > 0 && hdr_stop) return;            // block-uniform: the level has ended, the residual kernel did not run
  if (t < VIS_PSTRIDE) {
    double rr = scratch[t];
#pragma unroll
    for (int sl = 1; sl < 12; sl++) rr += scratch[sl * 41 + t];
    sums[t] = rr;
  }
  __syncthreads();
  VSPHASE(3);
  const int T = fc_threads(va.error_threads);
  const bool fcw = va.error_threads >= 0 && fc_use_waves(va.M, T, VS_FC_WAVES);         // block-uniform
  if (wave == 0) {                                           // the 19-dim algebra, speculative: LDS only
    if (lane < 49) {
      int rr = lane / 7, c = lane % 7;
      int u = rr < c ? rr : c, v = rr < c ? c : rr;
      int idx = u * 7 - (u * (u - 1)) / 2 + (v - u);
      s.hth[lane] = sums[idx];
    }
    if (lane < 7) s.htz[lane] = sums[28 + lane];
    wave_sync();
    if (mode != 0) esikf_solve_wave<7, false, false, true>(s, -1, lane);      // (the solution in the hv form, as the resident grid forms it: same bits)
  } else if (fcw ? (fc_wave_index(wave, VS_FC_WAVES) >= 0) : (wave == 2)) {      // frame error, part 2: OpenMP thread c's static block of patches, added in index order
    if (fcw) {                                                 // a chain per FC_W lanes (float_chain.hpp)
      const int c = fc_wave_index(wave, VS_FC_WAVES) * FC_CPW + lane / FC_W;
      int my_begin, my_end; fc_partition(va.M, T, c, my_begin, my_end);
      float priv = T == 1 ? (fc_wave_index(wave, VS_FC_WAVES) == 0 ? fc_one_chain(errs, n_stage, lane) : 0.0f) : float_chain_wave<FC_W, FC_LMAX>(errs, min(my_begin, n_stage), min(my_end, n_stage), 0.0f, lane);
      if ((lane & (FC_W - 1)) == 0 && c < T) {
        for (int i = max(my_begin, n_stage); i < my_end; i++) priv += xb_load<XB>(&va.errors[i]);
        err_chunk[c] = priv;
      }
    } else {                                                   // a chain per lane
      int my_begin, my_end; fc_partition(va.M, T, lane, my_begin, my_end);
      float priv = 0.0f;
      if (lane < T) {
        const int mid = min(my_end, n_stage);
        priv = float_chain(errs, min(my_begin, mid), mid, priv);
        for (int i = max(my_begin, n_stage); i < my_end; i++) priv += xb_load<XB>(&va.errors[i]);
        err_chunk[lane] = priv;
      }
    }
  }
  else if (wave == 1) {                                      // diagnostic double-precision sum of the patch errors (livo2_visual_sums.err_sum): a tree, off the chain
    double e = 0.0;
    for (int i = lane; i < n_stage; i += LIVO2_WAVE) e += (double)errs[i];
    for (int i = n_stage + lane; i < va.M; i += LIVO2_WAVE) e += (double)xb_load<XB>(&va.errors[i]);
    e = wave_sum(e);
    if (lane == 0) SL.err_sum_d = e;
  }
  VSPHASE(4);
  __syncthreads();
  VSPHASE(5);
  if (wave != 0) return;                                     // the rest is one wave; only wave-local synchronisation below
  const double err_sum = SL.err_sum_d;                       // (the partial rows do not carry the error: both sums are formed from errors[])
  const int n_meas = (int)sums[36];
  float error = 0.0f;
  for (int c = 0; c < T; c++) error += err_chunk[c];         // the threads' partial sums joined in thread order
  error = error / n_meas;                                   // float / int (vio.cpp:1636); NaN when n_meas == 0
  if (mode == 0) {
    if (lane < 49) ctl->sums_v.HtH[lane] = s.hth[lane];
    if (lane < 7) ctl->sums_v.Htz[lane] = s.htz[lane];
    if (lane == 0) { ctl->sums_v.err_sum = err_sum; ctl->sums_v.error = error; ctl->sums_v.n_meas = n_meas; }
    return;
  }
  const int nsl = sizeof(livo2_state) / sizeof(double);
  // level entry (iter == 0): old_state = *state, last_error = FLT_MAX (vio.cpp:1523,1528) — the first evaluation is always accepted
  // unless the error is NaN, in which case reverting to old_state == *state is a no-op.
  const float last_error = (iter == 0) ? FLT_MAX : hdr_last_error;
  const bool accepted = (mode == 2) ? true : (error <= last_error);
  const int step = hdr_steps;
  livo2_visual_step *st = (step < LIVO2_MAX_LEVELS * LIVO2_MAX_ITERS) ? &ctl->visual.steps[step] : nullptr;
  int stop = 0;
  if (accepted) {
    {                                                       // old_state = *state (vio.cpp:1650), from the prefetched copy
      double *dst = reinterpret_cast<double *>(&ctl->old);
      if (lane < 25) dst[lane] = s.cur[lane];
#pragma unroll
      for (int qq = 0; qq < 6; qq++) { const int e = lane + qq * LIVO2_WAVE; if (e < DS * DS) dst[25 + e] = craw[qq]; }
    }
    esikf_commit_wave(ctl, s, lane);
    const double rq = (s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1]) + s.sol[2] * s.sol[2], tq = (s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4]) + s.sol[5] * s.sol[5];
    if (esikf_norm_below(rq, (double)57.3f, (double)0.001f) && esikf_norm_below(tq, (double)100.0f, (double)0.001f)) stop = 1;   // vio.cpp:1675
    if (st) {
      if (lane < 49) st->HtH[lane] = s.hth[lane];
      if (lane < 7) st->Htz[lane] = s.htz[lane];
      if (lane < DS) st->solution[lane] = s.sol[lane];
    }
  } else {                                                  // (*state) = old_state ; EKF_end (vio.cpp:1679-1680)
    if (iter > 0) {
      const double *src = reinterpret_cast<const double *>(&ctl->old); double *dst = reinterpret_cast<double *>(&ctl->cur);
      for (int e = lane; e < nsl; e += LIVO2_WAVE) dst[e] = src[e];
    }
    stop = 1;
    if (st) {
      if (lane < 49) st->HtH[lane] = s.hth[lane];
      if (lane < 7) st->Htz[lane] = s.htz[lane];
      if (lane < DS) st->solution[lane] = 0.0;
    }
  }
  if (lane == 0) {
    if (accepted) ctl->hdr.last_error = error;
    if (mode == 1) ctl->hdr.stop = stop;
    if (st) { st->level = level; st->iteration = iter; st->accepted = accepted ? 1 : 0; st->n_meas = n_meas; st->error = error; st->pad = 0; }
    ctl->hdr.n_steps = step + 1;
  }
  VSPHASE(6);
  VSPHASE(7);
}

__global__ void __launch_bounds__(512) k_visual_solve(DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int level,
                                                             int iter, double img_point_cov, VisualSolveArgs va) {
  __shared__ VisSolveLds SL;
  visual_solve_body(SL, ctl, partials, nblocks, mode, level, iter, img_point_cov, va);
}

// one block per frame of a batch
__global__ void __launch_bounds__(512) k_visual_solve_batch(const VisualBatchEntry *__restrict__ entries, int mode, int level, int iter, double img_point_cov, int error_threads) {
  const VisualBatchEntry &e = entries[blockIdx.x];
  if (e.a.M == 0) return;                                    // total_points == 0 (vio.cpp:786): no step is taken, no step is recorded
  VisualSolveArgs va = {e.a.errors, e.a.M, error_threads};
  __shared__ VisSolveLds SL;
  visual_solve_body(SL, e.ctl, e.partials, e.nblocks, mode, level, iter, img_point_cov, va);
}

// One wave.  Every global read (covariance, G, the 25 pose / bias scalars) is issued in one batch and everything is written from registers / LDS: the first
// version re-read what it had just stored and walked G through HBM inside the 19-term loop (8.4 us for a 64-thread kernel).
__device__ __forceinline__ void visual_finish_body(DevCtl *__restrict__ ctl, const VisualKernelArgs &a, int update_cov) {
  __shared__ double cov[DS * DS];
  __shared__ double G[DS * DS];
  __shared__ double sc[25];
  const int lane = threadIdx.x;
  double c6[6], g6[6];
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; c6[q] = (e < DS * DS) ? ctl->cur.cov[e] : 0.0; g6[q] = (e < DS * DS) ? ctl->G[e] : 0.0; }
  const double sv = (lane < 25) ? reinterpret_cast<const double *>(&ctl->cur)[lane] : 0.0;
  const int n_steps = ctl->hdr.n_steps;
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; if (e < DS * DS) { cov[e] = c6[q]; G[e] = g6[q]; } }
  if (lane < 25) sc[lane] = sv;
  wave_sync();
  double *dst = reinterpret_cast<double *>(&ctl->visual.state);
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int e = lane + q * LIVO2_WAVE;
    if (e < DS * DS) {
      double v = c6[q];
      if (update_cov) {                                       // state->cov -= G * state->cov (vio.cpp:800)
        const int r = e / DS, c = e % DS;
        double g = G[r * DS] * cov[c];
        for (int k = 1; k < DS; k++) g = g + G[r * DS + k] * cov[k * DS + c];
        v = c6[q] - g;
        ctl->cur.cov[e] = v;
      }
      dst[25 + e] = v;
      ctl->visual.G[e] = g6[q];
    }
  }
  if (lane < 25) dst[lane] = sv;
  if (lane == 0) {                                          // updateFrameState (vio.cpp:1690-1697)
    double Rcw[9];
    mat3_mul_Bt(a.Rci, sc, Rcw);
    for (int j = 0; j < 9; j++) ctl->visual.Rcw[j] = Rcw[j];
    for (int j = 0; j < 3; j++) ctl->visual.Pcw[j] = a.Pci[j] - ((Rcw[j * 3] * sc[9] + Rcw[j * 3 + 1] * sc[10]) + Rcw[j * 3 + 2] * sc[11]);
    ctl->visual.n_steps = n_steps; ctl->visual.pad = 0;
  }
}

__global__ void __launch_bounds__(LIVO2_WAVE) k_visual_finish(DevCtl *__restrict__ ctl, VisualKernelArgs a, int update_cov) { visual_finish_body(ctl, a, update_cov); }

// ---- the whole computeJacobianAndUpdateEKF as ONE launch ---------------------------------------------------------------------------------------
// reference src/vio.cpp:784-802 (level loop, cov -= G*cov, updateFrameState) around updateState 1520-1688.
//
// Why: a frame-at-a-time visual update is 4 levels x <= 5 iterations of (residual grid, single-block solve) — 40 dependent launches of which ~13 execute,
// plus ~7 pairs that start only to read the stop flag (round 2: 42 % of a C4 frame ran on ONE compute unit, 10 % was launches of nothing).  Here the grid
// stays resident: G blocks (one group of VIS_PPB = 32 patches each while M <= 32 G), and per (level, iteration) step
//   1. every block evaluates its patches from the iterate it holds in LDS and publishes its partial row + per-patch errors,
//   2. EVERY block collects all G rows and the M errors and runs the reduction, the float error chain, the accept / revert decision and the 19-dim solve
//      REDUNDANTLY — same instructions on same data, so all blocks hold the same new iterate bit for bit and no second hand-off (solve -> publish ->
//      everyone reloads) exists.  Block 0 alone writes the step trace and, at the end, the result.
// Hand-off without a barrier: the exchange buffers hold plain doubles / floats, and a slot that has not been written yet holds an all-ones pattern (VP_EMPTY: a
// NaN with a payload no arithmetic produces).  A publisher writes each value with ONE write-through store (aligned 8- and 4-byte stores are single-copy atomic); a
// reader loads with cache-bypassing 16-byte loads (two doubles / four floats per request) and simply re-loads what is still empty: no arrival counter, no store-drain
// wait, no fence — the data is its own flag.  (Rounds 3-4 carried a 32-bit tag next to every 32 payload bits instead: twice the bytes and twice the requests of the
// collect, which is bound by exactly those.)  FOUR buffers rotate with the step number; during the collect of step k every block empties ITS OWN slots of buffer
// k + 2: its last readers finished step k - 2 before they published step k - 1, which this block has seen, and the emptying stores are acknowledged (s_waitcnt
// vmcnt(0) at the end of the phase) before this block publishes step k + 1 — so whoever looks into buffer k + 2 has seen that publication and cannot meet a value
// of step k - 2.  The step numbers run on from launch to launch (`*base`, advanced by block 0 when it leaves): a launch that ends with step L leaves buffers L + 1
// (emptied during step L - 1, or never written) and L + 2 (emptied during step L) empty — exactly what the first two steps of the next launch need, so the host
// clears nothing between updates.  Because the sub-map (M, G) changes from frame to frame, a launch empties its share of everything ANY launch since the last host
// clear may have written (clean_rows >= G rows, clean_m >= M errors: the host's high-water marks); after a timeout the host clears all four buffers.
// The loops end on the device: a level that stops after two iterations costs two steps, not five launch pairs.
// Bit-compatibility: the residual body, the order in which rows are added, the error chain and the solve are the code of the per-step kernels; with one
// patch group per block the partial rows are the same too, so the result equals the launch-per-step path bit for bit (tests/test_visual_gpu.py).
// Co-residency: the host launches at most as many blocks as the device holds at once (occupancy query, in-flight accounting across contexts, else it falls
// back to the per-step launches); a word that does not arrive within ~2 s sets hdr.pad[0] and every block leaves (livo2_visual_update_fetch reports it).
#define VP_MAX_BLOCKS 256
#define VP_MAX_ROWS 256
#ifndef VP_SLEEP0
#define VP_SLEEP0 0
#endif
#ifndef VP_SLEEP
#define VP_SLEEP 1
#endif
#define VP_RPT 21                // rows per thread and pass of the collect: 12 slices x 21 = 252 rows in ONE round of loads (a second pass is a second ~2-us round trip)
#define VP_BLOCK 512             // the collect / solve phases use 480 threads; the residual phase runs on waves 0..VIS_WAVES-1 (one per SIMD), the others wait at the barrier
struct VisPersistArgs {
  VisualKernelArgs a;
  unsigned long long *rows;     // [VP_NBUF][VP_MAX_ROWS][VIS_PSTRIDE] doubles (their bits); VP_EMPTY64 = not published yet
  uint32_t *errs;               // [VP_NBUF][err_pitch] floats (their bits); VP_EMPTY32 = not published yet
  int32_t levels, max_iterations, error_threads;
  int32_t n_rows, halves;       // rows published per step (= min(patch groups, 256)); rows per block: 1 (grid = n_rows, residual on 4 waves) or 2 (grid = n_rows / 2, all 8 waves)
  int32_t err_pitch;            // floats per error buffer (a multiple of 4: the collect loads four per request)
  int32_t clean_rows, clean_m;  // what the emptying covers: the largest n_rows / M of any launch since the host last cleared the buffers (>= n_rows, >= M)
  uint32_t *base;               // [0] step number of this launch's first step (mod 2^32; the buffer index is step % VP_NBUF), left behind by the launch before;
                                // [1] != 0: a grid gave up half-way and the buffers hold words of unfinished steps — every later launch gives up at once (and is re-run
                                //     per step by the fetch) until the host has cleared the buffers
  double img_point_cov;
  unsigned long long *prof;     // debug (LIVO2_VP_PROF=1): [block < 256][step < 32][16] stamps of the 100 MHz clock, else null
  unsigned long long timeout;   // 100 MHz ticks a block waits for a word before it gives the update up (VP_TIMEOUT; option "visual_persistent_debug_timeout" shortens it)
  int32_t debug_drop_block;     // -1; else this block leaves at once (a grid that is not co-resident, simulated: tests/test_visual_gpu.py)
  int32_t chained;              // 1: iterate = prior = the LiDAR posterior in ctl->lidar.state (what k_ctl_chain_visual would have copied into ctl->cur / ctl->prop: one launch less per frame)
  int32_t inverse;              // 1: updateStateInverse — precomputeReferencePatches at the head of every level, then the INV form of the wave body (round 6)
  VisualRefArgs r;              // (inverse)
};
#define VPP(k) do { if (p.prof && tid == 0 && blockIdx.x < VP_MAX_BLOCKS && step_global < 32) p.prof[((size_t)blockIdx.x * 32 + step_global) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define VPP_W(k, w) do { if (p.prof && tid == (w) * LIVO2_WAVE && blockIdx.x < VP_MAX_BLOCKS && step_global < 32) p.prof[((size_t)blockIdx.x * 32 + step_global) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
struct __attribute__((aligned(16))) VisPersistLds {
  union {
    struct { VisWaveLds lds[2 * VIS_WAVES]; double red[2 * VIS_WAVES][VIS_PSTRIDE]; } r;
    struct { float errs[VIS_ERR_STAGE]; double scratch[12 * 41]; double sums[64]; float err_chunk[LIVO2_WAVE]; double cov[DS * DS]; } s;
  } u;
  SolveLds s;                   // s.P = cov / img_point_cov (constant over the update: `state += solution` leaves cov alone), s.cur / s.prop = the iterate / the prior
  double old[25];               // old_state (vio.cpp:1523, 1650, 1679), scalars only (cov never differs)
  double Gfull[DS * DS];        // G as the reference keeps it: written on accepted steps only (vio.cpp:1655-1665), zero-padded 19 x 19
  float last_error, err_total;
  float last_error2[2];         // last_error as step k reads it ([k & 1]) and leaves it ([(k + 1) & 1]): the waves that decide in parallel never read a word another one writes
  int stop, n_steps, timed_out, stop_if_accepted;
  uint32_t base;                // *p.base as the launch found it
};

__device__ __forceinline__ void vis_log_lds(SolveLds &s) {           // Log(cur^T prop) from the LDS copies (esikf_log_lane reads HBM)
  double rotd[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) rotd[i * 3 + j] = (s.cur[i] * s.prop[j] + s.cur[3 + i] * s.prop[3 + j]) + s.cur[6 + i] * s.prop[6 + j];
  So3Mat m;
#pragma unroll
  for (int i = 0; i < 9; i++) m.v[i] = rotd[i];
  const So3Vec l = so3_log_call(m);
  s.vec[0] = l.v[0]; s.vec[1] = l.v[1]; s.vec[2] = l.v[2];
}

typedef unsigned long long vp_word;
#define VP_NBUF 4
#define VP_ROW_PITCH ((size_t)VP_MAX_ROWS * VIS_PSTRIDE)          // doubles per row buffer
#define VP_ERR_PITCH(M) (((size_t)(M) + 3) & ~(size_t)3)          // floats per error buffer
#define VP_EMPTY64 0xffffffffffffffffull
#define VP_EMPTY32 0xffffffffu
__device__ __forceinline__ void vp_st(vp_word *p, vp_word v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vp_st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t vp_ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes in ONE cache-bypassing request (every 8- / 4-byte part is validated on its own, so a pair that is half there is simply loaded again).  A relaxed
// agent-scope atomic load lowers to an sc1 load only up to 8 bytes, hence the asm; vp_ld2_land ties the destination registers to the wait.
typedef unsigned long long vp_word2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void vp_ld2_issue(vp_word2 &v, const void *p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void vp_ld2_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void vp_ld2_land(vp_word2 &v) { asm volatile("" : "+v"(v)); }

// The phases of a step are separate NON-inlined functions: inlined into one loop body the compiler hoists each phase's invariants (addresses, lane predicates,
// libm constants) above the step loop, where they are live across every other phase — 256 VGPRs + 112 spilled + 194 spilled SGPRs, reloaded in every phase.  A call
// costs ~0.1 us; a phase's registers now are its own (kernel maximum = the largest phase).  The kernel arguments are read where they are needed from the kernarg
// segment (scalar loads; VisPersistArgs is the kernel's first parameter, offset 0), the block's LDS through an address-space-3 pointer.
#ifndef VP_PHASE_ATTR
#define VP_PHASE_ATTR __forceinline__
#endif
typedef __attribute__((address_space(3))) VisPersistLds *VpLds;
__device__ __forceinline__ const VisPersistArgs &vp_args() {
  return *(const VisPersistArgs *)__builtin_amdgcn_kernarg_segment_ptr();
}
#define VP_TIMEOUT 200000000ull          // 100 MHz ticks: 2 s

// ---- 1. residual of this block's patch groups; the row and the errors are published into buffer step % VP_NBUF
template <bool INV> __device__ VP_PHASE_ATTR void vp_phase_residual(VpLds slp, int level_v, int step_v) {      // INV: a compile-time copy of the kernel per form (a run-time branch here cost the forward form 0.4 us per step)
  VisPersistLds &SL = *(VisPersistLds *)slp;
  const VisPersistArgs &p = vp_args();
  const int level = __builtin_amdgcn_readfirstlane(level_v), step_global = __builtin_amdgcn_readfirstlane(step_v);
  // Per-thread index arithmetic (window offsets, row addresses, lane predicates, ...) is invariant over the steps; hoisted out of the step loop it would live in
  // registers across every phase.  The thread index is therefore re-materialised opaquely in every phase, and everything derived from it stays inside the phase.
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));
  const int tid = tid_o, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.a.M, R = p.n_rows, halves = p.halves;
  const int ngroups = (M + VIS_PPB - 1) / VIS_PPB;
  const int buf = (int)((SL.base + (uint32_t)step_global) & (VP_NBUF - 1));
  vp_word *rows = p.rows + (size_t)buf * VP_ROW_PITCH;
  float *errs = reinterpret_cast<float *>(p.errs + (size_t)buf * p.err_pitch);
  VPP(0);
  double out_val = 0.0;
  // row r = blockIdx.x * halves + (wave / VIS_WAVES) sums the patch groups r, r + R, ... (a group = VIS_PPB = 16 patches = the row unit of k_visual_residual):
  // the rows — and with them every sum — do not depend on the block shape the host chose
  const int my_row = (int)blockIdx.x * halves + (halves == 1 ? 0 : wave / VIS_WAVES), wv = wave % VIS_WAVES;
  if (halves == 1) {
    // one row per block: waves 0-3 walk the critical path of their four patches, waves 4-7 (same SIMD, same VisWaveLds) form M and run the patch_error chains
    // (visual_wave_body, ROLE 1 / 2).  Every wave makes every call (the body holds a block barrier); a wave whose patches lie beyond M computes on the last patch
    // and contributes zeros, as invalid slots of a partly filled wave always did.
    if (my_row < R) {
      VPHASE(0);                                             // (profiling build: the stamps of the LAST step stay, tools/vis_phase.py --persistent)
      for (int g = my_row; g < ngroups; g += R) {
        const int patch0 = (g * VIS_WAVES + wv) * VIS_PPW;
        if (INV) {
          if (wave < VIS_WAVES) out_val += visual_wave_body<false, true, 1, true>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wv], patch0, lane, &p.r);
          else visual_wave_body<false, true, 2, true>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wv], patch0, lane, &p.r);
        }
        else if (wave < VIS_WAVES) out_val += visual_wave_body<false, true, 1>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wv], patch0, lane);
        else visual_wave_body<false, true, 2>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wv], patch0, lane);
        VPHASE(6);
        if (g + R < ngroups) __syncthreads();                // the partner is done with Rr, the main wave with Mx
      }
      if (wave < VIS_WAVES && lane < VIS_PSTRIDE) SL.u.r.red[wave][lane] = out_val;
    }
  } else if (wave < VIS_WAVES * halves && my_row < R) {
    for (int g = my_row; g < ngroups; g += R) {
      const int patch0 = (g * VIS_WAVES + wv) * VIS_PPW;
      if (patch0 < M) out_val += INV ? visual_wave_body<false, true, 0, true>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wave], patch0, lane, &p.r)
                                           : visual_wave_body<false, true>(p.a, level, SL.s.cur, SL.s.cur + 9, SL.s.cur + 12, errs, SL.u.r.lds[wave], patch0, lane);
      if (g + R < ngroups) wave_sync();
    }
    if (lane < VIS_PSTRIDE) SL.u.r.red[wave][lane] = out_val;
  }
  VPP(1);
  __syncthreads();
  if (tid < VIS_PSTRIDE * halves) {
    const int h = tid / VIS_PSTRIDE, k = tid % VIS_PSTRIDE, r = (int)blockIdx.x * halves + h;
    if (r < R) {                                               // all VIS_PSTRIDE entries (the padding beyond VIS_NSUM is 0.0): the collect reads whole pairs
      double v = SL.u.r.red[h * VIS_WAVES][k];
#pragma unroll
      for (int w = 1; w < VIS_WAVES; w++) v = v + SL.u.r.red[h * VIS_WAVES + w][k];
      if (v != v) v = __longlong_as_double(0x7ff8000000000000ll);                // (a NaN sum is published as the canonical NaN, never as the all-ones "empty" word)
      vp_st(rows + (size_t)r * VIS_PSTRIDE + k, (vp_word)__double_as_longlong(v));
    }
  }
  if (tid == LIVO2_WAVE) vis_log_lds(SL.s);                    // rotation part of vec = prior [-] iterate, while the words travel
  VPP(2);
}

// ---- 0. (inverse form) precomputeReferencePatches at the head of a level (has_ref_patch_cache = false at every level, vio.cpp:794-795) for the patch groups whose rows
// this block publishes: one patch per wave and turn, all 8 waves.  Written and read by this block only; the barrier at the end orders the two.
__device__ VP_PHASE_ATTR void vp_phase_precompute(VpLds slp, int level_v) {
  VisPersistLds &SL = *(VisPersistLds *)slp;
  const VisPersistArgs &p = vp_args();
  const int level = __builtin_amdgcn_readfirstlane(level_v);
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));
  const int tid = tid_o, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.a.M, R = p.n_rows, halves = p.halves;
  const int ngroups = (M + VIS_PPB - 1) / VIS_PPB;
  VisWaveLds &L = SL.u.r.lds[wave];
  for (int h = 0; h < halves; h++) {
    const int my_row = (int)blockIdx.x * halves + h;
    if (my_row >= R) break;
    for (int g = my_row; g < ngroups; g += R)
      for (int q = wave; q < VIS_PPB; q += VP_BLOCK / LIVO2_WAVE) {
        const int patch = g * VIS_PPB + q;
        if (patch < M) ref_precompute_patch(p.a, p.r, level, patch, lane, &L.st.Wf[0][0], &L.st.Bf[0][0]);
        wave_sync();                                             // (the next patch reuses the staging floats)
      }
  }
  __threadfence_block();
  __syncthreads();
}

// ---- 2. collect: all G rows (waves 0-3: one column pair per thread) and the M errors (waves 4-7: four per request); what is still empty is simply loaded again
__device__ VP_PHASE_ATTR void vp_phase_collect(VpLds slp, int step_v) {
  VisPersistLds &SL = *(VisPersistLds *)slp;
  const VisPersistArgs &p = vp_args();
  const int step_global = __builtin_amdgcn_readfirstlane(step_v);
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));
  const int tid_c = tid_o;
  const int tid = tid_c;                                     // (VPP)
  const int G = p.n_rows, M = p.a.M;                      // G rows (not blocks) from here on
  const int buf = (int)((SL.base + (uint32_t)step_global) & (VP_NBUF - 1)), cbuf = (buf + 2) & (VP_NBUF - 1);
  const vp_word *rows = p.rows + (size_t)buf * VP_ROW_PITCH;
  const uint32_t *errs = p.errs + (size_t)buf * p.err_pitch;
  const int n_stage = min(VIS_ERR_STAGE, M);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // this block's own slots of the buffer two steps ahead: empty again (see the head of this section for why nobody can still be reading them, and why nobody can
  // look at them before these stores have landed)
  {
    const int halves = p.halves, cm = p.clean_m, cgroups = (cm + VIS_PPB - 1) / VIS_PPB, crows_n = p.clean_rows;
    vp_word *crows = p.rows + (size_t)cbuf * VP_ROW_PITCH;
    uint32_t *cerrs = p.errs + (size_t)cbuf * p.err_pitch;
    if (tid_c < VIS_PSTRIDE * halves) {                      // (rows / patch groups r, r + G, ...: this block's own, then its share of what older launches left)
      const int r = (int)blockIdx.x * halves + tid_c / VIS_PSTRIDE;
      if (r < G) for (int rr = r; rr < crows_n; rr += G) vp_st(crows + (size_t)rr * VIS_PSTRIDE + tid_c % VIS_PSTRIDE, VP_EMPTY64);
    } else if (tid_c >= 2 * LIVO2_WAVE && tid_c < 2 * LIVO2_WAVE + VIS_PPB * halves) {
      const int q = tid_c - 2 * LIVO2_WAVE, r = (int)blockIdx.x * halves + q / VIS_PPB;
      if (r < G) for (int g = r; g < cgroups; g += G) { const int patch = g * VIS_PPB + q % VIS_PPB; if (patch < cm) vp_st32(cerrs + patch, VP_EMPTY32); }
    }
  }
  double acc0 = 0.0, acc1 = 0.0;
  const int pair = tid_c % (VIS_PSTRIDE / 2), slice = tid_c / (VIS_PSTRIDE / 2);       // rows: threads 0..239 = 12 slices x 20 column pairs
  vp_word2 ev4[VIS_ERR_STAGE / 1024];                                                      // errors: threads 256..511, four floats per request
  const int et = tid_c - 4 * LIVO2_WAVE;
  if (tid_c < VIS_SOLVE_THREADS / 2) {
    for (int base = 0; base < G; base += 12 * VP_RPT) {          // (same order of additions as k_visual_solve: rows slice, slice + 12, ... per thread and column)
      vp_word2 w2[VP_RPT];
      uint32_t need = 0, have = 0;
#pragma unroll
      for (int u = 0; u < VP_RPT; u++) if (base + slice + 12 * u < G) need |= 1u << u;
      while (have != need) {
        // unconditional loads (an asm statement cannot be predicated: a branch around each of the 21 costs an exec-mask pair apiece): an entry beyond the last row
        // reads row 0, an entry this thread already has reads the same — unchanged — values again (a buffer is not emptied before every block has left the next step)
#pragma unroll
        for (int u = 0; u < VP_RPT; u++) {
          const size_t row = (need >> u & 1u) ? (size_t)(base + slice + 12 * u) : 0;
          vp_ld2_issue(w2[u], rows + row * VIS_PSTRIDE + 2 * pair);
        }
        vp_ld2_wait();
#pragma unroll
        for (int u = 0; u < VP_RPT; u++) vp_ld2_land(w2[u]);
#pragma unroll
        for (int u = 0; u < VP_RPT; u++) if ((need >> u & 1u) && w2[u].x != VP_EMPTY64 && w2[u].y != VP_EMPTY64) have |= 1u << u;
        if (have != need) { __builtin_amdgcn_s_sleep(1); if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout) { SL.timed_out = 1; break; } }
      }
#pragma unroll
      for (int u = 0; u < VP_RPT; u++) if (need >> u & 1u) { acc0 += __longlong_as_double((long long)w2[u].x); acc1 += __longlong_as_double((long long)w2[u].y); }
    }
  } else if (et >= 0) {
    uint32_t eneed = 0, ehave = 0;
#pragma unroll
    for (int u = 0; u < VIS_ERR_STAGE / 1024; u++) if (4 * (et + 256 * u) < n_stage) eneed |= 1u << u;
    while (ehave != eneed) {
#pragma unroll
      for (int u = 0; u < VIS_ERR_STAGE / 1024; u++) vp_ld2_issue(ev4[u], errs + ((eneed >> u & 1u) ? 4 * (et + 256 * u) : 0));
      vp_ld2_wait();
#pragma unroll
      for (int u = 0; u < VIS_ERR_STAGE / 1024; u++) vp_ld2_land(ev4[u]);
#pragma unroll
      for (int u = 0; u < VIS_ERR_STAGE / 1024; u++) {           // (floats beyond the last patch of the step are not part of it: whatever they hold)
        const int i = 4 * (et + 256 * u);
        const bool ok = (uint32_t)ev4[u].x != VP_EMPTY32 && (i + 1 >= n_stage || (uint32_t)(ev4[u].x >> 32) != VP_EMPTY32) &&
                        (i + 2 >= n_stage || (uint32_t)ev4[u].y != VP_EMPTY32) && (i + 3 >= n_stage || (uint32_t)(ev4[u].y >> 32) != VP_EMPTY32);
        if ((eneed >> u & 1u) && ok) ehave |= 1u << u;
      }
      if (ehave != eneed) { __builtin_amdgcn_s_sleep(1); if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout) { SL.timed_out = 1; break; } }
    }
  }
  vp_ld2_wait();                                         // the emptying stores above have landed (threads that had nothing to load included)
  __syncthreads();                                       // every wave is done with the tiles of phase 1 (they alias the staging below)
  if (tid_c < VIS_SOLVE_THREADS / 2) { SL.u.s.scratch[slice * 41 + 2 * pair] = acc0; SL.u.s.scratch[slice * 41 + 2 * pair + 1] = acc1; }
  else if (et >= 0) {
#pragma unroll
    for (int u = 0; u < VIS_ERR_STAGE / 1024; u++) {
      const int i = 4 * (et + 256 * u);
      if (i < n_stage) {
        SL.u.s.errs[i] = __uint_as_float((uint32_t)ev4[u].x);
        if (i + 1 < n_stage) SL.u.s.errs[i + 1] = __uint_as_float((uint32_t)(ev4[u].x >> 32));
        if (i + 2 < n_stage) SL.u.s.errs[i + 2] = __uint_as_float((uint32_t)ev4[u].y);
        if (i + 3 < n_stage) SL.u.s.errs[i + 3] = __uint_as_float((uint32_t)(ev4[u].y >> 32));
      }
    }
  }
  __syncthreads();
  VPP(3);
}

// ---- 3. reduction + frame error + solve + accept / revert, redundantly in every block (the body of k_visual_solve on LDS state)
__device__ VP_PHASE_ATTR void vp_phase_solve(VpLds slp, DevCtl *ctl, int level_v, int it_v, int step_v) {
  VisPersistLds &SL = *(VisPersistLds *)slp;
  SolveLds &s = SL.s;
  const VisPersistArgs &p = vp_args();
  const int level = __builtin_amdgcn_readfirstlane(level_v), it = __builtin_amdgcn_readfirstlane(it_v), step_global = __builtin_amdgcn_readfirstlane(step_v);
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));
  const int tid = tid_o, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.a.M;
  const int n_stage = min(VIS_ERR_STAGE, M);
  const uint32_t *errs = p.errs + (size_t)((SL.base + (uint32_t)step_global) & (VP_NBUF - 1)) * p.err_pitch;
  if (tid < VIS_PSTRIDE) {
    double rr = SL.u.s.scratch[tid];
#pragma unroll
    for (int sl = 1; sl < 12; sl++) rr += SL.u.s.scratch[sl * 41 + tid];
    SL.u.s.sums[tid] = rr;
  }
  __syncthreads();
  VPP(4);
  const int T = fc_threads(p.error_threads);
  const bool fcw = p.error_threads >= 0 && fc_use_waves(M, T, VS_FC_WAVES);         // block-uniform
  if (wave == 0) {
    if (lane < 49) {
      const int rr = lane / 7, c = lane % 7, u = rr < c ? rr : c, v = rr < c ? c : rr;
      s.hth[lane] = SL.u.s.sums[u * 7 - (u * (u - 1)) / 2 + (v - u)];
    }
    if (lane < 7) s.htz[lane] = SL.u.s.sums[28 + lane];
    wave_sync();
    VPP(8);
    esikf_solve_wave<7, true, true>(s, -1, lane);         // speculative: LDS only (s.sol, s.Kc, s.newR; G = K_1 H_k is formed in the accept / revert phase, by other waves)
    if (lane == 0) {                                      // the convergence test of an accepted step (vio.cpp:1675), formed while the error chain still runs
      const double rq = (s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1]) + s.sol[2] * s.sol[2], tq = (s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4]) + s.sol[5] * s.sol[5];
      SL.stop_if_accepted = (esikf_norm_below(rq, (double)57.3f, (double)0.001f) && esikf_norm_below(tq, (double)100.0f, (double)0.001f)) ? 1 : 0;
    }
    VPP(9);
  } else if (fcw ? (fc_wave_index(wave, VS_FC_WAVES) >= 0) : (wave == 2)) {      // the frame error in the reference's float accumulation order (as in k_visual_solve)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int c, my_begin, my_end;
    float priv = 0.0f;
    bool owner;
    if (fcw) {                                            // a chain per FC_W lanes (float_chain.hpp)
      c = fc_wave_index(wave, VS_FC_WAVES) * FC_CPW + lane / FC_W;
      fc_partition(M, T, c, my_begin, my_end);
      priv = T == 1 ? (fc_wave_index(wave, VS_FC_WAVES) == 0 ? fc_one_chain(SL.u.s.errs, n_stage, lane) : 0.0f) : float_chain_wave<FC_W, FC_LMAX>(SL.u.s.errs, min(my_begin, n_stage), min(my_end, n_stage), 0.0f, lane);
      owner = (lane & (FC_W - 1)) == 0 && c < T;
    } else {                                              // a chain per lane
      c = lane;
      fc_partition(M, T, c, my_begin, my_end);
      owner = lane < T;
      if (owner) priv = float_chain(SL.u.s.errs, min(my_begin, n_stage), min(my_end, n_stage), priv);
    }
    if (owner) {
      for (int i = max(my_begin, n_stage); i < my_end; i++) {                 // sub-maps beyond the staging area: straight from the published words
        uint32_t w = vp_ld32(errs + i);
        while (w == VP_EMPTY32) { if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout) { SL.timed_out = 1; break; } w = vp_ld32(errs + i); }
        priv += __uint_as_float(w);
      }
      SL.u.s.err_chunk[c] = priv;
    }
    VPP_W(7, 2);
  }
  __syncthreads();
  VPP(5);
  if (wave < 5) {                                         // accept / revert (vio.cpp:1636, 1648-1681), on the LDS iterate
    // Five waves take the same decision from the same words (the float error, n_meas, last_error of the step's parity slot) and share what follows it: wave 0 the
    // iterate and the loop flags — the only part the next residual waits for —, waves 1, 3, 4 G = K_1 H_k (which the solve left undone), wave 2 (block 0) the step record.  (One wave did all of it: 1.0 us
    // per accepted step against 0.64 for a reverted one.)
    const int n_meas = (int)SL.u.s.sums[36];
    float error = 0.0f;
    for (int c = 0; c < T; c++) error += SL.u.s.err_chunk[c];       // the threads' partial sums joined in thread order
    error = error / n_meas;
    const float last_error = (it == 0) ? FLT_MAX : SL.last_error2[step_global & 1];
    const bool accepted = error <= last_error;
    const int step = step_global;                          // (= the number of steps taken so far: SL.n_steps)
    if (wave == 0) {
      int stop = 0;
      if (accepted) {
        stop = SL.stop_if_accepted;
        double nv = 0.0;
        if (lane < 9) nv = s.newR[lane]; else if (lane < 25) nv = s.cur[lane] + s.sol[lane - 6];
        if (lane < 25) { SL.old[lane] = s.cur[lane]; s.cur[lane] = nv; }                              // old_state = *state ; *state += solution
      } else {
        if (it > 0 && lane < 25) s.cur[lane] = SL.old[lane];                                           // *state = old_state ; EKF_end
        stop = 1;
      }
      if (lane == 0) { SL.last_error2[(step_global + 1) & 1] = accepted ? error : last_error; if (accepted) SL.last_error = error; SL.stop = stop; SL.n_steps = step + 1; }
    } else if (wave != 2) {                               // waves 1, 3, 4: one element of G each
      if (accepted) {
        for (int e = (wave == 1 ? 0 : wave - 2) * LIVO2_WAVE + lane; e < DS * KMAX; e += 3 * LIVO2_WAVE) {                  // G = K_1 H_k, left undone by the solve (DEFER_G): the FMA order of esikf_solve_wave
          const int r = e / KMAX, c = e % KMAX;
          double g = 0.0;
#pragma unroll
          for (int m = 0; m < KMAX; m++) g = fma(s.Kc[r * KMAX + m], s.hth[m * KMAX + c], g);
          SL.Gfull[r * DS + c] = g;
        }
      }
    } else if (blockIdx.x == 0 && step < LIVO2_MAX_LEVELS * LIVO2_MAX_ITERS) {
      livo2_visual_step *st = &ctl->visual.steps[step];
      if (lane < 49) st->HtH[lane] = s.hth[lane];
      if (lane < 7) st->Htz[lane] = s.htz[lane];
      if (lane < DS) st->solution[lane] = accepted ? s.sol[lane] : 0.0;
      if (lane == 0) { st->level = level; st->iteration = it; st->accepted = accepted ? 1 : 0; st->n_meas = n_meas; st->error = error; st->pad = 0; }
    }
  }
  __syncthreads();
  VPP(6);
}

template <bool INV> __global__ void __launch_bounds__(VP_BLOCK) k_visual_update_persistent(VisPersistArgs p, DevCtl *__restrict__ ctl) {
  __shared__ VisPersistLds SL;
  SolveLds &s = SL.s;
  const VpLds slp = (VpLds)&SL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = (int)gridDim.x, M = p.a.M;
  const int ngroups = (M + VIS_PPB - 1) / VIS_PPB;
  if ((int)blockIdx.x == p.debug_drop_block) return;
  // ---- entry: the iterate, the prior, P' = cov / img_point_cov, the G the reference would still hold (all blocks read the same words)
  const livo2_state *src_state = p.chained ? &ctl->lidar.state : &ctl->cur;        // (LIVMapper.cpp:135-136, 256, 371: `state` is shared; nobody writes either before block 0's last lines)
  {
    double craw[6];
    if (wave == 0) esikf_prefetch_wave(src_state, p.chained ? src_state : &ctl->prop, s, p.img_point_cov, lane, craw);
    for (int e = tid; e < DS * DS; e += VP_BLOCK) SL.Gfull[e] = ctl->G[e];
    if (tid == 0) { SL.last_error = FLT_MAX; SL.stop = 0; SL.n_steps = 0; SL.base = p.base[0]; SL.timed_out = p.base[1] != 0 ? 1 : 0; }      // (base[1]: the launch before gave up half-way)
  }
  __syncthreads();
  if (tid < 25) SL.old[tid] = s.cur[tid];
  __syncthreads();
  int step_global = 0, last_buf = 0;
  for (int level = p.levels - 1; level >= 0 && !SL.timed_out; level--) {
    if (INV) vp_phase_precompute(slp, level);
    for (int it = 0; it < p.max_iterations; it++) {
      vp_phase_residual<INV>(slp, level, step_global);
      last_buf = (int)((SL.base + (uint32_t)step_global) & (VP_NBUF - 1));
      vp_phase_collect(slp, step_global);
      if (SL.timed_out) break;
      vp_phase_solve(slp, ctl, level, it, step_global);
      step_global++;
      if (SL.stop) break;
    }
    if (SL.timed_out) break;
  }
  __syncthreads();
  const VisualKernelArgs &a = p.a;
  // ---- every block: errors[] of the LAST evaluated step for its own patches (visual_submap->errors, vio.cpp:1632)
  {
    const uint32_t *errs = p.errs + (size_t)last_buf * p.err_pitch;
    for (int g = blockIdx.x; g < ngroups; g += G) {
      const int patch = g * VIS_PPB + tid;
      if (tid < VIS_PPB && patch < M) a.errors[patch] = __uint_as_float(vp_ld32(errs + patch));
    }
  }
  if (SL.timed_out && tid == 0) p.base[1] = 1u;
  if (blockIdx.x != 0) return;
  // A grid that lost a block (not co-resident: admission is per process, advisor round 3) commits NOTHING: ctl->cur, cov and G stay what the launch found, only the
  // flag goes up, and livo2_visual_update_fetch re-runs the update as the launch-per-step sequence from the inputs it kept.
  if (SL.timed_out) { if (tid == 0) { ctl->hdr.pad[0] = 1; ctl->visual.pad = 1; } return; }      // (visual.pad: the flag travels with the result block — one copy command less per fetch)
  if (tid == 0) p.base[0] = SL.base + (uint32_t)step_global;     // (every other block read it before it published its first row, and this block has seen all of those)
  // ---- block 0: the result.  state->cov -= G * state->cov (vio.cpp:800), updateFrameState (vio.cpp:1690-1697)
  for (int e = tid; e < DS * DS; e += VP_BLOCK) SL.u.s.cov[e] = src_state->cov[e];
  __syncthreads();
  if (p.chained) {                                               // what k_ctl_chain_visual leaves behind: prop = the LiDAR posterior, header cleared (hdr.stop / n_steps / last_error follow below)
    const double *ls = reinterpret_cast<const double *>(&ctl->lidar.state); double *pp = reinterpret_cast<double *>(&ctl->prop);
    for (int k = tid; k < (int)(sizeof(livo2_state) / 8); k += VP_BLOCK) pp[k] = ls[k];
    if (tid == 0) { ctl->hdr.rematch_num = 0; ctl->hdr.reserved = 0; ctl->hdr.pad[1] = ctl->hdr.pad[2] = 0; }
    if (tid < 9) ctl->hdr.RE[tid] = 0.0;
  }
  double *dst = reinterpret_cast<double *>(&ctl->visual.state);
  for (int e = tid; e < DS * DS; e += VP_BLOCK) {
    const int r = e / DS, c = e % DS;
    double g = SL.Gfull[r * DS] * SL.u.s.cov[c];
    for (int k = 1; k < DS; k++) g = g + SL.Gfull[r * DS + k] * SL.u.s.cov[k * DS + c];
    const double v = SL.u.s.cov[e] - g;
    ctl->cur.cov[e] = v; dst[25 + e] = v;
    ctl->G[e] = SL.Gfull[e]; ctl->visual.G[e] = SL.Gfull[e];
  }
  if (tid < 25) { reinterpret_cast<double *>(&ctl->cur)[tid] = s.cur[tid]; dst[tid] = s.cur[tid]; }
  if (tid == 0) {
    double Rcw[9];
    mat3_mul_Bt(a.Rci, s.cur, Rcw);
    for (int j = 0; j < 9; j++) ctl->visual.Rcw[j] = Rcw[j];
    for (int j = 0; j < 3; j++) ctl->visual.Pcw[j] = a.Pci[j] - ((Rcw[j * 3] * s.cur[9] + Rcw[j * 3 + 1] * s.cur[10]) + Rcw[j * 3 + 2] * s.cur[11]);
    ctl->visual.n_steps = SL.n_steps; ctl->hdr.n_steps = SL.n_steps; ctl->hdr.last_error = SL.last_error; ctl->hdr.stop = SL.stop;
    ctl->hdr.pad[0] = 0; ctl->visual.pad = 0;
  }
}

// ---- livo2_debug_float_chain: the frame error's partial sums both ways, for the tests (tests/test_float_chain_gpu.py) ------------------------------------------------
// errors[n] are staged in LDS as the solve kernels stage them; out[0][c]: a chain per group of W lanes (waves 1..7), out[1][c]: a chain per lane (wave 0).
template <int W, int LMAX>
__global__ void __launch_bounds__(512) k_debug_float_chain(const float *__restrict__ errors, int n, int T, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float errs[VIS_ERR_STAGE + 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < VIS_ERR_STAGE + 64; i += 512) errs[i] = i < n ? errors[i] : 0.0f;
  __syncthreads();
  if (wave == 0) {
    int b, e; fc_partition(n, T, lane, b, e);
    if (lane < T) out[64 + lane] = float_chain(errs, b, e, 0.0f);
  } else {
    const int c = (wave - 1) * (LIVO2_WAVE / W) + lane / W;
    int b, e; fc_partition(n, T, c, b, e);
    const float v = float_chain_wave<W, LMAX>(errs, b, e, 0.0f, lane);
    if ((lane & (W - 1)) == 0 && c < T) out[c] = v;
  }
}
