// The frame error's serial float accumulation (reference src/vio.cpp:1554, 1634: `error += patch_error` inside an OpenMP static block, one float rounding per
// add) evaluated by W lanes per chain instead of one dependent chain of adds — SAME bits.  tools/float_chain_model.cpp is the CPU model of exactly this code
// (fuzzed against the serial loop, ties / binade crossings / zeros / subnormals / NaNs included: tests/test_float_chain_cpu.py); the GPU form is checked through
// livo2_debug_float_chain (tests/test_float_chain_gpu.py).
//
// Every number below is produced by real float adds; nothing is inferred about rounding.
//   0. lane l of a group of W lanes owns L consecutive elements (L odd: the lanes' LDS reads fall into distinct banks), held in registers.
//   1. guesses: the f64 prefix (DPP scan) over the float sums of the segments (four partial sums per lane: a short dependent path).
//   2. one PLAIN round: every lane adds its segment to its guessed start.  The increments it measures are increments in the accumulator's binade (a segment's float sum
//      from 0 is off by a few of the accumulator's ulps — more than step 3 absorbs); their f64 prefix behind the exact end of lane 0 gives the second guesses.
//   3. a TABLE round: every lane adds its segment to the EIGHT consecutive floats around its guess — bit patterns b .. b + 7, b = bits(guess) - 3 — as four chains of
//      v_pk_add_f32: a table of the lane's true start -> end function on that window.  What the next lane needs of it is an INDEX map: g_l[i] = where the end e_l[i]
//      lies in the window of lane l + 1 (bits(e_l[i]) - b_{l+1}; 0xff outside 0..7) — eight bytes, and index maps compose by byte permutation: ONE v_perm_b32 per four
//      entries, a selector byte 0xff yielding 0xff.  A log-step DPP scan (two moves + two permutes a level) gives every lane the map "index of the chain's start in
//      lane 0's window -> index of MY true start in my window", hence its exact end — or INVALID from the first lane on whose true start fell outside its window.  The
//      starts of all lanes up to that one are exact then, the guesses behind it are shifted by what that lane's guess was off, and step 3 is repeated (rare: the fuzz
//      of the model needs 1.00 table rounds per pass).  After FC_MAXT table rounds the rest is added serially.
// Any element that is negative, infinite or NaN sends the whole call to the serial loop (a table entry could equal the INVALID pattern otherwise).
// (Round 6 first shipped a form that certified segments as LINEAR — same increment from a start and from its twin with the last mantissa bit flipped, one binade, no tie
// — and accepted an exact f64 prefix of increments up to the first uncertified lane: one round per binade crossing of the running sum, 4.5 rounds per chain at C4,
// 2.8-5.6 us.  The tables need no reasoning about ties or binades and one round: profiles/r06_visual_error_waves_ab.txt.)
#pragma once
#include "livo2_device.hpp"

#define FC_MAXT 6              // table rounds before the rest of a pass is added serially
#define FC_INVALID 0xffffffffu
#define FC_MIN_N 768            // shorter chains stay on the serial loop (a round costs ~140 instructions whatever the length; measured break-even between 500 and 1 000)

typedef float fc_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fc_exp(float f) { return (int)((__float_as_uint(f) >> 23) & 0xffu); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double fc_dpp(double v) {          // lanes without a source (row edge, masked row) receive 0.0
  const int2 w = __builtin_bit_cast(int2, v);
  constexpr bool BC = ROW_MASK == 0xf;                        // every row written: a lane without a source reads 0 (bound_ctrl) and no `old` value has to be set up
  const int lo = __builtin_amdgcn_update_dpp(0, w.x, CTRL, ROW_MASK, 0xf, BC);
  const int hi = __builtin_amdgcn_update_dpp(0, w.y, CTRL, ROW_MASK, 0xf, BC);
  return __builtin_bit_cast(double, make_int2(lo, hi));
}
// 32-bit DPP move; lanes without a source keep `v`
template <int CTRL> __device__ __forceinline__ uint32_t fc_dpp_u32(uint32_t v) {
  constexpr int ROW_MASK = CTRL == 0x142 ? 0xa : CTRL == 0x143 ? 0xc : 0xf;
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// one step of the prefix scan over index maps (lo: entries 0-3, hi: entries 4-7, a byte each, 0xff = outside the window): the partner's lanes first, then this map's
// — R[i] = own[partner[i]] is a byte permute of {hi, lo} by the partner's bytes, and a selector byte >= 13 yields 0xff — where this lane has a partner
template <int CTRL> __device__ __forceinline__ void fc_compose(uint32_t &lo, uint32_t &hi, const bool has_partner) {
  const uint32_t plo = fc_dpp_u32<CTRL>(lo), phi = fc_dpp_u32<CTRL>(hi);
  const uint32_t nlo = __builtin_amdgcn_perm(hi, lo, plo), nhi = __builtin_amdgcn_perm(hi, lo, phi);
  if (has_partner) { lo = nlo; hi = nhi; }
}
// inclusive sum over the lanes of a group of W (16 / 32 / 64) consecutive lanes; every partial sum is the sum of a contiguous run of lanes
template <int W> __device__ __forceinline__ double fc_scan(double v) {
  v += fc_dpp<0x111, 0xf>(v); v += fc_dpp<0x112, 0xf>(v); v += fc_dpp<0x114, 0xf>(v); v += fc_dpp<0x118, 0xf>(v);      // row_shr 1, 2, 4, 8
  if (W >= 32) v += fc_dpp<0x142, 0xa>(v);                                                                             // row_bcast15 -> rows 1 and 3
  if (W >= 64) v += fc_dpp<0x143, 0xc>(v);                                                                             // row_bcast31 -> rows 2 and 3
  return v;
}

// acc + e[begin] + e[begin + 1] + ... + e[end - 1], one float rounding per add.  `e`: LDS.  Called by all 64 lanes of a wave; begin / end / acc are equal within
// a group of W lanes (a group without a chain passes begin == end); every lane of a group returns the group's sum.  LMAX (odd): elements per lane and pass.
template <int W, int LMAX>
__device__ __forceinline__ float float_chain_wave(const float *e_generic, const int begin, const int end, float acc, const int lane) {
  static_assert((LMAX & 1) == 1 && (W == 16 || W == 32 || W == 64), "odd segment length, 16 / 32 / 64 lanes per chain");
  constexpr int SLOTS = (LMAX + 3) & ~3;
  typedef __attribute__((address_space(3))) const float *lds_cfp;
  const lds_cfp e = (lds_cfp)e_generic;
  const int l = lane & (W - 1), gbase = lane & ~(W - 1);
  const unsigned long long gmask = W == 64 ? ~0ull : ((1ull << W) - 1ull);
  int pos = begin;
  for (;;) {
    const int nsb = min(end - pos, W * LMAX);                                  // (<= 0: this group has no elements left)
    int nmax = nsb;
#pragma unroll
    for (int g = W; g < 64; g += W) nmax = max(nmax, __builtin_amdgcn_readlane(nsb, g));
    nmax = max(nmax, __builtin_amdgcn_readlane(nsb, 0));
    if (nmax <= 0) break;
    const int Lu = ((nmax + W - 1) / W) | 1;                                   // wave-uniform, odd, <= LMAX
    const int mfull = min(Lu, nsb - l * Lu);                                   // valid elements of this lane (<= 0: none)
    float x[SLOTS];
    uint32_t worst = __float_as_uint(acc);
    {
      const lds_cfp src = e + max(pos, 0) + min(l * Lu, max(nsb, 0));          // (the lanes behind the end read what follows — masked below, inside LDS either way)
#pragma unroll
      for (int c = 0; c < SLOTS; c += 4) {
        if (c < Lu) {
#pragma unroll
          for (int m = c; m < c + 4; m++) { const float v = src[m]; x[m] = (m < mfull) ? v : 0.0f; }
          worst = max(worst, max(max(__float_as_uint(x[c]), __float_as_uint(x[c + 1])), max(__float_as_uint(x[c + 2]), __float_as_uint(x[c + 3]))));
        } else {
#pragma unroll
          for (int m = c; m < c + 4; m++) x[m] = 0.0f;
        }
      }
    }
    if (__ballot(worst >= 0x7f800000u) != 0ull) {                              // a negative, infinite or NaN element (or start): the serial loop, all groups
      float v = acc;
      if (l == 0) for (int i = begin > pos ? begin : pos; i < end; i++) v += e[i];
      return __shfl(v, gbase);
    }
    const uint32_t acc_bits = __float_as_uint(acc);
    float a = acc;
    {                                                                          // 1. guesses
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
#pragma unroll
      for (int m = 0; m < SLOTS; m += 4) { g0 += x[m]; g1 += x[m + 1]; g2 += x[m + 2]; g3 += x[m + 3]; }
      const double T = (double)((g0 + g1) + (g2 + g3)) + (l == 0 ? (double)acc : 0.0);
      const double s = fc_scan<W>(T) - T;
      if (l > 0) a = (float)s;
    }
    {                                                                          // 2. the plain round
      float v = a;
#pragma unroll
      for (int m = 0; m < SLOTS; m++) v += x[m];                               // (slots beyond Lu hold 0)
      const double T = l == 0 ? (double)v : (double)v - (double)a;
      const double s = fc_scan<W>(T) - T;
      if (l > 0) a = (float)s;
    }
    bool done = nsb <= 0;
    float out = acc;
    int lf = 0;                                                                // (a pass that ends unfinished: the first lane whose start is not confirmed, and that start)
    float xf = acc;
    for (int t = 0; t < FC_MAXT; t++) {                                        // 3. table rounds
      const uint32_t ba = __float_as_uint(a);
      const uint32_t fb = ba - min(ba, 3u);                                    // the window: b .. b + 7, the guess at index 3 (lane 0: the chain's start at index min(bits, 3))
      fc_v2 v01 = fc_v2{__uint_as_float(fb), __uint_as_float(fb + 1u)}, v23 = fc_v2{__uint_as_float(fb + 2u), __uint_as_float(fb + 3u)};
      fc_v2 v45 = fc_v2{__uint_as_float(fb + 4u), __uint_as_float(fb + 5u)}, v67 = fc_v2{__uint_as_float(fb + 6u), __uint_as_float(fb + 7u)};
#pragma unroll
      for (int m = 0; m < SLOTS; m++) { const fc_v2 xx = fc_v2{x[m], x[m]}; v01 += xx; v23 += xx; v45 += xx; v67 += xx; }
      const uint32_t e0 = __float_as_uint(v01.x), e1 = __float_as_uint(v01.y), e2 = __float_as_uint(v23.x), e3 = __float_as_uint(v23.y);
      const uint32_t e4 = __float_as_uint(v45.x), e5 = __float_as_uint(v45.y), e6 = __float_as_uint(v67.x), e7 = __float_as_uint(v67.y);
      // index map of this lane: where its ends lie in the NEXT lane's window
      const uint32_t nb = (uint32_t)__shfl_down((int)fb, 1, W);
      auto gi = [&](uint32_t en) -> uint32_t { const uint32_t d = en - nb; return d < 8u ? d : 0xffu; };
      uint32_t lo = gi(e0) | (gi(e1) << 8) | (gi(e2) << 16) | (gi(e3) << 24), hi = gi(e4) | (gi(e5) << 8) | (gi(e6) << 16) | (gi(e7) << 24);
      // prefix composition: after the scan (lo, hi) of lane l maps an index in lane 0's window to the index in lane l + 1's window
      fc_compose<0x111>(lo, hi, (lane & 15) >= 1); fc_compose<0x112>(lo, hi, (lane & 15) >= 2);
      fc_compose<0x114>(lo, hi, (lane & 15) >= 4); fc_compose<0x118>(lo, hi, (lane & 15) >= 8);
      if (W >= 32) fc_compose<0x142>(lo, hi, (lane & 16) != 0);                // row_bcast15: rows 1 and 3 take lane 15 of the row before
      if (W >= 64) fc_compose<0x143>(lo, hi, lane >= 32);                      // row_bcast31: rows 2 and 3 take lane 31
      const uint32_t i0 = min(acc_bits, 3u);                                   // the chain's start in lane 0's window
      const uint32_t plo = (uint32_t)__shfl_up((int)lo, 1, W);                 // (the map of the lanes before this one)
      const uint32_t idx = l == 0 ? i0 : ((plo >> (8u * i0)) & 0xffu);        // this lane's true start in its own window, or 0xff
      const uint32_t s01 = (idx & 1u) ? e1 : e0, s23 = (idx & 1u) ? e3 : e2, s45 = (idx & 1u) ? e5 : e4, s67 = (idx & 1u) ? e7 : e6;
      const uint32_t s03 = (idx & 2u) ? s23 : s01, s47 = (idx & 2u) ? s67 : s45;
      const uint32_t val = idx < 8u ? ((idx & 4u) ? s47 : s03) : FC_INVALID;   // the exact end of this lane, or INVALID
      const unsigned long long bm = (__ballot(val == FC_INVALID) >> gbase) & gmask;
      const int first_bad = bm ? (int)__builtin_ctzll(bm) : W;
      const uint32_t prev = (uint32_t)__shfl_up((int)val, 1, W);               // the exact start of this lane (lanes up to first_bad)
      if (!done) {
        if (first_bad >= W) { out = __uint_as_float((uint32_t)__shfl((int)val, gbase + W - 1)); done = true; }
        else { lf = first_bad; }
      }
      if (__ballot(!done) == 0ull) break;
      // (lane 0 cannot be invalid: its window holds the chain's start)  new guesses: exact starts up to lane lf, the guesses behind it shifted by what lane lf's was off
      const int lfc = lf > 0 ? lf : 1;
      const uint32_t xfb = (uint32_t)__shfl((int)val, gbase + lfc - 1);
      const float a_lf = __shfl(a, gbase + lfc);
      const double delta = (double)__uint_as_float(xfb) - (double)a_lf;
      if (!done) {
        xf = __uint_as_float(xfb);
        if (l > 0) a = (l <= lf) ? __uint_as_float(prev) : (float)((double)a + delta);
      }
    }
    if (__ballot(!done) != 0ull) {                                             // FC_MAXT table rounds were not enough: the rest serially, from the exact start of lane lf
      float v = lf > 0 ? xf : acc;
      const int i0 = lf > 0 ? lf * Lu : 0;
      if (!done && l == 0) for (int i = pos + i0; i < pos + nsb; i++) v += e[i];
      v = __shfl(v, gbase);
      if (!done) out = v;
    }
    if (nsb > 0) { acc = out; pos += nsb; }
  }
  return acc;
}
