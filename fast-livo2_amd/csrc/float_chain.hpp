// The frame error's serial float accumulation (reference src/vio.cpp:1554, 1634: `error += patch_error` inside an OpenMP static block, one float rounding per
// add) evaluated by W lanes per chain instead of one dependent chain of adds — SAME bits.  tools/float_chain_model.cpp is the CPU model of exactly this code
// (fuzzed against the serial loop, ties / binade crossings / zeros / subnormals / NaNs included: tests/test_float_chain_cpu.py); the GPU form is checked through
// livo2_debug_float_chain (tests/test_float_chain_gpu.py).
//
// Why it can be done: while the accumulator stays inside one binade (ulp u) and no add lands exactly half-way between two floats, fl(acc + e) = acc + D(e) u with
// D(e) = e rounded to a multiple of u — independent of acc.  A segment of the chain that neither leaves the binade nor meets a tie adds a CONSTANT, which a lane can
// measure from a guessed start; only segments that cross a binade or meet a tie need their true start.
//   1. lane l of a group of W lanes owns L consecutive elements (L odd: the lanes' LDS reads fall into distinct banks), held in registers.
//   2. a round: every lane runs its L adds twice, from its start `a` and from `a` with the last mantissa bit flipped.  Without a tie both runs make the same rounding
//      decisions and keep their distance of one ulp; a tie changes the distance to 0 or 2 ulps for good (round-to-even looks at the parity).  "Same increment in
//      both runs, start and both ends in one binade" therefore certifies the segment as linear (elements are >= 0: the run is monotone).
//   3. the true starts follow from an exact f64 prefix sum (DPP scan) over the increments, beginning at the end of the last lane whose start is known exactly
//      (`jx`; lane 0 at first); they are accepted up to the first lane whose segment is not certified for its true start.  That lane's start is exact now: the next
//      round runs it for real.  The first guesses are the f64 prefix over the float sums of the segments.  Every round advances jx; after FC_MAXR rounds
//      (tie-ridden or wildly ranged data) the rest is added serially.
// Exactness of the f64 sums: every term is a multiple of 2^(klow-23) and every accepted partial sum is below 2^(klow+30), klow = exponent field of the exact base.
// Any element that is negative, infinite or NaN sends the whole call to the serial loop.
// Measured at C4 (1 000 errors per OpenMP thread): ~5 rounds per chain; see DESIGN.md section 4.3.
#pragma once
#include "livo2_device.hpp"

#define FC_MAXR 24
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
    float a = (l == 0) ? acc : 0.0f;
    int jx = 0;
    bool done = nsb <= 0;
    float res = acc;
    // The loop is rotated: the acceptance of round r - 1 (shuffle, ballot, first bad lane: ~40 instructions whose results only the NEXT prefix sum needs) stands in
    // one basic block with the adds of round r (no branch inside either), so that it fills the issue slots the dependent adds leave; the loop-carried path is
    // run -> increment -> scan -> start.
    // Both runs of a round are ONE chain of packed adds (v_pk_add_f32: the same round-to-nearest-even add on both halves).
    fc_v2 v; float p_a, p_v0, p_v1; double p_C, p_s; int p_jx;
    auto run = [&]() {
      v = fc_v2{a, __uint_as_float(__float_as_uint(a) ^ 1u)};
#pragma unroll
      for (int m = 0; m < SLOTS; m++) v += fc_v2{x[m], x[m]};                  // (slots beyond Lu hold 0)
    };
    auto advance = [&]() {
      const float v0 = v.x;
      const double C = (double)v0 - (double)a;
      const double T = l < jx ? 0.0 : (l == jx ? (double)v0 : C);
      const double s = fc_scan<W>(T) - T;                                      // the true start of this lane if every lane between jx and here is certified
      p_a = a; p_v0 = v0; p_v1 = v.y; p_C = C; p_s = s; p_jx = jx;
      if (l > jx) a = (float)s;
    };
    {                                                                          // guesses: the float sum of each lane's segment (four partial sums: a short dependent path), f64 prefix
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
#pragma unroll
      for (int m = 0; m < SLOTS; m += 4) { g0 += x[m]; g1 += x[m + 1]; g2 += x[m + 2]; g3 += x[m + 3]; }
      const double T = (double)((g0 + g1) + (g2 + g3)) + (l == 0 ? (double)acc : 0.0);
      const double s = fc_scan<W>(T) - T;
      if (l > 0) a = (float)s;
    }
    run(); advance();
    for (int round = 1;; round++) {
      run();
      {                                                                        // acceptance of the round before (bitwise logic: no branches, one basic block with the run)
        const float q_a1 = __uint_as_float(__float_as_uint(p_a) ^ 1u);
        const double C1 = (double)p_v1 - (double)q_a1;
        const int k = fc_exp(p_a);
        const int reg = (int)(C1 == p_C) & (int)(fc_exp(p_v0) == k) & (int)(fc_exp(p_v1) == k) & (int)(k >= 2);
        const float rj = __shfl(p_v0, gbase + p_jx);                           // the end of lane jx: exact
        const int klow = fc_exp(rj);
        const double en = p_s + p_C;
        const float sf = (float)p_s, ef = (float)en;
        const int exact = (int)((double)sf == p_s) & (int)((double)ef == en) & (int)(k >= klow) & (int)(fc_exp(ef) - klow <= 28) & (int)(fc_exp(sf) >= klow);
        const int same = (int)(__float_as_uint(sf) == __float_as_uint(p_a)) & (int)(__float_as_uint(ef) == __float_as_uint(p_v0));      // the guess WAS the true start: its run is the true run
        const int lin = reg & (int)(fc_exp(sf) == k) & (int)(fc_exp(ef) == k);
        const int ok = exact & (same | lin);
        const unsigned long long bm = (__ballot((int)(l > p_jx) & (ok ^ 1)) >> gbase) & gmask;
        int first_bad = bm ? (int)__builtin_ctzll(bm) : W;
        first_bad = klow == 0 ? p_jx + 1 : first_bad;                          // zero or subnormal base: one lane per round (its start is the base itself, no sum involved)
        const bool fin = first_bad >= W;
        res = (!done && fin) ? ((p_jx == W - 1) ? p_v0 : ef) : res;            // (lane W - 1 holds the group's sum)
        jx = (!done && !fin) ? first_bad : jx;
        done = done || fin;
      }
      asm volatile("" : "+v"(v));                                              // the run belongs HERE, beside the acceptance (it would be sunk behind the exits otherwise)
      if (__ballot(!done) == 0ull) break;
      if (round >= FC_MAXR) break;
      advance();
    }
    float out = __shfl(res, gbase + W - 1);
    if (__ballot(!done) != 0ull) {                                             // FC_MAXR rounds were not enough: the rest serially, from the exact start of lane jx
      float v = __shfl(a, gbase + jx);
      if (!done && l == 0) for (int i = pos + jx * Lu; i < pos + nsb; i++) v += e[i];
      v = __shfl(v, gbase);
      if (!done) out = v;
    }
    if (nsb > 0) { acc = out; pos += nsb; }
  }
  return acc;
}
