// CPU model of float_chain_wave (fast-livo2_amd/csrc/float_chain.hpp): the reference's serial float accumulation of the per-patch errors
// (reference src/vio.cpp:1554, 1634: `error += patch_error` inside an OpenMP static block) evaluated by W lanes instead of one dependent chain of adds,
// with the SAME bits.  Test infrastructure: g++ -O2 -ffp-contract=off tools/float_chain_model.cpp && ./a.out [cases]  (tests/test_float_chain_cpu.py runs it).
//
// The method (lane by lane as the device code does it; every number below is produced by real float adds, nothing is inferred):
//   0. lane l of a group of W lanes owns L consecutive elements.
//   1. guesses: the f64 prefix over the float sums of the segments (four partial sums per lane).
//   2. one PLAIN round: every lane adds its segment to its guessed start; the increments it measures are increments in the right binade (a segment's float sum from 0
//      is off by a few ulps of the accumulator's binade, which step 3 could not absorb), their f64 prefix behind the exact end of lane 0 gives the second guesses.
//   3. a TABLE round: every lane adds its segment to the EIGHT consecutive floats around its guess, bit patterns b .. b + 7 with b = bits(guess) - 3 — a table of the
//      lane's true start -> end function on that window.  What the next lane needs of it is an INDEX map: g_l[i] = where the end e_l[i] lies in the window of lane
//      l + 1 (bits(e_l[i]) - b_{l+1}, INVALID = 0xff outside 0..7): eight bytes.  Index maps compose by byte permutation (one v_perm_b32 per four entries on the
//      device, 0xff selecting 0xff), so a log-step scan gives every lane the map "index of the chain's start in lane 0's window -> index of my true start in MY
//      window", and with it the exact end of the lane — or INVALID from the first lane on whose true start fell outside its window.  Then the starts of all lanes up
//      to that one are exact, the guesses behind it are shifted by what that lane's guess was off, and step 3 is repeated (rare).  After MAXT table rounds the rest is
//      added serially.
// Any element that is negative, infinite or NaN sends the whole call to the serial loop (a table entry could then equal the INVALID pattern).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float fromb(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static long g_ties = 0;
static float chain_seq(const float *e, int n, float acc) {
  for (int i = 0; i < n; i++) {
    const double s = (double)acc + (double)e[i];
    volatile float t = acc + e[i];
    if (s == s && std::fabs(s) < 1e38 && std::fabs(s - (double)t) * 2.0 == (double)std::nextafter(std::fabs(t), INFINITY) - (double)std::fabs(t)) g_ties++;
    acc = t;
  }
  return acc;
}

struct Stats { long table_rounds = 0, superblocks = 0, bails = 0, serial = 0; int max_rounds = 0; };
static const int MAXT = 6;
static const uint32_t INVALID = 0xffffffffu;
static int lmax_of(int W) { return W == 16 ? 63 : W == 32 ? 35 : 19; }      // the instantiations of float_chain_wave

static const int WIN = 8, WOFF = 3;
struct Map { uint8_t g[WIN]; };                             // index map of a run of lanes: index in the first lane's window -> index in the window of the lane behind the run
// own = the later lanes, partner = the lanes before them: partner first, then own (R[i] = own[partner[i]], 0xff stays 0xff)
static inline Map compose(const Map &own, const Map &partner) { Map r; for (int i = 0; i < WIN; i++) r.g[i] = partner.g[i] < WIN ? own.g[partner.g[i]] : 0xff; return r; }

static float chain_par(const float *e, int n, float acc, int W, Stats &st) {
  if (fbits(acc) >= 0x7f800000u) { st.serial++; return chain_seq(e, n, acc); }
  for (int i = 0; i < n; i++) if (fbits(e[i]) >= 0x7f800000u) { st.serial++; return chain_seq(e, n, acc); }      // negative, -0, inf, NaN: the serial loop
  int pos = 0;
  std::vector<float> a(W);
  std::vector<double> T(W);
  while (pos < n) {
    const int nsb = std::min(n - pos, W * lmax_of(W));
    const int L = ((nsb + W - 1) / W) | 1;
    const int SLOTS = (lmax_of(W) + 3) & ~3;
    const float *x = e + pos;
    auto elem = [&](int l, int m) -> float { const int i = l * L + m; return (m < L && i < nsb) ? x[i] : 0.0f; };
    auto run = [&](int l, float start) -> float { float v = start; for (int m = 0; m < SLOTS; m++) { volatile float t = v + elem(l, m); v = t; } return v; };
    st.superblocks++;
    // 1. guesses
    {
      double s = 0.0;
      for (int l = 0; l < W; l++) {
        a[l] = (float)s;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < SLOTS; m++) { volatile float t = g[m & 3] + elem(l, m); g[m & 3] = t; }
        volatile float g01 = g[0] + g[1], g23 = g[2] + g[3]; volatile float gs = g01 + g23;
        s += (double)gs + (l == 0 ? (double)acc : 0.0);
      }
      a[0] = acc;
    }
    // 2. the plain round
    {
      for (int l = 0; l < W; l++) { const float v = run(l, a[l]); T[l] = l == 0 ? (double)v : (double)v - (double)a[l]; }
      double s = 0.0;
      for (int l = 0; l < W; l++) { if (l > 0) a[l] = (float)s; s += T[l]; }
    }
    // 3. table rounds
    std::vector<uint32_t> b(W), val(W); std::vector<std::vector<uint32_t>> en(W, std::vector<uint32_t>(WIN));
    std::vector<Map> H(W), Hn(W);
    float result = 0.f; bool done = false; int t = 0;
    for (; !done; t++) {
      for (int l = 0; l < W; l++) {
        const uint32_t ba = fbits(a[l]);
        b[l] = ba - std::min<uint32_t>(ba, WOFF);
        for (int i = 0; i < WIN; i++) en[l][i] = fbits(run(l, fromb(b[l] + i)));
      }
      for (int l = 0; l < W; l++)
        for (int i = 0; i < WIN; i++) { const uint32_t idx = l + 1 < W ? en[l][i] - b[l + 1] : 0u; H[l].g[i] = idx < (uint32_t)WIN ? (uint8_t)idx : 0xff; }
      // the scan as the device runs it: row_shr 1, 2, 4, 8 inside rows of 16 lanes, then row_bcast15 (rows 1, 3 <- lane 15 of the row before), row_bcast31 (rows 2, 3 <- lane 31)
      for (int d = 1; d < 16 && d < W; d <<= 1) { for (int l = 0; l < W; l++) Hn[l] = (l % 16) >= d ? compose(H[l], H[l - d]) : H[l]; H = Hn; }
      if (W >= 32) { for (int l = 0; l < W; l++) Hn[l] = ((l / 16) & 1) ? compose(H[l], H[(l / 16) * 16 - 1]) : H[l]; H = Hn; }
      if (W >= 64) { for (int l = 0; l < W; l++) Hn[l] = (l >= 32) ? compose(H[l], H[31]) : H[l]; H = Hn; }
      const uint32_t i0 = fbits(acc) - b[0];               // (< WIN by construction: lane 0's guess IS the chain's start)
      int lf = W;
      for (int l = 0; l < W; l++) {
        const uint32_t idx = l == 0 ? i0 : H[l - 1].g[i0];
        val[l] = idx < (uint32_t)WIN ? en[l][idx] : INVALID;
        if (val[l] == INVALID && lf == W) lf = l;
      }
      if (lf == W) { result = fromb(val[W - 1]); done = true; break; }
      if (lf == 0 || t + 1 >= MAXT) {                      // (lane 0 cannot fail) the rest serially, from the exact start of lane lf
        float v = lf == 0 ? acc : fromb(val[lf - 1]);
        for (int i = lf * L; i < nsb; i++) { volatile float tt = v + x[i]; v = tt; }
        result = v; st.bails++; done = true; t++; break;
      }
      const float xf = fromb(val[lf - 1]);                 // the exact start of lane lf
      const double delta = (double)xf - (double)a[lf];
      for (int l = 1; l < W; l++) a[l] = l <= lf ? fromb(val[l - 1]) : (float)((double)a[l] + delta);
    }
    st.table_rounds += t + (done && t == 0 ? 1 : 0); if (t + 1 > st.max_rounds) st.max_rounds = t + 1;
    acc = result;
    pos += nsb;
  }
  return acc;
}

int main(int argc, char **argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long fails = 0;
  for (int W : {16, 32, 64}) {
    Stats st;
    for (int c = 0; c < cases; c++) {
      const int kind = c % 11;
      int n = 1 + (int)(rng() % 2600);
      if (kind == 0) n = 1000;
      if (kind == 9) n = 1 + (int)(rng() % 70);
      std::vector<float> e(n);
      const double scale = std::exp((U(rng) - 0.5) * 60.0);
      for (int i = 0; i < n; i++) {
        double v;
        switch (kind) {
          case 0: case 1: v = scale * (0.2 + U(rng)); break;                               // patch errors of similar size
          case 2: v = scale * std::exp((U(rng) - 0.5) * 20.0); break;                      // wide range
          case 3: v = (rng() % 4 == 0) ? 0.0 : scale * U(rng); break;                      // zeros in between
          case 4: v = (i < (int)(rng() % 200)) ? 0.0 : scale * U(rng); break;              // leading zeros
          case 5: v = std::ldexp((double)(1 + rng() % 7), -3 + (int)(rng() % 3)); break;   // few mantissa bits
          case 6: v = scale * 1e-38 * U(rng); break;                                       // subnormal neighbourhood
          case 7: v = std::exp((U(rng) - 0.5) * 150.0); break;                             // extreme range
          case 8: v = (double)(float)scale; break;                                         // constant
          case 10: { const float f = (float)(scale * (0.2 + U(rng))); const int keep = 12 + (int)(rng() % 5); v = (double)fromb(fbits(f) & ~((1u << (24 - keep)) - 1u)); break; }   // short mantissas: ties on most adds of some binades
          default: v = scale * U(rng); break;
        }
        e[i] = (float)v;
      }
      if (kind == 5 && c % 3 == 0) for (int i = 0; i < n; i++) e[i] = std::ldexp(e[i], 10);
      if (c % 997 == 0 && n > 3) e[rng() % n] = -1.0f;
      if (c % 1499 == 0 && n > 3) e[rng() % n] = NAN;
      if (c % 1999 == 0 && n > 3) e[rng() % n] = INFINITY;
      float acc0 = (c % 4 == 0) ? (float)(scale * 300.0 * U(rng)) : 0.0f;
      const float want = chain_seq(e.data(), n, acc0), got = chain_par(e.data(), n, acc0, W, st);
      if (fbits(want) != fbits(got) && !(want != want && got != got)) {
        if (fails < 10) std::printf("MISMATCH W %d case %d kind %d n %d: want %a got %a\n", W, c, kind, n, want, got);
        fails++;
      }
    }
    std::printf("W %2d: %d cases, super-blocks %ld, table rounds %ld (%.2f per super-block, max %d), bails %ld, serial %ld\n", W, cases, st.superblocks, st.table_rounds,
                (double)st.table_rounds / (double)st.superblocks, st.max_rounds, st.bails, st.serial);
  }
  // the C4 shape: 1 000 similar errors per chain, acc0 = 0
  for (int W : {16, 32, 64}) {
    Stats st;
    for (int c = 0; c < 2000; c++) {
      std::vector<float> e(1000);
      for (auto &v : e) v = (float)(300.0 * (0.05 + U(rng)) * (0.05 + U(rng)));
      const float want = chain_seq(e.data(), 1000, 0.f), got = chain_par(e.data(), 1000, 0.f, W, st);
      if (fbits(want) != fbits(got)) fails++;
    }
    std::printf("C4 shape W %2d: table rounds per chain %.2f (max %d), bails %ld\n", W, (double)st.table_rounds / 2000.0, st.max_rounds, st.bails);
  }
  std::printf("ties met by the serial loops: %ld\n", g_ties);
  std::printf("%s (%ld mismatches)\n", fails ? "FAIL" : "PASS", fails);
  return fails ? 1 : 0;
}
