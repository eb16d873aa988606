// Grouped fp32-MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// One launch executes up to MAXP independent GEMM "problems" (different
// shapes, operand layouts and epilogues) so that a whole dependency level of
// the training step (e.g. dW and dX of one layer) is a single kernel boundary.
// Arithmetic is v_mfma_f32_32x32x2_f32: exact fp32 products and fp32
// accumulation (the 1e-4 ELBO tolerance rules out bf16 INPUTS) -- or, for the
// largest launches, the same products as six exact bf16 piece products per
// fp32 product on operands their producers split once (plane_rounds3 below).
//
// Replaces: snt.nets.MLP MatMul/BiasAdd/Relu nodes (scripts/base.py:47-60,
// 67,135,198), their TF autodiff counterparts (scripts/runners.py:182) and the
// Independent(Bernoulli).log_prob expansion (scripts/base.py:143-146).
//
// Code-generation rules followed here (cdna_hip_programming.md 5.4 rule 20,
// 6 G5): every register array is indexed by compile-time constants only, no
// out-of-line calls, kernel-argument structs never escape by reference, and
// the staging loads are BRANCH-FREE: out-of-range coordinates are clamped to a
// valid address and the value is replaced by 0
// with a select, so edge tiles and K tails run the same code as interior ones.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux.hpp"

namespace gmvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXP = 8;         // problems per launch

enum { EPI_STORE = 0, EPI_BERNOULLI = 1 };

// One GEMM operand as seen by the kernel: a logical [mn][k] matrix.
//   k_contig: element(mn,k) = ptr[(mn/row_div)*ld + k]      (row-major [mn][k])
//   else    : element(mn,k) = ptr[(k /row_div)*ld + mn]     (row-major [k][mn])
// vec_ok : 16-byte (fp32) / 4-byte (uint8) loads of 4 consecutive elements are
//          legal: ld % 4 == 0 and the base is aligned (then every source row
//          holds pad4(extent) elements along the contiguous dimension).
struct Operand {
  const void* ptr;
  int ld;
  int n_mn;
  int row_div;
  unsigned char is_u8, k_contig, pad_, vec_ok;
};

struct Segment {
  Operand a, b;
  int K;
  const float* kscale;   // optional per-k scale applied to b (IWAE row weights)
};

struct Problem {
  int M, N;                // output extents
  int nseg;
  int tiles_m, tiles_n, splits, tile_begin;
  int epi;
  int ldc;
  int relu;                // activation of the epilogue: 0 none, 1 ReLU, 2 tanh, 3 sigmoid, 4 ELU (GMVAE_ACT_* + 1)
  int mask_act;            // whose derivative `mask` (the KEPT activation) stands for: 0 / 1 ReLU (keep where mask > 0), 2.. as above
  int ld_add, add_div;
  int ld_mask;
  int ldx, x_div, nparts;
  float addconst;
  long long split_stride;  // floats between split-K slabs
  float* C;
  int xbf16;               // dW of a uint8 activation (A = x^T, k-major rows of bytes; B = dY fp32 rows): bf16 MFMA path
  int xorder;              // tile order inside the launch (gemm_grouped: 0 split/tn/tm, 1 all tn of a tm on one XCD, 2 all tm of a (split, tn), 3 8 x 8 blocks, 4 all units of a split)
  int planes;              // operands of seg[0] are bf16 plane triples (hi, mid, lo as written by split_planes / the Bernoulli
                           // epilogue's C3): a.ptr / b.ptr name plane 0 (16-bit elements); plane_rounds3 below
                           // planes == 2: f16 PAIRS (plane_rounds2 below): a s = p1 + 2^-11 p2 with the power-of-two scale s of
                           // the tensor; the product is un-scaled by (*a_uns) (*b_uns) uns_c in front of the epilogue
  long long a_pstride, b_pstride;   // 16-bit elements between the planes of a / b
  const float* a_uns;      // pairs: device words holding 1 / s of each operand (written by split_pairs_b16) or nullptr (= 1)
  const float* b_uns;
  float uns_c;             // pairs: constant part of the un-scaling (fixed-scale operands: (sigmoid - x) x 2^15)
  float uns_cb;            // pairs: b's share of uns_c (the bias gradient's column sums are sums of b's pieces)
  float c3_scale;          // != 0: C3 receives f16 pairs of (value x c3_scale) in two planes instead of bf16 triples in three
  int n_padded;            // pairs: b's planes are zero-padded to a multiple of 128 columns (N itself need not be one)
  unsigned short* C3;      // EPI_BERNOULLI: (sigmoid - x) written as planes [3][M][ldc] of 16-bit pieces (beside or instead of C);
                           // EPI_STORE with per-element options (bias / ReLU ...): the stored values also as planes (beside C)
  long long c3_stride;
                           // (always in plane_rounds3's blocked-by-16 layout: element (m, n) at ((n >> 4) M + m) 16 + (n & 15))
  float* colsum_out;       // bias gradient riding on a dW problem: column sums of operand b over the tile's k
                           // range, written by the tiles of the first tile row to colsum_out[split][n]
  const float* bias;
  const float* bias2;      // second per-column constant (ConditionalBernoulli's vector bias_init, scripts/base.py:135) or nullptr
  const float* addsrc;
  const float* mask;       // keep where mask > 0
  const float* rowscale;
  const unsigned char* x;  // Bernoulli targets
  float* part;             // Bernoulli row partial sums [M][nparts]
  Segment seg[2];
};

// hidden_activation_fn in an epilogue (kind = GMVAE_ACT_* + 1; precise forms: these layers are not the ReLU fast paths) and
// its derivative as a function of the KEPT activation h = f(pre): tanh' = 1 - h^2, sigmoid' = h (1 - h), elu' = h > 0 ? 1 : h + 1
__device__ __forceinline__ float act_apply(const float x, const int kind) {
  if (kind <= 1) return fmaxf(x, 0.f);
  if (kind == 2) return tanhf(x);
  if (kind == 3) return 1.f / (1.f + expf(-x));
  return x > 0.f ? x : expm1f(x);
}
__device__ __forceinline__ float act_mask(const float x, const float h, const int kind) {
  if (kind <= 1) return h > 0.f ? x : 0.f;
  if (kind == 2) return x * (1.f - h * h);
  if (kind == 3) return x * (h * (1.f - h));
  return h > 0.f ? x : x * (h + 1.f);
}
struct Launch {
  // header: all a workgroup needs before it knows its problem, adjacent so that ONE scalar load fetches it (every
  // dependent round of kernel-argument loads costs ~0.3 us at a launch's cold start)
  int nprob;
  int total_tiles;         // the FIRST aux_nblocks workgroups run aux_block() (they start at once), the GEMM tiles follow
  int aux_nblocks;         // copy of aux.nblocks
  int aux_last;            // auxiliary workgroups after the tiles instead of before them
  unsigned long long* dbg; // diagnostic: [block][8] wall-clock stamps (100 MHz) or nullptr
  int tile_begin[MAXP];    // copy of p[i].tile_begin
  Problem p[MAXP];
  Aux aux;
};

// BK is sized so that every thread keeps 8 independent loads in flight per
// staging round (the small configurations are latency-, not MFMA-bound).
// NBUF = 1 (single staging buffer, one extra barrier per round) halves the LDS footprint: the small
// configuration mostly runs single-round split-K problems, where residency (4 instead of 2
// workgroups per CU) matters and double buffering buys nothing.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, int NBUF_ = 2>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_, NBUF = NBUF_;
  static constexpr int THREADS = 64 * WM * WN * WK;                  // 256; 512 for the 256 x 128 pair-plane instance (BIG = 3 only)
  static constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static constexpr int LDA = BM + 1, LDB = BN + 1, LDC = BN + 4;
  static constexpr int NSA = BM * BK / 4 / kThreads, NSB = BN * BK / 4 / kThreads;
  static constexpr int OPS = NBUF * (LDA + LDB) * BK;               // floats, NBUF staging buffers
  static constexpr int CST = WK * BM * LDC;                          // floats, C staging
  static constexpr int XBF = (BM == 64 && BN == 64 && BK == 64) ? 4 * 64 * 96 / 2 : 0;     // the bf16 path's 4 images [64][96] x 2 B
  static constexpr int CSP = BM >= 128 ? 8 : 4;                      // column-sum partials per thread (plane_rounds3: 8)
  static constexpr int LDS0 = OPS > CST + CSP * THREADS ? OPS : CST + CSP * THREADS;
  static constexpr int LDS1 = LDS0 > XBF ? LDS0 : XBF;
  static constexpr int P2R = (BM >= 128 && BN == 128) ? 4 * (BM * 16 + 2048) : 0;      // plane_rounds2's ring of 4 (floats: 2 planes x (BM + BN) x 16 k x 2 B per buffer)
  static constexpr int LDS_FLOATS = LDS1 > P2R ? LDS1 : P2R;
  // waves per SIMD the register allocator must leave room for: the small configuration's launches carry more
  // workgroups than 2 per CU (tiles + auxiliary blocks), and a workgroup that starts late ends the launch late
  static constexpr int WAVES_EU = (BM * BN <= 32 * 32 || (NBUF == 1 && BM * BN <= 64 * 64)) ? 3 : (BM * BN >= 128 * 128 ? 2 : 1);
  static_assert(WM * WN * WK == 4 || (BM == 256 && WM * WN * WK == 8), "4 waves per workgroup (8 for 256 x 128)");
  static_assert(NSA >= 1 && NSB >= 1, "tile too small for 256 threads");
  static_assert((BK / WK) % 2 == 0, "k slice per wave must be even");
};
typedef Cfg<32, 32, 128, 1, 1, 4, 1> CfgS;  // latency-bound: 4 waves split K inside the tile, single buffer
typedef Cfg<64, 64, 64, 2, 2, 1> CfgM;
typedef Cfg<64, 64, 64, 2, 2, 1, 1> CfgM1;  // single staging buffer (48 KB with the bf16 images): 3 workgroups per CU, for
                                            // launches whose tiles are one or two rounds long
typedef Cfg<128, 128, 32, 2, 2, 1> CfgL;    // MFMA-bound: 64x64 per wave
typedef Cfg<256, 128, 32, 4, 2, 1> CfgX;    // f16-pair planes only (gemm_grouped<CfgX, 3>): 8 waves of 64x64, one workgroup per CU
// (Single-round variants -- BK = the whole k range of a split, one batch of loads -- were measured and dropped:
// the staging phase of these launches scales with the bytes requested, not with the number of round trips.)

__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// ReLU that keeps a NaN (fmaxf would turn a diverged activation into 0 and hide it from the non-finite-loss check)
__device__ __forceinline__ float relu_nan(const float v) { return !(v <= 0.f) ? v : 0.f; }
// (the bare hardware forms, v_log_f32 / v_exp_f32 and one multiply: __logf / __expf wrap them in denormal-range fix-ups --
//  compare, select, ldexp: ~5 more dependent instructions per call -- that none of these kernels' arguments need: logs are taken
//  of normal numbers (uniforms >= 2^-24, sigmas >= sigma_min, sums >= 1, reciprocals in [0.5, 1]) and an exp whose result would
//  be denormal (probabilities below 1e-38) may flush to 0)
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// softplus(v) and sigmoid(v) of the per-row chains (one-launch, skinny, chain kernels) from ONE exp, ONE rcp and ONE log:
//   v >= -4:  max(v, 0) - log(rcp(1 + e)), e = exp(-|v|).  1 + e and the reciprocal each round at 6e-8 ABSOLUTE, so softplus
//             carries ~1.2e-7 / softplus(v) relative: 7e-6 at v = -4 (softplus = 0.018), less above.
//   v <  -4:  log1p(e) by its alternating series e (1 - e/2 + e^2/3 - e^3/4), e < 0.0184 (first dropped term e^4/5 = 2.3e-8
//             relative).  Nothing rounds against 1: sigma keeps full relative precision down to exp's own range (v > -87).
// Round 5's form took the first branch everywhere: a sigma of 3e-4 (raw = -8: a trained posterior) came out 2e-4 off -- 1 / sigma
// carries that into every gradient behind it (tests/test_saturated.py: 1.1 - 2.7e-4 of the tensor's largest entry) -- and nothing
// was left of a sigma below 6e-8 (raw < -16.6: log(0), NaN).  The series runs beside the rcp / log pair (four FMAs, independent of
// both quarter-rate instructions); the select is the only instruction added to the dependent chain.
__device__ __forceinline__ float softplus_sig(const float v, float& sig) {
  const float e = fexp(-fabsf(v));
  const float r = __builtin_amdgcn_rcpf(1.f + e);
  sig = v >= 0.f ? r : e * r;
  const float series = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.333333343f), -0.5f), 1.f);
  return v < -4.f ? series : fmaxf(v, 0.f) - flog(r);
}
__device__ __forceinline__ float fsoftplus(const float v) { float s; return softplus_sig(v, s); }
// log(1 + S) for S >= 0 without rounding 1 + S when S is small (the series of softplus_sig)
__device__ __forceinline__ float flog1p(const float S) {
  return S < 0.0184f ? S * fmaf(S, fmaf(S, fmaf(S, -0.25f, 0.333333343f), -0.5f), 1.f) : flog(1.f + S);
}
// ---- the categorical head's two row computations (scripts/gmvae.py:240,262-263), in forms that stay
// accurate when ONE class takes (almost) all of the mass.  A trained q(y|x) does: logits of +-40 after 300 steps on clustered
// pixels (tests/test_saturated.py) -- and there the textbook forms, TF's own fp32 kernels included, lose the small quantities
// the gradient is made of against a 1:
//   (1) log pi_k = (lg_k - m) - log(sum_j exp(lg_j - m)).  With one maximum the sum is 1 + S, S the OTHER classes' mass; rounded
//       against the 1 it keeps 6e-8 / S of S, and log pi_top = -S, H = sum pi log pi and the entropy gradient pi (log pi - H) all
//       inherit that (S = 2.5e-5: 2.4e-3 relative).  Here S is summed separately and log(1 + S) is log1p(S); (lg_k - m) is formed
//       before the normaliser is subtracted (m + log1p(S) would absorb S into m's ulp).  Several equal maxima: the plain form.
//   (2) da_k = y_k (dy_k - sum_j y_j dy_j).  With y_top -> 1 the top class's bracket is dy_top - (dy_top + small): cancellation
//       at 6e-8 / (1 - y_top).  Since sum_j y_j = 1 any constant c may be subtracted from dy first:
//       dy_k - dot = (dy_k - c) - sum_j y_j (dy_j - c); with c = dy at the largest y the top class's bracket is minus a sum of
//       small terms, every other class's an O(1) difference times a small y.  O(K), two more row reductions.
// R: the row's reduction policy (max / sum over the lanes that share the row); NJ classes per lane (absent: lg = -inf, y = dy = 0).
struct Sub16 {                     // reduction policy of gemm.hpp's cat_* helpers: the 16 lanes of a row, by shuffles
  static __device__ __forceinline__ float max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
  }
  static __device__ __forceinline__ float sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
};
struct Wave64 {                    // ... the whole wave holds one row
  static __device__ __forceinline__ float max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
  }
  static __device__ __forceinline__ float sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
};
template <class R, int NJ, bool LIBM = false>
__device__ __forceinline__ void cat_log_softmax(const float (&lg)[NJ], float (&lp)[NJ]) {
  float m = lg[0];
#pragma unroll
  for (int j = 1; j < NJ; ++j) m = fmaxf(m, lg[j]);
  m = R::max(m);
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float e = LIBM ? expf(lg[j] - m) : fexp(lg[j] - m);
    sa += e;
    sb += lg[j] == m ? 0.f : e;
  }
  sa = R::sum(sa); sb = R::sum(sb);
  const bool one = sa - sb < 1.5f;                 // ONE class holds the maximum
  const float l = LIBM ? (one ? log1pf(sb) : logf(sa)) : (one ? flog1p(sb) : flog(sa));
#pragma unroll
  for (int j = 0; j < NJ; ++j) lp[j] = (lg[j] - m) - l;
}
template <class R, int NJ>
__device__ __forceinline__ void cat_softmax_bwd(const float (&y)[NJ], const float (&dy)[NJ], float (&da)[NJ]) {
  float ym = y[0];
#pragma unroll
  for (int j = 1; j < NJ; ++j) ym = fmaxf(ym, y[j]);
  ym = R::max(ym);
  float c = -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; ++j) c = fmaxf(c, y[j] == ym ? dy[j] : -INFINITY);
  c = R::max(c);                                   // dy at (one of) the largest y
  float dc[NJ], dot = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) { dc[j] = dy[j] - c; dot += y[j] * dc[j]; }
  dot = R::sum(dot);
#pragma unroll
  for (int j = 0; j < NJ; ++j) da[j] = y[j] * (dc[j] - dot);
}
// The general schedule's z heads (51,200 x 64 elements per evaluation pass: z_head_fwd was 33 us of libm softplus / log /
// IEEE division, 17.5 us with these): the hardware forms where they are exact enough and the denormal-safe ones where they are
// not, chosen per element -- softplus through the compensated log1p, log(1 + e) e / ((1 + e) - 1) (a few ulp for every e in
// [0, 1]; the plain log(1 + e) of the fast paths is 2e-4 off at sigma = 2e-4), __expf instead of the bare exp where the bare
// one would flush (|v| > 80), logf instead of the bare log below the normal range.
__device__ __forceinline__ float softplus_r(const float v) {
  const float av = fabsf(v);
  const float e = av > 80.f ? __expf(-av) : fexp(-av);
  const float u = 1.f + e, d = u - 1.f;
  const float l1p = d == 0.f ? e : flog(u) * (e * __builtin_amdgcn_rcpf(d));
  return fmaxf(v, 0.f) + l1p;
}
__device__ __forceinline__ float log_r(const float x) { return x < 1.1754944e-38f ? logf(x) : flog(x); }
__device__ __forceinline__ float sigmoid_r(const float v) {
  const float av = fabsf(v);
  const float e = av > 80.f ? __expf(-av) : fexp(-av);
  const float r = __builtin_amdgcn_rcpf(1.f + e);
  return v >= 0.f ? r : e * r;
}
// (the other row kernels keep the denormal-safe libm forms: a sigma below 1e-38 is still a number there)
__device__ __forceinline__ float sigmoidf_(float v) {
  float e = __expf(-fabsf(v));
  float r = 1.0f / (1.0f + e);
  return v >= 0.f ? r : e * r;
}
__device__ __forceinline__ float softplusf_(float v) { return fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v))); }

// ---- global -> register staging ------------------------------------------
// A slot is 4 consecutive elements along the source's contiguous dimension.
// Thread -> slot map (Q = slots per line of the contiguous dimension): every 32
// consecutive slot ids cover 4 lines x 8 slots, so a half-wave touches 4 full
// 128-byte lines in memory AND 32 distinct LDS banks when the [k][mn] image has
// an odd leading dimension -- for k-contiguous and mn-contiguous sources alike.
//
// KIND bits: 1 = uint8 elements, 2 = mn-contiguous (else k-contiguous),
//            4 = element-wise loads (source not 16-byte vectorisable).
template <int KIND, int BMN, int BK, int NS>
__device__ __forceinline__ void op_load(const void* __restrict__ base, const int ld, const int n_mn,
                                        const int row_div, const float* __restrict__ kscale, const int K,
                                        const int mn0, const int k0, const int kend, const int tid, float4 (&r)[NS]) {
  constexpr bool U8 = (KIND & 1) != 0, MC = (KIND & 2) != 0, SC = (KIND & 4) != 0;
  const unsigned char* b8 = static_cast<const unsigned char*>(base);
  const float* b32 = static_cast<const float*>(base);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = tid + i * kThreads;
    float v[4];
    if (!MC) {
      constexpr int Q = BK / 4;
      const int mn = mn0 + (s & 3) + 4 * (s / (4 * Q));
      const int k = k0 + (((s >> 2) % Q) << 2);
      const int mnc = min(mn, n_mn - 1);
      const int row = (row_div == 1) ? mnc : mnc / row_div;
      const uint32_t rb = (uint32_t)row * (uint32_t)ld;
      if (!SC) {
        const uint32_t e = rb + (uint32_t)min(k, ((K + 3) & ~3) - 4);
        if (U8) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(b8 + e);
          v[0] = (float)(w & 0xff); v[1] = (float)((w >> 8) & 0xff); v[2] = (float)((w >> 16) & 0xff); v[3] = (float)(w >> 24);
        } else {
          const float4 q = *reinterpret_cast<const float4*>(b32 + e);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t e = rb + (uint32_t)min(k + j, K - 1);
          v[j] = U8 ? (float)b8[e] : b32[e];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = (k + j < kend) ? v[j] : 0.f;
        if (kscale) x *= kscale[min(k + j, K - 1)];
        v[j] = x;
      }
    } else {
      constexpr int Q = BMN / 4;
      const int k = k0 + (s & 3) + 4 * (s / (4 * Q));
      const int mn = mn0 + (((s >> 2) % Q) << 2);
      const int kc = min(k, K - 1);
      const int row = (row_div == 1) ? kc : kc / row_div;
      const uint32_t rb = (uint32_t)row * (uint32_t)ld;
      if (!SC) {
        const uint32_t e = rb + (uint32_t)min(mn, ((n_mn + 3) & ~3) - 4);
        if (U8) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(b8 + e);
          v[0] = (float)(w & 0xff); v[1] = (float)((w >> 8) & 0xff); v[2] = (float)((w >> 16) & 0xff); v[3] = (float)(w >> 24);
        } else {
          const float4 q = *reinterpret_cast<const float4*>(b32 + e);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t e = rb + (uint32_t)min(mn + j, n_mn - 1);
          v[j] = U8 ? (float)b8[e] : b32[e];
        }
      }
      const bool kin = k < kend;
      const float sc = kscale ? kscale[kc] : 1.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // columns/rows beyond the extent only reach unstored outputs (clamped loads: finite values)
        v[j] = kin ? v[j] * sc : 0.f;
      }
    }
    r[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// Lean variant for the common case -- tile completely inside the operand, chunk completely inside
// [kb, ke), 16-byte loads legal, no row broadcast / k scale: no clamps, no selects, shift-only
// slot arithmetic.  (SQ counters showed the grouped launches VALU-issue-bound on the predicated loader:
// ~580 VALU instructions per wave around 16 MFMAs, profiles/round1_pmc_sq_per_kernel.txt.)
// (A variant that also applied the per-k scale of the IWAE row weights here, so that the decoder's dW problem could
// take this path, made EVERY large-tile launch 6-16 % slower on the same box -- the staging code of this kernel is
// VALU-issue-sensitive -- and was dropped: A/B in profiles/round2_notes.md.)
template <int KIND, int BMN, int BK, int NS>
__device__ __forceinline__ void op_load_fast(const void* __restrict__ base, const int ld, const int mn0, const int k0,
                                             const int tid, float4 (&r)[NS]) {
  constexpr bool U8 = (KIND & 1) != 0, MC = (KIND & 2) != 0;
  const unsigned char* b8 = static_cast<const unsigned char*>(base);
  const float* b32 = static_cast<const float*>(base);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = tid + i * kThreads;
    uint32_t e;
    if (!MC) {
      constexpr int Q = BK / 4;
      e = (uint32_t)(mn0 + (s & 3) + 4 * (s / (4 * Q))) * (uint32_t)ld + (uint32_t)(k0 + (((s >> 2) % Q) << 2));
    } else {
      constexpr int Q = BMN / 4;
      e = (uint32_t)(k0 + (s & 3) + 4 * (s / (4 * Q))) * (uint32_t)ld + (uint32_t)(mn0 + (((s >> 2) % Q) << 2));
    }
    if (U8) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(b8 + e);
      r[i] = make_float4((float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24));
    } else {
      r[i] = *reinterpret_cast<const float4*>(b32 + e);
    }
  }
}

// LDS image of an operand tile is always [k][mn] with an odd leading dimension.
template <int BMN, int BK, int LD, int NS>
__device__ __forceinline__ void op_store(float* __restrict__ T, const bool mc, const int tid, const float4 (&r)[NS]) {
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = tid + i * kThreads;
    if (!mc) {
      constexpr int Q = BK / 4;
      const int mn = (s & 3) + 4 * (s / (4 * Q)), k = ((s >> 2) % Q) << 2;
      float* p = T + k * LD + mn;
      p[0] = r[i].x; p[LD] = r[i].y; p[2 * LD] = r[i].z; p[3 * LD] = r[i].w;
    } else {
      constexpr int Q = BMN / 4;
      const int k = (s & 3) + 4 * (s / (4 * Q)), mn = ((s >> 2) % Q) << 2;
      float* p = T + k * LD + mn;
      p[0] = r[i].x; p[1] = r[i].y; p[2] = r[i].z; p[3] = r[i].w;
    }
  }
}

__device__ __forceinline__ int op_kind(const unsigned char is_u8, const unsigned char k_contig,
                                       const unsigned char vec_ok) {
  return (is_u8 ? 1 : 0) | (k_contig ? 0 : 2) | (vec_ok ? 0 : 4);
}

// ---- uint8 x fp32 products on the bf16 matrix cores --------------------------------------------------------------
// dW = X^T dY with X uint8 (exact in bf16: 8 significant bits) and dY fp32 written as hi + mid + lo, three bf16
// pieces that reproduce its 24-bit significand exactly (truncation splits; the residuals are exact in fp32).  Every
// product is then exact and the fp32 accumulation is the only rounding, as on the fp32 MFMA path -- at 3 MFMAs of
// K = 16 per 16 k instead of 8 of K = 2 (v_mfma_f32_32x32x16_bf16 runs 16x the fp32 rate).  Both operands are
// k-major in memory (rows = batch index), so their LDS images stay [k][column] (coalesced 8-byte writes) and the
// fragments come through ds_read_b64_tr_b16, gfx950's transposing LDS read.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr int kXLD = 96;     // 16-bit elements per image row: 192 B = 48 banks, so the 4 rows of a transposed read tile the 64 banks

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned short* p) {     // rows +0..3 and +4..7 of this lane's 8 k
  typedef __attribute__((address_space(3))) s16x4_t* lp;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * kXLD));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// Two fp32 values -> their three 16-bit pieces, packed pairwise (element 0 in the low half): round-to-nearest-even pieces
// (v_cvt_pk_bf16_f32: one instruction per pair) with exact residuals -- v - hi is a multiple of ulp(v) below 2^-8 |v|, so
// it has <= 16 significant bits; the second residual <= 8, which a bf16 holds -- so hi + mid + lo == v bit for bit, as
// with the truncation splits above at 5.5 instead of 14 instructions per element.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(const float v0, const float v1, unsigned& hi, unsigned& mi, unsigned& lo) {
  const f32x2_t v = {v0, v1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t r1 = {v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xffff0000u)};
  mi = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  const f32x2_t r2 = {r1.x - __uint_as_float(mi << 16), r1.y - __uint_as_float(mi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}


// ---- operands split ONCE by their producer ("planes") ------------------------------------------------------------
// fp32 x fp32 products on the bf16 matrix cores: both operands are written as hi + mid + lo, three 16-bit pieces that
// reproduce the 24-bit significand EXACTLY (split_pair above); a*b = sum of nine piece products, each exact in fp32; the
// three smallest (mid*lo, lo*mid, lo*lo <= 2^-23 |a b|, below the rounding of the fp32 product itself) are dropped, so a
// k-step of 16 costs 6 v_mfma_f32_32x32x16_bf16 (192 cycles) instead of 8 v_mfma_f32_32x32x2_f32 (512 cycles) and the
// accumulation stays fp32.  The split happens ONCE, outside the GEMM: the three pieces of every operand element lie in
// memory as three 16-bit planes (written by split_planes_b16 below, by a producing GEMM's C3 epilogue or by the Bernoulli
// epilogue).  (Measured and removed in round 4: converting per tile inside the loop -- 1.1 - 1.3x SLOWER than the fp32 MFMA
// loop -- and a first plane loop over row-major planes staged through 48 registers -- 8 - 25 % slower than the one below;
// profiles/round3_notes.md.)
// The per-k scale of an IWAE weight gradient cannot ride on the pieces: the producer writes the planes of the SCALED
// activation (split_planes_b16's rowscale) and `kscale` only weighs the bias gradient's column sums here.

// ---- plane rounds: LDS-DMA into a ring of three 16-deep buffers ------------------------------------------------------
// (The removed register-staged loop, timing experiments with results discarded: without its global loads 17-22 % shorter,
// without its LDS stores 10 %, without both 25-34 % -- and what was left still ran the matrix pipes at ~65 %: 48 staging
// registers left no room to read the next k-step's fragments behind the current MFMAs.)  Here nothing is staged through registers:
//  * the planes lie in memory BLOCKED by 16 along their contiguous dimension ("B16": element (r, c) of a [Rows][Cols] matrix at
//    ((c >> 4) Rows + r) 16 + (c & 15)), so a [128 mn] x [16 k] tile of a k-contiguous use is ONE contiguous 4 KB piece and
//    a [16 k] x [128 mn] tile of an mn-contiguous use is eight contiguous 512-byte pieces: every wave-instruction of the
//    staging moves whole 128-byte lines in both orientations;
//  * global_load_lds_dwordx4 writes them straight into LDS (1 KB per wave-instruction, 6 per wave and round), three rounds
//    deep: round c + 3 is requested when round c's buffer has been read by every wave, and waited for two rounds later;
//  * the LDS position of a chunk inside its 1 KB piece is chosen per lane (the source address is per lane, the
//    destination is lane order): a k-contiguous image [128][16] swaps the two 16-byte halves of a row in every other
//    group of 8 rows, an mn-contiguous one [8 column blocks][16 k][16 mn] rotates the k rows of odd blocks by 4 -- both
//    make the fragment reads (ds_read_b128 / ds_read_b64_tr_b16) bank-conflict-free without padding;
//  * two sets of fragment registers: the next round's 12 fragments are read behind the round's barrier in the shadow of
//    its last 12 MFMAs (sched_group_barrier: 2 ds_read_b128 or 3 ds_read_b64_tr_b16 per MFMA; the reads are
//    UNCONDITIONAL so that they and the MFMAs stay one basic block -- under a uniform `if` the transposing forms ran
//    1.6x slower).  One barrier per 24 MFMAs, LDS counter only.
// Timing experiments on this loop (results discarded; 25600x512x3072 / 512x3072x25600, us): whole loop 404 / 429, without
// the DMAs 347 / 331, with neither DMAs nor fragment reads nor barriers 307 / 306 -- the MFMA stream, the tile prologue /
// epilogue and the tail of a launch whose tiles are no multiple of the 512 resident workgroups (800 tiles: 1.56 waves).
constexpr int kP3Op = 3 * 2048;                 // 16-bit elements per operand image: 3 planes of 128 x 16
constexpr int kP3Buf = 2 * kP3Op;               // A | B
constexpr int kP3Ring = 3;

template <bool MC>
__device__ __forceinline__ uint32_t p3_src(const int wave, const int lane, const uint32_t rows, const int mn0, const int k0) {
  const int p = 64 * wave + lane;               // this lane's 16-byte unit of the plane image (lane order = LDS order)
  if (!MC) {
    const int row = p >> 1, kh = (p & 1) ^ ((row >> 3) & 1);
    return ((uint32_t)(k0 >> 4) * rows + (uint32_t)(mn0 + row)) * 16u + 8u * kh;
  } else {
    const int blk = p >> 5, kr = (p >> 1) & 15, half = p & 1, k = (kr - 4 * (blk & 1)) & 15;
    return ((uint32_t)((mn0 >> 4) + blk) * rows + (uint32_t)(k0 + k)) * 16u + 8u * half;
  }
}

template <bool MC>
__device__ __forceinline__ bf16x8_t p3_frag(const unsigned short* __restrict__ op, const int base0, const int base1) {
  if (!MC) {
    return *reinterpret_cast<const bf16x8_t*>(op + base0);
  } else {
    typedef __attribute__((address_space(3))) s16x4_t* lp;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(op + base0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(op + base1));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

template <bool AMC, bool BMC>
__device__ __forceinline__ void plane_rounds3(unsigned short* __restrict__ img, const unsigned short* __restrict__ A,
                                              const uint32_t a_rows, const long long a_ps, const unsigned short* __restrict__ Bp,
                                              const uint32_t b_rows, const long long b_ps, const float* __restrict__ kscale,
                                              const bool do_cs, const int m0, const int n0, const int kb, const int NC16,
                                              const int tid, const int lane, const int wave, const int wm0, const int wn0,
                                              f32x16 (&acc)[2][2], float (&cs8)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // staging: source element offsets of this lane's unit (round 0) and their advance per round
  uint32_t ea = p3_src<AMC>(wave, lane, a_rows, m0, kb), eb = p3_src<BMC>(wave, lane, b_rows, n0, kb);
  const uint32_t a_round = AMC ? 256u : 16u * a_rows, b_round = BMC ? 256u : 16u * b_rows;
  // fragment reads: lane bases inside an operand image (+ 512 i + 2048 plane as immediates)
  int fa0, fa1, fb0, fb1;
  {
    const int l31 = lane & 31, g16 = lane >> 4, i16 = lane & 15;
    const int kc = (l31)*16 + 8 * ((lane >> 5) ^ ((l31 >> 3) & 1));
    const int kr0 = 8 * (g16 >> 1) + (i16 >> 2), rot = 4 * (g16 & 1);
    const int mc0 = (g16 & 1) * 256 + ((kr0 + rot) & 15) * 16 + 4 * (i16 & 3), mc1 = (g16 & 1) * 256 + ((kr0 + 4 + rot) & 15) * 16 + 4 * (i16 & 3);
    fa0 = AMC ? (wm0 >> 4) * 256 + mc0 : wm0 * 16 + kc;
    fa1 = AMC ? (wm0 >> 4) * 256 + mc1 : 0;
    fb0 = BMC ? (wn0 >> 4) * 256 + mc0 : wn0 * 16 + kc;
    fb1 = BMC ? (wn0 >> 4) * 256 + mc1 : 0;
  }
  // column sums (bias gradient; b mn-contiguous): thread (k = tid >> 4, 8 columns 8 (tid & 15)..) reads its chunk back from LDS
  const int cs_k = tid >> 4, cs_ch = tid & 15;
  const int cs_off = (cs_ch >> 1) * 256 + ((cs_k + 4 * ((cs_ch >> 1) & 1)) & 15) * 16 + 8 * (cs_ch & 1);
  int kk = kb + cs_k;
#define GMVAE_P3_DMA(buf_)                                                                                 \
  {                                                                                                        \
    unsigned short* const d_ = img + (buf_) * kP3Buf + wave * 512;                                         \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                     \
      __builtin_amdgcn_global_load_lds(A + pl * a_ps + ea, d_ + pl * 2048, 16, 0, 0);                      \
      __builtin_amdgcn_global_load_lds(Bp + pl * b_ps + eb, d_ + kP3Op + pl * 2048, 16, 0, 0);             \
    }                                                                                                      \
    ea += a_round; eb += b_round;                                                                          \
  }
#define GMVAE_P3_FRAGS(FA_, FB_, buf_)                                                                     \
  {                                                                                                        \
    const unsigned short* const ia_ = img + (buf_) * kP3Buf;                                               \
    const unsigned short* const ib_ = ia_ + kP3Op;                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                   \
        FA_[i][pl] = p3_frag<AMC>(ia_ + pl * 2048 + i * 512, fa0, fa1);                                    \
        FB_[i][pl] = p3_frag<BMC>(ib_ + pl * 2048 + i * 512, fb0, fb1);                                    \
      }                                                                                                    \
  }
#define GMVAE_P3_TILE(FA_, FB_, i_, j_)                                                                    \
  {                                                                                                        \
    f32x16 c_ = acc[i_][j_];                                                                               \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][0], FB_[j_][2], c_, 0, 0, 0);                     \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][2], FB_[j_][0], c_, 0, 0, 0);                     \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][1], FB_[j_][1], c_, 0, 0, 0);                     \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][0], FB_[j_][1], c_, 0, 0, 0);                     \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][1], FB_[j_][0], c_, 0, 0, 0);                     \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i_][0], FB_[j_][0], c_, 0, 0, 0);                     \
    acc[i_][j_] = c_;                                                                                      \
  }
  // LDS read instructions behind the first of a round's last 12 MFMAs -- 12 ds_read_b128 (both k-contiguous): 2 behind each of
  // 6; 6 + 12 ds_read_b64_tr_b16: 3 behind each of 6; 24 transposing: 3 behind each of 8 (the rates MI355X_MICROARCH.md gives
  // as free beside an MFMA) -- so that the last reads have 4-6 MFMAs to land before the next round multiplies them
  constexpr int kRdA = (AMC || BMC) ? 3 : 2, kRdB = (AMC && BMC) ? 3 : 0;
#define GMVAE_P3_SG(n_) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, n_, 0);
  // round c_: its fragments are in CUR; buffer bc_ = c_ % 3 holds its images, bn_ = (c_ + 1) % 3 the next round's
#define GMVAE_P3_ROUND(CA_, CB_, NA_, NB_, c_, bc_, bn_)                                                   \
  {                                                                                                        \
    if (BMC && do_cs) {                                                                                    \
      const unsigned short* const ib_ = img + (bc_) * kP3Buf + kP3Op + cs_off;                             \
      const u32x4 h_ = *reinterpret_cast<const u32x4*>(ib_), m_ = *reinterpret_cast<const u32x4*>(ib_ + 2048),   \
                  l_ = *reinterpret_cast<const u32x4*>(ib_ + 4096);                                        \
      const float sc = kscale ? kscale[kk] : 1.f;                                                          \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
        cs8[2 * q] += sc * ((__uint_as_float(l_[q] << 16) + __uint_as_float(m_[q] << 16)) + __uint_as_float(h_[q] << 16));                            \
        cs8[2 * q + 1] += sc * ((__uint_as_float(l_[q] & 0xffff0000u) + __uint_as_float(m_[q] & 0xffff0000u)) + __uint_as_float(h_[q] & 0xffff0000u)); \
      }                                                                                                    \
      kk += 16;                                                                                            \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    GMVAE_P3_TILE(CA_, CB_, 0, 0)                                                                          \
    GMVAE_P3_TILE(CA_, CB_, 0, 1)                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((c_) + 2 < NC16) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          \
    if ((c_) + 3 < NC16) GMVAE_P3_DMA(bc_)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    /* the next round's fragments, one (k-contiguous) or two (transposing) LDS reads behind each of the last 12 MFMAs; */ \
    /* in the last round they read a buffer nobody uses (unconditional: the reads and the MFMAs stay one block) */        \
    GMVAE_P3_FRAGS(NA_, NB_, bn_)                                                                          \
    GMVAE_P3_TILE(CA_, CB_, 1, 0)                                                                          \
    GMVAE_P3_TILE(CA_, CB_, 1, 1)                                                                          \
    GMVAE_P3_SG(kRdA) GMVAE_P3_SG(kRdA) GMVAE_P3_SG(kRdA) GMVAE_P3_SG(kRdA) GMVAE_P3_SG(kRdA) GMVAE_P3_SG(kRdA)            \
    GMVAE_P3_SG(kRdB) GMVAE_P3_SG(kRdB) GMVAE_P3_SG(0) GMVAE_P3_SG(0) GMVAE_P3_SG(0) GMVAE_P3_SG(0)                        \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  }
  bf16x8_t fa[2][3], fb[2][3], ga[2][3], gb[2][3];
  GMVAE_P3_DMA(0)
  GMVAE_P3_DMA(1)                                // (NC16 >= 2)
  // (the accumulators start here, inside the orientation's own copy of the loop -- see plane_rounds2)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (NC16 > 2) {
    GMVAE_P3_DMA(2)
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");    // round 0 has landed, two rounds stay in flight
  } else {
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  }
  GMVAE_P3_FRAGS(fa, fb, 0)
  int bc = 0;                                    // ring position of round c (runtime: three buffers, two fragment sets)
#pragma unroll 1
  for (int c = 0; c < NC16; c += 2) {            // (NC16 is even: k ranges are whole 32-deep rounds)
    const int b1 = bc == 2 ? 0 : bc + 1, b2 = b1 == 2 ? 0 : b1 + 1;
    GMVAE_P3_ROUND(fa, fb, ga, gb, c, bc, b1)
    GMVAE_P3_ROUND(ga, gb, fa, fb, c + 1, b1, b2)
    bc = b2;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the caller reuses LDS)
#undef GMVAE_P3_DMA
#undef GMVAE_P3_FRAGS
#undef GMVAE_P3_TILE
#undef GMVAE_P3_ROUND
#undef GMVAE_P3_SG
}


// ---- f16 pairs: fp32 x fp32 products as THREE piece products ------------------------------------------------------
// (round 5.)  The bf16 triples above make every product exact at six matrix instructions per 16 k.  Two f16 pieces carry 22
// of the 24 significand bits: with s the tensor's power-of-two scale (its largest magnitude lands in [2^14, 2^15): exact),
//     a s = p1 + 2^-11 p2 (1 + d),   p1 = f16(a s),   p2 = f16((a s - p1) 2^11),   |d| <= 2^-11
// -- the residual a s - p1 is exact in fp32 and is stored SHIFTED by 2^11, so that it has p1's exponent range and never
// lands among f16's subnormals: an element keeps 22 bits as long as |a| >= 2^-29 max|a| (p1 normal); below that it loses
// them gradually (absolute error <= 2^-40 max|a|).  A product then is
//     a b s t = p1 q1 + 2^-11 (p1 q2 + p2 q1) + O(2^-22 |a b s t|)
// three v_mfma_f32_32x32x16_f16 per 16 k; the main term and the cross terms accumulate in fp32 registers of their own
// (acc, accx) and meet once per tile: (acc + 2^-11 accx) / (s t).  Per product the relative error is <= 3 x 2^-22 (the two
// dropped residuals and p2 q2) and unbiased; the fp32 accumulation over k that follows rounds at 2^-24 per ADD of the running
// sum, which is what dominates the result in both forms.  Half the matrix instructions and two thirds of the operand bytes
// of the triples: the config-5 GEMMs are bound by the L2 -> LDS stream of their pieces.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// the scale of a tensor from the bits of its largest magnitude (uint order = float order for non-negative floats): 2^(14 - e)
__device__ __forceinline__ float pair_scale(const unsigned amax_bits) {
  int eb = (int)((amax_bits >> 23) & 0xffu);
  if (eb == 0xff) eb = 127 + 14;                   // inf / NaN somewhere: scale 1 (they propagate through p1)
  int sb = 127 + 14 - (eb - 127);
  sb = sb < 2 ? 2 : (sb > 252 ? 252 : sb);         // (1 / s stays a normal float)
  return __uint_as_float((unsigned)sb << 23);
}
// A tensor's largest magnitude (bits): every wave of the producing launch (or workgroup of amax_abs) leaves the maximum of
// what it wrote in a word of its own -- no atomics (thousands of waves adding to ONE address with device-scope atomics
// serialise at the memory side: a 13 M element reduction took 85 us that way; even a look-before-add cost the producer 15 us),
// nothing to zero -- and amax_final below folds the words into one and derives the scale.
__device__ __forceinline__ unsigned wave_umax(unsigned m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  return m;
}
// two values (already scaled) -> their packed first and second pieces (element 0 in the low half)
__device__ __forceinline__ void split_f16pair(const float v0, const float v1, unsigned& p1, unsigned& p2) {
  const f32x2_t v = {v0, v1};
  const f16x2_t h1 = __builtin_convertvector(v, f16x2_t);
  const f32x2_t r = {(v0 - (float)h1.x) * 2048.f, (v1 - (float)h1.y) * 2048.f};
  const f16x2_t h2 = __builtin_convertvector(r, f16x2_t);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
}
__device__ __forceinline__ float f16lo(const unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w).x; }
__device__ __forceinline__ float f16hi(const unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w).y; }

// The loop: plane_rounds3's layout, staging and fragment reads with two planes per operand, for workgroup tiles of BM x 128,
// BM = 128 (4 waves, two workgroups per CU) or 256 (8 waves, one workgroup per CU: every wave still multiplies 64 x 64, but a
// round's images are 24 KB for twice the products -- the loop is bound by the stream of pieces into LDS, measured below).  A ring
// of FOUR buffers of (BM + 128) x 16 k x 2 planes x 2 B (16 / 24 KB): round c + 4 is requested behind round c's barrier and
// waited for three rounds later (rings of 3 and 5 buffers ran the same / 2 % slower); per wave and round 12 matrix instructions,
// 4 / 3 LDS-DMA instructions and 8 (k-contiguous) / 16 (transposing) fragment reads.
// Timing experiments at BM = 128 (results discarded; 25600x512x3072 NT / 512x3072x25600 TN, us): whole loop 267 / 292; without
// its matrix instructions 185 / 216; with neither those nor the fragment reads 194 / 218 -- the DMA stream alone, 12.9 TB/s
// chip-wide, MI355X_MICROARCH.md's rate for an L2 / fabric mix like this one's; without the DMAs 147 / 168.  The two streams
// overlap only in part (both slow down with the clock the matrix pipes throttle to: 2.05 GHz in the step), so fewer bytes per
// product is what shortens the loop: the 256-row tile.
constexpr int kP2Ring = 4;
#ifndef GMVAE_P2_MFMA16
#define GMVAE_P2_MFMA16 1      // the pairs' loop on v_mfma_f32_16x16x32_f16 (plane_rounds2h); 0: the 32 x 32 x 16 loop (plane_rounds2)
#endif
template <int BM> struct P2 {
  static constexpr int PA = BM * 16;            // 16-bit elements per plane image of a (BM x 16 k); b's: 2048
  static constexpr int OpA = 2 * PA, Buf = OpA + 4096;
  static constexpr int NW = BM / 32;            // waves
  static constexpr int DPW = NW == 4 ? 4 : 3;   // LDS-DMA instructions per wave and round
};

template <bool AMC, bool BMC, int BM>
__device__ __forceinline__ void plane_rounds2(unsigned short* __restrict__ img, const unsigned short* __restrict__ A,
                                              const uint32_t a_rows, const long long a_ps, const unsigned short* __restrict__ Bp,
                                              const uint32_t b_rows, const long long b_ps, const float* __restrict__ kscale,
                                              const bool do_cs, const int m0, const int n0, const int kb, const int NC16,
                                              const int tid, const int lane, const int wave, const int wm0, const int wn0,
                                              f32x16 (&acc)[2][2], f32x16 (&accx)[2][2], float (&cs8)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef P2<BM> G;
  // staging: this wave's pieces (1 KB each).  BM = 128: piece `wave` of both planes of a and of b.  BM = 256: piece `wave` of both
  // planes of a (8 pieces per plane), and ONE of b's 8 pieces (plane wave >> 2, piece wave & 3)
  uint32_t ea = p3_src<AMC>(wave, lane, a_rows, m0, kb), eb = p3_src<BMC>(wave & 3, lane, b_rows, n0, kb);
  const uint32_t a_round = AMC ? 256u : 16u * a_rows, b_round = BMC ? 256u : 16u * b_rows;
  const unsigned short* const Bw = Bp + (G::NW == 8 ? (wave >> 2) * b_ps : 0);
  const int b_dst = G::OpA + (G::NW == 8 ? (wave >> 2) * 2048 : 0) + (wave & 3) * 512;
  int fa0, fa1, fb0, fb1;
  {
    const int l31 = lane & 31, g16 = lane >> 4, i16 = lane & 15;
    const int kc = (l31)*16 + 8 * ((lane >> 5) ^ ((l31 >> 3) & 1));
    const int kr0 = 8 * (g16 >> 1) + (i16 >> 2), rot = 4 * (g16 & 1);
    const int mc0 = (g16 & 1) * 256 + ((kr0 + rot) & 15) * 16 + 4 * (i16 & 3), mc1 = (g16 & 1) * 256 + ((kr0 + 4 + rot) & 15) * 16 + 4 * (i16 & 3);
    fa0 = AMC ? (wm0 >> 4) * 256 + mc0 : wm0 * 16 + kc;
    fa1 = AMC ? (wm0 >> 4) * 256 + mc1 : 0;
    fb0 = BMC ? (wn0 >> 4) * 256 + mc0 : wn0 * 16 + kc;
    fb1 = BMC ? (wn0 >> 4) * 256 + mc1 : 0;
  }
  // column sums (bias gradient; b mn-contiguous): thread (k = tid >> 4, 8 columns 8 (tid & 15)..) of the first 256 reads its chunk back
  const bool cs_on = BMC && do_cs && tid < 256;
  const int cs_k = (tid >> 4) & 15, cs_ch = tid & 15;
  const int cs_off = (cs_ch >> 1) * 256 + ((cs_k + 4 * ((cs_ch >> 1) & 1)) & 15) * 16 + 8 * (cs_ch & 1);
  int kk = kb + cs_k;
#define GMVAE_P2_DMA(buf_)                                                                                 \
  {                                                                                                        \
    unsigned short* const d_ = img + (buf_) * G::Buf;                                                      \
    __builtin_amdgcn_global_load_lds(A + ea, d_ + wave * 512, 16, 0, 0);                                   \
    __builtin_amdgcn_global_load_lds(A + a_ps + ea, d_ + G::PA + wave * 512, 16, 0, 0);                    \
    __builtin_amdgcn_global_load_lds(Bw + eb, d_ + b_dst, 16, 0, 0);                                       \
    if (G::NW == 4) __builtin_amdgcn_global_load_lds(Bw + b_ps + eb, d_ + b_dst + 2048, 16, 0, 0);         \
    ea += a_round; eb += b_round;                                                                          \
  }
  // (order: what the next round multiplies first -- row block 0 of a, both column blocks of b -- is read first)
#define GMVAE_P2_FRAGS(FA_, FB_, buf_)                                                                     \
  {                                                                                                        \
    const unsigned short* const ia_ = img + (buf_) * G::Buf;                                               \
    const unsigned short* const ib_ = ia_ + G::OpA;                                                        \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) FA_[0][pl] = __builtin_bit_cast(f16x8_t, p3_frag<AMC>(ia_ + pl * G::PA, fa0, fa1));        \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) FB_[0][pl] = __builtin_bit_cast(f16x8_t, p3_frag<BMC>(ib_ + pl * 2048, fb0, fb1));         \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) FB_[1][pl] = __builtin_bit_cast(f16x8_t, p3_frag<BMC>(ib_ + pl * 2048 + 512, fb0, fb1));   \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) FA_[1][pl] = __builtin_bit_cast(f16x8_t, p3_frag<AMC>(ia_ + pl * G::PA + 512, fa0, fa1));  \
  }
  // row block i_ against both column blocks: six instructions, consecutive ones on different accumulators
#define GMVAE_P2_TILES(FA_, FB_, i_)                                                                       \
  {                                                                                                        \
    f32x16 m0_ = acc[i_][0], m1_ = acc[i_][1], x0_ = accx[i_][0], x1_ = accx[i_][1];                       \
    m0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[0][0], m0_, 0, 0, 0);                     \
    m1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[1][0], m1_, 0, 0, 0);                     \
    x0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[0][1], x0_, 0, 0, 0);                     \
    x1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[1][1], x1_, 0, 0, 0);                     \
    x0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][1], FB_[0][0], x0_, 0, 0, 0);                     \
    x1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][1], FB_[1][0], x1_, 0, 0, 0);                     \
    acc[i_][0] = m0_; acc[i_][1] = m1_; accx[i_][0] = x0_; accx[i_][1] = x1_;                              \
  }
  // row block i_ with this round's LDS-DMA requests BETWEEN its matrix instructions, behind the round's barrier and before any
  // fragment read of the round is in flight: an LDS-DMA request costs the issuing wave ~60 cycles among bare MFMAs and 100 - 185
  // in a phase that already carries pieces and ds_reads (MI355X_MICROARCH.md).  (Round 5's order -- row block 0, barrier, the three
  // requests back to back, then the fragment reads among row block 1 -- against this one, interleaved on one box: config-5 shard
  // 1214 -> 1201 us, 1298 -> 1276 on a slower box; the requests between row block 1's instructions instead, behind the reads: 1213; the
  // fragment reads moved up behind row block 0's last two instructions: 1223 against 1214; moved down by one or two instructions: no change.)
#define GMVAE_P2_TILES_DMA(FA_, FB_, i_, on_, buf_)                                                            \
  {                                                                                                        \
    f32x16 m0_ = acc[i_][0], m1_ = acc[i_][1], x0_ = accx[i_][0], x1_ = accx[i_][1];                           \
    unsigned short* const d_ = img + (buf_) * G::Buf;                                                      \
    if (on_) __builtin_amdgcn_global_load_lds(A + ea, d_ + wave * 512, 16, 0, 0);                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    m0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[0][0], m0_, 0, 0, 0);                      \
    m1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[1][0], m1_, 0, 0, 0);                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (on_) __builtin_amdgcn_global_load_lds(A + a_ps + ea, d_ + G::PA + wave * 512, 16, 0, 0);           \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    x0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[0][1], x0_, 0, 0, 0);                      \
    x1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][0], FB_[1][1], x1_, 0, 0, 0);                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (on_) {                                                                                             \
      __builtin_amdgcn_global_load_lds(Bw + eb, d_ + b_dst, 16, 0, 0);                                     \
      if (G::NW == 4) __builtin_amdgcn_global_load_lds(Bw + b_ps + eb, d_ + b_dst + 2048, 16, 0, 0);       \
      ea += a_round; eb += b_round;                                                                        \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    x0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][1], FB_[0][0], x0_, 0, 0, 0);                      \
    x1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA_[i_][1], FB_[1][0], x1_, 0, 0, 0);                      \
    acc[i_][0] = m0_; acc[i_][1] = m1_; accx[i_][0] = x0_; accx[i_][1] = x1_;                                  \
  }
  constexpr int kRd = (AMC && BMC) ? 4 : ((AMC || BMC) ? 3 : 2);       // LDS reads behind each of the first four of a round's last six
#define GMVAE_P2_SG(n_) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, n_, 0);
  // round c_ (fragments in CUR; buffer bc_ holds its images, bn_ the next round's); rounds c_ + 1 .. c_ + 3 are in flight
  // when it waits for the next one, round c_ + 4 is requested behind the barrier into the buffer it has just finished
#define GMVAE_P2_ROUND(CA_, CB_, NA_, NB_, c_, bc_, bn_)                                                   \
  {                                                                                                        \
    if (cs_on) {                                                                                           \
      const unsigned short* const ib_ = img + (bc_) * G::Buf + G::OpA + cs_off;                            \
      const u32x4 h_ = *reinterpret_cast<const u32x4*>(ib_), l_ = *reinterpret_cast<const u32x4*>(ib_ + 2048);   \
      const float sc = kscale ? kscale[kk] : 1.f;                                                          \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
        cs8[2 * q] += sc * (f16lo(h_[q]) + 0x1p-11f * f16lo(l_[q]));                                       \
        cs8[2 * q + 1] += sc * (f16hi(h_[q]) + 0x1p-11f * f16hi(l_[q]));                                   \
      }                                                                                                    \
      kk += 16;                                                                                            \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((c_) + 3 < NC16) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * G::DPW) : "memory");      \
    else if ((c_) + 2 < NC16) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::DPW) : "memory");     \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          \
    GMVAE_P2_TILES_DMA(CA_, CB_, 0, (c_) + 4 < NC16, bc_)                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    GMVAE_P2_FRAGS(NA_, NB_, bn_)                                                                          \
    GMVAE_P2_TILES(CA_, CB_, 1)                                                                            \
    GMVAE_P2_SG(kRd) GMVAE_P2_SG(kRd) GMVAE_P2_SG(kRd) GMVAE_P2_SG(kRd) GMVAE_P2_SG(0) GMVAE_P2_SG(0)      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  }
  f16x8_t fa[2][2], fb[2][2], ga[2][2], gb[2][2];
  GMVAE_P2_DMA(0)
  GMVAE_P2_DMA(1)                                // (NC16 >= 2, even)
  // both accumulator sets start HERE, inside the orientation's own copy of the loop: zeroed by the caller above the four-way
  // branch, the compiler kept one set of zeros alive in 32 registers through every copy and spilled 46 registers inside the loop
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
  if (NC16 > 2) {
    GMVAE_P2_DMA(2)
    GMVAE_P2_DMA(3)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * G::DPW) : "memory");    // round 0 has landed, three rounds stay in flight
  } else {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G::DPW) : "memory");
  }
  GMVAE_P2_FRAGS(fa, fb, 0)
  int bc = 0;
#pragma unroll 1
  for (int c = 0; c < NC16; c += 2) {
    const int b1 = (bc + 1) & 3, b2 = (bc + 2) & 3;
    GMVAE_P2_ROUND(fa, fb, ga, gb, c, bc, b1)
    GMVAE_P2_ROUND(ga, gb, fa, fb, c + 1, b1, b2)
    bc = b2;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the caller reuses LDS)
#undef GMVAE_P2_DMA
#undef GMVAE_P2_FRAGS
#undef GMVAE_P2_TILES
#undef GMVAE_P2_TILES_DMA
#undef GMVAE_P2_ROUND
#undef GMVAE_P2_SG
}


// ---- the same loop on v_mfma_f32_16x16x32_f16 (round 6) ------------------------------------------------------------------------
// Same images, same LDS-DMA staging, same piece products; the matrix instruction is the 16 x 16 x 32 shape, which this part
// sustains at a higher clock than 32 x 32 x 16 at equal cycles per product (MI355X_MICROARCH.md, DVFS item 7: 1.12 - 1.15x on random
// data; timing experiment here with the operands of the loop above pushed through two 16x16x32 per instruction, results discarded:
// forward launch 290 -> 274 us, backward 455 -> 416).  An instruction contracts 32 k = the images of TWO 16-deep rounds; which 8
// k a lane group holds is free as long as both operands agree, and the choice here -- lane group lk: k 4 lk .. 4 lk + 3 of the
// first image, then the same four of the second -- makes every fragment two 8-byte reads that are bank-conflict-free on the
// EXISTING images (k-contiguous [rows][16 k] with its half swap: groups 0 / 1 and 2 / 3 of a 32-lane pass read different halves of
// 16 rows; mn-contiguous [blocks][16 k][16 mn] with its rotated rows: the four groups of ds_read_b64_tr_b16 read rows 4 lk ..
// 4 lk + 3).  A wave's 64 x 64 block is 4 x 4 tiles, 48 instructions per 32 k.  Registers: the accumulators (128), this 32-k step's
// 8 b fragments and the next step's (64), two a fragments of the row block being multiplied and of the next (16).  Order of a step:
// row blocks 0, 1, 2 (the next block's a fragment read behind the first instructions of each), then ONE wait + barrier per 48
// instructions (the next step's images have landed, this step's are read), the LDS-DMA requests of the step after next between the
// first instructions of row block 3 (as in the loop above), the next step's b fragments and first a fragment behind the rest.
// pairs of image buffers: three where the workgroup has the CU's LDS to itself (256-row tiles: 6 x 24 KB), two otherwise
template <int BM> struct P2H_PAIRS { static constexpr int v = BM == 256 ? 3 : 2; };
template <bool MC>
__device__ __forceinline__ f16x8_t p2h_frag(const unsigned short* __restrict__ i0, const unsigned short* __restrict__ i1) {
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  if (!MC) {
    const s16x4_t lo = *reinterpret_cast<const s16x4_t*>(i0), hi = *reinterpret_cast<const s16x4_t*>(i1);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  } else {
    typedef __attribute__((address_space(3))) s16x4_t* lp;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(i0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(i1));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  }
}

template <bool AMC, bool BMC, int BM>
__device__ __forceinline__ void plane_rounds2h(unsigned short* __restrict__ img, const unsigned short* __restrict__ A,
                                               const uint32_t a_rows, const long long a_ps, const unsigned short* __restrict__ Bp,
                                               const uint32_t b_rows, const long long b_ps, const float* __restrict__ kscale,
                                               const bool do_cs, const int m0, const int n0, const int kb, const int NC16,
                                               const int tid, const int lane, const int wave, const int wm0, const int wn0,
                                               f32x16 (&acc)[2][2], f32x16 (&accx)[2][2], float (&cs8)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef P2<BM> G;
  constexpr int NP = P2H_PAIRS<BM>::v;             // pairs of image buffers in the ring (a pair = one 32-k step)
  uint32_t ea = p3_src<AMC>(wave, lane, a_rows, m0, kb), eb = p3_src<BMC>(wave & 3, lane, b_rows, n0, kb);
  const uint32_t a_round = AMC ? 256u : 16u * a_rows, b_round = BMC ? 256u : 16u * b_rows;
  const unsigned short* const Bw = Bp + (G::NW == 8 ? (wave >> 2) * b_ps : 0);
  const int b_dst = G::OpA + (G::NW == 8 ? (wave >> 2) * 2048 : 0) + (wave & 3) * 512;
  const int ln = lane & 15, lk = lane >> 4;
  // fragment bases inside an operand image of one plane, for even and odd 16-blocks (+ 256 per block)
  // both operands k-contiguous (a data gradient): lane group lk takes k 8 (lk & 1) .. + 7 of image lk >> 1 -- ONE 16-byte read per
  // fragment, as the loop above reads them (with the 8-byte halves of two images such a launch was 15 % slower than there:
  // twice the LDS instructions)
  constexpr bool K8 = !AMC && !BMC;
  // (K8: image lk & 1, half lk >> 1 -- ds_read_b128's lane groups are {0-3, 12-15, 20-27}, ...: with the halves the other way round two
  //  lanes of a group met in every bank)
  const int kcn = K8 ? (lk & 1) * G::Buf + 8 * ((lk >> 1) ^ ((ln >> 3) & 1)) : 8 * ((lk >> 1) ^ ((ln >> 3) & 1)) + 4 * (lk & 1);
  // the second image's distance as a value the compiler cannot see: it fused a fragment's two 8-byte reads into ds_read2st64_b64, whose
  // lane groups and 32-bank map differ from ds_read_b64's (MI355X_MICROARCH.md, LDS table) -- SQ_LDS_BANK_CONFLICT rose 6 - 17 fold
  int buf2;
  asm volatile("s_mov_b32 %0, %1" : "=s"(buf2) : "i"(G::Buf));
  const int rE = ((4 * lk + (ln >> 2)) & 15) * 16 + 4 * (ln & 3), rO = ((4 * lk + (ln >> 2) + 4) & 15) * 16 + 4 * (ln & 3);
  const int faE = AMC ? (wm0 >> 4) * 256 + rE : (wm0 + ln) * 16 + kcn, faO = AMC ? (wm0 >> 4) * 256 + rO : faE;
  const int fbE = G::OpA + (BMC ? (wn0 >> 4) * 256 + rE : (wn0 + ln) * 16 + kcn), fbO = G::OpA + (BMC ? (wn0 >> 4) * 256 + rO : (wn0 + ln) * 16 + kcn);
  const bool cs_on = BMC && do_cs && tid < 256;
  const int cs_k = (tid >> 4) & 15, cs_ch = tid & 15;
  const int cs_off = (cs_ch >> 1) * 256 + ((cs_k + 4 * ((cs_ch >> 1) & 1)) & 15) * 16 + 8 * (cs_ch & 1);
  int kk = kb + cs_k;
  f32x4 m[4][4], x[4][4];
  // the LDS-DMA requests of one image (q = 0, 1: a's planes; 2: b; 3: b's second plane at four waves) into buffer buf
  auto dma = [&](const int buf, const int q, const uint32_t oa, const uint32_t ob) {
    unsigned short* const d_ = img + buf * G::Buf;
    if (q == 0) __builtin_amdgcn_global_load_lds(A + oa, d_ + wave * 512, 16, 0, 0);
    else if (q == 1) __builtin_amdgcn_global_load_lds(A + a_ps + oa, d_ + G::PA + wave * 512, 16, 0, 0);
    else if (q == 2) __builtin_amdgcn_global_load_lds(Bw + ob, d_ + b_dst, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds(Bw + b_ps + ob, d_ + b_dst + 2048, 16, 0, 0);
  };
  auto dma_pair_all = [&](const int pair) {
#pragma unroll
    for (int im = 0; im < 2; ++im) {
#pragma unroll
      for (int q = 0; q < G::DPW; ++q) dma(2 * pair + im, q, ea, eb);
      ea += a_round; eb += b_round;
    }
  };
  // fragments of the step whose images lie at ic (elements from img): block i / j, plane pl
  auto rd_a = [&](const unsigned short* ic, const int i, const int pl) {
    const unsigned short* const p0 = ic + pl * G::PA + 256 * i + ((i & 1) ? faO : faE);
    if constexpr (K8) return *reinterpret_cast<const f16x8_t*>(p0);
    else return p2h_frag<AMC>(p0, p0 + buf2);
  };
  auto rd_b = [&](const unsigned short* ic, const int j, const int pl) {
    const unsigned short* const p0 = ic + pl * 2048 + 256 * j + ((j & 1) ? fbO : fbE);
    if constexpr (K8) return *reinterpret_cast<const f16x8_t*>(p0);
    else return p2h_frag<BMC>(p0, p0 + buf2);
  };
  // the 12 instructions of row block i_: main terms, a's second piece x b's first, then a's first x b's SECOND piece (whose
  // fragments are read at the top of the step itself: single-buffered); HOOK_(k) behind the k-th instruction
#define GMVAE_P2H_BLOCK(i_, FA_, FB0_, HOOK_)                                                              \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
      m[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA_[0], FB0_[j], m[i_][j], 0, 0, 0);               \
      HOOK_(j) __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
      x[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA_[1], FB0_[j], x[i_][j], 0, 0, 0);               \
      HOOK_(4 + j) __builtin_amdgcn_sched_barrier(0);                                                      \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
      x[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA_[0], FB1[j], x[i_][j], 0, 0, 0);                \
      HOOK_(8 + j) __builtin_amdgcn_sched_barrier(0);                                                      \
    }                                                                                                      \
  }
  // hooks.  Row block 0: b's second-piece fragments of THIS step (needed from its ninth instruction on), then row block 1's a
  // fragment; row blocks 1, 2: the next block's a fragment; row block 3 (behind the barrier): the LDS-DMA requests of the step NP
  // ahead, one per instruction (first: MI355X_MICROARCH.md's issue cost of such a request is lowest among bare matrix
  // instructions; measured both ways), then the next step's first-piece b fragments and its first a fragment
#define H_B0(k) if ((k) < 4) FB1[k] = rd_b(ic, (k), 1); else if ((k) < 6) FO[(k) - 4] = rd_a(ic, 1, (k) - 4);
#define H_B1(k) if ((k) < 2) FE[k] = rd_a(ic, 2, (k));
#define H_B2(k) if ((k) < 2) FO[k] = rd_a(ic, 3, (k));
#define H_B3(k)                                                                                            \
  if ((k) < 2 * G::DPW) { if (dmaon) dma(2 * pc + (k) / G::DPW, (k) % G::DPW, ea + ((k) / G::DPW ? a_round : 0u), eb + ((k) / G::DPW ? b_round : 0u)); }   \
  else if (G::DPW == 3 && (k) < 10) FN[(k) - 6] = rd_b(in, (k) - 6, 0);                                    \
  else if (G::DPW == 3) FE[(k) - 10] = rd_a(in, 0, (k) - 10);                                              \
  else if ((k) < 10) { FN[2 * ((k) - 8)] = rd_b(in, 2 * ((k) - 8), 0); FN[2 * ((k) - 8) + 1] = rd_b(in, 2 * ((k) - 8) + 1, 0); }   \
  else FE[(k) - 10] = rd_a(in, 0, (k) - 10);
  // one 32-k step d_: its images in pair pc (b's first-piece fragments FBC_ and row block 0's a fragment FE in registers)
#define GMVAE_P2H_STEP(FBC_, d_)                                                                           \
  {                                                                                                        \
    const int pn = pc + 1 == NP ? 0 : pc + 1;                                                              \
    const unsigned short* const ic = img + 2 * pc * G::Buf;                                                \
    const unsigned short* const in = img + 2 * pn * G::Buf;                                                \
    if (cs_on) {                                                                                           \
      _Pragma("unroll") for (int im = 0; im < 2; ++im) {                                                   \
        const unsigned short* const ib_ = ic + im * G::Buf + G::OpA + cs_off;                              \
        const u32x4 h_ = *reinterpret_cast<const u32x4*>(ib_), l_ = *reinterpret_cast<const u32x4*>(ib_ + 2048);   \
        const float sc = kscale ? kscale[kk] : 1.f;                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
          cs8[2 * q] += sc * (f16lo(h_[q]) + 0x1p-11f * f16lo(l_[q]));                                     \
          cs8[2 * q + 1] += sc * (f16hi(h_[q]) + 0x1p-11f * f16hi(l_[q]));                                 \
        }                                                                                                  \
        kk += 16;                                                                                          \
      }                                                                                                    \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    GMVAE_P2H_BLOCK(0, FE, FBC_, H_B0)                                                                     \
    GMVAE_P2H_BLOCK(1, FO, FBC_, H_B1)                                                                     \
    GMVAE_P2H_BLOCK(2, FE, FBC_, H_B2)                                                                     \
    /* the next step's images have landed (NP = 3: the step after it may still be in flight); this step's are read */ \
    if (NP == 3 && (d_) + 2 < ND) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * G::DPW) : "memory");   \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          \
    const bool dmaon = (d_) + NP < ND;                                                                     \
    GMVAE_P2H_BLOCK(3, FO, FBC_, H_B3)                                                                     \
    if (dmaon) { ea += 2 * a_round; eb += 2 * b_round; }                                                   \
    pc = pn;                                                                                               \
  }
  f16x8_t FE[2], FO[2], F0[4], F1[4], FB1[4];
  const int ND = NC16 >> 1;                        // 32-k steps (NC16 is even)
  dma_pair_all(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (ND > 1) dma_pair_all(1);
  if (NP == 3 && ND > 2) dma_pair_all(2);
  // the first step's images have landed
  if (NP == 3 && ND > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * G::DPW) : "memory");
  else if (ND > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G::DPW) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) F0[j] = rd_b(img, j, 0);
  FE[0] = rd_a(img, 0, 0); FE[1] = rd_a(img, 0, 1);
  int pc = 0;
#pragma unroll 1
  for (int d = 0; d < ND; d += 2) {
    {
      auto& FN = F1;
      GMVAE_P2H_STEP(F0, d)
    }
    if (d + 1 < ND) {
      auto& FN = F0;
      GMVAE_P2H_STEP(F1, d + 1)
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the caller reuses LDS)
  // into the caller's registers: tile (i, j) = quarter 2 (i & 1) + (j & 1) of block (i >> 1, j >> 1) -- gemm_grouped's staging knows
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[i >> 1][j >> 1][4 * (2 * (i & 1) + (j & 1)) + r] = m[i][j][r];
        accx[i >> 1][j >> 1][4 * (2 * (i & 1) + (j & 1)) + r] = x[i][j][r];
      }
#undef GMVAE_P2H_BLOCK
#undef GMVAE_P2H_STEP
#undef H_B0
#undef H_B1
#undef H_B2
#undef H_B3
}

// ---- the 128x128x32 configuration's interior rounds ("big rounds") -----------------------------------------------
// For tiles completely inside both fp32 operands and k ranges that are whole 32-deep rounds.  LDS image of an operand
// round: [8 k-quads][128 mn] units of 16 bytes = the 4 consecutive k of one mn, unit index swizzled
//     U(kq, mn) = kq * 128 + (mn ^ ((mn >> 4) & 3) ^ ((kq & 3) << 2))
// so that all three access patterns are bank-conflict-free 16-byte operations: the staging store of a k-contiguous source
// (a slot IS a unit), of an mn-contiguous source (a thread owns 4 k x 4 mn and writes its 4 transposed units), and the
// fragment read (lane (h, l31) of v_mfma_f32_32x32x2_f32 takes unit (2g + h, mn0 + l31): its k for the 4 MFMA steps of
// k-group g -- both operands use the same k order, so the sum is the same set of products).  A round is then 64 MFMAs,
// 16 ds_read_b128, 8 ds_write_b128 and 8 global_load_dwordx4 per wave (the [k][mn] image of the general loop: 32
// ds_read2_b32, 32 ds_write_b32), the next k-group's fragments are in flight while one multiplies, and the staging stores
// of the next round sit in front of the last 16 MFMAs -- nothing but the barrier is outside the matrix pipe's shadow.
constexpr int kBigBuf = 8192;       // floats per staging buffer: A image | B image

template <bool AMC, bool BMC>
__device__ __forceinline__ void big_rounds(float* __restrict__ lds, const float* __restrict__ A, const uint32_t a_ld,
                                           const float* __restrict__ Bp, const uint32_t b_ld,
                                           const float* __restrict__ kscale, const int m0, const int n0, const int kb,
                                           const int NC, const int tid, const int lane, const int wm0, const int wn0,
                                           f32x16 (&acc)[2][2], float4& cs4, const bool first) {
  const int h = lane >> 5, l31 = lane & 31;
  // (a problem's first segment starts the accumulators here, inside the orientation's own copy of the loop -- see plane_rounds2)
  if (first) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // the per-k scale and the column sums ride only beside an mn-contiguous b; both are wave-uniform run-time options (as
  // template parameters they made ten copies of this loop in one kernel)
  const bool KS = BMC && kscale != nullptr;
  constexpr bool CS = BMC;
  // staging: element offsets of slot 0, slot stride, round stride; LDS float offsets of this thread's units
  const uint32_t kq_k = (tid >> 2) & 7, mn_k = (tid & 3) + 4 * (tid >> 5);      // k-contiguous map: (k-quad, mn), slots 32 mn apart
  const uint32_t kq_m = tid >> 5, q_m = tid & 31;                               // mn-contiguous map: (k-quad, mn-quad), slots = the 4 k
  uint32_t ea, eb, wa[4], wb[4];
  const uint32_t a_slot = AMC ? a_ld : 32u * a_ld, a_round = AMC ? 32u * a_ld : 32u;
  const uint32_t b_slot = BMC ? b_ld : 32u * b_ld, b_round = BMC ? 32u * b_ld : 32u;
  if (AMC) {
    ea = (uint32_t)(kb + 4 * kq_m) * a_ld + (uint32_t)m0 + 4 * q_m;
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = 4 * (kq_m * 128 + ((4 * q_m) ^ ((kq_m & 3) << 2)) + (j ^ ((q_m >> 2) & 3)));
  } else {
    ea = (uint32_t)(m0 + mn_k) * a_ld + (uint32_t)kb + 4 * kq_k;
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = 4 * (kq_k * 128 + 32 * i + (mn_k ^ (mn_k >> 4) ^ ((i & 1) << 1) ^ ((kq_k & 3) << 2)));
  }
  if (BMC) {
    eb = (uint32_t)(kb + 4 * kq_m) * b_ld + (uint32_t)n0 + 4 * q_m;
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[j] = 4096 + 4 * (kq_m * 128 + ((4 * q_m) ^ ((kq_m & 3) << 2)) + (j ^ ((q_m >> 2) & 3)));
  } else {
    eb = (uint32_t)(n0 + mn_k) * b_ld + (uint32_t)kb + 4 * kq_k;
#pragma unroll
    for (int i = 0; i < 4; ++i) wb[i] = 4096 + 4 * (kq_k * 128 + 32 * i + (mn_k ^ (mn_k >> 4) ^ ((i & 1) << 1) ^ ((kq_k & 3) << 2)));
  }
  // fragment reads: lane base per (tile parity, k-group parity); the rest is an immediate offset
  const uint32_t L0 = l31 ^ (l31 >> 4) ^ (h << 2);
  uint32_t fa_[2][2], fb_[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const uint32_t x = ((((wm0 >> 5) + i) & 1) << 1) ^ (gp << 3), y = ((((wn0 >> 5) + i) & 1) << 1) ^ (gp << 3);
      fa_[i][gp] = 4 * (h * 128 + wm0 + 32 * i + (L0 ^ x));
      fb_[i][gp] = 4096 + 4 * (h * 128 + wn0 + 32 * i + (L0 ^ y));
    }
  f32x4 ra[4], rb[4];          // (native vectors: whole-struct copies of HIP's float4 keep the array in scratch)
  float ks[4] = {1.f, 1.f, 1.f, 1.f};
  int kk = kb + 4 * (int)kq_m;
  // (macros, not lambdas: the staging registers must stay plain register arrays)
#define GMVAE_BIG_GLOAD()                                                                              \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + ea + i * a_slot);   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Bp + eb + i * b_slot);  \
    if (KS) {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) ks[i] = kscale[kk + i];                            \
      kk += 32;                                                                                        \
    }                                                                                                  \
    ea += a_round; eb += b_round;                                                                      \
  }
#define GMVAE_BIG_PUT(img_, w_, r_, MC_)                                                               \
  if (!(MC_)) {                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>((img_) + w_[i]) = r_[i];  \
  } else { /* 4 k x 4 mn in registers -> the 4 units (4 k of one mn each) */                           \
    *reinterpret_cast<f32x4*>((img_) + w_[0]) = f32x4{r_[0].x, r_[1].x, r_[2].x, r_[3].x};      \
    *reinterpret_cast<f32x4*>((img_) + w_[1]) = f32x4{r_[0].y, r_[1].y, r_[2].y, r_[3].y};      \
    *reinterpret_cast<f32x4*>((img_) + w_[2]) = f32x4{r_[0].z, r_[1].z, r_[2].z, r_[3].z};      \
    *reinterpret_cast<f32x4*>((img_) + w_[3]) = f32x4{r_[0].w, r_[1].w, r_[2].w, r_[3].w};      \
  }
#define GMVAE_BIG_LSTORE(buf_)                                                                         \
  {                                                                                                    \
    if (KS) {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { rb[i].x *= ks[i]; rb[i].y *= ks[i]; rb[i].z *= ks[i]; rb[i].w *= ks[i]; } \
    }                                                                                                  \
    if (CS) {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { cs4.x += rb[i].x; cs4.y += rb[i].y; cs4.z += rb[i].z; cs4.w += rb[i].w; } \
    }                                                                                                  \
    float* const img_w = lds + (buf_) * kBigBuf;                                                       \
    GMVAE_BIG_PUT(img_w, wa, ra, AMC)                                                                  \
    GMVAE_BIG_PUT(img_w, wb, rb, BMC)                                                                  \
  }
#define GMVAE_BIG_RD(dst_, base_, i_, g_) dst_[i_] = *reinterpret_cast<const f32x4*>(img + base_[i_][(g_) & 1] + 1024 * (g_));
#define GMVAE_BIG_STEP(FA_, FB_, c_)                                                                    \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA_[0].c_, FB_[0].c_, acc[0][0], 0, 0, 0);           \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA_[0].c_, FB_[1].c_, acc[0][1], 0, 0, 0);           \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA_[1].c_, FB_[0].c_, acc[1][0], 0, 0, 0);           \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA_[1].c_, FB_[1].c_, acc[1][1], 0, 0, 0);           \
  __builtin_amdgcn_sched_barrier(0);
  // one k-group's 16 MFMAs with the NEXT group's 4 fragment reads issued between its steps (each in the shadow of
  // the 4 MFMAs before it)
#define GMVAE_BIG_MUL_RD(FA_, FB_, NA_, NB_, g_)                                                        \
  GMVAE_BIG_STEP(FA_, FB_, x) GMVAE_BIG_RD(NA_, fa_, 0, g_) __builtin_amdgcn_sched_barrier(0);          \
  GMVAE_BIG_STEP(FA_, FB_, y) GMVAE_BIG_RD(NB_, fb_, 0, g_) __builtin_amdgcn_sched_barrier(0);          \
  GMVAE_BIG_STEP(FA_, FB_, z) GMVAE_BIG_RD(NA_, fa_, 1, g_) __builtin_amdgcn_sched_barrier(0);          \
  GMVAE_BIG_STEP(FA_, FB_, w) GMVAE_BIG_RD(NB_, fb_, 1, g_) __builtin_amdgcn_sched_barrier(0);
  f32x4 pa[2], pb[2], qa[2], qb[2];     // fragments of the even / odd k-groups
  GMVAE_BIG_GLOAD();
  GMVAE_BIG_LSTORE(0);
  __syncthreads();
  {
    const float* const img = lds;
    GMVAE_BIG_RD(pa, fa_, 0, 0) GMVAE_BIG_RD(pb, fb_, 0, 0) GMVAE_BIG_RD(pa, fa_, 1, 0) GMVAE_BIG_RD(pb, fb_, 1, 0)
  }
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    const bool more = c + 1 < NC;
    if (more) GMVAE_BIG_GLOAD();
    __builtin_amdgcn_sched_barrier(0);
    {
      const float* const img = lds + (c & 1) * kBigBuf;
      GMVAE_BIG_MUL_RD(pa, pb, qa, qb, 1)
      GMVAE_BIG_MUL_RD(qa, qb, pa, pb, 2)
      GMVAE_BIG_MUL_RD(pa, pb, qa, qb, 3)
    }
    // last k-group: the next round's staging stores, the round's barrier and the next round's first fragment reads all
    // sit between its steps (every wave finished reading the buffer being overwritten before the PREVIOUS barrier)
    // (the MFMA steps stay outside the uniform branches: inside them the compiler keeps two copies of the accumulators)
    float* const img_w = lds + ((c + 1) & 1) * kBigBuf;
    GMVAE_BIG_STEP(qa, qb, x)
    if (more) {
      if (KS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { rb[i].x *= ks[i]; rb[i].y *= ks[i]; rb[i].z *= ks[i]; rb[i].w *= ks[i]; }
      }
      if (CS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { cs4.x += rb[i].x; cs4.y += rb[i].y; cs4.z += rb[i].z; cs4.w += rb[i].w; }
      }
      GMVAE_BIG_PUT(img_w, wa, ra, AMC)
    }
    __builtin_amdgcn_sched_barrier(0);
    GMVAE_BIG_STEP(qa, qb, y)
    if (more) { GMVAE_BIG_PUT(img_w, wb, rb, BMC) }
    __builtin_amdgcn_sched_barrier(0);
    GMVAE_BIG_STEP(qa, qb, z)
    if (more) {
      __syncthreads();
      const float* const img = img_w;
      GMVAE_BIG_RD(pa, fa_, 0, 0) GMVAE_BIG_RD(pb, fb_, 0, 0) GMVAE_BIG_RD(pa, fa_, 1, 0) GMVAE_BIG_RD(pb, fb_, 1, 0)
    }
    __builtin_amdgcn_sched_barrier(0);
    GMVAE_BIG_STEP(qa, qb, w)
  }
#undef GMVAE_BIG_RD
#undef GMVAE_BIG_MUL_RD
#undef GMVAE_BIG_GLOAD
#undef GMVAE_BIG_PUT
#undef GMVAE_BIG_LSTORE
#undef GMVAE_BIG_STEP
}

// BIG = 1: the 128x128 instance for launches in which EVERY tile is interior, fp32 and made of whole rounds (the host
// checks: big_eligible); its only main loop is big_rounds -- a kernel of its own so that the general loop's loaders do
// not share its register budget.  BIG = 2: every problem of the launch reads pre-split planes (plane_rounds3).
template <class C, int BIG = 0>
__global__ __launch_bounds__(C::THREADS, C::WAVES_EU) void gemm_grouped(const Launch L) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  constexpr int kBK = C::BK;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Round 1 of kernel-argument loads: the whole header, before the first branch.
  const unsigned long long t_in = wall_clock64();
  unsigned long long* const dbg = L.dbg;
  const int auxn = L.aux_nblocks, nprob = L.nprob;
  // aux_last: the tiles come first and the auxiliary workgroups follow (a launch of few, long tiles: the Philox fill of
  // the general schedule then runs on the CUs the tiles leave idle instead of in front of them)
  const int ntl = L.total_tiles;
  const int aux_i = L.aux_last ? (int)blockIdx.x - ntl : (int)blockIdx.x;
  const int bid = L.aux_last ? ((int)blockIdx.x < ntl ? (int)blockIdx.x : -1) : (int)blockIdx.x - auxn;
  int pi = 0, tb = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i) {
    const int tbi = L.tile_begin[i];
    if (i < nprob && bid >= tbi) { pi = i; tb = tbi; }
  }
  // pin the header's consumers here: the compiler otherwise sinks these loads below the branches that follow
  // and every one of them becomes its own serialised round trip
  asm volatile("" ::"s"(pi), "s"(tb), "s"(dbg), "s"(auxn));
#define GMVAE_GSTAMP(i) if (dbg && tid == 0) dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64()
  if (dbg && tid == 0) dbg[(size_t)blockIdx.x * 8] = t_in;
  if (bid < 0) {                                // auxiliary work riding on this launch (aux.hpp)
    aux_block(L.aux, aux_i);
    GMVAE_GSTAMP(4);
    if (dbg && tid == 0) dbg[(size_t)blockIdx.x * 8 + 5] = 100;
    return;
  }

  // Round 2: every scalar of this tile's problem and of its first segment, requested back to back.
  const int splits = L.p[pi].splits, tiles_n = L.p[pi].tiles_n, nseg = L.p[pi].nseg;
  // segment descriptor -> scalars (nothing below takes a reference into the kernel arguments); the next
  // segment's are fetched at the top of its iteration
#define GMVAE_SEG_FIELDS(sg_)                                                                              \
  a_ptr = L.p[pi].seg[sg_].a.ptr; a_ld = L.p[pi].seg[sg_].a.ld; a_n = L.p[pi].seg[sg_].a.n_mn;             \
  a_div = L.p[pi].seg[sg_].a.row_div;                                                                      \
  a_mc = !L.p[pi].seg[sg_].a.k_contig;                                                                     \
  akind = op_kind(L.p[pi].seg[sg_].a.is_u8, L.p[pi].seg[sg_].a.k_contig, L.p[pi].seg[sg_].a.vec_ok);       \
  b_ptr = L.p[pi].seg[sg_].b.ptr; b_ld = L.p[pi].seg[sg_].b.ld; b_n = L.p[pi].seg[sg_].b.n_mn;             \
  b_div = L.p[pi].seg[sg_].b.row_div; b_mc = !L.p[pi].seg[sg_].b.k_contig;                                 \
  bkind = op_kind(0, L.p[pi].seg[sg_].b.k_contig, L.p[pi].seg[sg_].b.vec_ok);                              \
  kscale = L.p[pi].seg[sg_].kscale; K = L.p[pi].seg[sg_].K
  const void *a_ptr, *b_ptr;
  const float* kscale;
  int a_ld, a_n, a_div, akind, b_ld, b_n, b_div, bkind, K;
  bool a_mc, b_mc;
  GMVAE_SEG_FIELDS(0);
  asm volatile("" ::"s"(splits), "s"(tiles_n), "s"(nseg), "s"(a_ptr), "s"(a_ld), "s"(a_n), "s"(a_div),
               "s"(akind), "s"(b_ptr), "s"(b_ld), "s"(b_n), "s"(b_div), "s"(bkind), "s"(kscale), "s"(K));

  // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (observed; speed only), each with an L2 of its own, and
  // the ~64 tiles an XCD runs at a time walk k together: what they share is fetched once.  xorder 0: split fastest, then
  // tn, then tm (an XCD keeps a few B column blocks and streams every A row block: right when B is small).  xorder 1
  // (A much taller than B is wide: a data gradient): the tiles of one tm -- all tn -- are consecutive on ONE XCD, so its 64
  // concurrent tiles are 64 / tiles_n row blocks x all column blocks instead of 64 row blocks x one column block.  xorder 2
  // (B much wider than A is tall: a weight gradient): the tiles of one (split, tn) -- all tm -- likewise.  (Config 5's
  // decoder backward launch moved 3.0 GB through the fabric for 0.9 GB of operands with xorder 0.)
  int tm, tn, split;
  {
    const int t = bid - tb;
    const int xo = L.p[pi].xorder;
    if (xo == 0) {
      split = t % splits;
      const int u = t / splits;
      tn = u % tiles_n;
      tm = u / tiles_n;
    } else {
      const int tiles_m_ = L.p[pi].tiles_m;
      // (xorder 3, both operands large: groups of 8 column tiles of one tm, the groups of one block of 8 column tiles
      //  consecutive -- an XCD's 64 concurrent tiles are then 8 row blocks x 8 column blocks)
      // (xorder 4, a split-K weight gradient: the group is ONE split -- all its tm x tn units on one XCD, which then needs
      //  only that split's k slice of either operand)
      const int gs = xo == 1 ? tiles_n * splits : (xo == 2 ? tiles_m_ : (xo == 4 ? tiles_m_ * tiles_n : 8 * splits));                 // tiles per group
      const int ng = xo == 1 ? tiles_m_ : (xo == 2 ? tiles_n * splits : (xo == 4 ? splits : tiles_m_ * (tiles_n >> 3)));  // groups
      const int full = ng & ~7;                                             // groups dealt 8 at a time, one per XCD
      int grp, mem;
      if (t < full * gs) { const int j = t >> 3; grp = (j / gs) * 8 + (t & 7); mem = j % gs; }
      else { const int t2 = t - full * gs; grp = full + t2 / gs; mem = t2 % gs; }
      if (xo == 1) { tm = grp; split = mem % splits; tn = mem / splits; }
      else if (xo == 2) { tm = mem; split = grp % splits; tn = grp / splits; }
      else if (xo == 4) { tm = mem % tiles_m_; tn = mem / tiles_m_; split = grp; }
      else { tm = grp % tiles_m_; split = mem % splits; tn = (grp / tiles_m_) * 8 + mem / splits; }
    }
  }
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  const int wk = wave / (C::WM * C::WN);
  const int wmn = wave % (C::WM * C::WN);
  const int wm0 = (wmn / C::WN) * (C::TM * 32);
  const int wn0 = (wmn % C::WN) * (C::TN * 32);
  const int khalf = lane >> 5, l31 = lane & 31;

  f32x16 acc[C::TM][C::TN];
  if constexpr (BIG <= 1) {       // (the plane instances start their accumulators inside their loops: one segment)
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // bias gradient of a dW problem: column sums of b over this tile's k range (first tile row only)
  float* const colsum_out = L.p[pi].colsum_out;
  const bool do_colsum = colsum_out != nullptr && tm == 0;
  float csum = 0.f;
  constexpr int CSG = C::THREADS / C::BN;        // k groups of the column-sum threads
  const int cs_n = tid % C::BN, cs_k = tid / C::BN;
  bool did_bf16 = false, did_big = false, did_planes = false;
  float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (C::BM == 64 && C::BN == 64 && C::BK == 64) {
    if (L.p[pi].xbf16) {
      did_bf16 = true;
      unsigned short* const Xs = reinterpret_cast<unsigned short*>(lds);          // [64 k][kXLD]
      unsigned short* const Ds = Xs + 64 * kXLD;                                  // [3][64 k][kXLD]: hi, mid, lo
      const unsigned char* const xa = static_cast<const unsigned char*>(a_ptr);
      const float* const db = static_cast<const float*>(b_ptr);
      int kper = (K + splits - 1) / splits;
      kper = (kper + 63) / 64 * 64;
      const int kb = split * kper < K ? split * kper : K;
      const int ke = kb + kper < K ? kb + kper : K;
      const int NC = (ke - kb + 63) / 64;
      const int kr = tid >> 4, c4 = (tid & 15) << 2;          // this thread's (k row, 4 columns); 4 slots 16 rows apart
      const bool m_ok = m0 + c4 < a_n, n_ok = n0 + c4 < b_n;  // extents are multiples of 4 (vec_ok)
      unsigned xr[4];
      float4 dr[4];
      auto gload = [&](const int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = kb + c * 64 + kr + 16 * i;
          const bool kv = k < ke;
          const int kc = kv ? k : K - 1;
          xr[i] = (kv && m_ok) ? *reinterpret_cast<const unsigned*>(xa + (long long)kc * a_ld + m0 + c4) : 0u;
          dr[i] = (kv && n_ok) ? *reinterpret_cast<const float4*>(db + (long long)kc * b_ld + n0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = kr + 16 * i;
          const unsigned w = xr[i];
          const unsigned b0 = __float_as_uint((float)(w & 0xff)) >> 16, b1 = __float_as_uint((float)((w >> 8) & 0xff)) >> 16,
                         b2 = __float_as_uint((float)((w >> 16) & 0xff)) >> 16, b3 = __float_as_uint((float)(w >> 24)) >> 16;
          *reinterpret_cast<uint2*>(Xs + row * kXLD + c4) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
          const float v[4] = {dr[i].x, dr[i].y, dr[i].z, dr[i].w};
          unsigned hi[4], mi[4], lo[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned hb = __float_as_uint(v[j]) & 0xffff0000u;
            const float r1 = v[j] - __uint_as_float(hb);
            const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb);
            hi[j] = hb >> 16; mi[j] = mb >> 16; lo[j] = __float_as_uint(r2) >> 16;
          }
          *reinterpret_cast<uint2*>(Ds + (0 * 64 + row) * kXLD + c4) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          *reinterpret_cast<uint2*>(Ds + (1 * 64 + row) * kXLD + c4) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
          *reinterpret_cast<uint2*>(Ds + (2 * 64 + row) * kXLD + c4) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
          cs4.x += v[0]; cs4.y += v[1]; cs4.z += v[2]; cs4.w += v[3];
        }
      };
      // transposed-read addressing (cdna_hip_programming.md T10): lane 4q+p of a 16-lane group names row q, columns 4p..
      const int g16 = lane >> 4, i16 = lane & 15;
      const int fro = (8 * (g16 >> 1) + (i16 >> 2)) * kXLD + 16 * (g16 & 1) + 4 * (i16 & 3);
      if (NC > 0) gload(0);
      __syncthreads();
#pragma unroll 1
      for (int c = 0; c < NC; ++c) {
        lstore();
        __syncthreads();
        if (c + 1 < NC) gload(c + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8_t af = tr_frag(Xs + ks * 16 * kXLD + wm0 + fro);
          const bf16x8_t b2 = tr_frag(Ds + (2 * 64 + ks * 16) * kXLD + wn0 + fro);
          const bf16x8_t b1 = tr_frag(Ds + (1 * 64 + ks * 16) * kXLD + wn0 + fro);
          const bf16x8_t b0 = tr_frag(Ds + (0 * 64 + ks * 16) * kXLD + wn0 + fro);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[0][0], 0, 0, 0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[0][0], 0, 0, 0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[0][0], 0, 0, 0);
        }
        __syncthreads();
      }
    }
  }
#pragma unroll 1
  for (int sgi = 0; sgi < (did_bf16 ? 0 : nseg); ++sgi) {
    if (sgi > 0) { GMVAE_SEG_FIELDS(sgi); }

    int kper = (K + splits - 1) / splits;
    kper = (kper + kBK - 1) / kBK * kBK;
    const int kb = split * kper < K ? split * kper : K;
    const int ke = kb + kper < K ? kb + kper : K;
    const int NC = (ke - kb + kBK - 1) / kBK;
    if (NC == 0) continue;

    // wave-uniform eligibility of the lean loader for this tile
    const bool a_fast = akind < 4 && a_div == 1 && m0 + C::BM <= a_n;
    const bool b_fast = bkind < 4 && b_div == 1 && kscale == nullptr && n0 + C::BN <= b_n;

    float4 ra[C::NSA], rb[C::NSB];
#define GMVAE_FAST_A(KIND) op_load_fast<KIND, C::BM, kBK, C::NSA>(a_ptr, a_ld, m0, k0_, tid, ra)
#define GMVAE_FAST_B(KIND) op_load_fast<KIND, C::BN, kBK, C::NSB>(b_ptr, b_ld, n0, k0_, tid, rb)
#define GMVAE_LOAD_A(KIND) \
  op_load<KIND, C::BM, kBK, C::NSA>(a_ptr, a_ld, a_n, a_div, nullptr, K, m0, k0_, ke, tid, ra)
#define GMVAE_LOAD_B(KIND) \
  op_load<KIND, C::BN, kBK, C::NSB>(b_ptr, b_ld, b_n, b_div, kscale, K, n0, k0_, ke, tid, rb)
#define GMVAE_GLOAD(c_)                                                     \
  {                                                                         \
    const int k0_ = kb + (c_) * kBK;                                        \
    const bool full_ = k0_ + kBK <= ke;                                     \
    if (a_fast && full_) {                                                  \
      switch (akind) {                                                      \
        case 0: GMVAE_FAST_A(0); break;                                     \
        case 1: GMVAE_FAST_A(1); break;                                     \
        case 2: GMVAE_FAST_A(2); break;                                     \
        default: GMVAE_FAST_A(3); break;                                    \
      }                                                                     \
    } else                                                                  \
    switch (akind) {                                                        \
      case 0: GMVAE_LOAD_A(0); break;                                       \
      case 1: GMVAE_LOAD_A(1); break;                                       \
      case 2: GMVAE_LOAD_A(2); break;                                       \
      case 3: GMVAE_LOAD_A(3); break;                                       \
      case 4: GMVAE_LOAD_A(4); break;                                       \
      case 5: GMVAE_LOAD_A(5); break;                                       \
      case 6: GMVAE_LOAD_A(6); break;                                       \
      default: GMVAE_LOAD_A(7); break;                                      \
    }                                                                       \
    if (b_fast && full_) {                                                  \
      if (bkind == 0) GMVAE_FAST_B(0); else GMVAE_FAST_B(2);                \
    } else                                                                  \
    switch (bkind) {                                                        \
      case 0: GMVAE_LOAD_B(0); break;                                       \
      case 2: GMVAE_LOAD_B(2); break;                                       \
      case 4: GMVAE_LOAD_B(4); break;                                       \
      default: GMVAE_LOAD_B(6); break;                                      \
    }                                                                       \
  }
#define GMVAE_LSTORE(buf_)                                                  \
  {                                                                         \
    float* As_ = lds + (buf_) * (C::LDA + C::LDB) * kBK;                    \
    float* Bs_ = As_ + C::LDA * kBK;                                        \
    op_store<C::BM, kBK, C::LDA, C::NSA>(As_, a_mc, tid, ra);               \
    op_store<C::BN, kBK, C::LDB, C::NSB>(Bs_, b_mc, tid, rb);               \
  }

    if constexpr (BIG == 2) {                     // every problem of the launch has pre-split operands (host: planes_eligible)
      static_assert(C::BM == 128 && C::BN == 128 && C::BK == 32 && C::TM == 2 && C::TN == 2, "plane rounds: 128x128x32");
      __syncthreads();        // LDS is free
      if (sgi == 0) GMVAE_GSTAMP(6);
      const unsigned short* const Ah = static_cast<const unsigned short*>(a_ptr);
      const unsigned short* const Bh = static_cast<const unsigned short*>(b_ptr);
      const long long a_ps = L.p[pi].a_pstride, b_ps = L.p[pi].b_pstride;
      static_assert(kP3Ring * kP3Buf * 2 <= C::LDS_FLOATS * 4, "plane_rounds3's ring must fit the kernel's LDS");
      {
        // (B16 layout: the leading extent of an operand's matrix = its mn extent when k-contiguous, its k extent otherwise)
        const uint32_t a_rows = a_mc ? (uint32_t)K : (uint32_t)a_n, b_rows = b_mc ? (uint32_t)K : (uint32_t)b_n;
#define GMVAE_PL3(AMC_, BMC_) \
  plane_rounds3<AMC_, BMC_>(reinterpret_cast<unsigned short*>(lds), Ah, a_rows, a_ps, Bh, b_rows, b_ps, kscale, do_colsum, m0, n0, kb, \
                            2 * NC, tid, lane, wave, wm0, wn0, acc, cs8)
        if (!b_mc) {
          if (a_mc) GMVAE_PL3(true, false); else GMVAE_PL3(false, false);
        } else {
          if (a_mc) GMVAE_PL3(true, true); else GMVAE_PL3(false, true);
        }
#undef GMVAE_PL3
      }
      if (do_colsum) did_planes = true;
      if (sgi == 0) GMVAE_GSTAMP(1);
      continue;
    }
    if constexpr (BIG == 3) {                     // every problem of the launch reads f16 pairs (host: planes_eligible, planes == 2)
      static_assert((C::BM == 128 || C::BM == 256) && C::BN == 128 && C::BK == 32 && C::TM == 2 && C::TN == 2, "plane rounds: 128 / 256 x128x32");
      __syncthreads();        // LDS is free
      if (sgi == 0) GMVAE_GSTAMP(6);
      const unsigned short* const Ah = static_cast<const unsigned short*>(a_ptr);
      const unsigned short* const Bh = static_cast<const unsigned short*>(b_ptr);
      const long long a_ps = L.p[pi].a_pstride, b_ps = L.p[pi].b_pstride;
      static_assert(kP2Ring * P2<C::BM>::Buf * 2 <= C::LDS_FLOATS * 4, "plane_rounds2's ring must fit the kernel's LDS");
      static_assert(2 * P2H_PAIRS<C::BM>::v * P2<C::BM>::Buf * 2 <= C::LDS_FLOATS * 4, "plane_rounds2h's ring must fit the kernel's LDS");
      f32x16 accx[2][2];                          // (plane_rounds2 zeroes both sets: nseg == 1)
      {
        const uint32_t a_rows = a_mc ? (uint32_t)K : (uint32_t)a_n, b_rows = b_mc ? (uint32_t)K : (uint32_t)b_n;
#if GMVAE_P2_MFMA16
#define GMVAE_PL2(AMC_, BMC_) \
  plane_rounds2h<AMC_, BMC_, C::BM>(reinterpret_cast<unsigned short*>(lds), Ah, a_rows, a_ps, Bh, b_rows, b_ps, kscale, do_colsum, m0, n0, kb, \
                            2 * NC, tid, lane, wave, wm0, wn0, acc, accx, cs8)
#else
#define GMVAE_PL2(AMC_, BMC_) \
  plane_rounds2<AMC_, BMC_, C::BM>(reinterpret_cast<unsigned short*>(lds), Ah, a_rows, a_ps, Bh, b_rows, b_ps, kscale, do_colsum, m0, n0, kb, \
                            2 * NC, tid, lane, wave, wm0, wn0, acc, accx, cs8)
#endif
        if (!b_mc) {
          if (a_mc) GMVAE_PL2(true, false); else GMVAE_PL2(false, false);
        } else {
          if (a_mc) GMVAE_PL2(true, true); else GMVAE_PL2(false, true);
        }
#undef GMVAE_PL2
      }
      {
        // main term + cross terms, and both operands' scales off: (acc + 2^-11 accx) / (s t)
        const float ua = L.p[pi].a_uns ? *L.p[pi].a_uns : 1.f, ub = L.p[pi].b_uns ? *L.p[pi].b_uns : 1.f, uc = L.p[pi].uns_c;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = ((acc[i][j][r] + 0x1p-11f * accx[i][j][r]) * ua) * (ub * uc);
      }
      if (do_colsum) did_planes = true;
      if (sgi == 0) GMVAE_GSTAMP(1);
      continue;
    }
    if constexpr (BIG == 1) {
      static_assert(C::BM == 128 && C::BN == 128 && C::BK == 32 && C::TM == 2 && C::TN == 2, "big rounds: 128x128x32");
      __syncthreads();        // LDS is free
      if (sgi == 0) GMVAE_GSTAMP(6);
      const float* const Af = static_cast<const float*>(a_ptr);
      const float* const Bf = static_cast<const float*>(b_ptr);
#define GMVAE_BIG(AMC_, BMC_) \
  big_rounds<AMC_, BMC_>(lds, Af, (uint32_t)a_ld, Bf, (uint32_t)b_ld, kscale, m0, n0, kb, NC, tid, lane, wm0, wn0, acc, cs4, sgi == 0)
      if (!b_mc) {
        if (a_mc) GMVAE_BIG(true, false); else GMVAE_BIG(false, false);
      } else {
        if (a_mc) GMVAE_BIG(true, true); else GMVAE_BIG(false, true);
      }
#undef GMVAE_BIG
      if (do_colsum) did_big = true;
      if (sgi == 0) GMVAE_GSTAMP(1);
      continue;
    }
    if constexpr (BIG == 0) {       // (the instances above always `continue`: their kernels do not carry the general loop)
    __syncthreads();          // LDS is free (first segment: trivially; second: previous loop finished)
    if (sgi == 0) GMVAE_GSTAMP(6);
    GMVAE_GLOAD(0);
    if (L.dbg && sgi == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GMVAE_GSTAMP(7); }
    GMVAE_LSTORE(0);
    __syncthreads();
    if (sgi == 0) GMVAE_GSTAMP(1);
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) GMVAE_GLOAD(c + 1);
      const float* As = lds + (C::NBUF == 2 ? (c & 1) : 0) * (C::LDA + C::LDB) * kBK;
      const float* Bs = As + C::LDA * kBK;
      constexpr int KW = kBK / C::WK;
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < kBK / CSG; ++i) csum += Bs[(cs_k + i * CSG) * C::LDB + cs_n];
      }
#pragma unroll
      for (int kk = 0; kk < KW; kk += 2) {
        const int krow = wk * KW + kk + khalf;
        float a[C::TM], b[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) a[i] = As[krow * C::LDA + wm0 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) b[j] = Bs[krow * C::LDB + wn0 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (C::NBUF == 1 && c + 1 < NC) __syncthreads();     // everyone is done reading the only buffer
      if (c + 1 < NC) GMVAE_LSTORE(C::NBUF == 2 ? ((c + 1) & 1) : 0);
      __syncthreads();
    }
    }
#undef GMVAE_LOAD_A
#undef GMVAE_FAST_A
#undef GMVAE_FAST_B
#undef GMVAE_LOAD_B
#undef GMVAE_GLOAD
#undef GMVAE_LSTORE
  }

  // ---- stage the accumulators to LDS in row-major [BM][LDC] (one image per k-wave)
  __syncthreads();
  GMVAE_GSTAMP(2);
  {
    float* Cs = lds + wk * C::BM * C::LDC;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (BIG == 3 && GMVAE_P2_MFMA16) {
            // (plane_rounds2h: register r of block (i, j) is row r & 3 of the lane group's four in 16 x 16 tile r >> 2 of the block)
            const int row = wm0 + i * 32 + 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
            Cs[row * C::LDC + wn0 + j * 32 + 16 * ((r >> 2) & 1) + (lane & 15)] = acc[i][j][r];
          } else {
            const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            Cs[row * C::LDC + wn0 + j * 32 + l31] = acc[i][j][r];
          }
        }
    if (do_colsum) {
      if (did_planes) {                            // (k rows tid >> 4 + 16 i, columns 8 (tid & 15)..)
        if constexpr (C::CSP == 8) {
          *reinterpret_cast<float4*>(lds + C::CST + 8 * tid) = make_float4(cs8[0], cs8[1], cs8[2], cs8[3]);
          *reinterpret_cast<float4*>(lds + C::CST + 8 * tid + 4) = make_float4(cs8[4], cs8[5], cs8[6], cs8[7]);
        }
      } else if (did_bf16 || did_big) *reinterpret_cast<float4*>(lds + C::CST + 4 * tid) = cs4;     // (k rows tid>>4 + 16i, columns 4(tid&15)..)
      else lds[C::CST + tid] = csum;
    }
  }
  __syncthreads();
  if (do_colsum && tid < C::BN && n0 + tid < L.p[pi].N) {
    float v = 0.f;
    if (did_planes) {                              // column tid: threads (tid >> 3) + 16 g hold its chunk
#pragma unroll
      for (int g = 0; g < 16; ++g) v += lds[C::CST + 8 * ((tid >> 3) + 16 * g) + (tid & 7)];
    } else if (did_big) {                          // column tid: threads (tid >> 2) + 32 g hold its quad (k-quads g)
#pragma unroll
      for (int g = 0; g < 8; ++g) v += lds[C::CST + 4 * ((tid >> 2) + 32 * g) + (tid & 3)];
    } else if (did_bf16) {
#pragma unroll
      for (int g = 0; g < 16; ++g) v += lds[C::CST + 4 * ((tid >> 2) + 16 * g) + (tid & 3)];
    } else {
#pragma unroll
      for (int g = 0; g < CSG; ++g) v += lds[C::CST + g * C::BN + tid];
    }
    if constexpr (BIG == 3) v *= (L.p[pi].b_uns ? *L.p[pi].b_uns : 1.f) * L.p[pi].uns_cb;     // (the pieces carry b's scale)
    colsum_out[(long long)split * L.p[pi].split_stride + n0 + tid] = v;
  }

  GMVAE_GSTAMP(3);
  // ---- epilogue: each thread owns float4 groups of a row
  const int M = L.p[pi].M, N = L.p[pi].N, ldc = L.p[pi].ldc, epi = L.p[pi].epi;
  float* Cout = L.p[pi].C;
  const float* bias = L.p[pi].bias;
  const float* bias2 = L.p[pi].bias2;
  const float addconst = L.p[pi].addconst;
  constexpr int GPR = C::BN / 4;                 // groups per row (8, 16 or 32 consecutive lanes)
  constexpr int PASSES = C::BM * GPR / C::THREADS;
  if (epi == EPI_STORE) {
    const float* addsrc = L.p[pi].addsrc;
    const float* mask = L.p[pi].mask;
    const float* rowscale = L.p[pi].rowscale;
    const int relu = L.p[pi].relu, ld_add = L.p[pi].ld_add, add_div = L.p[pi].add_div, ld_mask = L.p[pi].ld_mask;
    const int mask_act = L.p[pi].mask_act;
    const long long soff = (long long)split * L.p[pi].split_stride;
    unsigned short* const C3s = L.p[pi].C3;        // (EPI_STORE with bias / ReLU: the activation also leaves as planes; never with splits)
    const long long c3ss = L.p[pi].c3_stride;
    // plain slab / matrix store of an interior tile (every weight-gradient tile but the edge ones): no per-element
    // options, all passes unrolled, 16-byte stores
    const bool plain = !bias && !bias2 && !addsrc && !mask && !rowscale && !relu && addconst == 0.f && m0 + C::BM <= M &&
                       n0 + C::BN <= N && (ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(Cout + soff) & 15) == 0);
    if (plain) {
      float* const dst0 = Cout + soff + (long long)m0 * ldc + n0;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int gidx = tid + ps * C::THREADS;
        const int row = gidx / GPR, c4 = gidx % GPR;
        float4 v4 = *reinterpret_cast<const float4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
        for (int w = 1; w < C::WK; ++w) {
          const float4 o = *reinterpret_cast<const float4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
          v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
        }
        *reinterpret_cast<float4*>(dst0 + (long long)row * ldc + 4 * c4) = v4;
      }
    } else if (m0 + C::BM <= M && n0 + C::BN <= N && (ldc & 3) == 0 && al16(Cout + soff) && (!bias || al16(bias)) && (!bias2 || al16(bias2)) &&
               (!addsrc || ((ld_add & 3) == 0 && al16(addsrc))) && (!mask || ((ld_mask & 3) == 0 && al16(mask)))) {
      // interior tile with options: 16-byte loads of every per-element option, ALL passes' loads in flight together (the
      // accumulators are staged, their registers are free): one pass at a time, each pass waits a full memory round trip --
      // ~20 % of a 16-round tile at the config-5 sizes, most of a 4-round one
      constexpr int RPP = C::THREADS / GPR;        // rows per pass; the column group is the same in every pass
      constexpr int PB = PASSES;          // every pass's loads in flight together: ONE memory round trip per tile
      const int c4 = tid % GPR, r0 = tid / GPR, nb = n0 + 4 * c4;
      float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias2) { const float4 c4v = *reinterpret_cast<const float4*>(bias2 + nb); b4.x += c4v.x; b4.y += c4v.y; b4.z += c4v.z; b4.w += c4v.w; }
#pragma unroll
      for (int pb = 0; pb < PASSES; pb += PB) {
        float4 a4[PB], k4[PB];
        float rs[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
          const int m = m0 + r0 + RPP * (pb + q);
          a4[q] = addsrc ? *reinterpret_cast<const float4*>(addsrc + (long long)(m / add_div) * ld_add + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
          k4[q] = mask ? *reinterpret_cast<const float4*>(mask + (long long)m * ld_mask + nb) : make_float4(1.f, 1.f, 1.f, 1.f);
          rs[q] = rowscale ? rowscale[m] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
          const int row = r0 + RPP * (pb + q);
          float4 v4 = *reinterpret_cast<const float4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
          for (int w = 1; w < C::WK; ++w) {
            const float4 o = *reinterpret_cast<const float4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
            v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
          }
          float v[4] = {v4.x + b4.x + a4[q].x + addconst, v4.y + b4.y + a4[q].y + addconst, v4.z + b4.z + a4[q].z + addconst,
                        v4.w + b4.w + a4[q].w + addconst};
          const float kk[4] = {k4[q].x, k4[q].y, k4[q].z, k4[q].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (relu) x = act_apply(x, relu);
            x = mask ? act_mask(x, kk[j], mask_act) : x;
            v[j] = x * rs[q];
          }
          *reinterpret_cast<float4*>(Cout + soff + (long long)(m0 + row) * ldc + nb) = make_float4(v[0], v[1], v[2], v[3]);
          if (C3s) {                                // the output also as planes of 16-bit pieces: the next layer's plane GEMM operand
            unsigned hi[2], mi[2], lo[2];
            split_pair(v[0], v[1], hi[0], mi[0], lo[0]);
            split_pair(v[2], v[3], hi[1], mi[1], lo[1]);
            unsigned short* const d3 = C3s + ((long long)(nb >> 4) * M + (m0 + row)) * 16 + (nb & 15);
            *reinterpret_cast<uint2*>(d3) = make_uint2(hi[0], hi[1]);
            *reinterpret_cast<uint2*>(d3 + c3ss) = make_uint2(mi[0], mi[1]);
            *reinterpret_cast<uint2*>(d3 + 2 * c3ss) = make_uint2(lo[0], lo[1]);
          }
        }
      }
    } else
#pragma unroll 1
    for (int ps = 0; ps < PASSES; ++ps) {
      const int gidx = tid + ps * C::THREADS;
      const int row = gidx / GPR, c4 = gidx % GPR;
      float4 v4 = *reinterpret_cast<const float4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
      for (int w = 1; w < C::WK; ++w) {
        const float4 o = *reinterpret_cast<const float4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
        v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
      }
      const int m = m0 + row, nb = n0 + 4 * c4;
      if (m < M && nb < N) {
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float* dst = Cout + soff + (long long)m * ldc + nb;
        const float rs = rowscale ? rowscale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = min(nb + j, N - 1);
          float x = v[j];
          if (bias) x += bias[n];
          if (bias2) x += bias2[n];
          if (addsrc) x += addsrc[(long long)(m / add_div) * ld_add + n];
          x += addconst;
          if (relu) x = act_apply(x, relu);
          if (mask) x = act_mask(x, mask[(long long)m * ld_mask + n], mask_act);
          v[j] = x * rs;
        }
        if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          dst[0] = v[0];
          if (nb + 1 < N) dst[1] = v[1];
          if (nb + 2 < N) dst[2] = v[2];
          if (nb + 3 < N) dst[3] = v[3];
        }
        if (C3s) {                                  // (edge / unaligned tiles: the pieces one by one)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (nb + j >= N) break;
            unsigned hi, mi, lo;
            split_pair(v[j], 0.f, hi, mi, lo);
            const long long o = ((long long)((nb + j) >> 4) * M + m) * 16 + ((nb + j) & 15);
            C3s[o] = (unsigned short)hi; C3s[o + c3ss] = (unsigned short)mi; C3s[o + 2 * c3ss] = (unsigned short)lo;
          }
        }
      }
    }
  } else {  // EPI_BERNOULLI: lambda -> (sigmoid(lambda) - x, sum_d x*lambda - softplus(lambda))
    const unsigned char* xp = L.p[pi].x;
    float* part = L.p[pi].part;
    const int ldx = L.p[pi].ldx, x_div = L.p[pi].x_div, nparts = L.p[pi].nparts;
    unsigned short* const C3 = L.p[pi].C3;         // (only set for launches whose tiles are all interior: planes_eligible)
    const long long c3s = L.p[pi].c3_stride;
    const float c3sc = L.p[pi].c3_scale;
    if (m0 + C::BM <= M && (n0 + C::BN <= N || ((N & 3) == 0 && !C3)) && (ldc & 3) == 0 && (!Cout || al16(Cout)) && al16(bias) && (!bias2 || al16(bias2)) &&
        (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & 3) == 0) {
      // (also the LAST column tile of an N that is no multiple of the tile -- 784 = 6 x 128 + 16: a seventh of the tiles of every
      //  Bernoulli launch at the reference's D --: column quads beyond N are predicated off instead of sending the whole tile
      //  down the element-by-element path below)
      // interior tile: the bias quad once, the 4 target bytes of a pass as one word; every pass's target word AND staged
      // accumulator quad are requested before the first is used (ONE memory and ONE LDS round trip per tile), and the row
      // sums meet through LDS at the end (a butterfly of 5 dependent cross-lane steps per pass was 80 LDS round trips per
      // tile, each several hundred cycles beside the co-resident workgroup's fragment reads: 22 us of a 77 us tile)
      constexpr int RPP = C::THREADS / GPR;
      const int c4 = tid % GPR, r0 = tid / GPR, nb = n0 + 4 * c4;
      const bool nv = nb < N;                      // (this thread's column quad exists)
      float4 b4 = nv ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias2 && nv) { const float4 c4v = *reinterpret_cast<const float4*>(bias2 + nb); b4.x += c4v.x; b4.y += c4v.y; b4.z += c4v.z; b4.w += c4v.w; }
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      unsigned xw[PASSES];
      f32x4 vv[PASSES];
#pragma unroll
      for (int q = 0; q < PASSES; ++q) {
        const int m = m0 + r0 + RPP * q;
        xw[q] = nv ? *reinterpret_cast<const unsigned*>(xp + (long long)(m / x_div) * ldx + nb) : 0u;
      }
#pragma unroll
      for (int q = 0; q < PASSES; ++q) {
        const int row = r0 + RPP * q;
        vv[q] = *reinterpret_cast<const f32x4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
        for (int w = 1; w < C::WK; ++w) vv[q] += *reinterpret_cast<const f32x4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
      }
      if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); GMVAE_GSTAMP(6); }
      // The passes, once per combination of outputs -- fp32 matrix or not, no planes / bf16 triples / f16 pairs -- with NO branch
      // inside: as run-time (uniform) conditions inside the loop they cut every pass into basic blocks of its own, and the
      // exp -> rcp -> log chain of a pass's 4 elements ran exposed, ~250 cycles per element (config 5: 12.5 us of a 31 us tile;
      // the forward-only evaluation's whole launch is this epilogue).  As one block the 64 chains of a thread interleave.
      const float bc[4] = {bb[0] + addconst, bb[1] + addconst, bb[2] + addconst, bb[3] + addconst};
      auto passes = [&](auto mode_c, auto cout_c) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool CO = decltype(cout_c)::value;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
          if (q == PASSES / 2) { GMVAE_GSTAMP(7); }
          const int row = r0 + RPP * q;
          // Per element ONE exp and ONE rcp: x l - softplus(l) = [x l - max(l, 0)] - log(1 + e), e = exp(-|l|), and the four
          // logs of a pass are one log of the product of the (1 + e) (each in [1, 2]) -- the bracket sums and the product run as
          // two chains of packed fp32 (v_pk_add / v_pk_fma / v_pk_mul_f32).  (The bare hardware forms: v_exp_f32 of an argument
          // <= 0 and v_log_f32 of a value in [1, 16] need none of the denormal-range fix-ups __expf / __logf wrap around them; e
          // flushes to 0 below 2^-126: sigmoid and softplus are exact to 1e-38 there.  Before: exp, rcp and log per element, 27
          // scalar-fp32 instructions -- the forward launch of the config-5 shard spent as many vector-pipe cycles in this epilogue
          // as matrix-pipe cycles in its 32 rounds, and the two do not overlap on a SIMD.)
          float v[4];
          f32x2_t brk = {0.f, 0.f}, prod = {1.f, 1.f};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2_t lam = f32x2_t{vv[q][2 * h], vv[q][2 * h + 1]} + f32x2_t{bc[2 * h], bc[2 * h + 1]};
            const f32x2_t xf = {(float)((xw[q] >> (16 * h)) & 0xffu), (float)((xw[q] >> (16 * h + 8)) & 0xffu)};
            const f32x2_t mx = {fmaxf(lam[0], 0.f), fmaxf(lam[1], 0.f)};
            brk += xf * lam - mx;
            const f32x2_t e = {__builtin_amdgcn_exp2f(fabsf(lam[0]) * -1.44269504088896341f), __builtin_amdgcn_exp2f(fabsf(lam[1]) * -1.44269504088896341f)};
            const f32x2_t ope = e + 1.f;
            prod *= ope;
            const f32x2_t rcp = {__builtin_amdgcn_rcpf(ope[0]), __builtin_amdgcn_rcpf(ope[1])};       // (only the sigmoid needs it)
            const f32x2_t er = e * rcp;
            const f32x2_t sg = f32x2_t{lam[0] >= 0.f ? rcp[0] : er[0], lam[1] >= 0.f ? rcp[1] : er[1]} - xf;
            v[2 * h] = sg[0]; v[2 * h + 1] = sg[1];
          }
          const float rsum = (brk[0] + brk[1]) - __builtin_amdgcn_logf(prod[0] * prod[1]) * 0.693147180559945309f;
          if constexpr (CO) { if (nv) *reinterpret_cast<float4*>(Cout + (long long)(m0 + row) * ldc + nb) = make_float4(v[0], v[1], v[2], v[3]); }
          if constexpr (MODE == 2) {                  // f16 pairs of (sigmoid - x) x c3_scale (|.| <= 1: a fixed scale), two planes
            unsigned q1[2], q2[2];
            split_f16pair(v[0] * c3sc, v[1] * c3sc, q1[0], q2[0]);
            split_f16pair(v[2] * c3sc, v[3] * c3sc, q1[1], q2[1]);
            unsigned short* const d3 = C3 + ((long long)(nb >> 4) * M + (m0 + row)) * 16 + (nb & 15);
            *reinterpret_cast<uint2*>(d3) = make_uint2(q1[0], q1[1]);
            *reinterpret_cast<uint2*>(d3 + c3s) = make_uint2(q2[0], q2[1]);
          } else if constexpr (MODE == 1) {           // the three 16-bit pieces of (sigmoid - x), one plane each (plane_rounds3's operand)
            unsigned hi[2], mi[2], lo[2];
            split_pair(v[0], v[1], hi[0], mi[0], lo[0]);
            split_pair(v[2], v[3], hi[1], mi[1], lo[1]);
            unsigned short* const d3 = C3 + ((long long)(nb >> 4) * M + (m0 + row)) * 16 + (nb & 15);
            *reinterpret_cast<uint2*>(d3) = make_uint2(hi[0], hi[1]);
            *reinterpret_cast<uint2*>(d3 + c3s) = make_uint2(mi[0], mi[1]);
            *reinterpret_cast<uint2*>(d3 + 2 * c3s) = make_uint2(lo[0], lo[1]);
          }
          lds[row * C::LDC + 4 * c4] = nv ? rsum : 0.f;          // (this thread's own, already consumed, slot of the staged tile)
        }
      };
      {
        typedef std::integral_constant<int, 0> M0; typedef std::integral_constant<int, 1> M1; typedef std::integral_constant<int, 2> M2;
        const int mode = !C3 ? 0 : (c3sc != 0.f ? 2 : 1);
        if (Cout) { if (mode == 2) passes(M2{}, std::true_type{}); else if (mode == 1) passes(M1{}, std::true_type{}); else passes(M0{}, std::true_type{}); }
        else { if (mode == 2) passes(M2{}, std::false_type{}); else if (mode == 1) passes(M1{}, std::false_type{}); else passes(M0{}, std::false_type{}); }
      }
      __syncthreads();
      if (tid < C::BM) {
        // (fp64: the row's partial sum grows to ~0.7 x BN; S > 1 turns the ABSOLUTE error of log w into a relative error
        //  of every gradient, and 32 sequential fp32 adds at that magnitude were its largest remaining piece)
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < GPR; ++c) t += (double)lds[tid * C::LDC + 4 * c];
        part[(long long)(m0 + tid) * nparts + tn] = (float)t;
      }
    } else
#pragma unroll 1
    for (int ps = 0; ps < PASSES; ++ps) {
      const int gidx = tid + ps * C::THREADS;
      const int row = gidx / GPR, c4 = gidx % GPR;
      float4 v4 = *reinterpret_cast<const float4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
      for (int w = 1; w < C::WK; ++w) {
        const float4 o = *reinterpret_cast<const float4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
        v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
      }
      const int m = m0 + row, nb = n0 + 4 * c4;
      float rsum = 0.f;
      if (m < M && nb < N) {
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const unsigned char* xr = xp + (long long)(m / x_div) * ldx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = min(nb + j, N - 1);
          // ONE exp, ONE rcp and ONE log per element give softplus AND the sigmoid (the hardware forms, ~1e-6 relative:
          // the ELBO tolerance is 1e-4; the libm log1pf / division forms cost ~4x the instructions and were 10 % of
          // this launch at the config-5 sizes)
          const float lam = v[j] + bias[n] + (bias2 ? bias2[n] : 0.f) + addconst;
          const float xv = (float)xr[n];
          const float e = fexp(-fabsf(lam));
          const float rcp = __builtin_amdgcn_rcpf(1.f + e);
          const float sp = fmaxf(lam, 0.f) - flog(rcp);
          rsum += (nb + j < N) ? xv * lam - sp : 0.f;
          v[j] = (lam >= 0.f ? rcp : e * rcp) - xv;
        }
        if (Cout) {
          float* dst = Cout + (long long)m * ldc + nb;
          if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            dst[0] = v[0];
            if (nb + 1 < N) dst[1] = v[1];
            if (nb + 2 < N) dst[2] = v[2];
            if (nb + 3 < N) dst[3] = v[3];
          }
        }
      }
#pragma unroll
      for (int o = GPR / 2; o > 0; o >>= 1) rsum += __shfl_xor(rsum, o, 64);
      if (m < M && c4 == 0) part[(long long)m * nparts + tn] = rsum;
    }
  }
  GMVAE_GSTAMP(4);
  if (L.dbg && tid == 0) L.dbg[(size_t)blockIdx.x * 8 + 5] = pi;
#undef GMVAE_GSTAMP
}

// The same split into plane_rounds3's layout: planes blocked by 16 along the contiguous dimension -- element (r, c) of the
// [rows][ld] source at ((c >> 4) rows + r) 16 + (c & 15).  A wave takes 16 rows x 32 columns (two blocks): lane l reads the 8
// floats (row l >> 2, columns 8 (l & 3)..) -- 4 lanes cover one whole 128-byte line -- and writes ONE 16-byte chunk per plane;
// a store instruction of the wave then covers two contiguous 512-byte pieces (rows % 16 == 0, ld % 16 == 0; the upper block of
// a 16-column remainder is masked).
__global__ __launch_bounds__(256) void split_planes_b16(const float* __restrict__ src, const float* __restrict__ rowscale,
                                                         const int ld, const int rows, unsigned short* __restrict__ dst,
                                                         const long long pstride) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int npair = (ld + 31) >> 5, ngrp = rows >> 4, lane = threadIdx.x & 63;
  const long long waves = (long long)ngrp * npair;
  for (long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < waves; wv += (long long)gridDim.x * 4) {
    const long long cp = wv / ngrp, rg = wv - cp * ngrp;       // row groups fastest: consecutive waves write consecutive pieces
    const long long r = rg * 16 + (lane >> 2);
    const int c = (int)cp * 32 + 8 * (lane & 3);
    if (c >= ld) continue;
    const float* const sp = src + r * ld + c;
    const float sc = rowscale ? rowscale[r] : 1.f;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(sp), q1 = *reinterpret_cast<const f32x4*>(sp + 4);
    const float v[8] = {q0.x * sc, q0.y * sc, q0.z * sc, q0.w * sc, q1.x * sc, q1.y * sc, q1.z * sc, q1.w * sc};
    unsigned hi[4], mi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], hi[j], mi[j], lo[j]);
    unsigned short* const dp = dst + ((long long)(c >> 4) * rows + r) * 16 + (c & 15);
    *reinterpret_cast<u32x4*>(dp) = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(dp + pstride) = u32x4{mi[0], mi[1], mi[2], mi[3]};
    *reinterpret_cast<u32x4*>(dp + 2 * pstride) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// The same for f16 pairs (plane_rounds2): dst receives two planes of (value x rowscale x s), s = pair_scale(*amax_bits) -- the
// power-of-two scale of the tensor, from the bits of its largest magnitude (amax_final below; of the UNWEIGHTED tensor: row
// weights are <= 1).
__global__ __launch_bounds__(256) void split_pairs_b16(const float* __restrict__ src, const float* __restrict__ rowscale,
                                                        const int ld, const int rows, unsigned short* __restrict__ dst,
                                                        const long long pstride, const unsigned* __restrict__ amax_bits,
                                                        const int ldp) {
  // (ldp >= ld: the planes hold ldp columns, the ones from ld on zero -- a GEMM whose N is no multiple of its tile reads whole
  //  tiles of the weight's planes; ld % 8 == 0)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const float s = pair_scale(*amax_bits);
  const int npair = (ldp + 31) >> 5, ngrp = rows >> 4, lane = threadIdx.x & 63;
  const long long waves = (long long)ngrp * npair;
  for (long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < waves; wv += (long long)gridDim.x * 4) {
    const long long cp = wv / ngrp, rg = wv - cp * ngrp;
    const long long r = rg * 16 + (lane >> 2);
    const int c = (int)cp * 32 + 8 * (lane & 3);
    if (c >= ldp) continue;
    const bool in = c < ld;
    const float* const sp = src + r * ld + (in ? c : 0);
    const float sc = in ? (rowscale ? rowscale[r] : 1.f) * s : 0.f;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(sp), q1 = *reinterpret_cast<const f32x4*>(sp + 4);
    const float v[8] = {q0.x * sc, q0.y * sc, q0.z * sc, q0.w * sc, q1.x * sc, q1.y * sc, q1.z * sc, q1.w * sc};
    unsigned p1[4], p2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_f16pair(v[2 * j], v[2 * j + 1], p1[j], p2[j]);
    unsigned short* const dp = dst + ((long long)(c >> 4) * rows + r) * 16 + (c & 15);
    *reinterpret_cast<u32x4*>(dp) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(dp + pstride) = u32x4{p2[0], p2[1], p2[2], p2[3]};
  }
}

constexpr int kAmaxBlocks = 256;      // workgroups of an amax_abs launch
// bits of the largest |src[i]| (n % 4 == 0, 16-byte aligned) of each workgroup's share into part[blockIdx.x]
__global__ __launch_bounds__(256) void amax_abs(const float* __restrict__ src, const long long n4, unsigned* __restrict__ part) {
  __shared__ unsigned wm[4];
  unsigned m = 0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f32x4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = i + j * stride < n4 ? reinterpret_cast<const f32x4*>(src)[i + j * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned a = __float_as_uint(q[j].x) & 0x7fffffffu, b = __float_as_uint(q[j].y) & 0x7fffffffu;
      const unsigned c = __float_as_uint(q[j].z) & 0x7fffffffu, d = __float_as_uint(q[j].w) & 0x7fffffffu;
      const unsigned ab = a > b ? a : b, cd = c > d ? c : d, e = ab > cd ? ab : cd;
      m = e > m ? e : m;
    }
  }
  m = wave_umax(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3];
    part[blockIdx.x] = a > b ? a : b;
  }
}

// workgroup t folds tensor t's partial maxima into bits[t] and leaves 1 / pair_scale in uns[t] (t = 0, 1)
struct AmaxFinalArgs {
  const unsigned* part[2];
  int n[2];
  unsigned* bits;
  float* uns;
};
__global__ __launch_bounds__(256) void amax_final(const AmaxFinalArgs a) {
  __shared__ unsigned wm[4];
  const int t = blockIdx.x;
  unsigned m = 0;
  for (int i = threadIdx.x; i < a.n[t]; i += 256) { const unsigned v = a.part[t][i]; m = v > m ? v : m; }
  m = wave_umax(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = wm[0] > wm[1] ? wm[0] : wm[1], y = wm[2] > wm[3] ? wm[2] : wm[3], mm = x > y ? x : y;
    a.bits[t] = mm;
    a.uns[t] = 1.f / pair_scale(mm);
  }
}

}  // namespace gmvae
