// The "skinny" schedule: the GMVAE training step for a SMALL batch and WIDE hidden layers -- the one training
// configuration the reference ships (bin/run_train.sh:3-14: batch 64, hidden 512, latent 128; scripts/gmvae.py:238-267 and
// its reverse pass, scripts/runners.py:181-183 for the optimizer).
//
// At 64 rows the activations are a few hundred KB and the 5 MB of weights are the operands that matter: every layer is
// a [B <= 128] x K x N product whose weight matrix should cross the fabric ONCE, spread over as many CUs as the layer
// has 16-column tiles, and whose activation operand is L2-resident.  The general schedule ran each such layer through
// the grouped GEMM's 32 x 32 x 128 tiles (7 serial staging rounds for K = 784: 11 us per launch, 19 launches, 185 us
// per step).  Here a workgroup owns a [16 rows] x [16 .. 64 columns] output tile for the whole contraction, its 8 waves
// split the contraction, and operands go global -> registers -> matrix core with 16-byte loads along the contraction:
// which element of the contraction is MFMA step q of lane group h is free as long as A and B agree, so a lane's four
// consecutive k (one 16-byte load of the activation row, or 4 bytes of the uint8 batch) serve four consecutive
// v_mfma_f32_16x16x4_f32 with k = 16 g + 4 h + q.  Partial tiles meet in LDS in wave order (bit-reproducible) and the
// layer's row-wise work (bias, ReLU, the q / prior heads, the Bernoulli term, their reverse forms) is the epilogue of the
// tile that produced its inputs.  10 launches:
//   F1 first layers over the uint8 batch (split over the contraction into ns1 slabs)   F2 y path, one wave per row
//   F3 q head + z + log q / log p partials   F4 decoder hidden   F5 decoder output + Bernoulli term + g = sigmoid - x
//   B1 dhd   B2 dz + the heads' reverse   B3 dhg   B4 dy -> dlogits -> dhy, one wave per row
//   W  every weight / bias gradient (contraction over the <= 128 batch rows: one 64 x 16 tile per WAVE) + TF-Adam + loss tail
// Bound: HBM/L2 latency per launch (a launch moves 0.1 - 2 MB), then the optimizer's 9 x 4 P bytes (47 MB at H = 512).
#pragma once
#include "dwadam.hpp"

namespace gmvae {

constexpr int kSkThreads = 512, kSkWaves = 8;
constexpr int kSkNs1x = 4;      // upper bound of SkArgs::ns1 (first-layer slabs)

struct SkTensor {              // one weight tensor of the W launch
  const void* A;               // [B][lda] uint8 or fp32: the layer's input rows
  const float* dY;             // [B][ldy]: the pre-activation gradients
  int lda, ldy, M, N, a_u8;
  int w_off, b_off;            // flat parameter offsets of W [M][N] and of its bias [N] (b_off < 0: none)
  int tiles_n, tile_begin;
  int vec;                     // N % 4 == 0: [16 x 64] tiles, 4 consecutive columns per lane (16-byte optimizer accesses)
};
constexpr int kSkMaxT = 10;

struct SkArgs {
  int model;                   // 2 GMVAE (ten launches); 0 VAE with the standard-normal prior (scripts/vae.py:167-185): no y path -- F1 is
                               // the whole first layer (bias + ReLU in its epilogue, the eps rows in extra workgroups), the "y" / "g"
                               // names below then all mean the one encoder: Wy0 / by0 its first layer, Wg1 / bg1 its head, hy = hg its
                               // hidden activation, dhg its gradient; pp, dpp, nent, s1 are unused (eight launches)
  int B, D, H, L, K, K4;       // K4 = pad4(K): row stride of y and dlogits
  int ns1, nparts;             // first-layer slabs; logpx partials per row
  float c, smin, invT, gen_bias;
  const float* gen_bias_vec;
  const unsigned char* x;
  const float* P;              // flat parameters
  long long Wy0, by0, Wy1, by1, Wp, bp, Wg0, bg0, Wg1, bg1, Wd0, bd0, Wd1, bd1;
  float *s1, *hy, *hg, *y, *logits, *nent, *pp, *qp, *z, *hd, *g, *part, *lqp;
  float *dhd, *dqp, *dpp, *dhg, *dlogits, *dhy;
  const float *eps, *u;        // the step's noise (external arrays, or the workspace arrays F2 fills when gen_noise)
  float *eps_w, *u_w;
  int gen_eps, gen_u;          // F2 draws that array with Philox (aux.hpp noise_vals) instead of reading an external one
  unsigned long long seed, step, row0;
  unsigned long long* step_dev;
  // W launch
  int ntens, total_tiles;
  const float* resp;           // VAE_GMP: the mixture prior's responsibilities [B][K] (mixture_logprob_*), and where its variables
  long long gmp_raw, gmp_mix;  //   lie in the flat layout (loc at gmp_off)
  float* dwp;                  // sk_dwc: dw_ks partial gradients (one per batch share) in the flat layout, dwp_stride floats apart,
  long long dwp_stride;        //   and a counter of arrived shares per tile (zeroed by the step's first launch)
  int dw_ks;
  unsigned* dw_cnt;
  int has_tail;                // W launch: its last workgroup is the loss tail (the launch that ends the step)
  SkTensor t[kSkMaxT];
  float *grads, *ap, *am, *av; // ap != null: TF-Adam in the epilogue
  float lr, b1, b2, aeps, ln_b1, ln_b2;
  float *tail, *tail_log;
  float *logpx, *logq, *logp, *logw;
  // model 1 (VAE_GMP, the learned mixture prior of scripts/vae.py:231-244): log p(z) is not column-local -- F3 leaves only
  // log q, mixture_logprob_* (kernels.hpp) writes logp[row] and the responsibilities, B2 leaves dz in `dz` for z_head_bwd,
  // gmp_param_bwd leaves gmp_n partial gradients of the prior's variables (rows of gmp_len floats at flat offset gmp_off),
  // which the loss-tail workgroup sums and updates
  float* dz;
  const float* gmp_part;
  int gmp_n, gmp_len;
  long long gmp_off;
  unsigned long long* dbg;     // diagnostic (GMVAE_SK_STAMPS): [10 launches][kSkDbgWgs blocks][8] device wall-clock stamps (100 MHz)
};
constexpr int kSkDwShares = 2;         // batch shares of a tile in sk_dwc (their partial gradients: w.slabs)
constexpr int kSkDwcMaxTiles = 4096;   // sk_dwc's per-tile arrival counters (H = 1024, D = 3072: 3 x 768 + small tensors)
constexpr int kSkDbgWgs = 1024;   // workgroups per launch that leave stamps
#define SK_STAMP(slot, i) if (a.dbg && threadIdx.x == 0 && blockIdx.x < kSkDbgWgs) a.dbg[((size_t)(slot) * kSkDbgWgs + blockIdx.x) * 8 + (i)] = wall_clock64()

// (F5W, B1W: the two D-wide layers in 64-column tiles, for batches that fill the chip without the narrow tiles' extra workgroups)
enum { SK_F1 = 0, SK_F3, SK_F4, SK_F5, SK_B1, SK_B2, SK_B3, SK_F5W, SK_B1W };

// ---- contraction pieces: one wave's share (k-groups kg = kg_lo + wave, + 8, ... < kg_hi; a k-group = 16 contraction steps).
// The wave's groups run in batches of 4, 2, 1 (compile-time sizes; MAXG bounds them): every load of a batch is in flight
// before its first MFMA, and no matrix instruction is spent on an absent group (a layer with K = 128 has ONE group per wave).
// RT: row tiles of 16 batch rows that share the wave's W fragments (1 at <= 128 rows; 2 or 4 where the batch fills the chip
// anyway: a W fragment loaded once then feeds RT x as many MFMAs -- at B = 1024 the one-tile form ran at the L2's rate).
template <int MAXG, class F>
__device__ __forceinline__ void sk_groups(const int kg_lo, const int kg_hi, const int wave, F&& body) {
  const int kg = kg_lo + wave;
  const int n = kg < kg_hi ? (kg_hi - kg + kSkWaves - 1) / kSkWaves : 0;
  int done = 0;
  if constexpr (MAXG >= 4) {
    for (; n - done >= 4; done += 4) body(std::integral_constant<int, 4>{}, kg + kSkWaves * done);
    if (n - done >= 2) { body(std::integral_constant<int, 2>{}, kg + kSkWaves * done); done += 2; }
  } else if constexpr (MAXG >= 2) {
    for (; n - done >= 2; done += 2) body(std::integral_constant<int, 2>{}, kg + kSkWaves * done);
  } else {
    for (; done < n; ++done) body(std::integral_constant<int, 1>{}, kg + kSkWaves * done);
  }
  if (n - done >= 1) body(std::integral_constant<int, 1>{}, kg + kSkWaves * done);
}
template <int RT> constexpr int sk_maxg() { return RT == 1 ? 4 : RT == 2 ? 2 : 1; }
// RT > 1: one group per stage, two register sets: the next group's loads are in flight under this group's RT x as many MFMAs.
// (Stages of two groups at RT = 2 measured neutral -- 160 VGPRs, one workgroup per CU -- and were dropped.)
template <int RT, int NB> struct SkFrag { float4 av[RT]; float4 bv[NB]; };
template <class Frag, bool PIPE = true, class LD, class MM>
__device__ __forceinline__ void sk_pipe(const int kg_lo, const int kg_hi, const int wave, LD&& ld, MM&& mm) {
  const int kg = kg_lo + wave;
  const int n = kg < kg_hi ? (kg_hi - kg + kSkWaves - 1) / kSkWaves : 0;
  if (n == 0) return;
  if constexpr (!PIPE) {                          // one register set: the 6-product forms (two sets: 188 registers at RT = 2 and
                                                  // measured 0.5 us slower there, spills at RT = 4)
    for (int i = 0; i < n; ++i) {
      Frag f;
      ld(f, kg + kSkWaves * i);
      __builtin_amdgcn_sched_barrier(0);
      mm(f);
    }
    return;
  }
  Frag fa, fb;
  ld(fa, kg);
  int i = 0;
  for (; i + 2 <= n; i += 2) {
    ld(fb, kg + kSkWaves * (i + 1));
    __builtin_amdgcn_sched_barrier(0);
    mm(fa);
    __builtin_amdgcn_sched_barrier(0);
    if (i + 2 < n) ld(fa, kg + kSkWaves * (i + 2));
    __builtin_amdgcn_sched_barrier(0);
    mm(fb);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (i < n) mm(fa);
}
// 8 fp32 values -> their three bf16 pieces (truncation splits with exact residuals: hi + mid + lo == v bit for bit, dwadam.hpp),
// packed pairwise: elements 2 j, 2 j + 1 in word j of each piece's MFMA operand
typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sk_split8(const float (&v)[8], sk_u32x4& h, sk_u32x4& m, sk_u32x4& l) {
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const float v0 = v[2 * jp], v1 = v[2 * jp + 1];
    h[jp] = pack_hi16(v1, v0);
    const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
    m[jp] = pack_hi16(r1, r0);
    const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    l[jp] = pack_hi16(s1, s0);
  }
}
// F1 above 128 rows (RT > 1): the uint8 batch times an fp32 weight on the bf16 matrix cores, exact as dw_adam's uint8 problems
// (dwadam.hpp): x is exact in bf16 and W = hi + mid + lo, so a group of 32 contraction steps is 3 v_mfma_f32_16x16x32_bf16 (16
// cycles) per [16 x 16] block instead of 8 v_mfma_f32_16x16x4_f32 (32 cycles); the lane's 8 steps are x[row][32 g + 8 lk ..
// + 7] (one 8-byte load) and W[32 g + 8 lk + e][4 strided columns] (8 16-byte loads, split into pieces beside the MFMAs).
// kmax (a multiple of 16): the last group may be half empty -- its addresses are clamped and its A values zero.
template <int RT> struct SkFragB { uint2 aw[RT]; float4 bv[8]; };
template <int RT>
__device__ __forceinline__ void sk_nn4_u8bf(const unsigned char* __restrict__ X, const long long (&arow)[RT], const float* __restrict__ W,
                                            const int ldw, const int ncol, const int g_lo, const int g_hi, const int wave, const int lk,
                                            const int kmax, f32x4 (&acc)[RT][4]) {
  sk_pipe<SkFragB<RT>>(g_lo, g_hi, wave,
    [&](SkFragB<RT>& f, const int gi) {
      const int k0 = 32 * gi + 8 * lk;
      const int k = min(k0, kmax - 8);
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        f.aw[j] = *reinterpret_cast<const uint2*>(X + arow[j] + k);
        if (k0 >= kmax) f.aw[j] = make_uint2(0u, 0u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f.bv[e] = *reinterpret_cast<const float4*>(W + (long long)(k + e) * ldw + ncol);
    },
    [&](const SkFragB<RT>& f) {
      sk_bf16x8 A[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        sk_u32x4 w;
        w[0] = pack_hi16((float)((f.aw[j].x >> 8) & 0xffu), (float)(f.aw[j].x & 0xffu));
        w[1] = pack_hi16((float)(f.aw[j].x >> 24), (float)((f.aw[j].x >> 16) & 0xffu));
        w[2] = pack_hi16((float)((f.aw[j].y >> 8) & 0xffu), (float)(f.aw[j].y & 0xffu));
        w[3] = pack_hi16((float)(f.aw[j].y >> 24), (float)((f.aw[j].y >> 16) & 0xffu));
        A[j] = __builtin_bit_cast(sk_bf16x8, w);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t == 0 ? f.bv[e].x : t == 1 ? f.bv[e].y : t == 2 ? f.bv[e].z : f.bv[e].w;
        sk_u32x4 h, m, l;
        sk_split8(v, h, m, l);
        const sk_bf16x8 Bh = __builtin_bit_cast(sk_bf16x8, h), Bm = __builtin_bit_cast(sk_bf16x8, m), Bl = __builtin_bit_cast(sk_bf16x8, l);
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], Bl, acc[j][t], 0, 0, 0);      // smallest pieces first
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], Bm, acc[j][t], 0, 0, 0);
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], Bh, acc[j][t], 0, 0, 0);
        }
      }
    });
}
// fp32 x fp32 on the bf16 matrix cores for the two D-wide layers above 128 rows: both operands as three pieces, 6 piece products
// per product (the three dropped ones <= 2^-23 of it: gemm.hpp's plane form), 6 x 16 MFMA cycles per group of 32 contraction steps
// and [16 x 16] block instead of 8 x 32; the splits run on the vector pipe beside the other waves' MFMAs.
__device__ __forceinline__ void sk_mma6(const sk_bf16x8 (&A)[3], const sk_bf16x8 (&B)[3], f32x4& c) {      // pieces [0] hi, [1] mid, [2] lo
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], c, 0, 0, 0);      // smallest products first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], c, 0, 0, 0);
}
__device__ __forceinline__ void sk_pieces(const float (&v)[8], sk_bf16x8 (&P)[3]) {
  sk_u32x4 h, m, l;
  sk_split8(v, h, m, l);
  P[0] = __builtin_bit_cast(sk_bf16x8, h); P[1] = __builtin_bit_cast(sk_bf16x8, m); P[2] = __builtin_bit_cast(sk_bf16x8, l);
}
// NN: A rows k-contiguous (two 16-byte loads per 8 steps), W k-major, 4 strided column tiles (eight 16-byte loads)
template <int RT> struct SkFragN6 { float4 av[RT][2]; float4 bv[8]; };
template <int RT>
__device__ __forceinline__ void sk_nn4_bf6(const float* __restrict__ A, const long long (&arow)[RT], const float* __restrict__ W, const int ldw,
                                           const int ncol, const int g_hi, const int wave, const int lk, f32x4 (&acc)[RT][4]) {
  sk_pipe<SkFragN6<RT>, false>(0, g_hi, wave,
    [&](SkFragN6<RT>& f, const int gi) {
      const int k = 32 * gi + 8 * lk;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        f.av[j][0] = *reinterpret_cast<const float4*>(A + arow[j] + k);
        f.av[j][1] = *reinterpret_cast<const float4*>(A + arow[j] + k + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f.bv[e] = *reinterpret_cast<const float4*>(W + (long long)(k + e) * ldw + ncol);
    },
    [&](const SkFragN6<RT>& f) {
      sk_bf16x8 Ap[RT][3];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float v[8] = {f.av[j][0].x, f.av[j][0].y, f.av[j][0].z, f.av[j][0].w, f.av[j][1].x, f.av[j][1].y, f.av[j][1].z, f.av[j][1].w};
        sk_pieces(v, Ap[j]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t == 0 ? f.bv[e].x : t == 1 ? f.bv[e].y : t == 2 ? f.bv[e].z : f.bv[e].w;
        sk_bf16x8 Bp[3];
        sk_pieces(v, Bp);
#pragma unroll
        for (int j = 0; j < RT; ++j) sk_mma6(Ap[j], Bp, acc[j][t]);
      }
    });
}
// NT: both operands contraction-contiguous (two 16-byte loads per 8 steps each)
template <int RT, int NU> struct SkFragT6 { float4 av[RT][2]; float4 bv[NU][2]; };
template <int RT, int NU>
__device__ __forceinline__ void sk_nt_bf6(const float* __restrict__ A, const long long (&arow)[RT], const float* __restrict__ W, const int ldw,
                                          const int (&wrow)[NU], const int g_hi, const int wave, const int lk, const int kmax,
                                          f32x4 (&acc)[RT][4]) {
  sk_pipe<SkFragT6<RT, NU>, false>(0, g_hi, wave,
    [&](SkFragT6<RT, NU>& f, const int gi) {
      const int k0 = 32 * gi + 8 * lk;
      const int k = min(k0, kmax - 8);            // (kmax a multiple of 16: the last group may be half empty: clamped, A zero)
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        f.av[j][0] = *reinterpret_cast<const float4*>(A + arow[j] + k);
        f.av[j][1] = *reinterpret_cast<const float4*>(A + arow[j] + k + 4);
        if (k0 >= kmax) { f.av[j][0] = make_float4(0.f, 0.f, 0.f, 0.f); f.av[j][1] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        f.bv[t][0] = *reinterpret_cast<const float4*>(W + (long long)wrow[t] * ldw + k);
        f.bv[t][1] = *reinterpret_cast<const float4*>(W + (long long)wrow[t] * ldw + k + 4);
      }
    },
    [&](const SkFragT6<RT, NU>& f) {
      sk_bf16x8 Ap[RT][3];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float v[8] = {f.av[j][0].x, f.av[j][0].y, f.av[j][0].z, f.av[j][0].w, f.av[j][1].x, f.av[j][1].y, f.av[j][1].z, f.av[j][1].w};
        sk_pieces(v, Ap[j]);
      }
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        const float v[8] = {f.bv[t][0].x, f.bv[t][0].y, f.bv[t][0].z, f.bv[t][0].w, f.bv[t][1].x, f.bv[t][1].y, f.bv[t][1].z, f.bv[t][1].w};
        sk_bf16x8 Bp[3];
        sk_pieces(v, Bp);
#pragma unroll
        for (int j = 0; j < RT; ++j) sk_mma6(Ap[j], Bp, acc[j][t]);
      }
    });
}
// NN, 4 strided column tiles: out[row][n0 + 4 i + t] for lane column i: W k-major [K][ldw], one 16-byte load of W per k
// MK: the contraction extent kmax is no multiple of 16 (a multiple of 4): quads at k >= kmax are loaded from the last valid quad
// (clamped address, branch-free) and their A values replaced by 0
template <bool U8, int RT, bool MK = false>
__device__ __forceinline__ void sk_nn4(const void* __restrict__ Ap, const long long (&arow)[RT], const float* __restrict__ W, const int ldw,
                                       const int ncol, const int kg_lo, const int kg_hi, const int wave, const int lk,
                                       f32x4 (&acc)[RT][4], const int kmax = 0) {
  if constexpr (RT > 1) {
    sk_pipe<SkFrag<RT, 4>>(kg_lo, kg_hi, wave,
      [&](SkFrag<RT, 4>& f, const int kgi) {
        const int k0 = 16 * kgi + 4 * lk;
        const int k = MK ? min(k0, kmax - 4) : k0;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if constexpr (U8) f.av[j].x = __uint_as_float(*reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(Ap) + arow[j] + k));
          else f.av[j] = *reinterpret_cast<const float4*>(static_cast<const float*>(Ap) + arow[j] + k);
          if (MK && k0 >= kmax) f.av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) f.bv[q] = *reinterpret_cast<const float4*>(W + (long long)(k + q) * ldw + ncol);
      },
      [&](const SkFrag<RT, 4>& f) {
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          float aq[4];
          if constexpr (U8) {
            const unsigned w = __float_as_uint(f.av[j].x);
            aq[0] = (float)(w & 0xffu); aq[1] = (float)((w >> 8) & 0xffu); aq[2] = (float)((w >> 16) & 0xffu); aq[3] = (float)(w >> 24);
          } else {
            aq[0] = f.av[j].x; aq[1] = f.av[j].y; aq[2] = f.av[j].z; aq[3] = f.av[j].w;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], f.bv[q].x, acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], f.bv[q].y, acc[j][1], 0, 0, 0);
            acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], f.bv[q].z, acc[j][2], 0, 0, 0);
            acc[j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], f.bv[q].w, acc[j][3], 0, 0, 0);
          }
        }
      });
    return;
  }
  sk_groups<sk_maxg<RT>()>(kg_lo, kg_hi, wave, [&](auto ng, const int kgb) {
    constexpr int NG = decltype(ng)::value;
    float4 av[NG][RT], bv[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k0 = 16 * (kgb + g * kSkWaves) + 4 * lk;
      const int k = MK ? min(k0, kmax - 4) : k0;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        if constexpr (U8) {
          const unsigned w = *reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(Ap) + arow[j] + k);
          av[g][j] = make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
        } else {
          av[g][j] = *reinterpret_cast<const float4*>(static_cast<const float*>(Ap) + arow[j] + k);
        }
        if (MK && k0 >= kmax) av[g][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[g][q] = *reinterpret_cast<const float4*>(W + (long long)(k + q) * ldw + ncol);
    }
    __builtin_amdgcn_sched_barrier(0);             // every load of the batch is issued before its first MFMA
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float aq[4] = {av[g][j].x, av[g][j].y, av[g][j].z, av[g][j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], bv[g][q].x, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], bv[g][q].y, acc[j][1], 0, 0, 0);
          acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], bv[g][q].z, acc[j][2], 0, 0, 0);
          acc[j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], bv[g][q].w, acc[j][3], 0, 0, 0);
        }
      }
    }
  });
}
// NN, NU plain column tiles at columns col[t] (4-byte loads of W: the q head's (mu, raw) column pairs)
template <int NU, int RT>
__device__ __forceinline__ void sk_nnp(const float* __restrict__ A, const long long (&arow)[RT], const float* __restrict__ W, const int ldw,
                                       const int (&col)[NU], const int kg_lo, const int kg_hi, const int wave, const int lk,
                                       f32x4 (&acc)[RT][4]) {
  sk_groups<sk_maxg<RT>()>(kg_lo, kg_hi, wave, [&](auto ng, const int kgb) {
    constexpr int NG = decltype(ng)::value;
    float4 av[NG][RT];
    float bv[NG][4][NU];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = 16 * (kgb + g * kSkWaves) + 4 * lk;
#pragma unroll
      for (int j = 0; j < RT; ++j) av[g][j] = *reinterpret_cast<const float4*>(A + arow[j] + k);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NU; ++t) bv[g][q][t] = W[(long long)(k + q) * ldw + col[t]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float aq[4] = {av[g][j].x, av[g][j].y, av[g][j].z, av[g][j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < NU; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q], bv[g][q][t], acc[j][t], 0, 0, 0);
      }
    }
  });
}
// NT (data gradients): out[row][j] = sum_c A[row][c] W[j][c], W rows contraction-contiguous: 16-byte loads of both
template <int NU, int RT, bool MK = false>
__device__ __forceinline__ void sk_nt(const float* __restrict__ A, const long long (&arow)[RT], const float* __restrict__ W, const int ldw,
                                      const int (&wrow)[NU], const int kg_lo, const int kg_hi, const int wave, const int lk,
                                      f32x4 (&acc)[RT][4], const int kmax = 0) {
  if constexpr (RT > 1) {
    sk_pipe<SkFrag<RT, NU>>(kg_lo, kg_hi, wave,
      [&](SkFrag<RT, NU>& f, const int kgi) {
        const int k0 = 16 * kgi + 4 * lk;
        const int k = MK ? min(k0, kmax - 4) : k0;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          f.av[j] = *reinterpret_cast<const float4*>(A + arow[j] + k);
          if (MK && k0 >= kmax) f.av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < NU; ++t) f.bv[t] = *reinterpret_cast<const float4*>(W + (long long)wrow[t] * ldw + k);
      },
      [&](const SkFrag<RT, NU>& f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)                  // (contraction step outermost: consecutive MFMAs never share an accumulator:
#pragma unroll                                       //  40 cycles of dependent latency against 32 of issue)
          for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int t = 0; t < NU; ++t) {
              const float av = q == 0 ? f.av[j].x : q == 1 ? f.av[j].y : q == 2 ? f.av[j].z : f.av[j].w;
              const float bv = q == 0 ? f.bv[t].x : q == 1 ? f.bv[t].y : q == 2 ? f.bv[t].z : f.bv[t].w;
              acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j][t], 0, 0, 0);
            }
      });
    return;
  }
  sk_groups<sk_maxg<RT>()>(kg_lo, kg_hi, wave, [&](auto ng, const int kgb) {
    constexpr int NG = decltype(ng)::value;
    float4 av[NG][RT], bv[NG][NU];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k0 = 16 * (kgb + g * kSkWaves) + 4 * lk;
      const int k = MK ? min(k0, kmax - 4) : k0;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        av[g][j] = *reinterpret_cast<const float4*>(A + arow[j] + k);
        if (MK && k0 >= kmax) av[g][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int t = 0; t < NU; ++t) bv[g][t] = *reinterpret_cast<const float4*>(W + (long long)wrow[t] * ldw + k);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < RT; ++j)
#pragma unroll
        for (int t = 0; t < NU; ++t) {
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][j].x, bv[g][t].x, acc[j][t], 0, 0, 0);
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][j].y, bv[g][t].y, acc[j][t], 0, 0, 0);
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][j].z, bv[g][t].z, acc[j][t], 0, 0, 0);
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][j].w, bv[g][t].w, acc[j][t], 0, 0, 0);
        }
    }
  });
}

// launch outputs leave write-through (mega2.hpp st4o): nothing waits dirty in L2 for the end-of-kernel write-back, which
// at ten short launches per step is most of a step (measured: tools/micro/launch_floor.hip, profiles/round3_notes.md)
// (st1o -- a 4-byte launch output, write-through -- lives in mega2.hpp)
// ReLU that keeps a NaN (fmaxf(NaN, 0) is 0: a non-finite pre-activation would vanish from the loss the runner watches)
// (relu_nan: gemm.hpp)

__device__ __forceinline__ float sk_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// One matrix-product launch of the schedule.  Grid: column tiles x groups of RT row tiles (x ns1 slabs for F1); a workgroup =
// RT 16-row output tiles of one column tile; tiles are NN4 (16 x 64, strided columns), pairs (F3) or plain 16-column tiles
// (NT forms).  The 8 waves split the contraction; their partial tiles meet in LDS two row tiles at a time (RT = 1: the one),
// thread (jl, er, el) of the meeting owning accumulator register er of lane slot el of row tile 2 c + jl and its epilogue.
template <int ST, int RT>
__global__ __launch_bounds__(kSkThreads) void sk_gemm(const SkArgs a) {
  constexpr int NU = (ST == SK_F1 || ST == SK_F4 || ST == SK_F5W || ST == SK_B1W) ? 4 : (ST == SK_F3 || ST == SK_B3) ? 2 : 1;
  constexpr int CJ = RT == 1 ? 1 : 2, NC = RT / CJ;                            // row tiles per meeting, meetings
  __shared__ __attribute__((aligned(16))) float red[kSkWaves * CJ * NU * 4 * 64];      // [wave][jl][4 t + r][lane]
  constexpr int SLOT = ST == SK_F1 ? 0 : ST == SK_F3 ? 2 : ST == SK_F4 ? 3 : (ST == SK_F5 || ST == SK_F5W) ? 4 : (ST == SK_B1 || ST == SK_B1W) ? 5 : ST == SK_B2 ? 6 : 7;
  SK_STAMP(SLOT, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, D = a.D, H = a.H, L = a.L;
  const int nrt = (((B + 15) >> 4) + RT - 1) / RT;      // groups of RT row tiles
  const bool vae = a.model != 2;                 // (VAE and VAE_GMP: one encoder, no y path)
  const int nct = ST == SK_F1 ? (vae ? H : 2 * H) / 64 : ST == SK_F3 ? (L + 15) / 16 : ST == SK_F4 ? H / 64 : ST == SK_F5 ? D / 16
                : ST == SK_B1 ? H / 16 : ST == SK_B2 ? (L + 15) / 16 : ST == SK_F5W ? (D + 63) / 64 : ST == SK_B1W ? H / 64 : H / 32;
  const int bid = blockIdx.x;
  if constexpr (ST == SK_F1) {
    if (bid >= nct * nrt * a.ns1) {               // VAE: extra workgroups draw the eps rows (the GMVAE's ride on F2)
      const unsigned long long step = a.step_dev ? a.step_dev[0] : a.step;
      const int qe = (L + 3) / 4;
      const long long i = (long long)(bid - nct * nrt * a.ns1) * kSkThreads + tid;
      if (i < (long long)B * qe) {
        const int r = (int)(i / qe), quad = (int)(i - (long long)r * qe);
        float nz[4];
        noise_vals(a.row0 + (unsigned long long)r, (unsigned)quad, false, a.seed, step, nz);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (quad * 4 + j < L) st1o(a.eps_w + (long long)r * L + quad * 4 + j, nz[j]);
      }
      return;
    }
    if (vae && bid == 0 && tid == 0 && a.step_dev) a.step_dev[1] = a.step_dev[0];     // the copy the W launch reads
    if (bid == 0 && a.dw_cnt)                     // sk_dwc's arrival counters (nine launches ahead of their use)
      for (int i = tid; i < kSkDwcMaxTiles; i += kSkThreads) a.dw_cnt[i] = 0u;
  }
  const int ct = bid % nct, rt = (bid / nct) % nrt, ks = bid / (nct * nrt);
  const int r0 = rt * 16 * RT;
  long long rowc[RT];                             // this lane's activation rows (clamped; rows >= B are masked at the stores)
#pragma unroll
  for (int j = 0; j < RT; ++j) rowc[j] = min(r0 + 16 * j + ln, B - 1);
  const float* const P = a.P;
  // ---- the epilogue's inputs (threads 0..255 own one accumulator register of one lane slot each: see below) are
  // requested FIRST: they do not depend on the contraction, and behind the waves' meeting in LDS they were a memory
  // round trip of their own (1.2 - 1.6 us of a 4 - 6 us launch, tools/skstamps.py)
  const int el = tid & 63, er = (tid >> 6) & 3;   // owner of accumulator register er of lane slot el
  const int jl = tid >> 8, ec = el & 15;          // ... of row tile 2 c + jl in meeting c
  const bool owner = tid < 256 * CJ;
  float pfa[NC][6];
  float4 pf4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float (&pf)[6] = pfa[c];
#pragma unroll
    for (int i = 0; i < 6; ++i) pf[i] = 0.f;
    const int row = r0 + 16 * (CJ * c + jl) + 4 * (el >> 4) + er;
    const long long rr = row < B ? row : B - 1;
  if (owner) {
    if constexpr (ST == SK_F1) {
      if (vae) pf4 = *reinterpret_cast<const float4*>(P + a.by0 + ct * 64 + 4 * ec);
    } else if constexpr (ST == SK_F3) {
      const int l = min(ct * 16 + ec, L - 1);      // (a ragged last tile of latent dimensions: clamped here, masked at the stores)
      pf[0] = P[a.bg1 + l]; pf[1] = P[a.bg1 + L + l]; pf[2] = a.eps[rr * L + l];
      if (!vae) { pf[3] = a.pp[rr * 2 * L + l]; pf[4] = a.pp[rr * 2 * L + L + l]; }
    } else if constexpr (ST == SK_F4) {
      pf4 = *reinterpret_cast<const float4*>(P + a.bd0 + ct * 64 + 4 * ec);
    } else if constexpr (ST == SK_F5) {
      const int n = ct * 16 + ec;
      pf[0] = P[a.bd1 + n] + (a.gen_bias_vec ? a.gen_bias_vec[n] : 0.f);
      pf[1] = (float)a.x[rr * D + n];
    } else if constexpr (ST == SK_F5W) {
      const int n = min(ct * 64 + 4 * ec, D - 4);  // (a ragged last tile of columns: clamped here, masked at the stores)
      pf4 = *reinterpret_cast<const float4*>(P + a.bd1 + n);
      if (a.gen_bias_vec) { const float4 q = *reinterpret_cast<const float4*>(a.gen_bias_vec + n); pf4.x += q.x; pf4.y += q.y; pf4.z += q.z; pf4.w += q.w; }
      pf[0] = __uint_as_float(*reinterpret_cast<const unsigned*>(a.x + rr * D + n));       // 4 pixels
    } else if constexpr (ST == SK_B1) {
      pf[0] = a.hd[rr * H + ct * 16 + ec];
    } else if constexpr (ST == SK_B1W) {
      const float4 q = *reinterpret_cast<const float4*>(a.hd + rr * H + ct * 64 + 4 * ec);
      pf[0] = q.x; pf[1] = q.y; pf[2] = q.z; pf[3] = q.w;
    } else if constexpr (ST == SK_B3) {
      pf[0] = a.hg[rr * H + ct * 32 + ec]; pf[1] = a.hg[rr * H + ct * 32 + 16 + ec];
    } else if constexpr (ST == SK_B2) {
      const int l = min(ct * 16 + ec, L - 1);
      pf[0] = a.qp[rr * 2 * L + L + l]; pf[1] = a.z[rr * L + l]; pf[4] = a.eps[rr * L + l];
      if (!vae) { pf[2] = a.pp[rr * 2 * L + l]; pf[3] = a.pp[rr * 2 * L + L + l]; }
    }
  }
  }
  f32x4 acc[RT][4];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long arow[RT];
  if constexpr (ST == SK_F1) {
    const int c0 = ct * 64, kgs = D / 16;
    const int kg_lo = (int)((long long)kgs * ks / a.ns1), kg_hi = (int)((long long)kgs * (ks + 1) / a.ns1);
    const float* W = c0 < H ? P + a.Wy0 + c0 : P + a.Wg0 + (c0 - H);
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * D;
    if constexpr (RT > 1) {                       // (no slabs where RT > 1: ns1 = 1)
      sk_nn4_u8bf<RT>(a.x, arow, W, H, 4 * ln, 0, (D + 31) / 32, wave, lk, D, acc);
    } else {
      sk_nn4<true, RT>(a.x, arow, W, H, 4 * ln, kg_lo, kg_hi, wave, lk, acc);
    }
  } else if constexpr (ST == SK_F3) {
    const int lc = min(ct * 16 + ln, L - 1);
    const int col[2] = {lc, L + lc};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * H;
    sk_nnp<2, RT>(a.hg, arow, P + a.Wg1, 2 * L, col, 0, H / 16, wave, lk, acc);
  } else if constexpr (ST == SK_F4) {
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * L;
    if (L & 15) sk_nn4<false, RT, true>(a.z, arow, P + a.Wd0 + ct * 64, H, 4 * ln, 0, (L + 15) / 16, wave, lk, acc, L);
    else sk_nn4<false, RT>(a.z, arow, P + a.Wd0 + ct * 64, H, 4 * ln, 0, L / 16, wave, lk, acc);
  } else if constexpr (ST == SK_F5) {             // 16-column tiles: D / 16 x row tiles workgroups (the widest layer on the most CUs)
    const int col[1] = {ct * 16 + ln};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * H;
    sk_nnp<1, RT>(a.hd, arow, P + a.Wd1, D, col, 0, H / 16, wave, lk, acc);
  } else if constexpr (ST == SK_F5W) {            // 64-column tiles (strided, as F1): 4 x the MFMAs per W fragment
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * H;
    if constexpr (RT > 1) {                       // (H % 64 = 0: whole groups of 32)
      sk_nn4_bf6<RT>(a.hd, arow, P + a.Wd1, D, min(ct * 64 + 4 * ln, D - 4), H / 32, wave, lk, acc);
    } else {
      sk_nn4<false, RT>(a.hd, arow, P + a.Wd1, D, min(ct * 64 + 4 * ln, D - 4), 0, H / 16, wave, lk, acc);
    }
  } else if constexpr (ST == SK_B1W) {            // 4 strided W rows per lane: out column ct 64 + 4 ln + t
    const int wr[4] = {ct * 64 + 4 * ln, ct * 64 + 4 * ln + 1, ct * 64 + 4 * ln + 2, ct * 64 + 4 * ln + 3};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * D;
    if constexpr (RT > 1) {
      sk_nt_bf6<RT, 4>(a.g, arow, P + a.Wd1, D, wr, (D + 31) / 32, wave, lk, D, acc);
    } else {
      sk_nt<4, RT>(a.g, arow, P + a.Wd1, D, wr, 0, D / 16, wave, lk, acc);
    }
  } else if constexpr (ST == SK_B1) {             // dhd = g Wd1^T: out column j = hidden unit, W row j of Wd1 [H][D]
    const int wr[1] = {ct * 16 + ln};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * D;
    sk_nt<1, RT>(a.g, arow, P + a.Wd1, D, wr, 0, D / 16, wave, lk, acc);
  } else if constexpr (ST == SK_B2) {             // dz = dhd Wd0^T: W row l of Wd0 [L][H]
    const int wr[1] = {min(ct * 16 + ln, L - 1)};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * H;
    sk_nt<1, RT>(a.dhd, arow, P + a.Wd0, H, wr, 0, H / 16, wave, lk, acc);
  } else {                                        // B3: dhg = dqp Wg1^T: W row h of Wg1 [H][2L]
    const int wr[2] = {ct * 32 + ln, ct * 32 + 16 + ln};
    for (int j = 0; j < RT; ++j) arow[j] = rowc[j] * 2 * L;
    if ((2 * L) & 15) sk_nt<2, RT, true>(a.dqp, arow, P + a.Wg1, 2 * L, wr, 0, (2 * L + 15) / 16, wave, lk, acc, 2 * L);
    else sk_nt<2, RT>(a.dqp, arow, P + a.Wg1, 2 * L, wr, 0, (2 * L) / 16, wave, lk, acc);
  }
  SK_STAMP(SLOT, 1);
  // ---- the waves' partial tiles meet in LDS (fixed order), CJ row tiles at a time
#pragma unroll
  for (int c = 0; c < NC; ++c) {
  if (c) __syncthreads();                         // (the previous meeting's readers are done)
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * CJ + j) * NU * 4 + 4 * t + r) * 64 + lane] = acc[CJ * c + j][t][r];
  __syncthreads();
  if (c == 0) { SK_STAMP(SLOT, 2); }
  if (!owner) continue;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < kSkWaves; ++w)
#pragma unroll
    for (int t = 0; t < NU; ++t) v[t] += red[((w * CJ + jl) * NU * 4 + 4 * t + er) * 64 + el];
  const float (&pf)[6] = pfa[c];
  const int row = r0 + 16 * (CJ * c + jl) + 4 * (el >> 4) + er;
  const bool rok = row < B;
  // ---- epilogues
  if constexpr (ST == SK_F1) {
    if (vae) {                                    // the whole layer: hidden activation of the encoder (scripts/base.py:47-60,67)
      if (rok) st4o(a.hy + (long long)row * H + ct * 64 + 4 * ec,
                    make_float4(fmaxf(v[0] + pf4.x, 0.f), fmaxf(v[1] + pf4.y, 0.f), fmaxf(v[2] + pf4.z, 0.f), fmaxf(v[3] + pf4.w, 0.f)));
    } else if (rok) st4o(a.s1 + ((long long)ks * B + row) * 2 * H + ct * 64 + 4 * ec, make_float4(v[0], v[1], v[2], v[3]));
  } else if constexpr (ST == SK_F3) {
    // ConditionalNormal heads (scripts/base.py:66-72), z = mu + sigma eps (gmvae.py:248), log q and log p(z|y) terms
    // (gmvae.py:258) of this tile's 16 latent dimensions; row sums over the tile -> lqp[.][tile][row]
    const int l = ct * 16 + ec;
    const bool lok = l < L;
    // (hardware exp / log / rcp forms as mega2.hpp S5: ~1e-6 relative, one pass of each per head)
    const float mu = v[0] + pf[0], raw = v[1] + pf[1];
    float sgq;
    const float sg = fmaxf(softplus_sig(raw + a.c, sgq), a.smin);
    const float zz = mu + sg * pf[2];
    const float e = pf[2];                        // (z - mu) / sigma IS eps (as mega2.hpp S5; formed from the rounded z it
                                                  //  loses every bit of eps once sigma |eps| < ulp(mu))
    float aq = -0.5f * e * e - 0.5f * kLog2Pi - flog(sg);
    float ap;
    if (a.model == 1) {                           // learned mixture prior: mixture_logprob_* after this launch
      ap = 0.f;
    } else if (vae) {                             // standard-normal prior (scripts/vae.py:247-250)
      ap = -0.5f * zz * zz - 0.5f * kLog2Pi;
    } else {
      float sgp;
      const float mp = pf[3], sp = fmaxf(softplus_sig(pf[4] + a.c, sgp), a.smin);
      const float t = (zz - mp) * __builtin_amdgcn_rcpf(sp);
      ap = -0.5f * t * t - 0.5f * kLog2Pi - flog(sp);
    }
    if (rok && lok) { st1o(a.qp + (long long)row * 2 * L + l, mu); st1o(a.qp + (long long)row * 2 * L + L + l, raw); st1o(a.z + (long long)row * L + l, zz); }
    aq = sk_row16_sum(lok ? aq : 0.f); ap = sk_row16_sum(lok ? ap : 0.f);
    if (ec == 0 && rok) { st1o(a.lqp + (long long)ct * B + row, aq); st1o(a.lqp + (long long)((L + 15) / 16 + ct) * B + row, ap); }
  } else if constexpr (ST == SK_F4) {
    const int n = ct * 64 + 4 * ec;
    if (rok) st4o(a.hd + (long long)row * H + n,
                  make_float4(fmaxf(v[0] + pf4.x, 0.f), fmaxf(v[1] + pf4.y, 0.f), fmaxf(v[2] + pf4.z, 0.f), fmaxf(v[3] + pf4.w, 0.f)));
  } else if constexpr (ST == SK_F5) {
    // logits = MLP(z) + bias_init (scripts/base.py:135); Independent(Bernoulli).log_prob (gmvae.py:254) and its gradient
    const float lam = v[0] + pf[0] + a.gen_bias;
    const float xv = pf[1];
    const float e = fexp(-fabsf(lam));
    const float rcp = __builtin_amdgcn_rcpf(1.f + e);
    const float sp = fmaxf(lam, 0.f) - flog(rcp);
    float rs = xv * lam - sp;
    if (rok) st1o(a.g + (long long)row * D + ct * 16 + ec, (lam >= 0.f ? rcp : e * rcp) - xv);
    rs = sk_row16_sum(rs);
    if (ec == 0 && rok) st1o(a.part + (long long)row * a.nparts + ct, rs);
  } else if constexpr (ST == SK_F5W) {
    const int n = ct * 64 + 4 * ec;
    const bool nok = n < D;
    const unsigned xw = __float_as_uint(pf[0]);
    const float bq[4] = {pf4.x, pf4.y, pf4.z, pf4.w};
    float gq[4], rs = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float lam = v[t] + bq[t] + a.gen_bias;
      const float xv = (float)((xw >> (8 * t)) & 0xffu);
      const float e = fexp(-fabsf(lam));
      const float rcp = __builtin_amdgcn_rcpf(1.f + e);
      const float sp = fmaxf(lam, 0.f) - flog(rcp);
      rs += xv * lam - sp;
      gq[t] = (lam >= 0.f ? rcp : e * rcp) - xv;
    }
    if (rok && nok) st4o(a.g + (long long)row * D + n, make_float4(gq[0], gq[1], gq[2], gq[3]));
    rs = sk_row16_sum(nok ? rs : 0.f);
    if (ec == 0 && rok) st1o(a.part + (long long)row * a.nparts + ct, rs);
  } else if constexpr (ST == SK_B1W) {
    if (rok) st4o(a.dhd + (long long)row * H + ct * 64 + 4 * ec,
                  make_float4(pf[0] > 0.f ? v[0] : 0.f, pf[1] > 0.f ? v[1] : 0.f, pf[2] > 0.f ? v[2] : 0.f, pf[3] > 0.f ? v[3] : 0.f));
  } else if constexpr (ST == SK_B1) {
    if (rok) st1o(a.dhd + (long long)row * H + ct * 16 + ec, pf[0] > 0.f ? v[0] : 0.f);
  } else if constexpr (ST == SK_B3) {
    if (rok) {
      st1o(a.dhg + (long long)row * H + ct * 32 + ec, pf[0] > 0.f ? v[0] : 0.f);
      st1o(a.dhg + (long long)row * H + ct * 32 + 16 + ec, pf[1] > 0.f ? v[1] : 0.f);
    }
  } else {                                        // B2: reverse of the two heads, per (row, latent dim)
    const int l = ct * 16 + ec;
    if (rok && l < L) {
      float sgq;                                  // sigmoid(raw_q + c): softplus' derivative
      const float rawq = pf[0] + a.c, spq = softplus_sig(rawq, sgq), sg = fmaxf(spq, a.smin);
      const float zz = pf[1];
      if (a.model == 1) {                         // learned mixture prior: z_head_bwd (kernels.hpp) takes it from here
        st1o(a.dz + (long long)row * L + l, v[0]);
      } else if (vae) {                           // standard-normal prior: d(-log p)/dz = z, nothing to send to a prior network
        const float dmu = v[0] + zz;
        const float dsg = dmu * pf[4] - __builtin_amdgcn_rcpf(sg);
        st1o(a.dqp + (long long)row * 2 * L + l, dmu);
        st1o(a.dqp + (long long)row * 2 * L + L + l, (spq > a.smin) ? dsg * sgq : 0.f);
      } else {
      float sgp;
      const float mp = pf[2], rawp = pf[3] + a.c, spp = softplus_sig(rawp, sgp), sp = fmaxf(spp, a.smin);
      const float isp = __builtin_amdgcn_rcpf(sp);
      const float t = (zz - mp) * isp;
      const float pterm = t * isp;
      const float dmu = v[0] + pterm;
      const float dsg = dmu * pf[4] - __builtin_amdgcn_rcpf(sg);
      st1o(a.dqp + (long long)row * 2 * L + l, dmu);
      st1o(a.dqp + (long long)row * 2 * L + L + l, (spq > a.smin) ? dsg * sgq : 0.f);
      st1o(a.dpp + (long long)row * 2 * L + l, -pterm);
      st1o(a.dpp + (long long)row * 2 * L + L + l, (spp > a.smin) ? (1.f - t * t) * isp * sgp : 0.f);
      }
    }
  }
  }
  SK_STAMP(SLOT, 3);
}

// block-wide sums of 16 per-thread partials (256 threads): DPP row sums (16 lanes), the 16 rows' sums meet in LDS and
// thread k < 16 adds them in row order: it returns the k-th total (other threads: 0).  One barrier.  (A butterfly of
// ds_bpermute shuffles per value was 2.5 us of these 6 - 9 us kernels.)
__device__ __forceinline__ float sk_block_sum16(const float (&p)[16], float* __restrict__ sh /* [16][16] */, const int tid) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = row16_sum(p[k]);
    if ((tid & 15) == 15) sh[k * 16 + (tid >> 4)] = v;       // (the DPP row-rotation sum ends in every lane of the row)
  }
  __syncthreads();
  float tot = 0.f;
  if (tid < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += sh[tid * 16 + j];
  }
  return tot;
}

// the K <= 16 entries of one row of a [.][K] matrix (surplus entries: copies of valid ones, they meet zeros) as 8-byte loads -- 8
// wave-instructions of 20 lines at K = 10 instead of 16: lane c reads row c, so every load of the row-major matrix touches a
// cache line per 3 lanes whatever its width.  (Rows of an odd K are only 4-byte aligned: the copy below compiles to the same
// global_load_dwordx2, which this target's unaligned access mode serves; no branch on K around the loads.)
__device__ __forceinline__ void sk_row16(const float* __restrict__ rowp, const int K, float (&w)[16]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float2 v;
    __builtin_memcpy(&v, rowp + min(2 * q, (K - 1) & ~1), sizeof(v));      // (odd K: the last pair's second element is the next row's first)
    w[2 * q] = v.x; w[2 * q + 1] = v.y;
  }
}

// F2: the y path, one workgroup (256 threads) per batch row: this row's Philox noise; hy = relu(sum of the first-layer
// slabs + b) and the x part of encoder_gmm's first layer; logits = hy Wy1 + b; RelaxedOneHotCategorical.sample and the
// entropy term (scripts/gmvae.py:238-240,262-263, as kernels.hpp y_head_fwd); hg = relu(gx + y Wg0[D:] + b); the prior
// head pp = y Wp + b (gmvae.py:243).  K <= 16.  A thread owns hidden units tid + 256 i (i < NI) and prior-head columns
// tid + 256 j (j < 2); EVERY global load of the kernel -- slabs, biases, the three small weight matrices' entries for the
// thread's columns, the uniform -- is issued at its start (none depends on another), so the kernel is ONE memory round
// trip, the logits' block sum, the softmax in wave 0 and the stores.
template <int NI>
__global__ __launch_bounds__(256) void sk_ypath(const SkArgs a) {
  __shared__ float rsh[256], ysh[16], ush[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, H = a.H, L = a.L, K = a.K, H2 = 2 * H, L2 = 2 * L;
  if ((int)blockIdx.x >= B) {                     // extra workgroups: the eps rows (Box-Muller) beside the rows' critical path
    const unsigned long long step = a.step_dev ? a.step_dev[0] : a.step;
    const int qe = (L + 3) / 4;
    const long long i = (long long)(blockIdx.x - B) * 256 + tid;
    if (i < (long long)B * qe) {
      const int r = (int)(i / qe), quad = (int)(i - (long long)r * qe);
      float nz[4];
      noise_vals(a.row0 + (unsigned long long)r, (unsigned)quad, false, a.seed, step, nz);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (quad * 4 + j < L) st1o(a.eps_w + (long long)r * L + quad * 4 + j, nz[j]);
    }
    return;
  }
  const int row = blockIdx.x;
  const float* const P = a.P;
  SK_STAMP(1, 0);
  const unsigned long long step = a.step_dev ? a.step_dev[0] : a.step;
  if (blockIdx.x == 0 && tid == 0 && a.step_dev) a.step_dev[1] = step;       // the copy the W launch reads
  // ---- all loads
  float sy[NI][kSkNs1x], sg[NI][kSkNs1x], by[NI], bg[NI], wy[NI][16], wg[NI][16], wp[2][16], bp[2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = min(tid + 256 * i, H - 1);
#pragma unroll
    for (int s2 = 0; s2 < kSkNs1x; ++s2) {
      const long long o = ((long long)min(s2, a.ns1 - 1) * B + row) * H2 + c;
      sy[i][s2] = a.s1[o];
      sg[i][s2] = a.s1[o + H];
    }
    by[i] = P[a.by0 + c];
    bg[i] = P[a.bg0 + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = min(k, K - 1);                 // (branch-free; surplus entries meet zeros below)
      wg[i][k] = P[a.Wg0 + (long long)(a.D + kc) * H + c];
    }
    sk_row16(P + a.Wy1 + (long long)c * K, K, wy[i]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jc = min(tid + 256 * j, L2 - 1);
    bp[j] = P[a.bp + jc];
#pragma unroll
    for (int k = 0; k < 16; ++k) wp[j][k] = P[a.Wp + (long long)min(k, K - 1) * L2 + jc];
  }
  const float b1v = P[a.by1 + min(lane, K - 1)];
  const float u_ext = a.gen_u ? 0.5f : a.u[(long long)row * K + min(lane, K - 1)];
  // ---- this row's uniforms (wave 3; the eps rows are drawn by the launch's extra workgroups)
  if (a.gen_u && tid >= 192 && tid - 192 < (K + 3) / 4) {
    const int quad = tid - 192;
    float nz[4];
    noise_vals(a.row0 + (unsigned long long)row, (unsigned)quad, true, a.seed, step, nz);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (quad * 4 + j < K) { st1o(a.u_w + (long long)row * K + quad * 4 + j, nz[j]); ush[quad * 4 + j] = nz[j]; }
  }
  // ---- hy, the logits' partial sums; gx stays in registers (the same thread finishes hg for the same hidden units)
  float p[16], gx[NI];
#pragma unroll
  for (int k = 0; k < 16; ++k) p[k] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float t = 0.f, g2 = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < kSkNs1x; ++s2) { t += s2 < a.ns1 ? sy[i][s2] : 0.f; g2 += s2 < a.ns1 ? sg[i][s2] : 0.f; }
    const bool on = tid + 256 * i < H;
    t = on ? fmaxf(t + by[i], 0.f) : 0.f;
    gx[i] = g2 + bg[i];
    if (on) st1o(a.hy + (long long)row * H + tid + 256 * i, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] += t * wy[i][k];
  }
  SK_STAMP(1, 1);
  const float ptot = sk_block_sum16(p, rsh, tid);  // (its barrier also publishes ush)
  SK_STAMP(1, 2);
  if (wave == 0) {
    const bool kv = lane < K;
    const float lg = kv ? ptot + b1v : -INFINITY;
    const float uu = kv ? (a.gen_u ? ush[lane] : u_ext) : 0.5f;
    // (hardware log / exp forms as mega2.hpp S2; lanes 0..15 hold the row: DPP row reductions)
    const float av = kv ? (lg - flog(-flog(uu))) * a.invT : -INFINITY;
    const float mx = row16_max(av);
    const float se = row16_sum(kv ? fexp(av - mx) : 0.f);
    const float lse = mx + flog(se);
    const float lga[1] = {lg};
    float lpa[1];
    cat_log_softmax<Row16, 1>(lga, lpa);           // log pi (gemm.hpp: accurate for a saturated q(y|x))
    float yv = 0.f, ne = 0.f;
    if (kv) {
      yv = fexp(av - lse);
      const float lp = lpa[0];
      ne = fexp(lp) * lp;
      st1o(a.logits + (long long)row * K + lane, lg);
    }
    ne = row16_sum(ne);
    if (lane < a.K4) st1o(a.y + (long long)row * a.K4 + lane, yv);            // rows of pad4(K) floats, the padding zero
    if (lane == 0) st1o(a.nent + row, ne);
    if (lane < 16) ysh[lane] = yv;
  }
  __syncthreads();
  float yk[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) yk[k] = ysh[k];      // (zero for k >= K)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float v = gx[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) v += yk[k] * wg[i][k];
    if (tid + 256 * i < H) st1o(a.hg + (long long)row * H + tid + 256 * i, fmaxf(v, 0.f));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v = bp[j];
#pragma unroll
    for (int k = 0; k < 16; ++k) v += yk[k] * wp[j][k];
    if (tid + 256 * j < L2) st1o(a.pp + (long long)row * L2 + tid + 256 * j, v);
  }
  SK_STAMP(1, 3);
}

// B4: one workgroup per row: dy = dhg Wg0[D:]^T + dpp Wp^T, the reverse of the Gumbel-softmax and of the entropy term
// (kernels.hpp y_head_bwd at S = 1) -> dlogits, then dhy = (dlogits Wy1^T) [hy > 0].  Same thread map and
// load discipline as sk_ypath: every global load at the start.
template <int NI>
__global__ __launch_bounds__(256) void sk_ybwd(const SkArgs a) {
  __shared__ float rsh[256], dls[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, L2 = 2 * a.L, K = a.K;
  const int row = blockIdx.x;
  const float* const P = a.P;
  SK_STAMP(8, 0);
  float dg[NI], hv[NI], wy[NI][16], wg[NI][16], wp[2][16], dp[2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = min(tid + 256 * i, H - 1);
    dg[i] = a.dhg[(long long)row * H + c];
    hv[i] = a.hy[(long long)row * H + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = min(k, K - 1);
      wg[i][k] = P[a.Wg0 + (long long)(a.D + kc) * H + c];
    }
    sk_row16(P + a.Wy1 + (long long)c * K, K, wy[i]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jc = min(tid + 256 * j, L2 - 1);
    dp[j] = a.dpp[(long long)row * L2 + jc];
#pragma unroll
    for (int k = 0; k < 16; ++k) wp[j][k] = P[a.Wp + (long long)min(k, K - 1) * L2 + jc];
  }
  const float lg_ = a.logits[(long long)row * K + min(lane, K - 1)];
  const float yv_ = a.y[(long long)row * a.K4 + min(lane, K - 1)];
  const float ne = a.nent[row];
  float p[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) p[k] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float dv = tid + 256 * i < H ? dg[i] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] += dv * wg[i][k];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float dv = tid + 256 * j < L2 ? dp[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] += dv * wp[j][k];
  }
  SK_STAMP(8, 1);
  const float dy = sk_block_sum16(p, rsh, tid);
  SK_STAMP(8, 2);
  if (wave == 0) {
    const bool kv = lane < K;
    const float lg = kv ? lg_ : -INFINITY;
    const float yv = kv ? yv_ : 0.f;
    const float lga[1] = {lg}, ya[1] = {yv}, dya[1] = {kv ? dy : 0.f};
    float lpa[1], daa[1];
    cat_log_softmax<Row16, 1>(lga, lpa);           // (gemm.hpp: the forms that survive a saturated softmax)
    cat_softmax_bwd<Row16, 1>(ya, dya, daa);
    float dl = 0.f;
    if (kv) {
      const float lp = lpa[0];
      dl = daa[0] * a.invT + fexp(lp) * (lp - ne);
    }
    if (lane < a.K4) st1o(a.dlogits + (long long)row * a.K4 + lane, dl);      // padding columns zero
    if (lane < 16) dls[lane] = dl;
  }
  __syncthreads();
  float dk[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dk[k] = dls[k];      // (zero for k >= K)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += dk[k] * wy[i][k];
    if (tid + 256 * i < H) st1o(a.dhy + (long long)row * H + tid + 256 * i, hv[i] > 0.f ? v : 0.f);
  }
  SK_STAMP(8, 3);
}

// ---- the y path at batches of more than 768 rows: R = 2 or 4 rows per workgroup.  One workgroup per row re-reads the three small
// weight matrices' entries (~100 scalar loads per thread) for every row and at B = 1024 ran in two rounds of workgroups (F2
// 15.5 us, B4 12.3 us of a 145 us step); here they are loaded once for R rows, the R x 16 block sums share one barrier, and the R
// rows' softmax work runs side by side in wave 0's R DPP rows (lane 16 j + k: row j, component k).
template <int R>
__device__ __forceinline__ float sk_block_sum16r(const float (&p)[R][16], float* __restrict__ sh /* [R][16][16] */, const int tid) {
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float v = row16_sum(p[j][k]);
      if ((tid & 15) == 15) sh[j * 256 + k * 16 + (tid >> 4)] = v;
    }
  __syncthreads();
  float tot = 0.f;
  if (tid < 16 * R) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += sh[(tid >> 4) * 256 + (tid & 15) * 16 + i];
  }
  return tot;                                      // thread 16 j + k: row j's k-th total
}

template <int NI, int R>
__global__ __launch_bounds__(256) void sk_ypath_r(const SkArgs a) {
  __shared__ float rsh[R * 256], ysh[R * 16], ush[R * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, H = a.H, L = a.L, K = a.K, H2 = 2 * H, L2 = 2 * L;
  const int nwr = (B + R - 1) / R;                // workgroups that carry rows
  if ((int)blockIdx.x >= nwr) {                   // extra workgroups: the eps rows (as sk_ypath)
    const unsigned long long step = a.step_dev ? a.step_dev[0] : a.step;
    const int qe = (L + 3) / 4;
    const long long i = (long long)(blockIdx.x - nwr) * 256 + tid;
    if (i < (long long)B * qe) {
      const int r = (int)(i / qe), quad = (int)(i - (long long)r * qe);
      float nz[4];
      noise_vals(a.row0 + (unsigned long long)r, (unsigned)quad, false, a.seed, step, nz);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (quad * 4 + j < L) st1o(a.eps_w + (long long)r * L + quad * 4 + j, nz[j]);
    }
    return;
  }
  const int row0 = blockIdx.x * R;
  const float* const P = a.P;
  SK_STAMP(1, 0);
  const unsigned long long step = a.step_dev ? a.step_dev[0] : a.step;
  if (blockIdx.x == 0 && tid == 0 && a.step_dev) a.step_dev[1] = step;       // the copy the W launch reads
  // ---- all loads: the weights once, the slabs of R rows
  float sy[R][NI], sg[R][NI], by[NI], bg[NI], wy[NI][16], wg[NI][16], wp[2][16], bp[2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = min(tid + 256 * i, H - 1);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long long rj = min(row0 + j, B - 1);
      float t = 0.f, g2 = 0.f;
      for (int s2 = 0; s2 < a.ns1; ++s2) {          // (one slab at these batch sizes)
        const long long o = ((long long)s2 * B + rj) * H2 + c;
        t += a.s1[o]; g2 += a.s1[o + H];
      }
      sy[j][i] = t; sg[j][i] = g2;
    }
    by[i] = P[a.by0 + c];
    bg[i] = P[a.bg0 + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = min(k, K - 1);
      wg[i][k] = P[a.Wg0 + (long long)(a.D + kc) * H + c];
    }
    sk_row16(P + a.Wy1 + (long long)c * K, K, wy[i]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jc = min(tid + 256 * j, L2 - 1);
    bp[j] = P[a.bp + jc];
#pragma unroll
    for (int k = 0; k < 16; ++k) wp[j][k] = P[a.Wp + (long long)min(k, K - 1) * L2 + jc];
  }
  const int sj = lane >> 4, sk = lane & 15;       // wave 0: lane 16 j + k works on (row j, component k)
  const long long srow = min(row0 + sj, B - 1);
  const float b1v = P[a.by1 + min(sk, K - 1)];
  const float u_ext = a.gen_u ? 0.5f : a.u[srow * K + min(sk, K - 1)];
  // ---- the rows' uniforms (wave 3)
  const int nq = (K + 3) / 4;
  if (a.gen_u && tid >= 192 && tid - 192 < R * nq) {
    const int j = (tid - 192) / nq, quad = (tid - 192) - j * nq;
    if (row0 + j < B) {
      float nz[4];
      noise_vals(a.row0 + (unsigned long long)(row0 + j), (unsigned)quad, true, a.seed, step, nz);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (quad * 4 + q < K) { st1o(a.u_w + (long long)(row0 + j) * K + quad * 4 + q, nz[q]); ush[j * 16 + quad * 4 + q] = nz[q]; }
    }
  }
  float p[R][16], gx[R][NI];
#pragma unroll
  for (int j = 0; j < R; ++j) {
#pragma unroll
    for (int k = 0; k < 16; ++k) p[j][k] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool on = tid + 256 * i < H;
      const float t = on ? fmaxf(sy[j][i] + by[i], 0.f) : 0.f;
      gx[j][i] = sg[j][i] + bg[i];
      if (on && row0 + j < B) st1o(a.hy + (long long)(row0 + j) * H + tid + 256 * i, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) p[j][k] += t * wy[i][k];
    }
  }
  SK_STAMP(1, 1);
  const float ptot = sk_block_sum16r<R>(p, rsh, tid);      // (its barrier also publishes ush)
  SK_STAMP(1, 2);
  if (wave == 0 && lane < 16 * R) {
    const bool rv = row0 + sj < B;
    const bool kv = sk < K;
    const float lg = kv ? ptot + b1v : -INFINITY;
    const float uu = kv && rv ? (a.gen_u ? ush[sj * 16 + sk] : u_ext) : 0.5f;
    const float av = kv ? (lg - flog(-flog(uu))) * a.invT : -INFINITY;
    const float mx = row16_max(av);
    const float se = row16_sum(kv ? fexp(av - mx) : 0.f);
    const float lse = mx + flog(se);
    const float lga[1] = {lg};
    float lpa[1];
    cat_log_softmax<Row16, 1>(lga, lpa);           // log pi (gemm.hpp: accurate for a saturated q(y|x))
    float yv = 0.f, ne = 0.f;
    if (kv) {
      yv = fexp(av - lse);
      const float lp = lpa[0];
      ne = fexp(lp) * lp;
      if (rv) st1o(a.logits + srow * K + sk, lg);
    }
    ne = row16_sum(ne);
    if (rv && sk < a.K4) st1o(a.y + srow * a.K4 + sk, yv);       // rows of pad4(K) floats, the padding zero
    if (rv && sk == 0) st1o(a.nent + srow, ne);
    ysh[sj * 16 + sk] = yv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (row0 + j >= B) break;
    float yk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) yk[k] = ysh[j * 16 + k];      // (zero for k >= K)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v = gx[j][i];
#pragma unroll
      for (int k = 0; k < 16; ++k) v += yk[k] * wg[i][k];
      if (tid + 256 * i < H) st1o(a.hg + (long long)(row0 + j) * H + tid + 256 * i, fmaxf(v, 0.f));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v = bp[q];
#pragma unroll
      for (int k = 0; k < 16; ++k) v += yk[k] * wp[q][k];
      if (tid + 256 * q < L2) st1o(a.pp + (long long)(row0 + j) * L2 + tid + 256 * q, v);
    }
  }
  SK_STAMP(1, 3);
}

template <int NI, int R>
__global__ __launch_bounds__(256) void sk_ybwd_r(const SkArgs a) {
  __shared__ float rsh[R * 256], dls[R * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, H = a.H, L2 = 2 * a.L, K = a.K;
  const int row0 = blockIdx.x * R;
  const float* const P = a.P;
  SK_STAMP(8, 0);
  float dg[R][NI], hv[R][NI], wy[NI][16], wg[NI][16], wp[2][16], dp[R][2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = min(tid + 256 * i, H - 1);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long long rj = min(row0 + j, B - 1);
      dg[j][i] = a.dhg[rj * H + c];
      hv[j][i] = a.hy[rj * H + c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = min(k, K - 1);
      wg[i][k] = P[a.Wg0 + (long long)(a.D + kc) * H + c];
    }
    sk_row16(P + a.Wy1 + (long long)c * K, K, wy[i]);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int jc = min(tid + 256 * q, L2 - 1);
#pragma unroll
    for (int j = 0; j < R; ++j) dp[j][q] = a.dpp[(long long)min(row0 + j, B - 1) * L2 + jc];
#pragma unroll
    for (int k = 0; k < 16; ++k) wp[q][k] = P[a.Wp + (long long)min(k, K - 1) * L2 + jc];
  }
  const int sj = lane >> 4, sk = lane & 15;
  const long long srow = min(row0 + sj, B - 1);
  const float lg_ = a.logits[srow * K + min(sk, K - 1)];
  const float yv_ = a.y[srow * a.K4 + min(sk, K - 1)];
  const float ne = a.nent[srow];
  float p[R][16];
#pragma unroll
  for (int j = 0; j < R; ++j) {
#pragma unroll
    for (int k = 0; k < 16; ++k) p[j][k] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float dv = tid + 256 * i < H ? dg[j][i] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) p[j][k] += dv * wg[i][k];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float dv = tid + 256 * q < L2 ? dp[j][q] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) p[j][k] += dv * wp[q][k];
    }
  }
  SK_STAMP(8, 1);
  const float dy = sk_block_sum16r<R>(p, rsh, tid);
  SK_STAMP(8, 2);
  if (wave == 0 && lane < 16 * R) {
    const bool rv = row0 + sj < B;
    const bool kv = sk < K;
    const float lg = kv ? lg_ : -INFINITY;
    const float yv = kv ? yv_ : 0.f;
    const float lga[1] = {lg}, ya[1] = {yv}, dya[1] = {kv ? dy : 0.f};
    float lpa[1], daa[1];
    cat_log_softmax<Row16, 1>(lga, lpa);           // (gemm.hpp: the forms that survive a saturated softmax)
    cat_softmax_bwd<Row16, 1>(ya, dya, daa);
    float dl = 0.f;
    if (kv) {
      const float lp = lpa[0];
      dl = daa[0] * a.invT + fexp(lp) * (lp - ne);
    }
    if (rv && sk < a.K4) st1o(a.dlogits + srow * a.K4 + sk, dl);      // padding columns zero
    dls[sj * 16 + sk] = dl;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (row0 + j >= B) break;
    float dk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dk[k] = dls[j * 16 + k];      // (zero for k >= K)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += dk[k] * wy[i][k];
      if (tid + 256 * i < H) st1o(a.dhy + (long long)(row0 + j) * H + tid + 256 * i, hv[j][i] > 0.f ? v : 0.f);
    }
  }
  SK_STAMP(8, 3);
}

// One [16 x 64] tile of dW = A^T dY (+ the bias gradient on the first tile row) and its TF-Adam update, by ONE wave, for
// tensors whose row length N is a multiple of 4: the lane's B operand is a 16-byte load of four consecutive columns (four
// strided column tiles) and its accumulators are dW[m0 + 4 lk + r][n0 + 4 ln .. + 3], so every access of the optimizer
// -- p, m, v in, p, m, v and the gradient out -- is a 16-byte access and a lane group covers 256 contiguous bytes of a row.
__device__ __forceinline__ void sk_dw_tile_v(const SkArgs& a, const SkTensor& T, const int tm, const int tn, const int ln, const int lk) {
  const int B = a.B, M = T.M, N = T.N, lda = T.lda, ldy = T.ldy;
  const int m0 = tm * 16, n0 = tn * 64;
  const int n = n0 + 4 * ln;
  const bool n_ok = n < N;
  const int nc = min(n, N - 4), mc = min(m0 + ln, M - 1);
  const bool upd = a.ap != nullptr;
  float4 pp[4], pm[4], pv[4];
  if (upd) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long i = (long long)T.w_off + (long long)min(m0 + 4 * lk + r, M - 1) * N + nc;
      pp[r] = *reinterpret_cast<const float4*>(a.ap + i);
      pm[r] = *reinterpret_cast<const float4*>(a.am + i);
      pv[r] = *reinterpret_cast<const float4*>(a.av + i);
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned char* const A8 = static_cast<const unsigned char*>(T.A);
  const float* const A32 = static_cast<const float*>(T.A);
  for (int b0 = 0; b0 < B; b0 += 64) {
    float av[16];
    float4 bv[16];
    if (T.a_u8) {
      unsigned char ab[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int b = min(b0 + 4 * s + lk, B - 1);
        ab[s] = A8[(long long)b * lda + mc];
        bv[s] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) av[s] = (float)ab[s];
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int b = min(b0 + 4 * s + lk, B - 1);
        av[s] = A32[(long long)b * lda + mc];
        bv[s] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool on = b0 + 4 * s + lk < B && n_ok;      // a zero B operand also voids clamped rows / columns
      const float4 bq = on ? bv[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float aq = m0 + ln < M ? av[s] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq.w, acc[3], 0, 0, 0);
      cs.x += bq.x; cs.y += bq.y; cs.z += bq.z; cs.w += bq.w;
    }
  }
  float lr_t = 0.f;
  if (upd) {
    const float tf = (float)((a.step_dev ? a.step_dev[1] : a.step) + 1ull);
    lr_t = a.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
  }
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / (float)B;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * lk + r;
    const float gq[4] = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    if (upd) {
      adam_update(pp[r].x, pm[r].x, pv[r].x, gq[0], gs, lr_t, omb1, omb2, a.aeps);
      adam_update(pp[r].y, pm[r].y, pv[r].y, gq[1], gs, lr_t, omb1, omb2, a.aeps);
      adam_update(pp[r].z, pm[r].z, pv[r].z, gq[2], gs, lr_t, omb1, omb2, a.aeps);
      adam_update(pp[r].w, pm[r].w, pv[r].w, gq[3], gs, lr_t, omb1, omb2, a.aeps);
    }
    if (m < M && n_ok) {
      const long long i = (long long)T.w_off + (long long)m * N + n;
      // write-through (mega2.hpp st4o): 5 x 4 P bytes of dirty lines would otherwise wait for the end-of-kernel write-back,
      // in front of the next step's first launch
      st4o(a.grads + i, make_float4(gq[0], gq[1], gq[2], gq[3]));
      if (upd) { st4o(a.ap + i, pp[r]); st4o(a.am + i, pm[r]); st4o(a.av + i, pv[r]); }
    }
  }
  if (tm == 0 && T.b_off >= 0) {                  // bias gradient: column sums of dY over the batch rows
    cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64); cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64); cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (lk == 0 && n_ok) {
      const long long i = (long long)T.b_off + n;
      *reinterpret_cast<float4*>(a.grads + i) = cs;
      if (upd) {
        float4 bp = *reinterpret_cast<const float4*>(a.ap + i), bm = *reinterpret_cast<const float4*>(a.am + i), bvv = *reinterpret_cast<const float4*>(a.av + i);
        adam_update(bp.x, bm.x, bvv.x, cs.x, gs, lr_t, omb1, omb2, a.aeps);
        adam_update(bp.y, bm.y, bvv.y, cs.y, gs, lr_t, omb1, omb2, a.aeps);
        adam_update(bp.z, bm.z, bvv.z, cs.z, gs, lr_t, omb1, omb2, a.aeps);
        adam_update(bp.w, bm.w, bvv.w, cs.w, gs, lr_t, omb1, omb2, a.aeps);
        *reinterpret_cast<float4*>(a.ap + i) = bp; *reinterpret_cast<float4*>(a.am + i) = bm; *reinterpret_cast<float4*>(a.av + i) = bvv;
      }
    }
  }
}

// VAE_GMP: the workgroups of a W launch that update the mixture prior's variables (idx: the workgroup's index among them)
__device__ __forceinline__ void sk_gmp_update(const SkArgs& a, const int idx, const int tid, const int nthr = kSkThreads) {
  const int B = a.B;
  const int i = idx * nthr + tid;
  // VAE_GMP: the mixture prior's variables (loc, raw scale, mixture logits: scripts/vae.py:233-238) have no matrix-product
  // gradient: gmp_param_bwd left gmp_n partials in the flat layout's order; sum them in partial order, update
  const int KL = a.K * a.L, KLp = (KL + 3) & ~3;
  const bool valid = i < a.gmp_len && (i < KL || (i >= KLp && i < KLp + KL) || (i >= 2 * KLp && i < 2 * KLp + a.K));
  if (!valid) return;                               // (alignment padding between the three tensors)
  const float tf = (float)((a.step_dev ? a.step_dev[1] : a.step) + 1ull);
  const float lr_t = a.ap ? a.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1)) : 0.f;
  const long long o = a.gmp_off + i;
  float pw = 0.f, pm = 0.f, pv = 0.f;
  if (a.ap) { pw = a.ap[o]; pm = a.am[o]; pv = a.av[o]; }
  float gsum = 0.f;
  for (int g0 = 0; g0 < a.gmp_n; g0 += 16) {        // 16 partials in flight at a time (one by one: 16 dependent round trips each)
    float pv16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) pv16[j] = a.gmp_part[(long long)min(g0 + j, a.gmp_n - 1) * a.gmp_len + i];
#pragma unroll
    for (int j = 0; j < 16; ++j) gsum += g0 + j < a.gmp_n ? pv16[j] : 0.f;
  }
  a.grads[o] = gsum;
  if (a.ap) {
    adam_update(pw, pm, pv, gsum, 1.f / (float)B, lr_t, 1.f - a.b1, 1.f - a.b2, a.aeps);
    a.ap[o] = pw; a.am[o] = pm; a.av[o] = pv;
  }
}

// the loss tail of a W launch (one workgroup): per-row terms from the partials, batch sums, counters
__device__ __forceinline__ void sk_loss_tail(const SkArgs& a, float (&red)[4][256], const int tid) {
  const int B = a.B;
  const unsigned long long dbg_c0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int nlt = (a.L + 15) / 16;
  if (tid < 256) {
    for (int b = tid; b < B; b += 256) {
      float lpx = 0.f, lq = 0.f, lp = 0.f;
      for (int i = 0; i < a.nparts; ++i) lpx += a.part[(long long)b * a.nparts + i];
      for (int i = 0; i < nlt; ++i) { lq += a.lqp[(long long)i * B + b]; lp += a.lqp[(long long)(nlt + i) * B + b]; }
      if (a.model == 1) lp = a.logp[b];               // (VAE_GMP: the mixture log-density, written by mixture_logprob_*)
      const float ne = a.nent ? a.nent[b] : 0.f;      // (VAE: no entropy term)
      const float lw = lpx + lp - lq - ne;
      a.logpx[b] = lpx; a.logq[b] = lq; a.logp[b] = lp; a.logw[b] = lw;
      a0 -= lw; a1 -= lpx; a2 += lq - lp; a3 += ne;
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2; red[3][tid] = a3;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
      for (int j = 0; j < 4; ++j) red[j][tid] += red[j][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const unsigned long long dbg_c1 = __builtin_amdgcn_s_memtime(), dbg_r1 = __builtin_amdgcn_s_memrealtime();
    const float tl[8] = {red[0][0], red[1][0], red[2][0], red[3][0], (float)B, 0.f, (float)(dbg_c1 - dbg_c0), (float)(dbg_r1 - dbg_r0)};
#pragma unroll
    for (int j = 0; j < 8; ++j) { a.tail[j] = tl[j]; if (a.tail_log) a.tail_log[j] = tl[j]; }
    if (a.step_dev) a.step_dev[0] = a.step_dev[1] + 1ull;
  }
}

// W: every weight and bias gradient + TF-Adam (scripts/runners.py:181-183).  The contraction runs over the <= 128 batch
// rows only, so a [64 x 16] tile of dW = A^T dY is ONE wave's work (16 MFMA steps of 4 rows per 64 rows, 4 strided 16-row
// tiles as in dwadam.hpp) and a workgroup carries 8 tiles; nothing meets in LDS.  The optimizer runs on the accumulator
// registers: per (tile, register) the 16 lanes of a lane group touch 64 contiguous bytes of p, m, v and the gradient.
// Bound: the optimizer's traffic (7 x 4 P bytes: 37 MB at H = 512).  The last workgroup is the loss tail.
__global__ __launch_bounds__(kSkThreads) void sk_dw(const SkArgs a) {
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B;
  const int gmp_wgs = (a.gmp_part && a.has_tail) ? (a.gmp_len + kSkThreads - 1) / kSkThreads : 0;
  const int ntw = (int)gridDim.x - (a.has_tail ? 1 : 0) - gmp_wgs;      // tile workgroups; then the loss tail, if this launch carries it; then
                                                                        // (VAE_GMP) the workgroups that update the mixture prior's variables
  if ((int)blockIdx.x >= ntw + (a.has_tail ? 1 : 0)) {
    sk_gmp_update(a, (int)blockIdx.x - ntw - (a.has_tail ? 1 : 0), tid);
    return;
  }
  SK_STAMP(9, 0);
  if ((int)blockIdx.x == ntw) {                   // ---- loss tail: per-row terms from the partials, batch sums, counters
    sk_loss_tail(a, red, tid);
    return;
  }
  // a workgroup's 8 waves take 8 CONSECUTIVE tiles (one row of tiles of a tensor: they share the A rows and write
  // neighbouring rows of p, m, v).  Measured alternatives, all slower: tiles dealt round-robin over 256 workgroups (no
  // shared lines: 70 vs 60 us per step), consecutive tiles walking down a column of tiles (62), 4 / 5 / 6 waves per workgroup
  // to spread the 1412 tiles of the bin/run_train.sh sizes over more than 177 CUs (63.6 / 70.1 / 61.2 against 60.7: round 4)
  for (int tile = blockIdx.x * kSkWaves + wave; tile < a.total_tiles; tile += ntw * kSkWaves) {
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kSkMaxT; ++i)
    if (i < a.ntens && tile >= a.t[i].tile_begin) ti = i;
  const SkTensor& T = a.t[ti];
  const int M = T.M, N = T.N, lda = T.lda, ldy = T.ldy;
  const int tl = tile - T.tile_begin, tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  if (T.vec) { sk_dw_tile_v(a, T, tm, tn, ln, lk); continue; }
  const int m0 = tm * 64, n0 = tn * 16;
  const int ma = m0 + 4 * ln;
  const bool a_ok = ma < M, n_ok = n0 + ln < N;
  const int mac = min(ma, ((M + 3) & ~3) - 4), nc = min(n0 + ln, N - 1);      // (source rows hold pad4(M) elements)
  // the lane's 16 (p, m, v) triples are requested BEFORE the contraction's operands (they depend on nothing): the optimizer's
  // 7 x 4 P bytes are the launch's bound, and behind the contraction they were a second, serial memory phase
  const bool upd = a.ap != nullptr;
  const int n = n0 + ln;
  float pp[4][4], pm[4][4], pv[4][4];
  if (upd) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(m0 + 4 * (4 * lk + r) + t, M - 1);
        const long long i = (long long)T.w_off + (long long)m * N + min(n, N - 1);
        pp[t][r] = a.ap[i]; pm[t][r] = a.am[i]; pv[t][r] = a.av[i];
      }
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs = 0.f;
  const unsigned char* const A8 = static_cast<const unsigned char*>(T.A);
  const float* const A32 = static_cast<const float*>(T.A);
  for (int b0 = 0; b0 < B; b0 += 64) {
    float4 av[16];
    float bv[16];
    if (T.a_u8) {                                  // (the operand type decides OUTSIDE the unrolled loads: a branch per load
      unsigned aw[16];                             //  serialises them, one memory round trip each)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int b = min(b0 + 4 * s + lk, B - 1);
        aw[s] = *reinterpret_cast<const unsigned*>(A8 + (long long)b * lda + mac);
        bv[s] = T.dY[(long long)b * ldy + nc];
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
        av[s] = make_float4((float)(aw[s] & 0xffu), (float)((aw[s] >> 8) & 0xffu), (float)((aw[s] >> 16) & 0xffu), (float)(aw[s] >> 24));
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int b = min(b0 + 4 * s + lk, B - 1);
        av[s] = *reinterpret_cast<const float4*>(A32 + (long long)b * lda + mac);
        bv[s] = T.dY[(long long)b * ldy + nc];
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float bq = (b0 + 4 * s + lk < B && n_ok) ? bv[s] : 0.f;       // a zero B operand also voids clamped rows
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].x : 0.f, bq, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].y : 0.f, bq, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].z : 0.f, bq, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].w : 0.f, bq, acc[3], 0, 0, 0);
      cs += bq;
    }
  }
  SK_STAMP(9, 1);
  // ---- TF-Adam (ApplyAdam form: kernels.hpp adam_update), then every store
  float lr_t = 0.f;
  if (upd) {                                      // 1 - b^t = -expm1(t ln b): no cancellation (dwadam.hpp)
    const float tf = (float)((a.step_dev ? a.step_dev[1] : a.step) + 1ull);
    lr_t = a.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
  }
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / (float)B;
  if (upd) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) adam_update(pp[t][r], pm[t][r], pv[t][r], acc[t][r], gs, lr_t, omb1, omb2, a.aeps);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * (4 * lk + r) + t;     // accumulator [t][r] of lane (ln, lk): strided tile t, tile row 4 lk + r
      if (m < M && n_ok) {
        const long long i = (long long)T.w_off + (long long)m * N + n;
        a.grads[i] = acc[t][r];
        if (upd) { a.ap[i] = pp[t][r]; a.am[i] = pm[t][r]; a.av[i] = pv[t][r]; }
      }
    }
  if (tm == 0 && T.b_off >= 0) {                  // bias gradient: column sums of dY over the batch rows
    cs += __shfl_xor(cs, 16, 64);
    cs += __shfl_xor(cs, 32, 64);
    if (lk == 0 && n_ok) {
      const long long i = (long long)T.b_off + n;
      a.grads[i] = cs;
      if (upd) {
        float pp = a.ap[i], pm = a.am[i], pv = a.av[i];
        adam_update(pp, pm, pv, cs, gs, lr_t, omb1, omb2, a.aeps);
        a.ap[i] = pp; a.am[i] = pm; a.av[i] = pv;
      }
    }
  }
  }
  SK_STAMP(9, 3);
}


// The uint8 problems (dW = x^T dY: 57 % of the stage's products at H = 512) run on the bf16 matrix cores, exact, as dw_adam's
// (dwadam.hpp dw_contract_u8x3): x is exact in bf16, dY = hi + mid + lo, so 3 v_mfma_f32_16x16x32_bf16 (16 cycles) per 32 batch
// rows and [16 x 16] block replace 8 v_mfma_f32_16x16x4_f32 (32 cycles); contraction index (lane group h, element e) of an MFMA
// is batch row b0 + 4 e + h -- the rows the lane's register-direct loads hold anyway.
// One wave's share [b_lo, b_hi) of the batch rows of a [64 x 64] tile of dW = A^T dY into acc[strided row tile][strided column
// tile] (+ the columns' sums of dY into cs where the tile carries the bias gradient): lane (ln, lk) loads 16 bytes of A (4 uint8
// or 4 floats: rows m0 + 4 ln + i) and 16 bytes of dY (columns n0 + 4 ln + j) per batch row.
__device__ __forceinline__ void sk_dw_contract64(const SkTensor& T, const int mac, const int nc, const int b_lo, const int b_hi,
                                                 const int lk, const bool bias, f32x4 (&acc)[4][4], float4& cs) {
  const int lda = T.lda, ldy = T.ldy;
  const unsigned char* const A8 = static_cast<const unsigned char*>(T.A);
  const float* const A32 = static_cast<const float*>(T.A);
  if (T.a_u8) {
    for (int b0 = b_lo; b0 < b_hi; b0 += 32) {
      unsigned aw[8];
      float4 bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int b = min(b0 + 4 * e + lk, b_hi - 1);
        aw[e] = *reinterpret_cast<const unsigned*>(A8 + (long long)b * lda + mac);
        bv[e] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (b0 + 4 * e + lk >= b_hi) aw[e] = 0u;    // a zero A operand voids a clamped batch row
      if (bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (b0 + 4 * e + lk < b_hi) { cs.x += bv[e].x; cs.y += bv[e].y; cs.z += bv[e].z; cs.w += bv[e].w; }
      }
      sk_bf16x8 At[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sk_u32x4 w;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp)
          w[jp] = pack_hi16((float)((aw[2 * jp + 1] >> (8 * i)) & 0xffu), (float)((aw[2 * jp] >> (8 * i)) & 0xffu));
        At[i] = __builtin_bit_cast(sk_bf16x8, w);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v[8] = {j == 0 ? bv[0].x : j == 1 ? bv[0].y : j == 2 ? bv[0].z : bv[0].w, j == 0 ? bv[1].x : j == 1 ? bv[1].y : j == 2 ? bv[1].z : bv[1].w,
                            j == 0 ? bv[2].x : j == 1 ? bv[2].y : j == 2 ? bv[2].z : bv[2].w, j == 0 ? bv[3].x : j == 1 ? bv[3].y : j == 2 ? bv[3].z : bv[3].w,
                            j == 0 ? bv[4].x : j == 1 ? bv[4].y : j == 2 ? bv[4].z : bv[4].w, j == 0 ? bv[5].x : j == 1 ? bv[5].y : j == 2 ? bv[5].z : bv[5].w,
                            j == 0 ? bv[6].x : j == 1 ? bv[6].y : j == 2 ? bv[6].z : bv[6].w, j == 0 ? bv[7].x : j == 1 ? bv[7].y : j == 2 ? bv[7].z : bv[7].w};
        sk_u32x4 h, m, l;
        sk_split8(v, h, m, l);
        const sk_bf16x8 Bh = __builtin_bit_cast(sk_bf16x8, h), Bm = __builtin_bit_cast(sk_bf16x8, m), Bl = __builtin_bit_cast(sk_bf16x8, l);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At[i], Bl, acc[i][j], 0, 0, 0);      // smallest pieces first
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At[i], Bm, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At[i], Bh, acc[i][j], 0, 0, 0);
        }
      }
    }
  } else {
    // fp32 x fp32: both operands as three bf16 pieces, 6 piece products per product (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi:
    // the three dropped ones are <= 2^-23 of the product, below its own fp32 rounding -- the plane GEMMs' form, gemm.hpp), 96 MFMA
    // cycles per 32 batch rows and [16 x 16] block instead of 256; the splits are VALU work beside the matrix pipe
    for (int b0 = b_lo; b0 < b_hi; b0 += 32) {
      float4 af[8], bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int b = min(b0 + 4 * e + lk, b_hi - 1);
        af[e] = *reinterpret_cast<const float4*>(A32 + (long long)b * lda + mac);
        bv[e] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (b0 + 4 * e + lk >= b_hi) af[e] = make_float4(0.f, 0.f, 0.f, 0.f);      // a zero A operand voids a clamped batch row
      if (bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (b0 + 4 * e + lk < b_hi) { cs.x += bv[e].x; cs.y += bv[e].y; cs.z += bv[e].z; cs.w += bv[e].w; }
      }
      sk_bf16x8 Ah[4], Am[4], Al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = i == 0 ? af[e].x : i == 1 ? af[e].y : i == 2 ? af[e].z : af[e].w;
        sk_u32x4 h, m, l;
        sk_split8(v, h, m, l);
        Ah[i] = __builtin_bit_cast(sk_bf16x8, h); Am[i] = __builtin_bit_cast(sk_bf16x8, m); Al[i] = __builtin_bit_cast(sk_bf16x8, l);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = j == 0 ? bv[e].x : j == 1 ? bv[e].y : j == 2 ? bv[e].z : bv[e].w;
        sk_u32x4 h, m, l;
        sk_split8(v, h, m, l);
        const sk_bf16x8 Bh = __builtin_bit_cast(sk_bf16x8, h), Bm = __builtin_bit_cast(sk_bf16x8, m), Bl = __builtin_bit_cast(sk_bf16x8, l);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[i], Bl, c, 0, 0, 0);      // smallest products first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[i], Bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[i], Bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[i], Bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[i], Bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[i], Bh, c, 0, 0, 0);
          acc[i][j] = c;
        }
      }
    }
  }
}

// ... the same with fp32 matrix instructions (16 v_mfma_f32_16x16x4_f32 per batch row quad): 126 registers instead of 170, so TWO
// 8-wave workgroups fit a CU -- the form for more tiles than CUs at batches too short for batch shares (sk_dwb<0>)
__device__ __forceinline__ void sk_dw_contract64_f32(const SkTensor& T, const int mac, const int nc, const int b_lo, const int b_hi,
                                                     const int lk, f32x4 (&acc)[4][4], float4& cs) {
  const int lda = T.lda, ldy = T.ldy;
  const unsigned char* const A8 = static_cast<const unsigned char*>(T.A);
  const float* const A32 = static_cast<const float*>(T.A);
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    float4 av[4], bv[4];
    if (T.a_u8) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int b = min(b0 + 4 * s + lk, b_hi - 1);
        av[s].x = __uint_as_float(*reinterpret_cast<const unsigned*>(A8 + (long long)b * lda + mac));
        bv[s] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned w = __float_as_uint(av[s].x);
        av[s] = make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int b = min(b0 + 4 * s + lk, b_hi - 1);
        av[s] = *reinterpret_cast<const float4*>(A32 + (long long)b * lda + mac);
        bv[s] = *reinterpret_cast<const float4*>(T.dY + (long long)b * ldy + nc);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool on = b0 + 4 * s + lk < b_hi;                    // a zero B operand voids a clamped batch row
      const float bq[4] = {on ? bv[s].x : 0.f, on ? bv[s].y : 0.f, on ? bv[s].z : 0.f, on ? bv[s].w : 0.f};
      const float aq[4] = {av[s].x, av[s].y, av[s].z, av[s].w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[i], bq[j], acc[i][j], 0, 0, 0);
      cs.x += bq[0]; cs.y += bq[1]; cs.z += bq[2]; cs.w += bq[3];
    }
  }
}

// W for batches above 128 rows: sk_dw's one-wave tiles re-read their operands per [16 x 64] tile -- at B = 1024, H = 512 that is
// 450 MB through L2 for 2.9 GFLOP and the launch ran at the L2's rate (63 us).  Here a WORKGROUP owns a [64 x 64] tile of
// dW = A^T dY: lane (ln, lk) loads 16 bytes of A (4 uint8 or 4 floats: rows m0 + 4 ln + tm) and 16 bytes of dY (columns
// n0 + 4 ln + tn) per batch row and feeds 16 MFMAs from them; the 8 waves split the batch rows and their partial tiles meet
// in LDS in wave order (two halves of 32 accumulator registers: 64 KB).  The optimizer runs on the summed tile, spread over
// all 512 threads: thread (tml, r, lane) of half p owns dW[m0 + 16 lk + 4 r + 2 p + tml][n0 + 4 ln .. + 3], so p, m, v and the
// gradient move in 16-byte accesses, 256 contiguous bytes per lane group (rows of N % 4 != 0: element by element).
// Tiles are dealt so that the 8 workgroups an XCD receives in turn hold neighbouring tiles (shared operand columns in its L2).
// (Measured and dropped: [32 x 64] tiles to balance 354 tiles over 256 CUs -- twice the workgroups, but only two of them fit a CU,
// so the second half queued: 43 us against 46.  Where the tiles do not balance, the stage runs as sk_dwc below.)
template <int BF>
__global__ __launch_bounds__(kSkThreads) void sk_dwb(const SkArgs a) {
  constexpr int TM = 4, NP = 2;                    // strided 16-row tiles; meetings (halves of 32 accumulator registers)
  __shared__ __attribute__((aligned(16))) float part[kSkWaves * 8 * 64 * 4];      // [wave][tml * 4 + r][lane][tn]
  __shared__ __attribute__((aligned(16))) float csl[kSkWaves * 16 * 4];           // column sums [wave][ln][tn]
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B;
  const int gmp_wgs = (a.gmp_part && a.has_tail) ? (a.gmp_len + kSkThreads - 1) / kSkThreads : 0;
  const int ntw = (int)gridDim.x - (a.has_tail ? 1 : 0) - gmp_wgs;
  if ((int)blockIdx.x >= ntw + (a.has_tail ? 1 : 0)) {
    sk_gmp_update(a, (int)blockIdx.x - ntw - (a.has_tail ? 1 : 0), tid);
    return;
  }
  SK_STAMP(9, 0);
  if ((int)blockIdx.x == ntw) {
    sk_loss_tail(a, red, tid);
    return;
  }
  const int per = ntw >> 3;                        // (ntw = 8 ceil(total_tiles / 8))
  const int tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile >= a.total_tiles) return;
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kSkMaxT; ++i)
    if (i < a.ntens && tile >= a.t[i].tile_begin) ti = i;
  const SkTensor& T = a.t[ti];
  const int M = T.M, N = T.N;
  const int tl = tile - T.tile_begin, tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  const int m0 = tm * 16 * TM, n0 = tn * 64;
  const int mac = min(m0 + TM * ln, ((M + 3) & ~3) - TM);        // (operand rows hold pad4(M), pad4(N) elements; rows m >= M and
  const int nc = min(n0 + 4 * ln, ((N + 3) & ~3) - 4);         //  columns n >= N of the tile are computed from clamped loads and never stored)
  const int rows_p = (((B + kSkWaves - 1) / kSkWaves) + 3) & ~3;
  const int b_lo = min(B, wave * rows_p), b_hi = min(B, b_lo + rows_p);
  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool bias = tm == 0 && T.b_off >= 0;
  if constexpr (BF) sk_dw_contract64(T, mac, nc, b_lo, b_hi, lk, bias, acc, cs);
  else sk_dw_contract64_f32(T, mac, nc, b_lo, b_hi, lk, acc, cs);
  SK_STAMP(9, 1);
  // ---- the optimizer's operands of this thread's two rows (one per half), requested before the waves meet
  const bool upd = a.ap != nullptr;
  const bool vecn = (N & 3) == 0;
  const int tml = wave >> 2, er = wave & 3;
  const int n = n0 + 4 * ln;
  float pp[NP][4], pm[NP][4], pv[NP][4];
  long long ei[NP];
  bool eok[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m = m0 + TM * (4 * lk + er) + 2 * p + tml;       // MFMA row 4 lk + er of strided tile 2 p + tml
    eok[p] = m < M && n < N;
    ei[p] = (long long)T.w_off + (long long)min(m, M - 1) * N + min(n, vecn ? N - 4 : N - 1);
    if (upd) {
      if (vecn) {
        const float4 q0 = *reinterpret_cast<const float4*>(a.ap + ei[p]), q1 = *reinterpret_cast<const float4*>(a.am + ei[p]),
                     q2 = *reinterpret_cast<const float4*>(a.av + ei[p]);
        pp[p][0] = q0.x; pp[p][1] = q0.y; pp[p][2] = q0.z; pp[p][3] = q0.w;
        pm[p][0] = q1.x; pm[p][1] = q1.y; pm[p][2] = q1.z; pm[p][3] = q1.w;
        pv[p][0] = q2.x; pv[p][1] = q2.y; pv[p][2] = q2.z; pv[p][3] = q2.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long i = ei[p] + min(j, N - 1 - min(n, N - 1));
          pp[p][j] = a.ap[i]; pm[p][j] = a.am[i]; pv[p][j] = a.av[i];
        }
      }
    }
  }
  float lr_t = 0.f;
  if (upd) {
    const float tf = (float)((a.step_dev ? a.step_dev[1] : a.step) + 1ull);
    lr_t = a.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
  }
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / (float)B;
  if (bias) {
    cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64); cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64); cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (lk == 0) *reinterpret_cast<float4*>(csl + (wave * 16 + ln) * 4) = cs;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p) __syncthreads();                        // (the first half's readers are done)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(part + (((wave * 8 + i * 4 + r) * 64) + lane) * 4) =
            make_float4(acc[2 * p + i][0][r], acc[2 * p + i][1][r], acc[2 * p + i][2][r], acc[2 * p + i][3][r]);
    __syncthreads();
    if (p == 0) { SK_STAMP(9, 2); }
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kSkWaves; ++w) {
      const float4 q = *reinterpret_cast<const float4*>(part + (((w * 8 + tml * 4 + er) * 64) + lane) * 4);
      g[0] += q.x; g[1] += q.y; g[2] += q.z; g[3] += q.w;
    }
    if (upd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_update(pp[p][j], pm[p][j], pv[p][j], g[j], gs, lr_t, omb1, omb2, a.aeps);
    }
    if (eok[p]) {
      if (vecn) {
        st4o(a.grads + ei[p], make_float4(g[0], g[1], g[2], g[3]));
        if (upd) {
          st4o(a.ap + ei[p], make_float4(pp[p][0], pp[p][1], pp[p][2], pp[p][3]));
          st4o(a.am + ei[p], make_float4(pm[p][0], pm[p][1], pm[p][2], pm[p][3]));
          st4o(a.av + ei[p], make_float4(pv[p][0], pv[p][1], pv[p][2], pv[p][3]));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < N) {
            st1o(a.grads + ei[p] + j, g[j]);
            if (upd) { st1o(a.ap + ei[p] + j, pp[p][j]); st1o(a.am + ei[p] + j, pm[p][j]); st1o(a.av + ei[p] + j, pv[p][j]); }
          }
      }
    }
    if (p == 0 && bias && tid < 16 && n0 + 4 * tid < N) {      // bias gradient: column sums of dY over the batch rows, in wave order
      float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < kSkWaves; ++w) {
        const float4 q = *reinterpret_cast<const float4*>(csl + (w * 16 + tid) * 4);
        c4[0] += q.x; c4[1] += q.y; c4[2] += q.z; c4[3] += q.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = n0 + 4 * tid + j;
        if (nb < N) {
          const long long i = (long long)T.b_off + nb;
          a.grads[i] = c4[j];
          if (upd) {
            float bp = a.ap[i], bm = a.am[i], bvv = a.av[i];
            adam_update(bp, bm, bvv, c4[j], gs, lr_t, omb1, omb2, a.aeps);
            a.ap[i] = bp; a.am[i] = bm; a.av[i] = bvv;
          }
        }
      }
    }
  }
  SK_STAMP(9, 3);
}

// W where one workgroup per tile cannot balance (H = 512: 354 [64 x 64] tiles on 256 CUs, and a CU that carries two sets the
// launch's time): workgroups of 4 waves; workgroup (tile, share) contracts one of dw_ks shares of the batch rows for a [64 x 64]
// tile (operand fragments and accumulators as sk_dwb), its 4 waves meet in LDS and the summed partial tile -- and, on the first
// tile row, the partial column sums (bias gradient) -- goes to dwp[share] in the flat layout (write-through).  The workgroup then
// counts itself in (agent-scope atomic add behind a fence and a barrier); the one that arrives LAST for its tile adds the shares
// in share order -- all of them read back from memory, so the sum does not depend on who was last -- and runs TF-Adam on the
// tile: no second launch.  (As sk_dwc + an element-wise optimizer launch the stage took 32 us at B = 1024, H = 512: the second
// launch moved 50 MB behind a 1.7 us boundary.)  The loss tail and the mixture prior's workgroups ride along.
constexpr int kDwcThreads = 256, kDwcWaves = 4;
__global__ __launch_bounds__(kDwcThreads, 3) void sk_dwc(const SkArgs a) {
  constexpr int TM = 4, NP = 2;
  __shared__ __attribute__((aligned(16))) float part[kDwcWaves * 8 * 64 * 4];     // [wave][tml * 4 + r][lane][tn]: 32 KB
  __shared__ __attribute__((aligned(16))) float csl[kDwcWaves * 16 * 4];
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B;
  const int gmp_wgs = (a.gmp_part && a.has_tail) ? (a.gmp_len + kDwcThreads - 1) / kDwcThreads : 0;
  const int ntw = (int)gridDim.x - (a.has_tail ? 1 : 0) - gmp_wgs;
  if ((int)blockIdx.x >= ntw + (a.has_tail ? 1 : 0)) {
    sk_gmp_update(a, (int)blockIdx.x - ntw - (a.has_tail ? 1 : 0), tid, kDwcThreads);
    return;
  }
  SK_STAMP(9, 0);
  if ((int)blockIdx.x == ntw) {
    sk_loss_tail(a, red, tid);
    return;
  }
  __shared__ int lastf;
  const int per = ntw >> 3;                        // (ntw = 16 ceil(total_tiles dw_ks / 16): a tile's shares on ONE XCD)
  const int unit = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  const int tile = unit / a.dw_ks, ks = unit - tile * a.dw_ks;
  if (tile >= a.total_tiles) return;
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kSkMaxT; ++i)
    if (i < a.ntens && tile >= a.t[i].tile_begin) ti = i;
  const SkTensor& T = a.t[ti];
  const int M = T.M, N = T.N;
  const int tl = tile - T.tile_begin, tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  const int m0 = tm * 16 * TM, n0 = tn * 64;
  const int mac = min(m0 + TM * ln, ((M + 3) & ~3) - TM);
  const int nc = min(n0 + 4 * ln, ((N + 3) & ~3) - 4);
  const int rows_s = (((B + a.dw_ks - 1) / a.dw_ks) + 3) & ~3;                    // this workgroup's share of the batch rows ...
  const int s_lo = min(B, ks * rows_s), s_hi = min(B, s_lo + rows_s);
  const int rows_p = (((s_hi - s_lo + kDwcWaves - 1) / kDwcWaves) + 3) & ~3;      // ... and this wave's share of that
  const int b_lo = min(s_hi, s_lo + wave * rows_p), b_hi = min(s_hi, b_lo + rows_p);
  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool bias = tm == 0 && T.b_off >= 0;
  sk_dw_contract64(T, mac, nc, b_lo, b_hi, lk, bias, acc, cs);
  SK_STAMP(9, 1);
  float* const out = a.dwp + (long long)ks * a.dwp_stride;
  const bool vecn = (N & 3) == 0;
  if (bias) {
    cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64); cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64); cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (lk == 0) *reinterpret_cast<float4*>(csl + (wave * 16 + ln) * 4) = cs;
  }
  const int n = n0 + 4 * ln;
  float gown[4][4];                                // this thread's four units of ITS share: the finisher does not read them back
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p) __syncthreads();                        // (the first half's readers are done)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(part + (((wave * 8 + i * 4 + r) * 64) + lane) * 4) =
            make_float4(acc[2 * p + i][0][r], acc[2 * p + i][1][r], acc[2 * p + i][2][r], acc[2 * p + i][3][r]);
    __syncthreads();
    if (p == 0) { SK_STAMP(9, 2); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                  // this thread's two (strided tile, MFMA row) pairs of the half
      const int q8 = wave * 2 + h, tml = q8 >> 2, er = q8 & 3;
      float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < kDwcWaves; ++w) {
        const float4 q = *reinterpret_cast<const float4*>(part + (((w * 8 + q8) * 64) + lane) * 4);
        g[0] += q.x; g[1] += q.y; g[2] += q.z; g[3] += q.w;
      }
      const int m = m0 + TM * (4 * lk + er) + 2 * p + tml;
#pragma unroll
      for (int j = 0; j < 4; ++j) gown[2 * p + h][j] = g[j];
      if (m < M && n < N) {
        float* const o = out + (long long)T.w_off + (long long)m * N + n;
        if (vecn) st4o(o, make_float4(g[0], g[1], g[2], g[3]));
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < N) st1o(o + j, g[j]);
        }
      }
    }
    if (p == 0 && bias && tid < 16 && n0 + 4 * tid < N) {      // the share's column sums of dY, in wave order
      float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < kDwcWaves; ++w) {
        const float4 q = *reinterpret_cast<const float4*>(csl + (w * 16 + tid) * 4);
        c4[0] += q.x; c4[1] += q.y; c4[2] += q.z; c4[3] += q.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + 4 * tid + j < N) st1o(out + (long long)T.b_off + n0 + 4 * tid + j, c4[j]);
    }
  }
  // ---- count this share in; the last one of the tile finishes it.  No fence (an agent-scope fence writes the L2 back: with two
  // per workgroup the launch took 133 us): the partials left as sc1 write-through stores, every wave waits for its own
  // (vmcnt(0)), ONE lane adds behind the barrier, and the finisher reads with sc1 loads (MI355X_MICROARCH.md, hand-off table row 1)
  // The hardware contract this rests on (the same one as mega2.hpp's granules and mega3.hpp's flags; not a C++ release/acquire
  // pair -- an acq_rel add at agent scope IS the two fences measured above):
  //   * st4o / st1o are sc1 WRITE-THROUGH stores (asserted below: with -DM2_WT=0 they would sit dirty in this XCD's L2 and the
  //     finisher on another XCD would read stale partials), and vmcnt counts such a store until the memory side has
  //     acknowledged it -- so after s_waitcnt vmcnt(0) in every wave and the barrier, the share's partials are in memory;
  //   * the counter add is performed at the memory side (agent scope: sc1), after that barrier in program order of the ONE lane
  //     that issues it: the share that reads dw_ks - 1 finds every other share's add, hence its stores, complete;
  //   * the finisher's sc1 loads are served from memory, never from another XCD's (or its own, older) L2 line.
  // Nothing assumes which XCD a workgroup runs on.
  static_assert(M2_WT == 1, "sk_dwc's last-arriver hand-off needs the partial tiles stored write-through (sc1)");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) lastf = __hip_atomic_fetch_add(a.dw_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.dw_ks - 1) ? 1 : 0;
  __syncthreads();
  if (!lastf) { SK_STAMP(9, 3); return; }
  const bool upd = a.ap != nullptr;
  float lr_t = 0.f;
  if (upd) {
    const float tf = (float)((a.step_dev ? a.step_dev[1] : a.step) + 1ull);
    lr_t = a.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
  }
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / (float)B;
  long long ei[4];
  bool eok[4];
  float g[4][4], pp[4][4], pm[4][4], pv[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                    // this thread's four rows of the tile (the ones it stored above)
    const int p = q >> 1, q8 = wave * 2 + (q & 1), tml = q8 >> 2, er = q8 & 3;
    const int m = m0 + TM * (4 * lk + er) + 2 * p + tml;
    eok[q] = m < M && n < N;
    ei[q] = (long long)T.w_off + (long long)min(m, M - 1) * N + min(n, vecn ? N - 4 : N - 1);
    if (upd) {
      if (vecn) {
        const float4 q0 = *reinterpret_cast<const float4*>(a.ap + ei[q]), q1 = *reinterpret_cast<const float4*>(a.am + ei[q]),
                     q2 = *reinterpret_cast<const float4*>(a.av + ei[q]);
        pp[q][0] = q0.x; pp[q][1] = q0.y; pp[q][2] = q0.z; pp[q][3] = q0.w;
        pm[q][0] = q1.x; pm[q][1] = q1.y; pm[q][2] = q1.z; pm[q][3] = q1.w;
        pv[q][0] = q2.x; pv[q][1] = q2.y; pv[q][2] = q2.z; pv[q][3] = q2.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long i = ei[q] + min(j, N - 1 - min(n, N - 1));
          pp[q][j] = a.ap[i]; pm[q][j] = a.am[i]; pv[q][j] = a.av[i];
        }
      }
    }
  }
  // the shares' partials: every load in flight before the first sum (agent-coherent loads: the other share came from another CU)
  // (only the OTHER share is read back -- 11 MB of agent-scope loads at H = 512 instead of 22: such loads move at the fabric's
  //  uncached rate, profiles/round5_notes.md; this share's sums are still in registers, and a + b == b + a bit for bit)
  const long long oth = (long long)(1 - ks) * a.dwp_stride;
  if (vecn) {
    u32x4_t sv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = granule2_load(reinterpret_cast<const unsigned long long*>(a.dwp + oth + ei[q]));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3])::"memory");
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[q][j] = ks == 0 ? gown[q][j] + __uint_as_float(sv[q][j]) : __uint_as_float(sv[q][j]) + gown[q][j];
  } else {
    float t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        t[q][j] = __hip_atomic_load(a.dwp + oth + ei[q] + min(j, N - 1 - min(n, N - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[q][j] = ks == 0 ? gown[q][j] + t[q][j] : t[q][j] + gown[q][j];
  }
  static_assert(kSkDwShares == 2, "the sums above add two shares");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (upd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_update(pp[q][j], pm[q][j], pv[q][j], g[q][j], gs, lr_t, omb1, omb2, a.aeps);
    }
    if (eok[q]) {
      if (vecn) {
        st4o(a.grads + ei[q], make_float4(g[q][0], g[q][1], g[q][2], g[q][3]));
        if (upd) {
          st4o(a.ap + ei[q], make_float4(pp[q][0], pp[q][1], pp[q][2], pp[q][3]));
          st4o(a.am + ei[q], make_float4(pm[q][0], pm[q][1], pm[q][2], pm[q][3]));
          st4o(a.av + ei[q], make_float4(pv[q][0], pv[q][1], pv[q][2], pv[q][3]));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < N) {
            st1o(a.grads + ei[q] + j, g[q][j]);
            if (upd) { st1o(a.ap + ei[q] + j, pp[q][j]); st1o(a.am + ei[q] + j, pm[q][j]); st1o(a.av + ei[q] + j, pv[q][j]); }
          }
      }
    }
  }
  if (bias && tid < 64 && n0 + tid < N) {          // the bias gradient: the shares' column sums, in share order
    const long long i = (long long)T.b_off + n0 + tid;
    const float c0 = __hip_atomic_load(a.dwp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                c1 = __hip_atomic_load(a.dwp + a.dwp_stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float c = c0 + c1;
    a.grads[i] = c;
    if (upd) {
      float bp = a.ap[i], bm = a.am[i], bvv = a.av[i];
      adam_update(bp, bm, bvv, c, gs, lr_t, omb1, omb2, a.aeps);
      a.ap[i] = bp; a.am[i] = bm; a.av[i] = bvv;
    }
  }
  SK_STAMP(9, 3);
}

// VAE_GMP behind B2, ONE launch for everything the learned mixture prior adds to the step (scripts/vae.py:231-244, consumed at
// vae.py:181): log p(z) and the responsibilities (kernels.hpp mixture_logprob_lse), the prior's share of dz with the
// q head's reverse (z_head_bwd at PRIOR_GMP, S = 1) and the partial gradients of the prior's variables per strip of rows
// (gmp_param_bwd).  As three launches of the general schedule's row kernels -- written for 10^4 - 10^5 rows: softplus(raw_scale)
// per (row, component, dim), one lane walking a component's L terms -- they were 24 us of an 80 us step at B = 256.  Here every
// workgroup stages (loc, 1 / s) and the components' constants once in LDS; a row's K component terms are K wave sums over the
// latent dims (lane = dim), its logsumexp one more; the responsibilities never leave LDS.  Workgroups [0, nz): 4 rows each (a
// wave per row) -> logp, dqp; [nz, nz + gmp_n): strip i of the rows -> partial i in the flat layout's order [loc KLp | raw_scale
// KLp | mixture_logits pad4(K)], summed in strip order by the W launch.  K <= 64, 2 K (L | 1) floats of LDS (else the row kernels).
constexpr int kGmpStripRows = 64;                  // rows of a strip whose responsibilities LDS holds (B <= 4096 / 64 strips)
__global__ __launch_bounds__(256) void sk_gmp_bwd(const SkArgs a, const int nz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, L = a.L, K = a.K, LD = L | 1, KL = K * L;
  float* const s_loc = sm;                         // [K][LD]
  float* const s_is = sm + K * LD;                 // [K][LD]: 1 / s
  float* const s_c = s_is + K * LD;                // [64]: log_softmax(mixture_logits)_k - sum_l log s_kl - L/2 log 2 pi
  float* const s_w = s_c + 64;                     // [64]: softmax(mixture_logits)
  float* const s_r = s_w + 64;                     // responsibilities [kGmpStripRows or 4][64]
  const float* const P = a.P;
  SK_STAMP(8, 0);
  // (every global load of the prologue in flight before the first use: the variables were written by the previous step's last
  //  launch, one memory round trip is ~2 us)
  const float ml = (wave == 0 && lane < K) ? P[a.gmp_mix + lane] : -INFINITY;
  for (int i0 = 0; i0 < KL; i0 += 1024) {
    float lc[4], rw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = min(i0 + tid + 256 * j, KL - 1);
      lc[j] = P[a.gmp_off + i]; rw[j] = P[a.gmp_raw + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + tid + 256 * j;
      if (i < KL) {
        const int k = i / L, l = i - k * L;
        s_loc[k * LD + l] = lc[j];
        s_is[k * LD + l] = 1.f / softplusf_(rw[j]);
      }
    }
  }
  __syncthreads();
  for (int k = wave; k < K; k += 4) {
    float ls = 0.f;
    for (int l = lane; l < L; l += 64) ls += logf(s_is[k * LD + l]);       // = -log s
    ls = row16_sum(ls); ls += __shfl_xor(ls, 16, 64); ls += __shfl_xor(ls, 32, 64);
    if (lane == 0) s_c[k] = ls;
  }
  __syncthreads();
  if (wave == 0) {
    const float mx = wave_max(ml);
    const float em = lane < K ? expf(ml - mx) : 0.f;
    const float se = wave_sum(em);
    if (lane < K) { s_c[lane] = ml - (mx + logf(se)) + s_c[lane] - 0.5f * kLog2Pi * (float)L; s_w[lane] = em / se; }
  }
  __syncthreads();
  SK_STAMP(8, 1);
  // row r's responsibilities into rs[0..K) (and log p(z) returned in every lane): lane = latent dim (+ 64 q; L <= 256).  The K
  // sums over the dims run 16 components at a time -- DPP row sums, then two shuffles across the wave's four rows, 16 independent
  // chains (one wave_sum of six dependent shuffles per component made a row 2.5 us)
  auto wsum = [](float v) { v = row16_sum(v); v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
  auto wmax = [](float v) { v = row16_max(v); v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); };
  auto row_resp = [&](const int r, float* __restrict__ rs) {
    float zr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zr[q] = a.z[(long long)r * L + min(lane + 64 * q, L - 1)];
    float comp = -INFINITY;                        // lane k ends with component k's term
    for (int k0 = 0; k0 < K; k0 += 16) {
      float p[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = min(k0 + j, K - 1);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = min(lane + 64 * q, L - 1);
          const float t = (zr[q] - s_loc[k * LD + l]) * s_is[k * LD + l];
          acc += lane + 64 * q < L ? t * t : 0.f;
        }
        p[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) p[j] = wsum(p[j]);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (lane == k0 + j && lane < K) comp = s_c[lane] - 0.5f * p[j];
    }
    const float mx = wmax(comp);
    const float se = wsum(lane < K ? expf(comp - mx) : 0.f);
    const float lse = mx + logf(se);
    if (lane < K) rs[lane] = expf(comp - lse);
    return lse;
  };
  if ((int)blockIdx.x < nz) {
    const int r = blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* const rs = s_r + wave * 64;
    const float lse = row_resp(r, rs);
    if (lane == 0) st1o(a.logp + r, lse);
    SK_STAMP(8, 2);
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < L; l += 64) {
      const long long o = (long long)r * L + l;
      const float zz = a.z[o], dzv = a.dz[o], ev = a.eps[o];
      const float rawq = a.qp[(long long)r * 2 * L + L + l] + a.c;
      float pterm = 0.f;
      for (int k = 0; k < K; ++k) {
        const float is = s_is[k * LD + l];
        pterm += rs[k] * (zz - s_loc[k * LD + l]) * is * is;
      }
      const float spq = softplusf_(rawq), sg = fmaxf(spq, a.smin);
      const float dmu = dzv + pterm;
      const float dsg = dmu * ev - 1.f / sg;
      st1o(a.dqp + (long long)r * 2 * L + l, dmu);
      st1o(a.dqp + (long long)r * 2 * L + L + l, (spq > a.smin) ? dsg * sigmoidf_(rawq) : 0.f);
    }
    SK_STAMP(8, 3);
    return;
  }
  const int strip = blockIdx.x - nz;
  const int KLp = (KL + 3) & ~3;
  const int rows_per = (B + a.gmp_n - 1) / a.gmp_n;      // (<= kGmpStripRows: the host's strip count)
  const int rb = strip * rows_per, re = min(B, rb + rows_per);
  for (int r = rb + wave; r < re; r += 4) row_resp(r, s_r + (r - rb) * 64);
  __syncthreads();
  SK_STAMP(8, 2);
  float* const out = const_cast<float*>(a.gmp_part) + (long long)strip * a.gmp_len;      // (SkArgs keeps it const: the W launch only reads it)
  for (int i = tid; i < KL; i += 256) {
    const int k = i / L, l = i - k * L;
    const float is = s_is[k * LD + l], lc = s_loc[k * LD + l];
    const float rawv = P[a.gmp_raw + i];
    float ga = 0.f, gb = 0.f;
    for (int r0 = rb; r0 < re; r0 += 4) {          // four rows' loads in flight (a row at a time: a round trip per row)
      float zv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) zv[j] = a.z[(long long)min(r0 + j, re - 1) * L + l];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (r0 + j < re) {
          const float wr = s_r[(r0 + j - rb) * 64 + k];
          const float t = (zv[j] - lc) * is;
          ga -= wr * t * is;
          gb += wr * (1.f - t * t) * is;
        }
    }
    st1o(out + i, ga);
    st1o(out + KLp + i, gb * sigmoidf_(rawv));
  }
  if (tid < K) {
    const float wk = s_w[tid];
    float ga = 0.f;
    for (int r = rb; r < re; ++r) ga -= s_r[(r - rb) * 64 + tid] - wk;
    st1o(out + 2 * KLp + tid, ga);
  }
  SK_STAMP(8, 3);
}

// ---- the general schedule's first layers over the uint8 batch (enc layer 0: bias (+ ReLU); GMVAE: and the x part of enc_gmm layer 0,
// raw) in the form F1 takes above 128 rows: RT row tiles x 64 strided columns per workgroup, the contraction split over 8 waves as
// 3 exact bf16 piece products on the matrix cores (sk_nn4_u8bf), partial tiles meeting in LDS in wave order.  The grouped fp32 GEMM
// ran these at 46 TFLOP/s (config-5 shard: [512 x 3072] x [3072 x 1024] in 70 us).  D % 16 = 0, widths % 64 = 0.
struct FlxArgs {
  const unsigned char* x;
  const float *W0, *b0, *W1;   // [D][H0] (+ bias [H0]); [D][H1] or null
  float *out0, *out1;          // [B][H0], [B][H1]
  int H0, H1, relu0, B, D;
};
template <int RT>
__global__ __launch_bounds__(kSkThreads) void first_layers_u8bf(const FlxArgs a) {
  constexpr int CJ = RT == 1 ? 1 : 2, NC = RT / CJ;
  __shared__ __attribute__((aligned(16))) float red[kSkWaves * CJ * 16 * 64];      // [wave][jl][4 t + r][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, D = a.D;
  const int nct = (a.H0 + a.H1) / 64;
  const int ct = blockIdx.x % nct, rt = blockIdx.x / nct;
  const bool second = ct * 64 >= a.H0;
  const int c0 = second ? ct * 64 - a.H0 : ct * 64, ldw = second ? a.H1 : a.H0;
  const float* const W = (second ? a.W1 : a.W0) + c0;
  const int r0 = rt * 16 * RT;
  long long arow[RT];
#pragma unroll
  for (int j = 0; j < RT; ++j) arow[j] = (long long)min(r0 + 16 * j + ln, B - 1) * D;
  const int el = tid & 63, er = (tid >> 6) & 3, jl = tid >> 8, ec = el & 15;
  const bool owner = tid < 256 * CJ;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (owner && !second && a.b0) bias = *reinterpret_cast<const float4*>(a.b0 + c0 + 4 * ec);
  f32x4 acc[RT][4];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  sk_nn4_u8bf<RT>(a.x, arow, W, ldw, 4 * ln, 0, (D + 31) / 32, wave, lk, D, acc);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c) __syncthreads();
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * CJ + j) * 16 + 4 * t + r) * 64 + lane] = acc[CJ * c + j][t][r];
    __syncthreads();
    if (!owner) continue;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kSkWaves; ++w)
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] += red[((w * CJ + jl) * 16 + 4 * t + er) * 64 + el];
    const int row = r0 + 16 * (CJ * c + jl) + 4 * (el >> 4) + er;
    if (row < B) {
      if (second) {
        *reinterpret_cast<float4*>(a.out1 + (long long)row * a.H1 + c0 + 4 * ec) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        float4 o = make_float4(v[0] + bias.x, v[1] + bias.y, v[2] + bias.z, v[3] + bias.w);
        if (a.relu0) o = make_float4(relu_nan(o.x), relu_nan(o.y), relu_nan(o.z), relu_nan(o.w));
        *reinterpret_cast<float4*>(a.out0 + (long long)row * a.H0 + c0 + 4 * ec) = o;
      }
    }
  }
}

// ---- the general schedule's row-panel layers at thousands of rows (IWAE: R = B S): out[R][N] = act(A[R][K] W[K][N] + bias (+ the
// row group's addend)), K and N in the dozens to hundreds.  The grouped GEMM's LDS tiles ran these at 35 - 65 fp32 TFLOP/s and a
// quarter of the HBM rate their [R x 512] outputs need (config-5 shard: fwd_y_layers 48, fwd_enc_gmm 53, fwd_dec 49 us).  Here
// a WAVE owns RT row tiles x 64 strided columns for the whole contraction -- operands register-direct, both split into bf16 pieces
// beside the MFMAs (6 piece products per product, sk_mma6) -- so nothing meets in LDS and every store is 16 bytes of one row;
// units (row group, column tile) are dealt to waves in order, up to two problems of the same A side by side (the prior net beside
// encoder_gmm's y part).  An output may also leave as plane_rounds3's bf16 planes (gemm.hpp C3: the next layer's operand).
// K % 32 = 0, N % 64 = 0.
struct RowsProb {
  const float *W, *bias, *addsrc;   // [K][N]; [N] or null; [R / add_div][ld_add] or null
  float* out;                       // [R][N]
  unsigned short* C3;               // or null
  long long c3_stride;
  int N, relu, ld_add, add_div;
  unsigned* amax;                   // or null (np == 1 only): word [unit] receives the bits of the largest |out| that wave wrote (gemm.hpp amax_final)
};
struct RowsArgs {
  const float* A;                   // [R][K]
  int R, K, np;
  RowsProb p[2];
};
template <int RT>
__global__ __launch_bounds__(kSkThreads) void rows_nn_bf6(const RowsArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int R = a.R, K = a.K;
  const int nct0 = a.p[0].N >> 6, nct = nct0 + (a.np > 1 ? a.p[1].N >> 6 : 0);
  const int unit = blockIdx.x * kSkWaves + wave;
  const int rg = unit / nct, ct = unit - rg * nct;
  const int r0 = rg * 16 * RT;
  if (r0 >= R) return;
  const RowsProb& P = ct < nct0 ? a.p[0] : a.p[1];
  const int c0 = (ct < nct0 ? ct : ct - nct0) * 64;
  const int N = P.N;
  const float* const W = P.W + c0 + 4 * ln;
  long long arow[RT];
#pragma unroll
  for (int j = 0; j < RT; ++j) arow[j] = (long long)min(r0 + 16 * j + ln, R - 1) * K;
  f32x4 acc[RT][4];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 8 * lk; k0 < K; k0 += 32) {
    float4 av[RT][2], bv[8];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      av[j][0] = *reinterpret_cast<const float4*>(a.A + arow[j] + k0);
      av[j][1] = *reinterpret_cast<const float4*>(a.A + arow[j] + k0 + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = *reinterpret_cast<const float4*>(W + (long long)(k0 + e) * N);
    __builtin_amdgcn_sched_barrier(0);
    sk_bf16x8 Ap[RT][3];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const float v[8] = {av[j][0].x, av[j][0].y, av[j][0].z, av[j][0].w, av[j][1].x, av[j][1].y, av[j][1].z, av[j][1].w};
      sk_pieces(v, Ap[j]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = t == 0 ? bv[e].x : t == 1 ? bv[e].y : t == 2 ? bv[e].z : bv[e].w;
      sk_bf16x8 Bp[3];
      sk_pieces(v, Bp);
#pragma unroll
      for (int j = 0; j < RT; ++j) sk_mma6(Ap[j], Bp, acc[j][t]);
    }
  }
  // ---- epilogue: accumulator [j][t][r] = out[r0 + 16 j + 4 lk + r][c0 + 4 ln + t]
  const int nb = c0 + 4 * ln;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.bias) bias = *reinterpret_cast<const float4*>(P.bias + nb);
  unsigned vmax = 0;
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * j + 4 * lk + r;
      if (row >= R) continue;
      float v[4] = {acc[j][0][r] + bias.x, acc[j][1][r] + bias.y, acc[j][2][r] + bias.z, acc[j][3][r] + bias.w};
      if (P.addsrc) {
        const float4 q = *reinterpret_cast<const float4*>(P.addsrc + (long long)(row / P.add_div) * P.ld_add + nb);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
      if (P.relu) { v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]); v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]); }
      *reinterpret_cast<float4*>(P.out + (long long)row * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) { const unsigned b = __float_as_uint(v[t]) & 0x7fffffffu; vmax = b > vmax ? b : vmax; }
      if (P.C3) {
        unsigned hi[2], mi[2], lo[2];
        split_pair(v[0], v[1], hi[0], mi[0], lo[0]);
        split_pair(v[2], v[3], hi[1], mi[1], lo[1]);
        unsigned short* const d3 = P.C3 + ((long long)(nb >> 4) * R + row) * 16 + (nb & 15);
        *reinterpret_cast<uint2*>(d3) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(d3 + P.c3_stride) = make_uint2(mi[0], mi[1]);
        *reinterpret_cast<uint2*>(d3 + 2 * P.c3_stride) = make_uint2(lo[0], lo[1]);
      }
    }
  if (P.amax) { vmax = wave_umax(vmax); if (lane == 0) P.amax[unit] = vmax; }
}

}  // namespace gmvae
