// evalf_rows: the forward-only evaluation of the GMVAE at the reference's default sizes (run_gmvae.py: hidden 64, latent 64,
// K = 10; MNIST D = 784) with S importance samples per batch row -- scripts/runners.py:324-333 reuses run_model
// (scripts/gmvae.py:238-267) for the bound BASELINE.json's metric names -- in ONE launch behind the first layers
// (skinny.hpp first_layers_u8bf) and a parameter-only preparation (evalf_prep): three launches per pass instead of twelve,
// and no [B S x 64..128] intermediate ever touches memory.
//
// Design.  A WAVE owns a panel of 16 sample rows (row r = b S + s) and keeps the whole per-row chain in REGISTERS: every
// product runs transposed on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products; weight = A operand from an
// [k/4][outputs][4] LDS image, activation = B operand), whose accumulator holds, in lane (row j = lane % 16, lk = lane / 16),
// columns 16 t + 4 lk + {0..3} of row j -- EXACTLY the B operand the next layer's k-steps (t, component) want from that lane.
// So Gumbel-softmax y -> prior head and encoder_gmm hidden -> q head -> z, log q, log p -> decoder hidden -> logits never leave
// the lane that computed them; row reductions (softmax, the row sums of the log-densities) are two lane shuffles (lk pairs).
// The Philox stream falls the same way: quad 4 t + lk of row j is the four eps values that lane needs for tile t.
// The decoder's output layer (64 -> 784: 77 % of the matrix work) streams its 200 KB weight image through a two-chunk LDS ring
// (LDS-DMA, 10 chunks of 5 column tiles) shared by the workgroup's 8 waves -- the only workgroup-wide synchronisation: one barrier
// per chunk -- and ends in a Bernoulli epilogue of ten issue slots per logit instead of the general kernel's 36:
//     x l - softplus(l) = [x l - max(l, 0)] - log(1 + e^-|l|)
//   * the bracket is exact in fp32 and <= 0, small unless the pixel is mispredicted: it sums without cancellation;
//   * the logs are taken of PRODUCTS: prod (1 + e^-|l|) over the lane's 28 logits of a chunk stays below 2^28, one v_log_f32
//     per chunk instead of one per logit (one fused multiply-add per logit: p <- p + p e).
// A workgroup owns whole batch rows, so the IWAE bound logsumexp_s(log w) - log S is finished in the launch;
// the batch sums meet through per-workgroup slots and a last-arriver (fixed order: the same bits whatever the timing).
//
// The output layer multiplies as bf16 piece products (hi + mid + lo of either operand reproduce its 24-bit significand; 6 of
// the 9 piece products per k step, the dropped ones <= 2^-24 of the product: the arithmetic of dwadam.hpp's and gemm.hpp's triples),
// 12 v_mfma_f32_16x16x32_bf16 (192 cycles) per tile instead of 16 v_mfma_f32_16x16x4_f32 (512); so do the q head and the decoder's
// hidden layer (their pieces live in the LDS image); the two layers whose contraction is y (10 classes) stay fp32 MFMA.
#pragma once
#include "mega3.hpp"

namespace gmvae {

struct EV {
  static constexpr int H = 64, L = 64, K = 10, D = 784, L2 = 128;
  static constexpr int NT = 49, CH = 5, NCH = 10, NTP = NCH * CH;      // 49 column tiles of 16, streamed in 10 chunks of 5 (the 50th tile is padding)
  // A 64-deep layer's weight as THREE bf16 planes (hi, mid, lo: truncation splits with exact residuals, so hi + mid + lo is the fp32
  // weight bit for bit), in tiles of 16 output columns: tile T at T * TW floats; plane p at + p * 512; in a plane the 16-byte unit of
  // (k32 step m, lane group lk, column n) at ((m * 4 + lk) * 16 + n) * 4 floats holds the eight contraction indices
  // k = 16 (2 m + (e >> 2)) + 4 lk + (e & 3), e = 0..7 -- the columns of the layer's INPUT that lane (row, lk) holds in its
  // accumulators of tiles 2 m and 2 m + 1 (v_mfma_f32_16x16x32_bf16's eight-per-lane operands: WHICH index is (lane group,
  // element) is free as long as A and B agree).
  static constexpr int TW = 3 * 512;
  // LDS-resident image (floats).  The two layers whose contraction is y (10 classes) stay fp32 MFMA: element (k, n) of a [16 x N]
  // weight at ((k >> 2) * N + n) * 4 + (k & 3); the q head and the decoder's hidden layer as piece planes.
  static constexpr int Wp = 0;                     // prior head:           16 (10) -> 128, fp32
  static constexpr int Wg0y = Wp + 16 * 128;       // encoder_gmm layer 0, y rows: 16 (10) -> 64, fp32
  static constexpr int Wg1 = Wg0y + 16 * 64;       // q head:               64 -> 128: 8 tiles of pieces
  static constexpr int Wd0 = Wg1 + 8 * TW;         // decoder hidden:       64 -> 64: 4 tiles of pieces
  static constexpr int b_p = Wd0 + 4 * TW;
  static constexpr int b_g0 = b_p + 128;
  static constexpr int b_g1 = b_g0 + 64;
  static constexpr int b_d0 = b_g1 + 128;
  static constexpr int img = (b_d0 + 64 + 255) / 256 * 256;      // 22,016 floats = 86 KB, DMA'd into LDS once per workgroup
  // global only: the output layer's pieces (64 -> 784), tile T at d1 + T * TW, streamed through the ring
  static constexpr int d1 = img;
  static constexpr int b_d1 = d1 + NTP * TW;       // [800]: bias + gen_bias_init (zero beyond 784)
  static constexpr int total = b_d1 + 800;
  // LDS map
  static constexpr int NB = 8;                     // batch rows per table pass
  static constexpr int ring = img;                 // 2 x [5 tiles][TW]
  static constexpr int T_lg = ring + 2 * CH * TW;  // [NB][16] logits (-inf beyond K)
  static constexpr int T_gx = T_lg + NB * 16;      // [NB][64]
  static constexpr int T_ne = T_gx + NB * 64;      // [NB]
  static constexpr int red = T_ne + NB;            // [8 waves][4]
  static constexpr int B1 = red + 32;              // [800] the output layer's bias (+ gen_bias_init)
  static constexpr int T_x = B1 + 800;             // [NB][784] bytes (+ 16): the pass's batch rows of x
  static constexpr int lds = T_x + NB * 784 / 4 + 4;
};
static_assert(EV::lds * 4 <= 160 * 1024, "LDS budget");

struct EvalArgs {
  int B, S;
  const unsigned char* x;        // [B][784] 0/1
  const float *he1, *gx;         // first layers' outputs: relu(x W_y0 + b) [B][64], x W_g0[:D] [B][64]
  const float *Wy1, *by1;        // encoder_y's output layer [64][10], [10] (read once per batch row, straight from the parameters)
  const float* img;              // evalf_prep's image (EV::total floats)
  const float *eps, *u;          // external noise [R][64], [R][10], or null: Philox (aux.hpp noise_vals)
  unsigned long long seed, step, row_base;
  float c, smin, invT;
  float *rows4, *z_out, *y_out, *logits_out;      // optional outputs: [R][4] (log p(x|z), log q, log p, log w), [R][64], [R][10], [B][10]
  float* rows_ws;                // [R][4] (workspace; = rows4 when that is given)
  float* slots;                  // [grid][4] per-workgroup sums
  unsigned* counter;             // arrived workgroups (evalf_prep zeroes it)
  float* tail;                   // [8]: sum_b -bound_b, sum nll, sum kl, sum nent, B
  unsigned long long* dbg;       // diagnostic (GMVAE_EV_STAMPS, tools/evstamps.py): [workgroup][16] device-clock stamps, or null
  // the VAE family (evalf_rows_v): the encoder's output layer [64][2 L], [2 L] (applied once per batch row); VAE_GMP: the mixture
  // prior's variables loc [K][L], raw_scale_diag [K][L], mixture_logits [K] (scripts/vae.py:233-244)
  const float *We1, *be1, *loc, *raw_scale, *mixlog;
};

// evalf_prep: the operand images from the parameters (one thread per image element) and the arrival counter.
struct EvalPrepArgs {
  const float *Wp, *bp, *Wg0, *bg0, *Wg1, *bg1, *Wd0, *bd0, *Wd1, *bd1;
  float gen_bias;
  float* img;
  unsigned* counter;
  int family;                    // 0: GMVAE; 1: the VAE family (only the decoder's sections are built); 2: ... at latent size 2
};
__global__ __launch_bounds__(256) void evalf_prep(const EvalPrepArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e == 0) *a.counter = 0u;
  if (e >= EV::total) return;
  float v = 0.f;
  auto kn = [](const int idx, const int N, int& k, int& n) { const int q = idx >> 2; k = 4 * (q / N) + (idx & 3); n = q % N; };
  int k, n;
  // one float of a piece plane = two bf16 pieces (elements el, el + 1 of a lane's eight) of W[k][16 T + n], W row-major [64][ldw]
  auto plane = [](const float* W, const int ldw, const int ncols, const int off) -> float {
    const int T = off / EV::TW, idx = off % EV::TW;
    const int pl = idx >> 9, bi = (idx & 511) * 2;
    const int m = bi >> 9, lk = (bi >> 7) & 3, ln = (bi >> 3) & 15, el = bi & 7;
    if (16 * T + ln >= ncols) return 0.f;
    unsigned pc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = 16 * (2 * m + ((el + j) >> 2)) + 4 * lk + ((el + j) & 3);
      float r = W[kk * ldw + 16 * T + ln];
      for (int q = 0; q < pl; ++q) r -= __uint_as_float(__float_as_uint(r) & 0xffff0000u);       // exact residuals
      pc[j] = __float_as_uint(r) >> 16;
    }
    return __uint_as_float(pc[0] | (pc[1] << 16));
  };
  if (e < EV::Wd0 && a.family) v = 0.f;           // (the VAE family has no y path and no q head over sample rows)
  else if (e < EV::Wg0y) { kn(e - EV::Wp, 128, k, n); v = k < EV::K ? a.Wp[k * 128 + n] : 0.f; }
  else if (e < EV::Wg1) { kn(e - EV::Wg0y, 64, k, n); v = k < EV::K ? a.Wg0[(EV::D + k) * 64 + n] : 0.f; }
  else if (e < EV::Wd0) v = plane(a.Wg1, 128, 128, e - EV::Wg1);
  else if (e < EV::b_p && a.family == 2) {         // latent size 2: the decoder's hidden layer as a [16 (2) x 64] fp32 image
    if (e - EV::Wd0 < 16 * 64) { kn(e - EV::Wd0, 64, k, n); v = k < 2 ? a.Wd0[k * 64 + n] : 0.f; }
  }
  else if (e < EV::b_p) v = plane(a.Wd0, 64, 64, e - EV::Wd0);
  else if (e < EV::b_d0 && a.family) v = 0.f;
  else if (e < EV::b_g0) v = a.bp[e - EV::b_p];
  else if (e < EV::b_g1) v = a.bg0[e - EV::b_g0];
  else if (e < EV::b_d0) v = a.bg1[e - EV::b_g1];
  else if (e < EV::b_d0 + 64) v = a.bd0[e - EV::b_d0];
  else if (e < EV::d1) v = 0.f;
  else if (e < EV::b_d1) v = plane(a.Wd1, EV::D, EV::D, e - EV::d1);
  else v = e - EV::b_d1 < EV::D ? a.bd1[e - EV::b_d1] + a.gen_bias : 0.f;
  a.img[e] = v;
}

__device__ __forceinline__ float ev_lk_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float ev_lk_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ f32x4 ev_mfma4(const float4 a, const f32x4 b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x4 ev_ld(const float* p) { const float4 v = *reinterpret_cast<const float4*>(p); return f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ f32x4 ev_relu(f32x4 v) { return f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])}; }
typedef __bf16 ev_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ev_u32x4 __attribute__((ext_vector_type(4)));
// a layer's 64-wide input (four accumulator tiles of a lane) as the B operands of the two k32 steps, in three bf16 pieces
struct EvPieces { ev_u32x4 h[2], m[2], l[2]; };
__device__ __forceinline__ void ev_split(const f32x4 (&v)[4], EvPieces& p) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {               // elements e = 2 jp, 2 jp + 1: tile 2 m + (jp >> 1), components 2 (jp & 1), + 1
      const float v0 = v[2 * m + (jp >> 1)][2 * (jp & 1)], v1 = v[2 * m + (jp >> 1)][2 * (jp & 1) + 1];
      p.h[m][jp] = pack_hi16(v1, v0);
      const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
      p.m[m][jp] = pack_hi16(r1, r0);
      const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      p.l[m][jp] = pack_hi16(s1, s0);
    }
}
// one 16-column output tile of a 64-deep layer: 12 piece products (6 of the 9 per k32 step: the dropped ones are <= 2^-24 of the
// product), smallest first; `tile` = the tile's pieces in LDS + this lane's unit offset ((lk * 16 + ln) * 4 floats)
__device__ __forceinline__ f32x4 ev_tile6(const float* tile, const EvPieces& b, f32x4 acc) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const ev_bf16x8 Ah = __builtin_bit_cast(ev_bf16x8, *reinterpret_cast<const ev_u32x4*>(tile + m * 256));
    const ev_bf16x8 Am = __builtin_bit_cast(ev_bf16x8, *reinterpret_cast<const ev_u32x4*>(tile + 512 + m * 256));
    const ev_bf16x8 Al = __builtin_bit_cast(ev_bf16x8, *reinterpret_cast<const ev_u32x4*>(tile + 1024 + m * 256));
    const ev_bf16x8 bh = __builtin_bit_cast(ev_bf16x8, b.h[m]), bm = __builtin_bit_cast(ev_bf16x8, b.m[m]), bl = __builtin_bit_cast(ev_bf16x8, b.l[m]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bh, acc, 0, 0, 0);
  }
  return acc;
}

// The decoder's output layer + Independent(Bernoulli).log_prob for the (up to) two panels of a wave; every wave of the workgroup
// calls it (the chunk barriers), `actv` says which panels exist.  lpx[pi] = this LANE's share of sum_d x l - softplus(l) of its row
// (the caller adds the four lk lanes).  cc: chunks streamed so far (ring slot cc & 1 holds chunk 0 on entry); more_after: another
// call follows (the last chunk then prefetches chunk 0 again).
template <int MODE>
__device__ __forceinline__ void ev_output_layer(const float* const gd1, float* const ring, const float* const B1, unsigned& cc,
                                                const bool more_after, const EvPieces (&Bp)[2], const bool (&actv)[2],
                                                const unsigned char* const (&xrA)[2], const int wave, const int lane, float (&lpx)[2]) {
  typedef ev_bf16x8 bf16x8;
  typedef ev_u32x4 u32x4;
  const int ln = lane & 15, lk = lane >> 4;
  // ---- output layer + Independent(Bernoulli).log_prob (gmvae.py:254): 10 chunks of 5 column tiles through the LDS ring.
  // A tile: 12 piece products per panel (6 of the 9 per k32 step: the dropped ones are <= 2^-24 of the product), smallest
  // first.  One tile per iteration, software-pipelined: the weight pieces of tile i + 1 are requested from LDS while tile i's
  // matrix instructions run, each followed by a piece of tile i - 1's epilogue (sched_barrier: nothing moves across).
  f32x2_t brk2[2] = {{0.f, 0.f}, {0.f, 0.f}};    // per panel: sum of [x l - max(l, 0)] (two chains: packed fp32)
  float lg2[2] = {0.f, 0.f};                   // ... sum of log2 prod (1 + e^-|l|)
  f32x4 accp[2] = {{-1e30f, -1e30f, -1e30f, -1e30f}, {-1e30f, -1e30f, -1e30f, -1e30f}};   // "tile -1": with x = 0 its epilogue adds exactly nothing
  unsigned xprev[2] = {0u, 0u};
  f32x2_t prod2[2] = {{1.f, 1.f}, {1.f, 1.f}};
  const bool any = actv[0];                    // (panel 1 is active only if panel 0 is)
#pragma unroll 1
  for (int c = 0; c < EV::NCH; ++c, ++cc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                           // chunk cc has landed for every wave; everyone is done with chunk cc - 1
    const bool more = c + 1 < EV::NCH || more_after;
    if (more) dma_copy_m(ring + ((cc + 1) & 1) * (EV::CH * EV::TW), gd1 + ((c + 1) % EV::NCH) * (EV::CH * EV::TW), EV::CH * EV::TW, wave, lane);
    if (any) {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        lg2[pi] += __builtin_amdgcn_logf(prod2[pi][0] * prod2[pi][1]);      // (<= 20 factors in (1, 2] since the last one)
        prod2[pi] = f32x2_t{1.f, 1.f};
      }
      const float* const cb = ring + (cc & 1) * (EV::CH * EV::TW) + ((lk * 16 + ln) << 2);
      u32x4 fr[6];                             // [m][hi, mid, lo]
#pragma unroll
      for (int q = 0; q < 6; ++q) fr[q] = *reinterpret_cast<const u32x4*>(cb + (q % 3) * 512 + (q / 3) * 256);
      f32x4 bias = ev_ld(B1 + 16 * (c * EV::CH) + 4 * lk);
      unsigned xcur[2];
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) xcur[pi] = *reinterpret_cast<const unsigned*>(xrA[pi] + 16 * (c * EV::CH));
#pragma unroll
      for (int i = 0; i < EV::CH; ++i) {       // (unrolled: the loop-carried operand registers rename instead of moving)
        const int in = min(i + 1, EV::CH - 1), Tn = c * EV::CH + in;
        u32x4 fn[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) fn[q] = *reinterpret_cast<const u32x4*>(cb + in * EV::TW + (q % 3) * 512 + (q / 3) * 256);
        const f32x4 biasn = ev_ld(B1 + 16 * Tn + 4 * lk);
        unsigned xnext[2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) xnext[pi] = *reinterpret_cast<const unsigned*>(xrA[pi] + 16 * Tn);
        if (c * EV::CH + i < EV::NT) {         // (uniform: the 50th tile is padding)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          if (pi == 1 && !actv[1]) break;      // (wave-uniform)
          f32x4 acc = bias;
          f32x2_t e2_[2];
          auto piece = [&](const int k) {      // pair pr = k / 6 of the previous tile's four logits, stage k % 6 (packed fp32 where the ISA has it)
            const int pr = k / 6, st = k % 6;
            if (MODE == 1) { __builtin_amdgcn_sched_barrier(0); return; }
            const f32x2_t lam = {accp[pi][2 * pr], accp[pi][2 * pr + 1]};
            if (st == 0) {
              const f32x2_t xf = {(float)((xprev[pi] >> (16 * pr)) & 0xffu), (float)((xprev[pi] >> (16 * pr + 8)) & 0xffu)};
              const f32x2_t mx = {__builtin_amdgcn_fmed3f(lam[0], 0.f, INFINITY), __builtin_amdgcn_fmed3f(lam[1], 0.f, INFINITY)};
              brk2[pi] += xf * lam - mx;       // x l - max(l, 0): exact, <= 0
            } else if (st == 1) {
              e2_[pr][0] = fexp(-fabsf(lam[0]));
            } else if (st == 2) {
              e2_[pr][1] = fexp(-fabsf(lam[1]));
            } else if (st == 3) {
              prod2[pi] = prod2[pi] * e2_[pr] + prod2[pi];
            }
            __builtin_amdgcn_sched_barrier(0);
          };
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fr[3 * m]), Am = __builtin_bit_cast(bf16x8, fr[3 * m + 1]), Al = __builtin_bit_cast(bf16x8, fr[3 * m + 2]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bp[pi].h[m]), bm = __builtin_bit_cast(bf16x8, Bp[pi].m[m]), bl = __builtin_bit_cast(bf16x8, Bp[pi].l[m]);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, bh, acc, 0, 0, 0); piece(6 * m + 0);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, bm, acc, 0, 0, 0); piece(6 * m + 1);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bl, acc, 0, 0, 0); piece(6 * m + 2);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, bh, acc, 0, 0, 0); piece(6 * m + 3);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bm, acc, 0, 0, 0); piece(6 * m + 4);
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, bh, acc, 0, 0, 0); piece(6 * m + 5);
          }
          accp[pi] = acc; xprev[pi] = xcur[pi]; xcur[pi] = xnext[pi];
        }
        }
        bias = biasn;
#pragma unroll
        for (int q = 0; q < 6; ++q) fr[q] = fn[q];
      }
    }
  }
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    lpx[pi] = 0.f;
    if (!actv[pi]) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                  // the last tile's epilogue
      const float lam = accp[pi][r];
      const float xf = (float)((xprev[pi] >> (8 * r)) & 0xffu);
      brk2[pi][r & 1] += fmaf(xf, lam, -fmaxf(lam, 0.f));
      prod2[pi][r & 1] = fmaf(prod2[pi][r & 1], fexp(-fabsf(lam)), prod2[pi][r & 1]);
    }
    lg2[pi] += __builtin_amdgcn_logf(prod2[pi][0] * prod2[pi][1]);
    lpx[pi] = (brk2[pi][0] + brk2[pi][1]) - 0.693147180559945309f * lg2[pi];
  }
}

template <int MODE>      // 0: the kernel; 1 / 2: timing experiments (tools/evstamps.py): the output layer without its epilogue / without its matrix instructions
__global__ __launch_bounds__(kMT) void evalf_rows(const EvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int H = EV::H, L = EV::L, K = EV::K, D = EV::D;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, S = a.S;
  float* const img = sm;
  float* const ring = sm + EV::ring;
  float *T_lg = sm + EV::T_lg, *T_gx = sm + EV::T_gx, *T_ne = sm + EV::T_ne, *red = sm + EV::red, *T_x = sm + EV::T_x;
#define EV_ST(i) if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + (i)] = wall_clock64()
  EV_ST(0);
  const int nbt = (B + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_begin = min(B, (int)blockIdx.x * nbt), b_end = min(B, b_begin + nbt);
  dma_copy_m(img, a.img, EV::img, wave, lane);
  const float* const gd1 = a.img + EV::d1;
  float* const B1 = sm + EV::B1;
  for (int i = tid; i < 784 / 4; i += kMT) *reinterpret_cast<float4*>(B1 + 4 * i) = *reinterpret_cast<const float4*>(a.img + EV::b_d1 + 4 * i);
  float w_loss = 0.f, w_nl = 0.f, w_kl = 0.f, w_ne = 0.f;      // this wave's share of the workgroup's sums (lane 0)
  unsigned cc = 0;                                              // chunks of the output layer's image streamed so far (ring slot cc & 1)
  bool ring_primed = false;
  for (int bb = b_begin; bb < b_end; bb += EV::NB) {
    const int nb = min(EV::NB, b_end - bb);
    __syncthreads();                               // the tables of the previous pass are dead
    // ---- per batch row: gx; logits = relu(x W_y0 + b) W_y1 + b (scripts/gmvae.py:238), log pi, sum pi log pi (utils.py:165-170)
    for (int i = tid; i < nb * 16; i += kMT) {
      const int b = i >> 4, q = i & 15;
      *reinterpret_cast<float4*>(T_gx + b * 64 + 4 * q) = *reinterpret_cast<const float4*>(a.gx + (long long)(bb + b) * H + 4 * q);
    }
    for (int i = tid; i < nb * (D / 16); i += kMT)  // (784 = 49 x 16 bytes per row; rows are 16-byte aligned: 784 % 16 = 0)
      *reinterpret_cast<float4*>(T_x + 4 * i) = *reinterpret_cast<const float4*>(a.x + (long long)bb * D + 16ll * i);
    {
      // logits: the batch rows' hidden activations and encoder_y's output layer staged in the (not yet used) ring -- a 64-deep dot
      // product per (row, class) from LDS instead of 64 dependent global loads (3 us of the pass)
      // (the ring slot that is NOT receiving the next chunk: slot cc & 1 holds -- from the second pass on -- the prefetched chunk)
      float* const S_h = ring + ((cc + 1) & 1) * (EV::CH * EV::TW);       // [NB][64]
      float* const S_w = S_h + EV::NB * 64;        // [64][10] + [10]
      for (int i = tid; i < nb * 16; i += kMT)
        *reinterpret_cast<float4*>(S_h + 4 * i) = *reinterpret_cast<const float4*>(a.he1 + (long long)bb * H + 4 * i);
      for (int i = tid; i < H * K + K; i += kMT) S_w[i] = i < H * K ? a.Wy1[i] : a.by1[i - H * K];
      __syncthreads();
      const int b = tid >> 4, k = tid & 15;        // 16 lanes per batch row (NB <= 32 rows in one pass of the 512 threads)
      const bool bv = b < nb, kv = k < K;
      float lg = -INFINITY;
      if (bv && kv) {
        float acc = S_w[H * K + k];
#pragma unroll 16
        for (int j = 0; j < H; ++j) acc = fmaf(S_h[b * 64 + j], S_w[j * K + k], acc);
        lg = acc;
      }
      const float lga[1] = {bv ? lg : (kv ? 0.f : -INFINITY)};
      float lpa[1];
      cat_log_softmax<Row16, 1>(lga, lpa);
      const float ne = row16_sum(kv ? fexp(lpa[0]) * lpa[0] : 0.f);
      if (bv) {
        T_lg[b * 16 + k] = lg;
        if (k == 0) T_ne[b] = ne;
        if (kv && a.logits_out) a.logits_out[(long long)(bb + b) * K + k] = lg;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (first pass: the small image has landed)
    __syncthreads();
    const long long r_begin = (long long)bb * S, r_end = (long long)(bb + nb) * S;
    const int panels = (int)((r_end - r_begin + 15) >> 4);
    if (!ring_primed) {                            // chunk 0 of the output layer's image
      dma_copy_m(ring, gd1, EV::CH * EV::TW, wave, lane);
      ring_primed = true;
    }
    EV_ST(1);
    // A wave owns TWO panels per round (panels p0 + 2 wave, + 1): it runs the per-row chain for one, then the other, and takes
    // both through the output layer together -- every weight piece read from LDS feeds two matrix instructions (the layer ran at the
    // rate of the 8 waves' LDS reads: 6 KB per wave, tile and panel), the image streams through the ring once per 16 panels
    // instead of once per 8, and a workgroup's 13 panels (200 sample rows) take one round with 7 barriers instead of two.
    for (int p0 = 0; p0 < panels; p0 += 2 * kMW) {
      EV_ST(2 + 4 * min(p0 / (2 * kMW), 2));
      typedef ev_bf16x8 bf16x8;
      typedef ev_u32x4 u32x4;
      EvPieces Bp[2];                              // per panel: the decoder's hidden layer as bf16 pieces (the output layer's B operands)
      float lqA[2] = {0.f, 0.f}, lpA[2] = {0.f, 0.f};
      bool actv[2], rvA[2];
      long long rowA[2];
      int bjA[2];
      const unsigned char* xrA[2];
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int p = p0 + wave + kMW * pi;       // (panels w and w + 8: a round of <= 8 panels puts one on every wave)
        const bool active = p < panels;
        const long long row = r_begin + 16ll * p + ln;
        const bool rv = active && row < r_end;
        const long long rowc = rv ? row : r_end - 1;              // (clamped: lanes of absent rows compute on the last row, store nothing)
        const int bj = (int)(rowc / S) - bb;
        const unsigned long long grow = a.row_base + (unsigned long long)rowc;
        actv[pi] = active; rvA[pi] = rv; rowA[pi] = row; bjA[pi] = bj;
        xrA[pi] = reinterpret_cast<const unsigned char*>(T_x) + bj * D + 4 * lk;       // this row's x bytes (LDS)
        float lq = 0.f, lp_ = 0.f;
        if (active) {
          // ---- y = softmax((logits + Gumbel) / T) (gmvae.py:240): lane (j, lk) holds classes 4 lk + r
          f32x4 y4;
          {
            const f32x4 lg4 = ev_ld(T_lg + bj * 16 + 4 * lk);
            float u4[4] = {0.5f, 0.5f, 0.5f, 0.5f};
            if (a.u) {
  #pragma unroll
              for (int r = 0; r < 4; ++r)
                if (4 * lk + r < K) u4[r] = a.u[rowc * K + 4 * lk + r];
            } else if (lk < (K + 3) / 4) {
              noise_vals(grow, (unsigned)lk, true, a.seed, a.step, u4);
            }
            float av[4], m = -INFINITY;
  #pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = (lg4[r] - flog(-flog(u4[r]))) * a.invT; m = fmaxf(m, av[r]); }
            m = ev_lk_max(m);
            float se = 0.f;
  #pragma unroll
            for (int r = 0; r < 4; ++r) se += fexp(av[r] - m);
            se = ev_lk_sum(se);
            const float lse = m + flog(se);
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              y4[r] = fexp(av[r] - lse);
              if (a.y_out && rv && 4 * lk + r < K) a.y_out[row * K + 4 * lk + r] = y4[r];
            }
          }
          // ---- prior head (gmvae.py:243) and encoder_gmm's hidden layer (gmvae.py:246): contraction over y (one k tile)
          f32x4 pp[8], hg[4], qp[8];
  #pragma unroll
          for (int nt = 0; nt < 8; ++nt) {
            const float4 w4 = *reinterpret_cast<const float4*>(img + EV::Wp + ((lk * 128 + nt * 16 + ln) << 2));
            pp[nt] = ev_mfma4(w4, y4, ev_ld(img + EV::b_p + nt * 16 + 4 * lk));
          }
  #pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float4 w4 = *reinterpret_cast<const float4*>(img + EV::Wg0y + ((lk * 64 + nt * 16 + ln) << 2));
            const f32x4 b0 = ev_ld(img + EV::b_g0 + nt * 16 + 4 * lk), g0 = ev_ld(T_gx + bj * 64 + nt * 16 + 4 * lk);
            hg[nt] = ev_relu(ev_mfma4(w4, y4, b0 + g0));
          }
          // ---- q head (exact bf16 piece products: ev_tile6)
          {
            EvPieces hp;
            ev_split(hg, hp);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              qp[nt] = ev_tile6(img + EV::Wg1 + nt * EV::TW + ((lk * 16 + ln) << 2), hp, ev_ld(img + EV::b_g1 + nt * 16 + 4 * lk));
          }
          // ---- z = mu + sigma eps, log q(z|x,y), log p(z|y) (gmvae.py:248,258; base.py:66-72)
          f32x4 z[4];
  #pragma unroll
          for (int t = 0; t < 4; ++t) {
            float e4[4];
            if (a.eps) {
              const float4 v = *reinterpret_cast<const float4*>(a.eps + rowc * L + 16 * t + 4 * lk);
              e4[0] = v.x; e4[1] = v.y; e4[2] = v.z; e4[3] = v.w;
            } else {
              noise_vals(grow, (unsigned)(4 * t + lk), false, a.seed, a.step, e4);
            }
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              float s_;
              const float sg = fmaxf(softplus_sig(qp[t + 4][r] + a.c, s_), a.smin);
              const float ee = e4[r];
              const float zz = fmaf(sg, ee, qp[t][r]);
              lq += -0.5f * ee * ee - 0.5f * kLog2Pi - flog(sg);          // (z - mu) / sigma IS eps
              const float sp = fmaxf(softplus_sig(pp[t + 4][r] + a.c, s_), a.smin);
              const float tt = (zz - pp[t][r]) * __builtin_amdgcn_rcpf(sp);
              lp_ += -0.5f * tt * tt - 0.5f * kLog2Pi - flog(sp);
              z[t][r] = zz;
            }
            if (a.z_out && rv) *reinterpret_cast<float4*>(a.z_out + row * L + 16 * t + 4 * lk) = make_float4(z[t][0], z[t][1], z[t][2], z[t][3]);
          }
          lqA[pi] = ev_lk_sum(lq); lpA[pi] = ev_lk_sum(lp_);
          // ---- decoder hidden layer (gmvae.py:251), its output as the pieces the output layer multiplies
          {
            EvPieces zp;
            ev_split(z, zp);
            f32x4 hd[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              hd[nt] = ev_relu(ev_tile6(img + EV::Wd0 + nt * EV::TW + ((lk * 16 + ln) << 2), zp, ev_ld(img + EV::b_d0 + nt * 16 + 4 * lk)));
            ev_split(hd, Bp[pi]);
          }
        }
      }
      EV_ST(3 + 4 * min(p0 / (2 * kMW), 2));
      float lpxl[2];
      ev_output_layer<MODE>(gd1, ring, B1, cc, p0 + 2 * kMW < panels || bb + EV::NB < b_end, Bp, actv, xrA, wave, lane, lpxl);
      EV_ST(4 + 4 * min(p0 / (2 * kMW), 2));
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        if (!actv[pi]) break;
        const float lpx = ev_lk_sum(lpxl[pi]);
        const float lw = lpx + lpA[pi] - lqA[pi] - T_ne[bjA[pi]];
        if (lk == 0 && rvA[pi]) {
          const float4 o = make_float4(lpx, lqA[pi], lpA[pi], lw);
          st4o(a.rows_ws + rowA[pi] * 4, o);
          if (a.rows4 && a.rows4 != a.rows_ws) *reinterpret_cast<float4*>(a.rows4 + rowA[pi] * 4) = o;
        }
      }
    }
    EV_ST(14);
    // ---- the IWAE bound of this pass's batch rows: logsumexp_s(log w) - log S; a wave per batch row, lanes over s
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float logS = flog((float)S);
    for (int b = wave; b < nb; b += kMW) {
      const float* const rw = a.rows_ws + ((long long)(bb + b) * S) * 4;
      float mx = -INFINITY, se = 0.f, nl = 0.f, kl = 0.f;
      if (S <= 64) {                               // one sample row per lane: ONE 16-byte agent-scope load each
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (lane < S) { v = granule2_load(reinterpret_cast<const unsigned long long*>(rw + 4 * lane)); asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
        const float lw = lane < S ? __uint_as_float(v[3]) : -INFINITY;
        mx = Wave64::max(lw);
        se = lane < S ? fexp(lw - mx) : 0.f;
        nl = lane < S ? -__uint_as_float(v[0]) : 0.f;
        kl = lane < S ? __uint_as_float(v[1]) - __uint_as_float(v[2]) : 0.f;
      } else {
        for (int s = lane; s < S; s += 64) mx = fmaxf(mx, ld_sc(rw + 4 * s + 3));
        mx = Wave64::max(mx);
        for (int s = lane; s < S; s += 64) {
          se += fexp(ld_sc(rw + 4 * s + 3) - mx);
          nl -= ld_sc(rw + 4 * s);
          kl += ld_sc(rw + 4 * s + 1) - ld_sc(rw + 4 * s + 2);
        }
      }
      se = Wave64::sum(se); nl = Wave64::sum(nl); kl = Wave64::sum(kl);
      const float bound = mx + flog(se) - logS;
      w_loss -= bound; w_nl += nl / (float)S; w_kl += kl / (float)S; w_ne += T_ne[b];
    }
  }
  // ---- batch sums: waves -> workgroup slot -> the last workgroup to arrive adds the slots in order
  __syncthreads();
  if (lane == 0) { red[wave * 4] = w_loss; red[wave * 4 + 1] = w_nl; red[wave * 4 + 2] = w_kl; red[wave * 4 + 3] = w_ne; }
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kMW; ++w) s += red[w * 4 + tid];
    st1o(a.slots + (long long)blockIdx.x * 4 + tid, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* const last = reinterpret_cast<unsigned*>(sm + EV::T_ne);      // (the tables are dead)
  if (tid == 0) *last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  EV_ST(15);
  if (*last) {                                     // (workgroup-uniform) every slot in flight at once, then a fixed-order sum
    float* const sl = sm + EV::ring;               // [grid <= 1024][4] (the ring is dead)
    for (unsigned g = tid; g < gridDim.x * 4; g += kMT) sl[g] = ld_sc(a.slots + g);
    __syncthreads();
    if (tid < 4) {
      float s = 0.f;
      for (unsigned g = 0; g < gridDim.x; ++g) s += sl[g * 4 + tid];
      a.tail[tid] = s;
      if (tid == 0) { a.tail[4] = (float)B; a.tail[5] = 0.f; a.tail[6] = 0.f; a.tail[7] = 0.f; *a.counter = 0u; }
    }
  }
}


// evalf_rows_v: the same one-launch evaluation for the VAE family at the reference's sizes (scripts/vae.py:167-185 -- BASELINE
// configs[0]: VAE, latent 2; configs[1]: VAE_GMP, latent 64, K = 10; also the plain VAE at latent 64).  q(z|x) depends on the batch
// row alone: the encoder's output layer runs once per batch row (table T_qp), a sample row draws z = mu + sigma eps, takes log q,
// log p (N(0, I), or the K-component mixture: 16 latent dimensions per lane, components reduced over the four lk lanes, logsumexp
// in the lane) and the decoder (hidden layer: piece products at L = 64, one fp32 k tile at L = 2; output layer: ev_output_layer).
struct EVV {
  static constexpr int imgv = 4 * EV::TW + 64;     // the decoder's hidden layer (pieces or fp32 image) + its bias
  static constexpr int ring = (imgv + 255) / 256 * 256;
  static constexpr int T_qp = ring + 2 * EV::CH * EV::TW;     // [NB][128]
  static constexpr int red = T_qp + EV::NB * 128;
  static constexpr int B1 = red + 32;
  static constexpr int T_x = B1 + 800;
  static constexpr int M_loc = T_x + EV::NB * 784 / 4 + 4;    // [10][64]
  static constexpr int M_inv = M_loc + 640;
  static constexpr int M_c = M_inv + 640;          // [16]
  static constexpr int lds = M_c + 16;
};
template <int MODEL, int L>      // MODEL 0: VAE (N(0, I) prior), 1: VAE_GMP (K = 10 mixture prior); L = 2 or 64
__global__ __launch_bounds__(kMT) void evalf_rows_v(const EvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int H = EV::H, D = EV::D, K = 10, L2 = 2 * L, LT = (L + 15) / 16;
  static_assert(L == 2 || L == 64, "latent sizes of the reference's configurations");
  static_assert(MODEL == 0 || L == 64, "VAE_GMP: latent 64");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, S = a.S;
  float* const img = sm;                           // [0, 4 TW): Wd0; then b_d0
  float* const ring = sm + EVV::ring;
  float *T_qp = sm + EVV::T_qp, *red = sm + EVV::red, *B1 = sm + EVV::B1, *T_x = sm + EVV::T_x;
  float *M_loc = sm + EVV::M_loc, *M_inv = sm + EVV::M_inv, *M_c = sm + EVV::M_c;
  const int nbt = (B + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_begin = min(B, (int)blockIdx.x * nbt), b_end = min(B, b_begin + nbt);
  dma_copy_m(img, a.img + EV::Wd0, 4 * EV::TW, wave, lane);
  if (tid < 64) img[4 * EV::TW + tid] = a.img[EV::b_d0 + tid];
  const float* const gd1 = a.img + EV::d1;
  for (int i = tid; i < 784 / 4; i += kMT) *reinterpret_cast<float4*>(B1 + 4 * i) = *reinterpret_cast<const float4*>(a.img + EV::b_d1 + 4 * i);
  if (MODEL == 1) {                                // mixture constants (scripts/vae.py:233-244): 1 / s, loc, log w_k - sum log s - L/2 log 2 pi
    float lnw = 0.f;
    {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, a.mixlog[k]);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(a.mixlog[k] - m);
      lnw = m + logf(se);                          // the normaliser; log w_k = mixlog[k] - lnw
    }
    for (int k = wave; k < K; k += kMW) {
      const float sc = softplusf_(a.raw_scale[k * L + lane % L]);
      const float iv = 1.f / sc;
      M_inv[k * 64 + lane] = iv;
      M_loc[k * 64 + lane] = a.loc[k * L + lane % L];
      const float ls = Wave64::sum(lane < L ? logf(iv) : 0.f);
      if (lane == 0) M_c[k] = a.mixlog[k] - lnw + ls - 0.5f * (float)L * kLog2Pi;
    }
  }
  float w_loss = 0.f, w_nl = 0.f, w_kl = 0.f;
  unsigned cc = 0;
  bool ring_primed = false;
  for (int bb = b_begin; bb < b_end; bb += EV::NB) {
    const int nb = min(EV::NB, b_end - bb);
    __syncthreads();
    for (int i = tid; i < nb * (D / 16); i += kMT)
      *reinterpret_cast<float4*>(T_x + 4 * i) = *reinterpret_cast<const float4*>(a.x + (long long)bb * D + 16ll * i);
    {
      // q's parameters of this pass's batch rows: [mu | raw] = relu(x W_e0 + b) W_e1 + b (scripts/vae.py:170, base.py:66-72)
      float* const S_h = ring + ((cc + 1) & 1) * (EV::CH * EV::TW);       // [NB][64] (the ring slot not receiving a chunk)
      for (int i = tid; i < nb * 16; i += kMT)
        *reinterpret_cast<float4*>(S_h + 4 * i) = *reinterpret_cast<const float4*>(a.he1 + (long long)bb * H + 4 * i);
      __syncthreads();
      for (int i = tid; i < nb * L2; i += kMT) {
        const int b = i / L2, col = i - b * L2;
        float acc = a.be1[col];
#pragma unroll 16
        for (int j = 0; j < H; ++j) acc = fmaf(S_h[b * 64 + j], a.We1[j * L2 + col], acc);
        T_qp[b * 128 + col] = acc;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long r_begin = (long long)bb * S, r_end = (long long)(bb + nb) * S;
    const int panels = (int)((r_end - r_begin + 15) >> 4);
    if (!ring_primed) { dma_copy_m(ring, gd1, EV::CH * EV::TW, wave, lane); ring_primed = true; }
    for (int p0 = 0; p0 < panels; p0 += 2 * kMW) {
      EvPieces Bp[2];
      float lqA[2] = {0.f, 0.f}, lpA[2] = {0.f, 0.f};
      bool actv[2], rvA[2];
      long long rowA[2];
      const unsigned char* xrA[2];
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int p = p0 + wave + kMW * pi;       // (panels w and w + 8: a round of <= 8 panels puts one on every wave)
        const bool active = p < panels;
        const long long row = r_begin + 16ll * p + ln;
        const bool rv = active && row < r_end;
        const long long rowc = rv ? row : r_end - 1;
        const int bj = (int)(rowc / S) - bb;
        const unsigned long long grow = a.row_base + (unsigned long long)rowc;
        actv[pi] = active; rvA[pi] = rv; rowA[pi] = row;
        xrA[pi] = reinterpret_cast<const unsigned char*>(T_x) + bj * D + 4 * lk;
        if (active) {
          // ---- z = mu + sigma eps, log q(z|x) (vae.py:171,181; base.py:66-72); lane (j, lk) holds latent dimensions 16 t + 4 lk + r
          f32x4 z[4];
          float lq = 0.f, lp_ = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < LT; ++t) {
            float e4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.eps) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * t + 4 * lk + r < L) e4[r] = a.eps[rowc * L + 16 * t + 4 * lk + r];
            } else if (4 * (4 * t + lk) < L) {
              noise_vals(grow, (unsigned)(4 * t + lk), false, a.seed, a.step, e4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int col = 16 * t + 4 * lk + r;
              if (col < L) {
                float s_;
                const float sg = fmaxf(softplus_sig(T_qp[bj * 128 + L + col] + a.c, s_), a.smin);
                const float ee = e4[r];
                const float zz = fmaf(sg, ee, T_qp[bj * 128 + col]);
                lq += -0.5f * ee * ee - 0.5f * kLog2Pi - flog(sg);
                if (MODEL == 0) lp_ += -0.5f * zz * zz - 0.5f * kLog2Pi;     // N(0, I): vae.py:247-250
                z[t][r] = zz;
              }
            }
            if (a.z_out && rv) {
              if (L >= 16) *reinterpret_cast<float4*>(a.z_out + row * L + 16 * t + 4 * lk) = make_float4(z[t][0], z[t][1], z[t][2], z[t][3]);
              else if (lk == 0) { a.z_out[row * L] = z[0][0]; a.z_out[row * L + 1] = z[0][1]; }
            }
          }
          lqA[pi] = ev_lk_sum(lq);
          if (MODEL == 0) lpA[pi] = ev_lk_sum(lp_);
          else {                                     // MixtureSameFamily.log_prob (vae.py:240-244): logsumexp_k [log w_k + log N(z; loc_k, s_k)]
            float comp[K], m = -INFINITY;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              float acc = 0.f;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const f32x4 lc = ev_ld(M_loc + k * 64 + 16 * t + 4 * lk), iv = ev_ld(M_inv + k * 64 + 16 * t + 4 * lk);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float tt = (z[t][r] - lc[r]) * iv[r]; acc = fmaf(-0.5f * tt, tt, acc); }
              }
              comp[k] = M_c[k] + ev_lk_sum(acc);
              m = fmaxf(m, comp[k]);
            }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) se += fexp(comp[k] - m);
            lpA[pi] = m + flog(se);
          }
          // ---- decoder hidden layer (vae.py:174)
          f32x4 hd[4];
          if (L == 64) {
            EvPieces zp;
            ev_split(z, zp);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              hd[nt] = ev_relu(ev_tile6(img + nt * EV::TW + ((lk * 16 + ln) << 2), zp, ev_ld(img + 4 * EV::TW + nt * 16 + 4 * lk)));
          } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              hd[nt] = ev_relu(ev_mfma4(*reinterpret_cast<const float4*>(img + ((lk * 64 + nt * 16 + ln) << 2)), z[0], ev_ld(img + 4 * EV::TW + nt * 16 + 4 * lk)));
          }
          ev_split(hd, Bp[pi]);
        }
      }
      float lpxl[2];
      ev_output_layer<0>(gd1, ring, B1, cc, p0 + 2 * kMW < panels || bb + EV::NB < b_end, Bp, actv, xrA, wave, lane, lpxl);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        if (!actv[pi]) break;
        const float lpx = ev_lk_sum(lpxl[pi]);
        const float lw = lpx + lpA[pi] - lqA[pi];
        if (lk == 0 && rvA[pi]) {
          const float4 o = make_float4(lpx, lqA[pi], lpA[pi], lw);
          st4o(a.rows_ws + rowA[pi] * 4, o);
          if (a.rows4 && a.rows4 != a.rows_ws) *reinterpret_cast<float4*>(a.rows4 + rowA[pi] * 4) = o;
        }
      }
    }
    // ---- the IWAE bound of this pass's batch rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float logS = flog((float)S);
    for (int b = wave; b < nb; b += kMW) {
      const float* const rw = a.rows_ws + ((long long)(bb + b) * S) * 4;
      float mx = -INFINITY, se = 0.f, nl = 0.f, kl = 0.f;
      for (int s = lane; s < S; s += 64) mx = fmaxf(mx, ld_sc(rw + 4 * s + 3));
      mx = Wave64::max(mx);
      for (int s = lane; s < S; s += 64) {
        se += fexp(ld_sc(rw + 4 * s + 3) - mx);
        nl -= ld_sc(rw + 4 * s);
        kl += ld_sc(rw + 4 * s + 1) - ld_sc(rw + 4 * s + 2);
      }
      se = Wave64::sum(se); nl = Wave64::sum(nl); kl = Wave64::sum(kl);
      w_loss -= mx + flog(se) - logS; w_nl += nl / (float)S; w_kl += kl / (float)S;
    }
  }
  __syncthreads();
  if (lane == 0) { red[wave * 4] = w_loss; red[wave * 4 + 1] = w_nl; red[wave * 4 + 2] = w_kl; red[wave * 4 + 3] = 0.f; }
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kMW; ++w) s += red[w * 4 + tid];
    st1o(a.slots + (long long)blockIdx.x * 4 + tid, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* const last = reinterpret_cast<unsigned*>(red);
  if (tid == 0) *last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (*last) {
    float* const sl = ring;
    for (unsigned g = tid; g < gridDim.x * 4; g += kMT) sl[g] = ld_sc(a.slots + g);
    __syncthreads();
    if (tid < 4) {
      float s = 0.f;
      for (unsigned g = 0; g < gridDim.x; ++g) s += sl[g * 4 + tid];
      a.tail[tid] = s;
      if (tid == 0) { a.tail[4] = (float)B; a.tail[5] = 0.f; a.tail[6] = 0.f; a.tail[7] = 0.f; *a.counter = 0u; }
    }
  }
}
static_assert(EVV::lds * 4 <= 160 * 1024, "LDS budget");

#undef EV_ST

}  // namespace gmvae
