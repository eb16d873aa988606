// mega_fwd_bwd: the whole per-row part of the GMVAE training step in ONE launch.
//
// A workgroup owns 16 batch rows and runs, without leaving the CU:
//   F  the forward chain of chain.hpp (slab reduce -> logits -> Gumbel-softmax -> heads -> z -> decoder hidden),
//   D  the decoder output layer streamed in 128-column chunks through an LDS ring filled by LDS-DMA:
//      lambda = hd1*Wd1 + b (scripts/base.py:133-135), Bernoulli log-likelihood (base.py:143-146,
//      gmvae.py:254), g = sigmoid(lambda) - x, and IN THE SAME PASS the data gradient
//      dhd1 += g_chunk * Wd1_chunk^T (both products read the same chunk image),
//   B  the backward chain of chain.hpp on the accumulated dhd1.
// Only the batch-reduced weight gradients (one grouped TN launch) and the optimiser remain outside.
//
// ONE weight image serves both directions: every matrix is stored row-major with an ODD leading
// dimension, so the forward operand B(k,n) = W[k*ld + n] and the transposed backward operand
// B(k,n) = W[n*ld + k] are both bank-conflict-free fragment reads.
#pragma once
#include "chain.hpp"

namespace gmvae {

constexpr int kCW = 128;       // decoder columns per streamed chunk (finalize_adam's image scatter shifts by 7)
constexpr int kFlLda = 226;    // row stride of the in-launch first layer's x image: = 2 (mod 32), >= D/4 rounded up to 4
constexpr int kMW = 8;         // wavefronts per workgroup (2 per SIMD: the partner hides LDS/DMA/VALU latency)
constexpr int kMT = kMW * 64;

struct MegaLay {
  int KP, K2, L2, LP;
  int ldY1, ldG0, ldP, ldG1, ldD0, ldc;
  int W_y1, W_g0y, W_p, W_g1, W_d0, b_y0, b_y1, b_g0, b_p, b_g1, b_d0, img;   // small-weight image [0, img)
  int img_early;                     // [0, img_early): what the first stages need; the rest lands while they run
  int fl_kq, fl_ld, fl_W, fl_A, fl_ok;   // in-launch first layer: k rows per quarter, staging [kq][fl_ld] + x image [kq][17]
  int chunk, nch;                    // floats per decoder chunk image ([H+1][ldc], row H = bias slice), chunks
  int ring, xring;                   // LDS: ring of 2 chunk images (aliases the small-weight image), 2 x-chunks
  int A_hy, A_y, A_hg, A_z, A_hd, A_g;          // forward / decoder operand images ([k][17])
  int A_dhd, A_dqp, A_cat, A_dl;                // backward operand images (overlay the same region)
  int P_lg, P_qp, P_pp, P_eps, P_z, P_hd, P_hg, P_hy, P_y, P_gx, P_u, P_dz, P_dy, nll, red, total;
  int ldM, M_loc, M_raw, M_mix;      // VAE_GMP: prior variables in the image ([K][L+1], [K][L+1], [KP])
  int M_inv, M_c, M_w, P_r;          // VAE_GMP: 1/s [K][L+1], per-component constants [KP], softmax weights [KP], resp [16][KP]
};
// model: 0 VAE (std-normal prior), 1 VAE_GMP (learned mixture prior), 2 GMVAE
__host__ __device__ inline MegaLay mega_lay(int H, int L, int K, int D, int model = 2) {
  MegaLay m;
  m.KP = (K + 15) & ~15; m.K2 = (K + 3) & ~3; m.L2 = 2 * L; m.LP = (L + 15) & ~15;
  m.ldY1 = m.KP + 1; m.ldG0 = H + 1; m.ldP = m.L2 + 1; m.ldG1 = m.L2 + 1; m.ldD0 = H + 1; m.ldc = kCW + 1;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += GMVAE_P4(n); return r; };
  m.b_y0 = take(H); m.b_y1 = take(m.KP); m.b_g0 = take(H); m.b_p = take(m.L2); m.b_g1 = take(m.L2); m.b_d0 = take(H);
  m.W_y1 = take(H * m.ldY1);       // [H][KP+1]
  m.W_g0y = take(m.K2 * m.ldG0);   // [K2][H+1]
  m.W_p = take(m.K2 * m.ldP);      // [K2][L2+1]
  m.ldM = L + 1;
  m.M_loc = m.M_raw = m.M_mix = m.M_inv = m.M_c = m.M_w = m.P_r = 0;
  if (model == 1) { m.M_loc = take(K * m.ldM); m.M_raw = take(K * m.ldM); m.M_mix = take(m.KP); }
  m.img_early = o = GMVAE_P256(o);
  m.W_g1 = take(H * m.ldG1);       // [H][L2+1]
  m.W_d0 = take(GMVAE_P4(L) * m.ldD0);   // [L][H+1] (+ zero rows up to a multiple of 4: the k-steps of z * Wd0)
  m.img = GMVAE_P256(o);
  m.chunk = GMVAE_P256((H + 1) * m.ldc);
  m.nch = (D + kCW - 1) / kCW;
  m.ring = 0;
  o = (m.img > 2 * m.chunk) ? m.img : 2 * m.chunk;
  m.xring = take(2 * kPanel * kCW / 4);          // 2 x [16][128] bytes
  const int abase = o;
  m.A_hy = take(H * kLDA); m.A_y = take(m.K2 * kLDA); m.A_hg = take(H * kLDA); m.A_z = take(GMVAE_P4(L) * kLDA);
  m.A_hd = take(H * kLDA); m.A_g = take(kCW * kLDA);
  const int aend_f = o;
  o = abase;
  m.A_dhd = take(H * kLDA); m.A_dqp = take(m.L2 * kLDA); m.A_cat = take((H + m.L2) * kLDA); m.A_dl = take(m.K2 * kLDA);
  o = o > aend_f ? o : aend_f;
  m.P_lg = take(kPanel * m.KP); m.P_qp = take(kPanel * m.L2); m.P_pp = take(kPanel * m.L2); m.P_eps = take(kPanel * L);
  m.P_z = take(kPanel * L); m.P_hd = take(kPanel * H); m.P_hg = take(kPanel * H); m.P_hy = take(kPanel * H);
  m.P_y = take(kPanel * K);
  const int pb = o;
  m.P_gx = take(kPanel * H); m.P_u = take(kPanel * K);
  const int pe_f = o;
  o = pb;
  m.P_dz = take(kPanel * m.LP); m.P_dy = take(kPanel * m.KP);
  o = o > pe_f ? o : pe_f;
  m.nll = take(4 * kPanel);
  m.red = take(kMW * 256);
  if (model == 1) { m.M_inv = take(K * m.ldM); m.M_c = take(m.KP); m.M_w = take(m.KP); m.P_r = take(kPanel * m.KP); }
  m.total = o;
  // first layer inside the launch: each of the 4 workgroups of a panel takes a quarter of the D input columns.
  // Its weight rows are staged as they lie in memory ([kq][H] per tensor: linear 1 KB DMA bursts; the 2-way bank
  // conflict of the MFMA operand reads costs far less than padded per-row DMAs did).  The staging overlays
  // everything below the row-sum / reduction scratch (nothing else is live yet).
  {
    const int nw = model == 2 ? 2 : 1;
    m.fl_kq = (((D + 3) / 4) + 3) & ~3;
    m.fl_ld = H;
    m.fl_W = 0;
    m.fl_A = GMVAE_P256(nw * m.fl_kq * H);
    m.fl_ok = (m.fl_A + kPanel * kFlLda <= m.nll) && m.fl_kq <= kFlLda && D % 16 == 0 && nw * H <= 128 && H % 16 == 0 && L % 2 == 0 &&
              H == 64 && (m.fl_kq * H) % 256 == 0 && kPanel * m.fl_kq / 4 <= 2 * kMT &&
              kPanel * ((L + 3) / 4) + (model == 2 ? kPanel * ((K + 3) / 4) : 0) <= kMT;
  }
  return m;
}

// All-reduce over the 16 lanes of a DPP row by rotations (row_ror 8, 4, 2, 1): 4 VALU instructions instead of
// 4 dependent trips through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(const float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
// A launch OUTPUT in global memory (nothing in this launch reads it back; the next launch does, from other CUs): stored
// WRITE-THROUGH (sc1), so the bytes leave for memory while the launch is still running instead of waiting dirty in the
// XCD's L2 for the end-of-kernel write-back that the next launch's start sits behind.  (Inline asm: uncounted by the
// compiler's vmcnt bookkeeping, which only makes its own waits stricter -- retirement is in order; `s_nop 1`: the data
// registers may be rewritten right after, cdna_hip_programming.md 5.7 item 1.)
#ifndef M2_WT
#define M2_WT 1
#endif
__device__ __forceinline__ void st4o(float* p, const float4 v) {
#if M2_WT
  const f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
  return v;
}
struct Row16 {                     // reduction policy of gemm.hpp's cat_* helpers: 16 consecutive lanes share a row
  static __device__ __forceinline__ float max(float v) { return row16_max(v); }
  static __device__ __forceinline__ float sum(float v) { return row16_sum(v); }
};
__device__ __forceinline__ float row32_sum(float v) { v = row16_sum(v); return v + __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float row32_max(float v) { v = row16_max(v); return fmaxf(v, __shfl_xor(v, 16, 64)); }

// partial sums of one 16x16 tile over k-steps [s0, s1) -> acc.  Operand reads run 8 steps ahead of the MFMA chain
// (a 16-deep batch and a blocked k-assignment were both measured SLOWER: tools/stamps.py).  Addresses advance by
// pointer increments: the multiply-and-clamp form cost ~2x the MFMA chain in quarter-rate integer VALU.
// A(row, k) = A[k * ASK + row * ASM]: the [k][17] activation images (ASK = 17, ASM = 1) or a row-major image (ASK = 1).
template <int ASK = kLDA, int ASM = 1>
__device__ __forceinline__ f32x4 tile_ksteps(const float* __restrict__ A, const float* __restrict__ Bw, const int sk,
                                             const int sn, const int tile, const int s0, const int s1, const int /*smax*/,
                                             const int lane, f32x4 acc) {
  const int ln = lane & 15, lk = lane >> 4;
  const float* pa = A + (s0 * 4 + lk) * ASK + ln * ASM;
  const float* pb = Bw + (s0 * 4 + lk) * sk + (tile * 16 + ln) * sn;
  const int sb4 = 4 * sk;
  int n = s1 - s0;
  for (; n >= 8; n -= 8) {
    float av[8], bv[8];
    const float* q = pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      av[j] = pa[j * 4 * ASK];
      bv[j] = *q;
      q += sb4;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
    pa += 32 * ASK;
    pb = q;
  }
  if (n > 0) {                                   // 1..7 steps left (wave-uniform)
    float av[7], bv[7];
    const float* q = pb;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      av[j] = pa[j * 4 * ASK < (n - 1) * 4 * ASK ? j * 4 * ASK : (n - 1) * 4 * ASK];
      bv[j] = *q;
      if (j + 1 < n) q += sb4;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < n) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
  }
  return acc;
}

// 16x16 tiles of A[K][17] x B with B(k,n) = Bw[k*sk + n*sn]; tile t = wave, wave + kMW, ...
template <class Epi>
__device__ __forceinline__ void panel_gemm_s(const float* __restrict__ A, const float* __restrict__ Bw, const int sk,
                                             const int sn, const int K4, const int ntiles, const int wave,
                                             const int lane, Epi epi) {
  const int ln = lane & 15, lk = lane >> 4;
  for (int t = wave; t < ntiles; t += kMW) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_ksteps(A, Bw, sk, sn, t, 0, K4 / 4, K4 / 4, lane, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) epi(lk * 4 + r, t * 16 + ln, acc[r]);
  }
}

template <class Epi>
__device__ __forceinline__ void ksplit_finish(const f32x4 acc, const int tile, float* __restrict__ red, const int wave,
                                              const int lane, Epi epi) {
  const int ln = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lk * 4 + r) * 16 + ln] = acc[r];
  __syncthreads();
  const int tid = wave * 64 + lane;
  if (tid < 256) {
    float v = red[tid];
#pragma unroll
    for (int w = 1; w < kMW; ++w) v += red[w * 256 + tid];
    epi(tid >> 4, tile * 16 + (tid & 15), v);
  }
}

// asynchronous linear copy by the kMW waves of the mega workgroup
__device__ __forceinline__ void dma_copy_m(float* __restrict__ lds_dst, const float* __restrict__ g, const int nfloats,
                                           const int wave, const int lane) {
  for (int c = wave * 256; c < nfloats; c += kMW * 256) {
    const int idx = c + lane * 4;
    if (idx < nfloats) __builtin_amdgcn_global_load_lds(g + idx, lds_dst + c, 16, 0, 0);
  }
}

// First-layer weight staging: [rows][64-float row pieces] copied in 1 KB bursts (4 rows of 64 floats), burst order
// rotated by `rot` (the workgroups of an XCD would otherwise walk the same L2 lines in lockstep), and the two
// 16-column halves of every 32 columns swapped in odd rows: the MFMA operand read of a half-wave (k rows lk = 0,1
// or 2,3, 16 columns each) then covers 32 distinct banks although the row stride is a multiple of 32 floats.
__device__ __forceinline__ void dma_stage_w(float* __restrict__ lds_dst, const float* __restrict__ g, const int nfloats,
                                            const int rot, const int wave, const int lane) {
  const int nb = nfloats >> 8;                     // bursts (nfloats is a multiple of 256 here)
  const int row = lane >> 4, piece = lane & 15;    // H = 64: 16 pieces of 16 bytes per row, 4 rows per burst
  const int src = row * 64 + ((piece ^ ((row & 1) << 2)) << 2);
  for (int b = wave; b < nb; b += kMW) {
    int bb = b + rot;
    bb = bb >= nb ? bb - nb : bb;
    __builtin_amdgcn_global_load_lds(g + (bb << 8) + src, lds_dst + (bb << 8), 16, 0, 0);
  }
}

struct MegaArgs {
  int model;                  // 0 VAE, 1 VAE_GMP, 2 GMVAE
  int B, H, L, K, D, NS;
  int Q;                      // workgroups per panel (see xchg below)
  float c, smin, invT, gen_bias;
  const float* s1;            // [NS][B][2H] split-K partials of X*[Wy0 | Wg0x]
  const float* img;           // small-weight image (mega_lay [0, img))
  const float* dimg;          // decoder chunk images [nch][chunk]
  const unsigned char* x;     // [B][D]
  const float *eps, *u;
  float *hy1, *y, *hg1, *z, *hd1, *g;                           // saved for the weight gradients
  float *dhd1, *dqp, *dpp, *dhg1, *dlogits, *dhy1;              // pre-activation gradients
  float *nent, *logq, *logp, *logpx, *logw;                     // per-row loss terms
  float* gmp_part;            // VAE_GMP: per-workgroup partial gradients of (loc, raw_scale_diag, mixture_logits)
  // Decoder chunks of one panel are spread over Q workgroups (all run the cheap forward chain); quarters
  // 1..Q-1 hand their partial dhd1 tile and Bernoulli row sums to quarter 0 through 8-byte {epoch, value}
  // granules (cdna_hip_programming.md G16 recipe R2: the data is its own flag, no fences).
  unsigned long long* xchg;   // [panels][Q-1][16*H + 16] granules
  const unsigned* epoch_word; // tag of this step (bumped by the first launch of the step)
  unsigned* err_word;         // set to 1 if a bounded spin gives up (results are then invalid)
  unsigned long long* dbg;
  unsigned long long* span;   // measurement: [block][2] wall-clock (100 MHz) at a workgroup's first and last instruction
  int fine;                   // diagnostic: slots 8.. of a block's stamp record take intra-stage stamps instead
  MegaLay lay;                // mega_lay(H, L, K, D, model), computed once on the host
  // fl = 1: the launch also runs the first layer (no separate split-K GEMM launch, no noise / image launch):
  // the panel's 4 workgroups each reduce a quarter of the D input columns and exchange their [16][H2] partials
  // through tagged granules (all four then hold identical sums), draw their own Philox noise, and read weight
  // images the previous step's finalize_adam wrote.
  int fl;
  const float *w0a, *w0b;     // first-layer weights [D][H]: encoder(_y) and, GMVAE, encoder_gmm's x rows
  unsigned long long* xfl;    // [panels][4][16 * H2] granules
  unsigned long long seed;
  unsigned long long row0;    // global index of this device's first batch row (GmvaeDims::row0): Philox counter only
  unsigned long long* step_dev;
  const float *img2f, *img2b, *dimg2;   // mega2_fwd_bwd (mega2.hpp): forward / backward operand images, decoder operand images
  // mega2_fwd_bwd also leaves this step's Adam step size alpha_t = lr sqrt(1 - b2^t) / (1 - b1^t) (fp64, adam_tf's form) for
  // dw_adam: one thread of a workgroup that is done early computes it, instead of every thread of the optimizer launch
  float lr, b1, b2;
  float* lr_t_out;             // (null: not wanted)
};

// HT, LT, KT, DT, MODEL: compile-time sizes of a specialised instance (0 / -1 = read them from the arguments).
// With runtime sizes the launch spends ~1100 instructions on pointer and index set-up before its first DMA and
// keeps ~100 scalars live (spilled to VGPR lanes); with the sizes folded in, the layout is a table of constants.
// FLT = 1: the instance that runs the first layer itself (MegaArgs::fl); a compile-time switch, because the two
// forms of the first stage together push the kernel over its 256 registers.
template <int HT, int LT, int KT, int DT, int MODEL, int FLT>
__global__ __launch_bounds__(kMT) void mega_fwd_bwd(const MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = HT ? HT : a.H, L = LT ? LT : a.L, K = KT ? KT : a.K, D = DT ? DT : a.D, B = a.B;
  const int model = MODEL >= 0 ? MODEL : a.model;
  const bool gm = model == 2, gmp = model == 1;
  // producers (quarters 1..Q-1) take the LOWER block ids: a consumer can then never keep its producers off the chip
  const int nP = (B + kPanel - 1) / kPanel, Q = a.Q;
  const int bid = blockIdx.x;
  const int q = bid < nP * (Q - 1) ? 1 + bid / nP : 0;
  const int pnl = bid < nP * (Q - 1) ? bid % nP : bid - nP * (Q - 1);
  const int r0 = pnl * kPanel;
  if (a.span && tid == 0) a.span[2 * bid] = wall_clock64();
  // Bounded spins: ~2^19 polls (a few 100 ms) before a hand-off gives up, sets the error word and poisons the step
  // with NaN; once the word is set every later wait of this workspace gives up at once (fail fast, no stalls).
  const unsigned spin_limit = (Q > 1 && *a.err_word) ? 0u : (1u << 19);
#define GMVAE_SPAN_END() if (a.span && threadIdx.x == 0) a.span[2 * blockIdx.x + 1] = wall_clock64()
  // The split-K partials of the first layer are the longest (coldest) wait of the launch: their loads go out
  // before anything else (the scheduling barrier keeps the ~1000 instructions of pointer set-up below them).
  const int H2f = gm ? 2 * H : H;
  const int nitem = kPanel * H2f / 4;             // <= 512 for H <= 64: one item per thread
  const int fi = min(tid, nitem - 1);
  const int frow = (fi * 4) / H2f, fcol = (fi * 4) % H2f;
  const float* const sp = a.s1 + (long long)min(r0 + frow, B - 1) * H2f + fcol;
  const long long sstride = (long long)B * H2f;
  float4 so[4];                                   // raw: summed after the DMA issue
  if (!FLT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) so[j] = *reinterpret_cast<const float4*>(sp + (long long)min(j, a.NS - 1) * sstride);
  }
  __builtin_amdgcn_sched_barrier(0);
  const MegaLay f = HT ? mega_lay(H, L, K, D, model) : a.lay;
  const int KP = f.KP, K2 = f.K2, L2 = f.L2, LP = f.LP;
  float *W_y1 = sm + f.W_y1, *W_g0y = sm + f.W_g0y, *W_p = sm + f.W_p, *W_g1 = sm + f.W_g1, *W_d0 = sm + f.W_d0;
  float *b_y0 = sm + f.b_y0, *b_y1 = sm + f.b_y1, *b_g0 = sm + f.b_g0, *b_p = sm + f.b_p, *b_g1 = sm + f.b_g1, *b_d0 = sm + f.b_d0;
  float *A_hy = sm + f.A_hy, *A_y = sm + f.A_y, *A_hg = sm + f.A_hg, *A_z = sm + f.A_z, *A_hd = sm + f.A_hd, *A_g = sm + f.A_g;
  float *A_dhd = sm + f.A_dhd, *A_dqp = sm + f.A_dqp, *A_cat = sm + f.A_cat, *A_dl = sm + f.A_dl;
  float *P_lg = sm + f.P_lg, *P_qp = sm + f.P_qp, *P_pp = sm + f.P_pp, *P_eps = sm + f.P_eps, *P_z = sm + f.P_z;
  float *P_hd = sm + f.P_hd, *P_hg = sm + f.P_hg, *P_hy = sm + f.P_hy, *P_y = sm + f.P_y;
  float *P_gx = sm + f.P_gx, *P_u = sm + f.P_u, *P_dz = sm + f.P_dz, *P_dy = sm + f.P_dy;
  float *nllp = sm + f.nll, *red = sm + f.red;
  float* nllp2 = sm + f.P_gx;                  // [kMW][16] Bernoulli row sums per wave (P_gx is dead after the forward chain)
  unsigned char* xring = reinterpret_cast<unsigned char*>(sm + f.xring);
  float* A_dhg = A_cat;
  float* A_dpp = A_cat + H * kLDA;
  // the hidden activations feeding the q head: encoder_gmm's (GMVAE) or the encoder's own first layer (VAE)
  float* A_hq = gm ? A_hg : A_hy;
  float* P_hq = gm ? P_hg : P_hy;
  float *M_loc = sm + f.M_loc, *M_raw = sm + f.M_raw, *M_mix = sm + f.M_mix, *M_inv = sm + f.M_inv, *M_c = sm + f.M_c,
        *M_w = sm + f.M_w, *P_r = sm + f.P_r;
  const int ldM = f.ldM;

  const bool lead = q == 0;                    // quarter 0 owns the panel: saves activations, runs phase B
  const int nrow = min(kPanel, B - r0);
  const int ln = lane & 15, lk = lane >> 4;
  GMVAE_STAMP(0);
#define GMVAE_FL(i) if (a.dbg && a.fine == 4 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime()
#define GMVAE_FS(i) if (a.dbg && a.fine >= 2 && a.fine != 4 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime()
  // ======================================================================= F: forward chain
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FLT) {
    // ---------------------------------------------------------------- FL: first layer over this quarter's columns
    const int KQ = f.fl_kq, kq4 = KQ / 4;
    float* const Wst = sm + f.fl_W;
    float* const A_x = sm + f.fl_A;
    const int k0 = q * KQ;
    const int kn = max(0, min(KQ, D - k0));        // rows of this quarter that exist (the last quarter may be short)
    const unsigned epoch_fl = *a.epoch_word;
    const unsigned long long step = a.step_dev[0];
    // noise items of this panel: one quad of one row each (L and K are padded to quads per row, as in gmvae_noise_fill)
    const int qer = (L + 3) / 4, qur = gm ? (K + 3) / 4 : 0;
    const int qe = kPanel * qer, qu = kPanel * qur;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    const int ntile = H2f / 16, tpw = H / 16;      // tiles; tiles per weight tensor
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned xw[2];                                // this thread's x bytes: (row, 4 consecutive columns), <= 2 items
    auto x_store = [&]() {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int i = tid + it * kMT;
        if (i < kPanel * kq4) {
          const int row = i / kq4, k4 = (i - row * kq4) * 4;
          const unsigned w = xw[it];               // row-major [16][kFlLda]: consecutive lanes, consecutive 16 bytes
          float2* const dst = reinterpret_cast<float2*>(A_x + row * kFlLda + k4);
          dst[0] = make_float2((float)(w & 0xff), (float)((w >> 8) & 0xff));
          dst[1] = make_float2((float)((w >> 16) & 0xff), (float)(w >> 24));
        }
      }
    };
    auto noise_draw = [&]() {
      // this panel's rows of the Philox streams (the values gmvae_noise_fill produces for the same seed, step and
      // global row), drawn while the loads are in flight and parked in registers until the staging area is dead
      if (tid < qe + qu) {
        const bool is_u = tid >= qe;
        const int li = is_u ? tid - qe : tid;
        const int qpr = max(is_u ? qur : qer, 1);
        const int row = li / qpr, quad = li - row * qpr;
        noise_vals(a.row0 + (unsigned long long)(r0 + row), (unsigned)quad, is_u, a.seed, step, nz);
      }
    };
    if constexpr (HT == 64 && DT == 784 && MODEL == 2) {
      // Specialised sizes: 49 bursts of 4 rows per tensor, staged as 2 sub-chunks (24 and 25 bursts per tensor);
      // every wave issues exactly 6 + 7 bursts (straight-line code, so the vmcnt below is a constant) and the MFMAs
      // of the first half run while the second is still landing.  (4 sub-chunks spilled registers and lost 3 us.)
      const int row_ = lane >> 4, piece = lane & 15;
      const int src = row_ * 64 + ((piece ^ ((row_ & 1) << 2)) << 2);
      auto burst = [&](const int first, const int b, const int per) {      // b-th burst of a sub-chunk, `per` per tensor
        const int t = b / per, idx = first + b % per;
        __builtin_amdgcn_global_load_lds((t ? a.w0b : a.w0a) + (long long)k0 * H + (idx << 8) + src, Wst + t * KQ * H + (idx << 8), 16,
                                         0, 0);
      };
#pragma unroll
      for (int j = 0; j < 6; ++j) burst(0, wave + 8 * j, 24);
#pragma unroll
      for (int it = 0; it < 2; ++it) {             // always two loads per thread (clamped): uniform vmcnt accounting
        const int i = min(tid + it * kMT, kPanel * kq4 - 1);
        const int row = i / kq4, k4 = (i - row * kq4) * 4;
        const unsigned wv = *reinterpret_cast<const unsigned*>(a.x + (long long)min(r0 + row, B - 1) * D + k0 + k4);
        xw[it] = row < nrow ? wv : 0u;
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) burst(24, min(wave + 8 * j, 49), 25);    // 50 bursts: waves 2..7 repeat the last one
      noise_draw();
      // (no diagnostic stamp in here: a stamp is a global store, younger than the bursts, and would be counted)
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                      // the first half and the x bytes are in
      x_store();
      __syncthreads();
      const int tl = wave & 3;
      const int swz = ((tl * 16 + ln) ^ ((lk & 1) << 4)) - (tl * 16 + ln);  // this lane's k rows are all odd or all even
      const float* const Wt = (wave < 4 ? Wst : Wst + KQ * H) + swz;
      acc = tile_ksteps<1, kFlLda>(A_x, Wt, H, 1, tl, 0, 24, 24, lane, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      GMVAE_FL(4);
      acc = tile_ksteps<1, kFlLda>(A_x, Wt, H, 1, tl, 24, 49, 49, lane, acc);
    } else {
    const int rot = ((pnl >> 3) * 6) % max(1, (kn * H) >> 8);
    dma_stage_w(Wst, a.w0a + (long long)k0 * H, kn * H, rot, wave, lane);
    if (gm) dma_stage_w(Wst + KQ * H, a.w0b + (long long)k0 * H, kn * H, rot, wave, lane);
    GMVAE_FL(0);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * kMT;
      const int row = i / kq4, k4 = (i - row * kq4) * 4;
      xw[it] = 0;
      if (i < kPanel * kq4 && row < nrow && k4 < kn)
        xw[it] = *reinterpret_cast<const unsigned*>(a.x + (long long)(r0 + row) * D + k0 + k4);
    }
    GMVAE_FL(1);
    noise_draw();
    GMVAE_FL(2);
    if (kn < KQ) {                                 // zero the staging rows past D (their x columns are zero: no NaN * 0)
      for (int i = kn * H + tid; i < KQ * H; i += kMT) { Wst[i] = 0.f; if (gm) Wst[KQ * H + i] = 0.f; }
    }
    x_store();
    GMVAE_FS(0);
    GMVAE_FL(3);
    dma_wait();
    __syncthreads();
    GMVAE_FL(4);
    if (wave < ntile) {
      const int tl = wave < tpw ? wave : wave - tpw;
      const int swz = ((tl * 16 + ln) ^ ((lk & 1) << 4)) - (tl * 16 + ln);      // this lane's k rows are all odd or all even
      acc = tile_ksteps<1, kFlLda>(A_x, (wave < tpw ? Wst : Wst + KQ * H) + swz, H, 1, tl, 0, kq4, kq4, lane, acc);
    }
    }
    GMVAE_FL(5);
    const int ngr = kPanel * H2f;
    if (wave < ntile) {
      unsigned long long* xo = a.xfl + ((long long)pnl * 4 + q) * ngr;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __hip_atomic_store(xo + (lk * 4 + r) * H2f + wave * 16 + ln, ((unsigned long long)epoch_fl << 32) | __float_as_uint(acc[r]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                               // the staging area is dead: the weight image may land on it
    GMVAE_FS(1);
    GMVAE_FL(6);
    dma_copy_m(sm, a.img, f.img_early, wave, lane);
    if (tid < qe) {
      if ((L & 3) == 0) {
        *reinterpret_cast<float4*>(P_eps + tid * 4) = make_float4(nz[0], nz[1], nz[2], nz[3]);
      } else {                                       // ragged latent size (the reference's VAE example: L = 2)
        const int row = tid / qer, l0 = (tid - row * qer) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (l0 + j < L) P_eps[row * L + l0 + j] = nz[j];
      }
    } else if (tid < qe + qu) {
      const int li = tid - qe, row = li / max(qur, 1), k0u = (li - row * qur) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0u + j < K) P_u[row * K + k0u + j] = nz[j];
    }
    if (bid == gridDim.x - 1 && tid == 0) a.step_dev[1] = step;       // the copy finalize_adam reads
    // the quarters' partials (this workgroup's own included), two quarters per sweep: 8 granules of a lane are
    // requested together and re-read until every tag carries this step's epoch; summed in quarter order, so every
    // workgroup of the panel gets the same bits
    float flt[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < ntile) {
      const unsigned long long* xp = a.xfl + (long long)pnl * 4 * ngr + (lk * 4) * H2f + wave * 16 + ln;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        unsigned long long gv[2][4];
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              gv[pq][r] = __hip_atomic_load(xp + (long long)(2 * half + pq) * ngr + r * H2f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && (unsigned)(gv[pq][r] >> 32) == epoch_fl;
            }
          }
          if (__all(ok)) break;
          if (++spins > spin_limit) {
            if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
            for (int pq = 0; pq < 2; ++pq)
#pragma unroll
              for (int r = 0; r < 4; ++r) gv[pq][r] = 0x7fc00000ull;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          flt[r] += __uint_as_float((unsigned)gv[0][r]);
          flt[r] += __uint_as_float((unsigned)gv[1][r]);
        }
      }
    }
    GMVAE_FL(7);
    dma_wait();                                    // the biases (early image part) have landed
    __syncthreads();
    dma_copy_m(sm + f.img_early, a.img + f.img_early, f.img - f.img_early, wave, lane);   // q head / decoder hidden weights
    if (wave < ntile) {                            // bias + ReLU straight from the accumulator layout
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lk * 4 + r, col = wave * 16 + ln;
        if (col < H) {
          const float h = fmaxf(flt[r] + b_y0[col], 0.f);
          A_hy[col * kLDA + row] = h;
          P_hy[row * H + col] = h;
          if (lead && row < nrow) a.hy1[(long long)(r0 + row) * H + col] = h;
        } else {
          P_gx[row * H + (col - H)] = flt[r];
        }
      }
    }
  } else {
  dma_copy_m(sm, a.img, f.img_early, wave, lane);
  dma_copy_m(P_eps, a.eps + (long long)r0 * L, (nrow * L) & ~3, wave, lane);
  for (int e = ((nrow * L) & ~3) + tid; e < nrow * L; e += kMT) P_eps[e] = a.eps[(long long)r0 * L + e];   // ragged tail
  if (gm) dma_copy_m(P_u, a.u + (long long)r0 * K, (nrow * K) & ~3, wave, lane);
    if (gm)
      for (int e = ((nrow * K) & ~3) + tid; e < nrow * K; e += kMT) P_u[e] = a.u[(long long)r0 * K + e];
    GMVAE_FS(0);
    dma_wait();
    __syncthreads();
    GMVAE_FS(1);
    dma_copy_m(sm + f.img_early, a.img + f.img_early, f.img - f.img_early, wave, lane);   // q head / decoder hidden weights
    v = so[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const float w = j < a.NS ? 1.f : 0.f;
      v.x += w * so[j].x; v.y += w * so[j].y; v.z += w * so[j].z; v.w += w * so[j].w;
    }
    if (a.NS > 4) {
      const float4 t = slab_sum4(sp + 4 * sstride, sstride, a.NS - 4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  if (!FLT) {
    const int row = frow, col = fcol;
    if (tid < nitem) {
      float vv[4] = {v.x, v.y, v.z, v.w};
      if (col < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vv[j] = fmaxf(vv[j] + b_y0[col + j], 0.f);
          A_hy[(col + j) * kLDA + row] = vv[j];
        }
        *reinterpret_cast<float4*>(P_hy + row * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (lead && row < nrow) *reinterpret_cast<float4*>(a.hy1 + (long long)(r0 + row) * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      } else {
        *reinterpret_cast<float4*>(P_gx + row * H + (col - H)) = v;
      }
    }
  }
  {                               // (closes at the end of the kernel) laundered thread index, see phase B
  int tidf_ = threadIdx.x;
  asm volatile("" : "+v"(tidf_));
  const int tid = tidf_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 15, lk = lane >> 4;
  if (gmp) {                      // mixture constants: 1/s, log-softmax weights, per-component constant
    // one WAVE per component: its lanes walk the latent dimensions, so 1/s and -sum log s of a component are one pass and
    // one wave reduction (round 4: ten threads each walked their component's L dimensions serially -- 64 dependent LDS
    // reads and logs, ~9 k cycles = 4 us at the head of every VAE_GMP launch, tools/stamps.py)
    for (int k = wave; k < K; k += kMW) {
      float ls = 0.f;
      for (int l = lane; l < L; l += 64) {
        const float iv = 1.f / fsoftplus(M_raw[k * ldM + l]);
        M_inv[k * ldM + l] = iv;
        ls += flog(iv);                            // = -log s
      }
      ls = wave_sum(ls);
      if (lane == 0) M_c[k] = ls;                  // (finished below, once the mixture weights' normaliser is known)
    }
  }
  __syncthreads();
  if (gmp && tid < 64) {
    float mx = -INFINITY;
    for (int k = tid; k < K; k += 64) mx = fmaxf(mx, M_mix[k]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int k = tid; k < K; k += 64) se += fexp(M_mix[k] - mx);
    se = wave_sum(se);
    const float lse = mx + flog(se);
    for (int k = tid; k < K; k += 64) {
      const float lw = M_mix[k] - lse;
      M_w[k] = fexp(lw);
      M_c[k] = lw + M_c[k] - 0.5f * kLog2Pi * (float)L;
    }
  }
  GMVAE_STAMP(1);
  if (gm) {
  // logits
  for (int t = 0; t < KP / 16; ++t) {
    const int steps = H / 4, per = (steps + kMW - 1) / kMW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_ksteps(A_hy, W_y1, f.ldY1, 1, t, min(steps, wave * per), min(steps, wave * per + per), steps, lane, acc);
    ksplit_finish(acc, t, red, wave, lane, [&](int row, int col, float v) { P_lg[row * KP + col] = v + b_y1[col]; });
    __syncthreads();
  }
  GMVAE_FS(2);
  // Gumbel-softmax + entropy (16 lanes per row: the first 4 waves)
  if (tid < 256) {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    float lgv[4], av[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      lgv[j] = -INFINITY; av[j] = -INFINITY;
      if (k < K) {
        lgv[j] = P_lg[row * KP + k];
        const float uu = ok ? P_u[row * K + k] : 0.5f;
        av[j] = (lgv[j] - flog(-flog(uu))) * a.invT;
        mx = fmaxf(mx, av[j]);
      }
    }
    mx = row16_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (sub + 16 * j < K) se += fexp(av[j] - mx);
    se = row16_sum(se);
    const float lse = mx + flog(se);
    float lpv[4];
    cat_log_softmax<Row16, 4>(lgv, lpv);           // log pi (gemm.hpp: accurate for a saturated q(y|x))
    float ne = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      if (k < K2) {
        float yv = 0.f;
        if (k < K) {
          yv = fexp(av[j] - lse);
          const float lp = lpv[j];
          ne += fexp(lp) * lp;
          P_y[row * K + k] = yv;
          if (ok && lead) a.y[(long long)(r0 + row) * K2 + k] = yv;        // rows of pad4(K) floats
        }
        A_y[k * kLDA + row] = yv;
      }
    }
    ne = row16_sum(ne);
    if (sub == 0) { nllp[3 * kPanel + row] = ne; if (ok && lead) a.nent[r0 + row] = ne; }     // nllp[48..63]: nent per row
  }
  __syncthreads();
  GMVAE_STAMP(2);
  // hg1 and prior head
  panel_gemm_s(A_y, W_g0y, f.ldG0, 1, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    const float h = fmaxf(v + P_gx[row * H + col] + b_g0[col], 0.f);
    A_hg[col * kLDA + row] = h;
    P_hg[row * H + col] = h;
    if (lead && row < nrow) a.hg1[(long long)(r0 + row) * H + col] = h;
  });
  panel_gemm_s(A_y, W_p, f.ldP, 1, K2, (L2 + 15) / 16, wave, lane,
               [&](int row, int col, float v) { if (col < L2) P_pp[row * L2 + col] = v + b_p[col]; });
  }  // gm
  dma_wait();                      // the late part of the weight image
  __syncthreads();
  GMVAE_FS(3);
  // q head
  panel_gemm_s(A_hq, W_g1, f.ldG1, 1, H, (L2 + 15) / 16, wave, lane,
               [&](int row, int col, float v) { if (col < L2) P_qp[row * L2 + col] = v + b_g1[col]; });
  __syncthreads();
  GMVAE_STAMP(3);
  // z, log q, log p: 32 lanes per row.  One exp, one rcp and one log give softplus AND its derivative (the
  // sigmoid); the backward chain's inputs replace the head outputs in place -- P_qp[row] = [sigmoid(raw_q) | sigma_q],
  // P_pp[row] = [t = (z - mu_p) / sigma_p | sigma_p], P_z[row] = sigmoid(raw_p) (GMVAE; the others keep z) -- so that
  // phase B needs no transcendental at all.
  {
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
    float aq = 0.f, ap = 0.f;
    if (sub >= L && sub < GMVAE_P4(L)) A_z[sub * kLDA + row] = 0.f;      // k rows that pad L to a multiple of 4
    for (int l = sub; l < L; l += 32) {
      const float mu = P_qp[row * L2 + l];
      const float vq = P_qp[row * L2 + L + l] + a.c;
      float sgq;
      const float sg = fmaxf(softplus_sig(vq, sgq), a.smin);
      const float ee = ok ? P_eps[row * L + l] : 0.f;
      const float zz = mu + sg * ee;
      A_z[l * kLDA + row] = zz;
      if (ok && lead) a.z[(long long)(r0 + row) * L + l] = zz;
      P_qp[row * L2 + l] = sgq;
      P_qp[row * L2 + L + l] = sg;
      aq += -0.5f * ee * ee - 0.5f * kLog2Pi - flog(sg);     // (z - mu) / sigma IS eps
      if (gm) {                                             // p(z|y): gmvae.py:258
        const float vp = P_pp[row * L2 + L + l] + a.c;
        float sgp;
        const float sp = fmaxf(softplus_sig(vp, sgp), a.smin);
        const float t = (zz - P_pp[row * L2 + l]) * __builtin_amdgcn_rcpf(sp);
        ap += -0.5f * t * t - 0.5f * kLog2Pi - flog(sp);
        P_pp[row * L2 + l] = t;
        P_pp[row * L2 + L + l] = sp;
        P_z[row * L + l] = sgp;
      } else {                                              // N(0, I): vae.py:247-250 (VAE_GMP: mixture stage below)
        P_z[row * L + l] = zz;
        ap += -0.5f * zz * zz - 0.5f * kLog2Pi;
      }
    }
    aq = row32_sum(aq); ap = row32_sum(ap);
    if (sub == 0) {
      nllp[row] = aq;
      if (!gmp) nllp[kPanel + row] = ap;
      if (ok && lead) { a.logq[r0 + row] = aq; if (!gmp) a.logp[r0 + row] = ap; }
    }
  }
  __syncthreads();
  GMVAE_FS(4);
  if (gmp) {
    // MixtureSameFamily.log_prob (vae.py:240-244,181): 32 lanes per row, lane = component (+32), each lane
    // walks l over the LDS-resident (loc, 1/s) image; K-way logsumexp by wavefront shuffles; responsibilities kept.
    // (round 4: the 32 lanes of a row split the LATENT dimensions and every component's squared distance is a 32-lane
    //  reduction -- with lane = component, K = 10 left 22 of 32 lanes idle behind L serial steps of three LDS reads each)
    const int row = tid >> 5, sub = tid & 31;
    float comp[2] = {-INFINITY, -INFINITY};
    {
      float zl[4];                                 // this lane's latent dimensions l = sub + 32 i (L <= 128)
#pragma unroll
      for (int i = 0; i < 4; ++i) zl[i] = sub + 32 * i < L ? P_z[row * L + sub + 32 * i] : 0.f;
      for (int k = 0; k < K; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int l = sub + 32 * i;
          if (l < L) {
            const float t = (zl[i] - M_loc[k * ldM + l]) * M_inv[k * ldM + l];
            acc += t * t;
          }
        }
        acc = row32_sum(acc);
        const float c = M_c[k] - 0.5f * acc;
        if (k < 32) { if (sub == k) comp[0] = c; }           // lane k (mod 32) keeps component k, as the reductions below expect
        else if (sub == k - 32) comp[1] = c;
      }
    }
    float mx = fmaxf(comp[0], comp[1]);
    mx = row32_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (sub + 32 * j < K) se += fexp(comp[j] - mx);
    se = row32_sum(se);
    const float lse = mx + flog(se);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (sub + 32 * j < K) P_r[row * KP + sub + 32 * j] = fexp(comp[j] - lse);
    if (sub == 0) {
      nllp[kPanel + row] = lse;
      if (lead && row < nrow) a.logp[r0 + row] = lse;
    }
    __syncthreads();
  }
  // decoder hidden
  panel_gemm_s(A_z, W_d0, f.ldD0, 1, GMVAE_P4(L), H / 16, wave, lane, [&](int row, int col, float v) {
    const float h = fmaxf(v + b_d0[col], 0.f);
    A_hd[col * kLDA + row] = h;
    P_hd[row * H + col] = h;
    if (lead && row < nrow) a.hd1[(long long)(r0 + row) * H + col] = h;
  });
  __syncthreads();                 // the small-weight image is dead from here until phase B
  GMVAE_STAMP(4);

  // ======================================================================= D: decoder output, streamed
  {                                              // (closes at the end of the kernel)
  int tidd_ = threadIdx.x;                       // laundered thread index (see phase B)
  asm volatile("" : "+v"(tidd_));
  const int tid = tidd_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 15, lk = lane >> 4;
  const int nch = f.nch, ldc = f.ldc;
  auto issue_chunk = [&](int i) {             // i = local index; chunk c = q + i * Q
    const int c = q + i * Q;
    dma_copy_m(sm + f.ring + (i & 1) * f.chunk, a.dimg + (long long)c * f.chunk, f.chunk, wave, lane);
    // x chunk: [16 rows][128 bytes]; lane -> (row, 16-byte piece); columns beyond D are masked off
    if (tid < kPanel * (kCW / 16)) {
      const int row = tid >> 3, piece = tid & 7;
      const int col = c * kCW + piece * 16;
      if (col < D && row < nrow)
        __builtin_amdgcn_global_load_lds(a.x + (long long)(r0 + row) * D + col,
                                         reinterpret_cast<float*>(xring + (i & 1) * kPanel * kCW + (tid >> 6) * 1024), 16, 0, 0);
    }
  };
  f32x4 dacc = {0.f, 0.f, 0.f, 0.f};           // partial 16x16 tile of dhd1: tile = wave & 3, column half = wave >> 2
  const int dtile = wave & 3, dhalf = wave >> 2;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};          // Bernoulli terms of rows lk*4 + r, this lane's columns
  // this workgroup's chunks: q, q + Q, q + 2Q, ...  (ring buffer parity follows the LOCAL index)
  const int nloc = (nch - q + Q - 1) / Q;
  if (nloc > 0) issue_chunk(0);
  unsigned long long tseg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = a.dbg ? __builtin_amdgcn_s_memtime() : 0;   // diagnostic only
#define GMVAE_SEG(i) if (a.dbg) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tseg[i] += tn_ - tprev; tprev = tn_; }
  // the backward chain's weight image was overwritten by the ring: its first part returns during the last chunk
  const int pre_img = (lead && nloc > 0 && ((nloc - 1) & 1)) ? min(f.img, f.chunk) : 0;
  for (int i = 0; i < nloc; ++i) {
    dma_wait();
    __syncthreads();                            // chunk i landed; A_g and buffer (i+1)&1 are free
    GMVAE_SEG(0);
    if (i + 1 < nloc) issue_chunk(i + 1);
    else if (lead && (i & 1)) dma_copy_m(sm, a.img, pre_img, wave, lane);   // last chunk sits in buffer 1: buffer 0 is free
    GMVAE_SEG(4);
    const float* Wc = sm + f.ring + (i & 1) * f.chunk;
    const unsigned char* xc = xring + (i & 1) * kPanel * kCW;
    const int c0 = (q + i * Q) * kCW;
    // lambda tiles (2 per wave) + Bernoulli epilogue.  One wave per SIMD: nothing hides latency, so the
    // operand-independent LDS reads (bias, x bytes) are issued before the MFMA chain, the element math is
    // branch-free with ONE exp, ONE rcp and ONE log per element, and g leaves through the A_g image as
    // 16-byte row-major stores after the barrier (cdna_hip_programming.md T21).
    for (int t = wave; t < kCW / 16; t += kMW) {
      const int cl = t * 16 + ln, col = c0 + cl;
      const float bias = Wc[H * ldc + cl] + a.gen_bias;
      float xv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[r] = (float)xc[(lk * 4 + r) * kCW + cl];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_ksteps(A_hd, Wc, ldc, 1, t, 0, H / 4, H / 4, lane, acc);
      if (a.dbg) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); }
      GMVAE_SEG(5);
      const bool cok = col < D;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lk * 4 + r;
        const bool ok = cok && row < nrow;
        const float lam = acc[r] + bias;
        const float e = fexp(-fabsf(lam));
        const float rcp = __builtin_amdgcn_rcpf(1.f + e);          // 1/(1+e): sigmoid(|lam|)
        const float sp = fmaxf(lam, 0.f) - flog(rcp);             // softplus = max(lam,0) + log(1+e)
        const float sg = lam >= 0.f ? rcp : e * rcp;
        const float x_ = ok ? xv[r] : 0.f;
        rs[r] += ok ? x_ * lam - sp : 0.f;
        A_g[cl * kLDA + row] = ok ? sg - x_ : 0.f;
      }
    }
    GMVAE_SEG(1);
    __syncthreads();
    GMVAE_SEG(2);
    // g chunk -> HBM as 16-byte row-major stores (for the dWd1 GEMM): thread -> (row, 4 columns).  (A conflict-free
    // read mapping with four 4-byte stores per thread was measured SLOWER: every later vmcnt(0) drains 4x the stores.)
    {
      const int row = tid >> 5, c4 = (tid & 31) << 2;     // 512 float4 per chunk: one per thread
      if (row < nrow && c0 + c4 < D) {
        const float4 gv = make_float4(A_g[(c4 + 0) * kLDA + row], A_g[(c4 + 1) * kLDA + row], A_g[(c4 + 2) * kLDA + row],
                                      A_g[(c4 + 3) * kLDA + row]);
        st4o(a.g + (long long)(r0 + row) * D + c0 + c4, gv);             // write-through: read by the NEXT launch only
      }
    }
    // dhd1 += g_chunk * Wd1_chunk^T   (B(k = column, n = h) = Wc[h*ldc + column]); each tile's K split over 2 waves
    if (dtile * 16 < H)
      dacc = tile_ksteps(A_g, Wc, 1, ldc, dtile, dhalf * (kCW / 8), (dhalf + 1) * (kCW / 8), kCW / 4, lane, dacc);
    GMVAE_SEG(3);
  }
  if (a.dbg && a.fine < 2 && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) a.dbg[(size_t)blockIdx.x * 16 + 8 + i] = tseg[i];
  GMVAE_STAMP(5);
  // per-row log p(x|z): reduce over the 16 column lanes, then over the 4 waves
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rs[r] = row16_sum(rs[r]);
  }
  __syncthreads();                               // every wave is done with the ring and A_g
  const unsigned epoch = Q > 1 ? *a.epoch_word : 0u;
  const int ngr = kPanel * H + kPanel;           // granules one producer publishes: dhd1 tile partials + row sums
  if (!lead) {
    // ------------------------------------------------------------- producer: publish the partials and leave
    if (ln == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) nllp2[wave * kPanel + lk * 4 + r] = rs[r];
    }
    if (dhalf == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[dtile * 256 + (lk * 4 + r) * 16 + ln] = dacc[r];
    }
    __syncthreads();
    unsigned long long* xo = a.xchg + ((long long)pnl * (Q - 1) + (q - 1)) * ngr;
    if (dhalf == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lk * 4 + r, col = dtile * 16 + ln;
        if (col < H) {
          const float dv = dacc[r] + red[dtile * 256 + (lk * 4 + r) * 16 + ln];
          __hip_atomic_store(xo + row * H + col, ((unsigned long long)epoch << 32) | __float_as_uint(dv), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (tid >= 256 && tid < 256 + kPanel) {
      const int row = tid - 256;
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < kMW; ++w) s_ += nllp2[w * kPanel + row];
      __hip_atomic_store(xo + kPanel * H + row, ((unsigned long long)epoch << 32) | __float_as_uint(s_), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    GMVAE_SPAN_END();
    return;
  }
  // ======================================================================= B: backward chain
  // The thread index is "laundered" at the phase boundary: index arithmetic of this phase can then not be merged
  // with (and hoisted to) the kernel's start, where it used to stay live across every other phase and spill.
  {
  int tidb_ = threadIdx.x;
  asm volatile("" : "+v"(tidb_));
  const int tid = tidb_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 15, lk = lane >> 4;
  const int dtile = wave & 3, dhalf = wave >> 2;
  dma_copy_m(sm + pre_img, a.img + pre_img, f.img - pre_img, wave, lane);      // the rest of the small-weight image
  if (ln == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) nllp2[wave * kPanel + lk * 4 + r] = rs[r];
  }
  if (dhalf == 1) {                              // upper column halves hand their partial tiles over
#pragma unroll
    for (int r = 0; r < 4; ++r) red[dtile * 256 + (lk * 4 + r) * 16 + ln] = dacc[r];
  }
  // the other quarters' partials: sweep the granules until every tag carries this step's epoch (bounded)
  float xd[4] = {0.f, 0.f, 0.f, 0.f}, xn = 0.f;
  if (Q > 1) {
    const unsigned long long* xi = a.xchg + (long long)pnl * (Q - 1) * ngr;
    const bool wd = dhalf == 0 && dtile * 16 + ln < H;      // this lane waits for dhd1 granules
    const bool wn = tid >= 256 && tid < 256 + kPanel;       // ... for a row-sum granule
    for (int pq = 0; pq < Q - 1; ++pq) {
      const unsigned long long* xp = xi + (long long)pq * ngr;
      unsigned long long v[5] = {0, 0, 0, 0, 0};
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
        if (wd) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = __hip_atomic_load(xp + (lk * 4 + r) * H + dtile * 16 + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(v[r] >> 32) == epoch;
          }
        }
        if (wn) {
          v[4] = __hip_atomic_load(xp + kPanel * H + (tid - 256), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (unsigned)(v[4] >> 32) == epoch;
        }
        if (__all(ok)) break;
        if (++spins > spin_limit) {                          // the producer never ran; flag and go on
          if (lane == 0) atomicExch(a.err_word, 1u);
          v[0] = v[1] = v[2] = v[3] = v[4] = 0x7fc00000ull;  // NaN: the step's loss and gradients say so loudly
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) xd[r] += __uint_as_float((unsigned)v[r]);
      xn += __uint_as_float((unsigned)v[4]);
    }
  }
  dma_wait();
  __syncthreads();
  if (dhalf == 0) {                              // masked top gradient -> A_dhd (+ saved for dWd0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = lk * 4 + r, col = dtile * 16 + ln;
      if (col < H) {
        const float dv = dacc[r] + red[dtile * 256 + (lk * 4 + r) * 16 + ln] + xd[r];
        const float d = (row < nrow && P_hd[row * H + col] > 0.f) ? dv : 0.f;
        A_dhd[col * kLDA + row] = d;              // overlays A_hy.. (dead)
        if (row < nrow) a.dhd1[(long long)(r0 + row) * H + col] = d;
      }
    }
  }
  if (tid >= 256 && tid < 256 + kPanel) {
    const int row = tid - 256;
    if (row < nrow) {
      float s_ = xn;
#pragma unroll
      for (int w = 0; w < kMW; ++w) s_ += nllp2[w * kPanel + row];
      a.logpx[r0 + row] = s_;
      a.logw[r0 + row] = s_ + nllp[kPanel + row] - nllp[row] - (gm ? nllp[3 * kPanel + row] : 0.f);
    }
  }
  __syncthreads();
  GMVAE_STAMP(6);
  // dz_dec = dhd1 * Wd0^T
  panel_gemm_s(A_dhd, W_d0, 1, f.ldD0, H, LP / 16, wave, lane, [&](int row, int col, float v) { P_dz[row * LP + col] = v; });
  __syncthreads();
  GMVAE_FS(5);
  {                                             // 32 lanes per row; inputs prepared by the forward z stage
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
    for (int l = sub; l < L; l += 32) {
      float dmu = 0.f, draw = 0.f, dmup = 0.f, drawp = 0.f;
      if (ok) {
        const float sg = P_qp[row * L2 + L + l];
        float pterm;                                   // d(-log p)/dz
        if (gm) {
          const float sp = P_pp[row * L2 + L + l], t = P_pp[row * L2 + l];
          const float isp = __builtin_amdgcn_rcpf(sp);
          pterm = t * isp;
          dmup = -pterm;
          drawp = (sp > a.smin) ? (1.f - t * t) * isp * P_z[row * L + l] : 0.f;
          float* dp = a.dpp + (long long)(r0 + row) * L2;
          dp[l] = dmup; dp[L + l] = drawp;
        } else if (gmp) {                              // sum_k r_k (z - loc_k) / s_k^2
          const float zz = P_z[row * L + l];
          pterm = 0.f;
          for (int k = 0; k < K; ++k) {
            const float iv = M_inv[k * ldM + l];
            pterm += P_r[row * KP + k] * (zz - M_loc[k * ldM + l]) * iv * iv;
          }
        } else {
          pterm = P_z[row * L + l];
        }
        dmu = P_dz[row * LP + l] + pterm;
        const float dsg = dmu * P_eps[row * L + l] - __builtin_amdgcn_rcpf(sg);
        draw = (sg > a.smin) ? dsg * P_qp[row * L2 + l] : 0.f;
        float* dq = a.dqp + (long long)(r0 + row) * L2;
        dq[l] = dmu; dq[L + l] = draw;
      }
      A_dqp[l * kLDA + row] = dmu;
      A_dqp[(L + l) * kLDA + row] = draw;
      if (gm) {
        A_dpp[l * kLDA + row] = dmup;
        A_dpp[(L + l) * kLDA + row] = drawp;
      }
    }
  }
  __syncthreads();
  GMVAE_FS(6);
  // dh = (dqp * W^T) * [h > 0]: encoder_gmm's hidden layer (GMVAE) or the encoder's (VAE)
  panel_gemm_s(A_dqp, W_g1, 1, f.ldG1, L2, H / 16, wave, lane, [&](int row, int col, float v) {
    const float d = (row < nrow && P_hq[row * H + col] > 0.f) ? v : 0.f;
    A_dhg[col * kLDA + row] = d;
    if (row < nrow) a.dhg1[(long long)(r0 + row) * H + col] = d;
  });
  if (gmp) {
    // gradients of the learned mixture prior over this panel's rows (A12): one partial per workgroup
    const int KL = K * L, KLp = (KL + 3) & ~3;
    float* out = a.gmp_part + (long long)pnl * (2 * KLp + ((K + 3) & ~3));
    for (int i = tid; i < KL; i += kMT) {
      const int k = i / L, l = i - k * L;
      const float iv = M_inv[k * ldM + l], lc = M_loc[k * ldM + l];
      float ga = 0.f, gb = 0.f;
      float wr[kPanel], zz[kPanel];                // all 32 LDS reads in flight (a runtime row bound made them 16 dependent rounds)
#pragma unroll
      for (int row = 0; row < kPanel; ++row) { wr[row] = P_r[row * KP + k]; zz[row] = P_z[row * L + l]; }
#pragma unroll
      for (int row = 0; row < kPanel; ++row) {
        const float w_ = row < nrow ? wr[row] : 0.f;
        const float t = row < nrow ? (zz[row] - lc) * iv : 0.f;
        ga -= w_ * t * iv;
        gb += w_ * (1.f - t * t) * iv;
      }
      out[i] = ga;
      out[KLp + i] = gb * sigmoidf_(M_raw[k * ldM + l]);
    }
    for (int k = tid; k < K; k += kMT) {
      float ga = 0.f;
#pragma unroll
      for (int row = 0; row < kPanel; ++row) ga -= row < nrow ? P_r[row * KP + k] - M_w[k] : 0.f;
      out[2 * KLp + k] = ga;
    }
  }
  if (!gm) { GMVAE_STAMP(7); GMVAE_SPAN_END(); return; }
  __syncthreads();
  GMVAE_FS(7);
  // dy = dhg1 * Wg0[D:,:]^T + dpp * Wp^T : K = H then K = 2L, split over the 4 waves
  for (int t = 0; t < KP / 16; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
      const int steps = H / 4, per = (steps + kMW - 1) / kMW;
      acc = tile_ksteps(A_dhg, W_g0y, 1, f.ldG0, t, min(steps, wave * per), min(steps, wave * per + per), steps, lane, acc);
    }
    {
      const int steps = L2 / 4, per = (steps + kMW - 1) / kMW;
      acc = tile_ksteps(A_dpp, W_p, 1, f.ldP, t, min(steps, wave * per), min(steps, wave * per + per), steps, lane, acc);
    }
    ksplit_finish(acc, t, red, wave, lane, [&](int row, int col, float v) { P_dy[row * KP + col] = v; });
    __syncthreads();
  }
  // softmax backward + entropy gradient
  if (tid < 256) {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    float lgv[4], yv[4], dyv[4], lpv[4], dav[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      lgv[j] = -INFINITY; yv[j] = 0.f; dyv[j] = 0.f;
      if (k < K && ok) {
        lgv[j] = P_lg[row * KP + k];
        yv[j] = P_y[row * K + k];
        dyv[j] = P_dy[row * KP + k];
      }
    }
    cat_log_softmax<Row16, 4>(lgv, lpv);           // (gemm.hpp: the forms that survive a saturated softmax)
    cat_softmax_bwd<Row16, 4>(yv, dyv, dav);
    const float ne = nllp[3 * kPanel + row];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      if (k < K2) {
        float dl = 0.f;
        if (k < K && ok) {
          const float lp = lpv[j];
          dl = dav[j] * a.invT + fexp(lp) * (lp - ne);
          a.dlogits[(long long)(r0 + row) * K2 + k] = dl;
        }
        A_dl[k * kLDA + row] = dl;
      }
    }
  }
  __syncthreads();
  // dhy1 = (dlogits * Wy1^T) * [hy1 > 0]
  panel_gemm_s(A_dl, W_y1, 1, f.ldY1, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    if (row < nrow) a.dhy1[(long long)(r0 + row) * H + col] = P_hy[row * H + col] > 0.f ? v : 0.f;
  });
  GMVAE_STAMP(7);
  GMVAE_SPAN_END();
  }
  }
  }
}

}  // namespace gmvae
