// mega2_fwd_bwd: the steady-state per-row launch of the GMVAE training step at the reference's default sizes
// (run_gmvae.py: hidden 64, latent 64, K = 10; MNIST D = 784; batch <= 1024) -- the same work as
// mega_fwd_bwd<64, 64, 10, 784, 2, 1> (mega.hpp: first layer, forward chain, decoder layer with its data gradient,
// backward chain; scripts/gmvae.py:238-267 and its reverse pass), re-laid for the 16-row panel's latency chain:
//
//   * every matrix product runs TRANSPOSED on the matrix cores: the WEIGHT is the MFMA A operand (16 output features
//     per tile) and the panel's activations are the B operand (16 batch rows).  With the contraction index of four
//     consecutive v_mfma_f32_16x16x4_f32 assigned as k = 16 kt + 4 (lane / 16) + s, a lane's four steps are ONE
//     16-byte LDS read of the weight image ([k / 4][outputs][4], written by finalize_adam's scatter) and ONE 16-byte
//     read of a ROW-MAJOR activation panel -- and the accumulator it ends with, out[row = lane % 16][4 (lane / 16) + r],
//     is four consecutive columns of a row: one 16-byte store puts it where the next product and the row-wise stages
//     read it.  (mega_fwd_bwd kept [k][17] images: 32 four-byte LDS reads and 8 four-byte writes per tile and stage.)
//   * the decoder layer needs no LDS at all: each wave owns 16-column tiles of the 784 outputs; its weights (forward and
//     transposed operand images, 2 x 4 KB per tile) are fetched straight into registers long before they are used,
//     lambda = hd1 Wd1 + b comes out of the matrix core as [4 columns of one row] per lane, the Bernoulli term,
//     g = sigmoid(lambda) - x and -- the accumulator being exactly the B operand of the data gradient's contraction
//     over those columns -- dhd1 += g Wd1^T follow without a barrier or an LDS round trip.
//
// Grid and hand-offs are mega_fwd_bwd's: 4 workgroups per panel (the decoder's column tiles dealt round-robin; quarters
// 1..3 publish their dhd1 and row-sum partials and leave, quarter 0 runs the backward chain).  The FIRST LAYER (round 4) is
// split over the EIGHT workgroups of two neighbouring panels: workgroup (panel 2s + h, quarter q) multiplies the 32 rows of
// both panels by rows [196 q, 196 q + 196) of ONE of the two first-layer weight tensors (h = 0: encoder_y's, 1: encoder_gmm's
// x rows) -- 50 KB of weights per workgroup instead of 100 (its wait for them was 6.3 k of the stage's 17 k cycles) for the
// same 392 matrix instructions and the same exchange (2048 granules out, 16 per lane in).
#pragma once
#include "mega.hpp"

namespace gmvae {

struct M2 {
  static constexpr int H = 64, L = 64, K = 10, KP = 16, L2 = 128, D = 784, K2 = 12;
  static constexpr int NT = D / 16;               // 49 decoder column tiles
  static constexpr int DC = 16 * kM2ProdTiles;    // columns of one workgroup's part: 14 tiles (kernels.hpp m2_dec_part)
  // ---- forward operand image (floats): biases, then [contraction / 4][outputs][4] weights
  static constexpr int b_y0 = 0, b_y1 = 64, b_g0 = 80, b_p = 144, b_g1 = 272, b_d0 = 400;
  static constexpr int Wy1f = 512;                // [16][16][4]   logits = hy * Wy1      (contraction h, 16 >= K outputs)
  static constexpr int Wg0yf = Wy1f + 1024;       // [4][64][4]    hg    += y * Wg0[D:]   (contraction k padded to 16)
  static constexpr int Wpf = Wg0yf + 1024;        // [4][128][4]   pp     = y * Wp
  static constexpr int imgF_early = Wpf + 2048;   // 4608: what the first stages need
  static constexpr int Wg1f = imgF_early;         // [16][128][4]  qp     = hg * Wg1
  static constexpr int Wd0f = Wg1f + 8192;        // [16][64][4]   hd     = z * Wd0
  static constexpr int imgF = Wd0f + 4096;        // 16896
  // ---- backward operand image
  static constexpr int Wd0b = 0;                  // [16][64][4]   dz   = dhd * Wd0^T     (contraction h, outputs l)
  static constexpr int Wg1b = Wd0b + 4096;        // [32][64][4]   dhg  = dqp * Wg1^T     (contraction 2L, outputs h)
  static constexpr int Wg0yb = Wg1b + 8192;       // [16][16][4]   dy   = dhg * Wg0[D:]^T (contraction h, outputs k)
  static constexpr int Wpb = Wg0yb + 1024;        // [32][16][4]   dy  += dpp * Wp^T      (contraction 2L, outputs k)
  static constexpr int Wy1b = Wpb + 2048;         // [4][64][4]    dhy  = dl * Wy1^T      (contraction k, outputs h)
  static constexpr int imgB = Wy1b + 1024;        // 16384
  // ---- decoder operand images in global memory, per workgroup part q: forward [16][DC][4], transposed [DC/4][64][4], bias [DC]
  static constexpr int dF = 0, dFq = 16 * DC * 4;
  static constexpr int dB = 4 * dFq, dBq = (DC / 4) * 64 * 4;
  static constexpr int dbias = dB + 4 * dBq, dbq = DC;
  static constexpr int dimg = dbias + 4 * dbq;    // floats
  // ---- LDS map (floats).  Row-major panels of the 16 batch rows; strides = 4 (mod 64): a wave's 16-byte operand reads
  // (16 rows x 4 column groups) are then nearly conflict-free.
  static constexpr int ld64 = 68, ld128 = 132, ldk = 20;
  static constexpr int IMG = 0;
  static constexpr int P_h1 = IMG + imgF;         // [16][132]  relu(hy) | gx (raw)
  static constexpr int P_y = P_h1 + 16 * ld128;   // [16][20]
  static constexpr int P_hg = P_y + 16 * ldk;     // [16][68]
  static constexpr int P_pp = P_hg + 16 * ld64;   // [16][132]  prior head (mu | raw) -> (t | sigma_p) for the backward chain
  static constexpr int P_qp = P_pp + 16 * ld128;  // [16][132]  q head -> (sigmoid(raw_q) | sigma_q)
  static constexpr int P_z = P_qp + 16 * ld128;   // [16][68]
  static constexpr int P_sp = P_z + 16 * ld64;    // [16][64]   sigmoid(raw_p)
  static constexpr int P_hd = P_sp + 16 * 64;     // [16][68]
  static constexpr int P_eps = P_hd + 16 * ld64;  // [16][64]
  static constexpr int P_u = P_eps + 16 * 64;     // [16][16]
  static constexpr int P_lg = P_u + 256;          // [16][20]
  static constexpr int nll = P_lg + 16 * ldk;     // [4][16]: log q, log p, -, nent per row
  static constexpr int rsum = nll + 64;           // [8][16] Bernoulli row sums per wave
  static constexpr int red = rsum + 128;          // [8][256] K-split partial tiles
  static constexpr int dred = red + 2048;         // [8][1024] per-wave partial dhd1 tiles; later the backward panels
  static constexpr int P_dhd = dred + 8192;       // [16][68]
  static constexpr int total = P_dhd + 16 * ld64;
  // backward panels (inside the dred region, which is dead once the partial tiles are summed)
  static constexpr int P_dz = dred;               // [16][68]
  static constexpr int P_dqp = P_dz + 16 * ld64;  // [16][132]
  static constexpr int P_dpp = P_dqp + 16 * ld128;
  static constexpr int P_dhg = P_dpp + 16 * ld128;
  static constexpr int P_dl = P_dhg + 16 * ld64;  // [16][20]
  static_assert(P_dl + 16 * ldk <= P_dhd, "backward panels must fit the dred region");
  static_assert(total * 4 <= 160 * 1024, "LDS budget");
  // the in-launch first layer's staging ([196][64] weights + [32][226] x image) lies BEHIND the forward operand image (over
  // the panels, which nothing uses yet): the image's DMA is then issued at the kernel's start, beside the weight bursts,
  // instead of after the exchange's publish -- where its 66 KB sat in front of every wave's polls (vmcnt retires in order:
  // the exchange took 2.4 us, the time of that DMA, for a hand-off whose latency is ~1 us)
  static constexpr int fl_kq = 196, fl_W = imgF, fl_A = fl_W + fl_kq * 64;
  static_assert(fl_A + 32 * kFlLda <= total, "first-layer staging must fit");
  static constexpr int imgF_pieces = (imgF + 2047) / 2048;      // 1 KB LDS-DMA pieces per wave (the last ones clamped: uniform counts)
};

// one 16-output tile `nt` of  out[row][n] = sum_k in[row][k] W[k][n]:  W as a [K/4][NPAD][4] operand image, `in` a
// row-major panel; KT contraction tiles of 16 starting at kt0.  All operand reads go out before the first MFMA.
template <int KT, int NPAD>
__device__ __forceinline__ f32x4 m2_tile(const float* __restrict__ Wimg, const int nt, const float* __restrict__ Pin,
                                         const int ld, const int kt0, const int ln, const int lk, f32x4 acc) {
  float4 a[KT], b[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    a[t] = *reinterpret_cast<const float4*>(Wimg + ((((kt0 + t) * 4 + lk) * NPAD + nt * 16 + ln) << 2));
    b[t] = *reinterpret_cast<const float4*>(Pin + ln * ld + (kt0 + t) * 16 + 4 * lk);
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b[t].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b[t].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b[t].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b[t].w, acc, 0, 0, 0);
  }
  return acc;
}
// Hand-off granules: one naturally aligned 8-byte {epoch, value} word written by ONE write-through (sc1) store: the
// data is its own flag (mega.hpp; cdna_hip_programming.md G16 recipe R2).  Measured (tools/handoff_clock.py): a granule
// is readable on another CU 2.3-2.8 us after its store, whatever the reader does; a second, L2-resident copy written
// with an ordinary store (readable sooner by a same-XCD reader in isolation) bought nothing once both stores were issued,
// so there is one copy.  What does matter is the readers' side: all of a lane's granule loads in flight together
// (branch-free sweeps) and a wave's granules contiguous in memory.
__device__ __forceinline__ void granule_publish(unsigned long long* p, const unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Two granules {value, epoch} {value, epoch} of one lane as ONE 16-byte write-through store / sc1 load (round 4: the sweeps
// of the two hand-offs were bound by the CU's vector-memory ISSUE -- 128 and 96 eight-byte wave-loads per sweep at ~16
// cycles each, 1.3 us per sweep whatever had been published when; each 8-byte half is still its own flag, cdna_hip_programming
// G16 R2).  The loads are inline asm: g2_wait() makes their registers the outputs of the wait, so no use can move above it.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule2_publish(unsigned long long* p, const unsigned epoch, const float v0, const float v1) {
  const u32x4_t t = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ u32x4_t granule2_load(const unsigned long long* p) {
  u32x4_t r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}
// The same load served by the XCD's L2 (sc0: only the CU's own cache is bypassed).  A write-through store updates the writer's
// L2 on its way to memory, so a reader ON THE SAME XCD sees it an L2 round trip later instead of a trip through the fabric;
// a reader on another XCD would spin on its own L2's stale copy -- hence M2_L2_SWEEPS failed sweeps at most, then sc1.
#ifndef M2_L2_SWEEPS
#define M2_L2_SWEEPS 0
#endif
__device__ __forceinline__ u32x4_t granule2_load_l2(const unsigned long long* p) {
  u32x4_t r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void g2_wait(u32x4_t (&v)[N]) {
  static_assert(N == 6 || N == 8, "operand lists below");
  if constexpr (N == 8)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
}
__device__ __forceinline__ float4 f4(const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
// (st4o -- a launch output as a 16-byte write-through store -- lives in mega.hpp)
// a 4-byte launch output, write-through: mega3_step's workers on other XCDs read it in the SAME launch (behind the lead's flag)
__device__ __forceinline__ void st1o(float* p, const float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The launch's body.  FUSE = 0: the whole of mega2_fwd_bwd (below).  FUSE = 1: the per-row part of mega3_step (mega3.hpp), whose
// workgroups go on to the weight-gradient tiles: the body then leaves alpha_t and the closing span stamp to the caller.  Returns
// what the workgroup was: 0 a phantom panel's (its share of the pair's first layer is out), 1 a producer (partials published),
// 2 the panel's lead (backward chain done, every output stored write-through).
template <int FUSE>
__device__ __forceinline__ int mega2_body(const MegaArgs& a, float* const sm, unsigned* const m3_flags = nullptr) {
  constexpr int H = M2::H, L = M2::L, K = M2::K, D = M2::D, L2 = M2::L2, K2 = M2::K2;
  constexpr int Q = 4, H2f = 2 * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  // (the grid covers an EVEN number of panels: the first layer works on pairs of them; with an odd count the last pair's
  //  second panel is a phantom whose four workgroups take their part in the first layer and leave)
  const int B = a.B, nP = ((B + kPanel - 1) / kPanel + 1) & ~1;
  const int bid = blockIdx.x;
  // producers (quarters 1..3) take the LOWER block ids: a consumer can then never keep its producers off the chip
  const int q = bid < nP * (Q - 1) ? 1 + bid / nP : 0;
  const int pnl = bid < nP * (Q - 1) ? bid % nP : bid - nP * (Q - 1);
  const int r0 = pnl * kPanel;
  const int nrow = min(kPanel, B - r0);            // (<= 0: a phantom panel)
  const bool lead = q == 0;
  // who stores the panel's forward activations (kept for the weight gradients; every quarter holds the same bits): the lead,
  // or -- mega3_step -- quarter 1, whose end-of-role flag then covers them and who is off the launch's critical path
  const bool act = FUSE ? q == 1 : lead;
  if (a.span && tid == 0) a.span[2 * bid] = wall_clock64();
#define M2_SPAN_END() if (!FUSE && a.span && threadIdx.x == 0) a.span[2 * blockIdx.x + 1] = wall_clock64()
  const unsigned spin_limit = *a.err_word ? 0u : (1u << 19);       // bounded spins (see mega.hpp)
  const unsigned epoch0 = *a.epoch_word;           // tag of this step's hand-offs
  float* const img = sm + M2::IMG;
  float *P_h1 = sm + M2::P_h1, *P_y = sm + M2::P_y, *P_hg = sm + M2::P_hg, *P_pp = sm + M2::P_pp, *P_qp = sm + M2::P_qp;
  float *P_z = sm + M2::P_z, *P_sp = sm + M2::P_sp, *P_hd = sm + M2::P_hd, *P_eps = sm + M2::P_eps, *P_u = sm + M2::P_u;
  float *P_lg = sm + M2::P_lg, *nllp = sm + M2::nll, *rsum = sm + M2::rsum, *red = sm + M2::red, *dred = sm + M2::dred;
  float* P_dhd = sm + M2::P_dhd;
  GMVAE_STAMP(0);
  // diagnostics (tools/handoff_clock.py, GMVAE_STAMPS=5): device wall clock around the two in-launch hand-offs
#define M2_WC(i) if (a.dbg && a.fine == 5 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 8 + (i)] = wall_clock64()
  // GMVAE_STAMPS=6 (tools/flstamps.py): shader-clock stamps inside the first-layer stage
#define M2_FC(i) if (a.dbg && a.fine == 6 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime()

  // ======================================================================= FL: first layer over this quarter's columns
  // (the staging of mega_fwd_bwd's specialised instance: 49 bursts of 4 weight rows per tensor in two sub-chunks)
  float flt[4] = {0.f, 0.f, 0.f, 0.f};
  {
    constexpr int KQ = M2::fl_kq, kq4 = KQ / 4;
    float* const Wst = sm + M2::fl_W;
    float* const A_x = sm + M2::fl_A;
    const int k0 = q * KQ;
    const int sp = pnl >> 1, hh = pnl & 1;         // the pair of panels; this workgroup's weight tensor (and its own row tile)
    const int rs0 = sp * 2 * kPanel;               // first of the pair's 32 rows
    const float* const W0 = (hh ? a.w0b : a.w0a) + (long long)k0 * H;
    const unsigned epoch_fl = epoch0;
    const unsigned long long step = a.step_dev[0];
    constexpr int qer = L / 4, qur = (K + 3) / 4, qe = kPanel * qer, qu = kPanel * qur;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned xw[4];
    const int row_ = lane >> 4, piece = lane & 15;
    const int src = row_ * 64 + ((piece ^ ((row_ & 1) << 2)) << 2);
    // 49 bursts of 4 weight rows (1 KB each) in two sub-chunks of 24 and 25; always 3 + 4 per wave (clamped duplicates)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int idx = wave + 8 * j;
      __builtin_amdgcn_global_load_lds(W0 + (idx << 8) + src, Wst + (idx << 8), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {               // always four loads per thread (clamped): uniform vmcnt accounting
      const int i = min(tid + it * kMT, 2 * kPanel * kq4 - 1);
      const int row = i / kq4, k4 = (i - row * kq4) * 4;
      const unsigned wv = *reinterpret_cast<const unsigned*>(a.x + (long long)min(rs0 + row, B - 1) * D + k0 + k4);
      xw[it] = rs0 + row < B ? wv : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = 24 + min(wave + 8 * j, 24);
      __builtin_amdgcn_global_load_lds(W0 + (idx << 8) + src, Wst + (idx << 8), 16, 0, 0);
    }
    // the forward operand image: M2::imgF_pieces pieces per wave, behind the weight bursts in every wave's queue
#pragma unroll
    for (int j = 0; j < M2::imgF_pieces; ++j) {
      const int c = min(wave * 256 + j * 2048, M2::imgF - 256);
      __builtin_amdgcn_global_load_lds(a.img2f + c + lane * 4, img + c, 16, 0, 0);
    }
    M2_FC(0);
    if (tid < qe + qu && nrow > 0) {               // this panel's rows of the Philox streams (= gmvae_noise_fill's)
      const bool is_u = tid >= qe;
      const int li = is_u ? tid - qe : tid;
      const int qpr = is_u ? qur : qer;
      const int row = li / qpr, quad = li - row * qpr;
      noise_vals(a.row0 + (unsigned long long)(r0 + row), (unsigned)quad, is_u, a.seed, step, nz);
    }
    static_assert(M2::imgF_pieces == 9, "the counted waits below assume 9 image pieces per wave");
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");                     // the first half and the x bytes are in
    M2_FC(1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * kMT;
      if (i < 2 * kPanel * kq4) {
        const int row = i / kq4, k4 = (i - row * kq4) * 4;
        const unsigned w = xw[it];
        float2* const dst = reinterpret_cast<float2*>(A_x + row * kFlLda + k4);
        dst[0] = make_float2((float)(w & 0xff), (float)((w >> 8) & 0xff));
        dst[1] = make_float2((float)((w >> 16) & 0xff), (float)(w >> 24));
      }
    }
    __syncthreads();
    M2_FC(2);
    const int tl = wave & 3, rt = wave >> 2;        // this wave's 16 columns of the tensor, its row tile (= panel of the pair)
    const int swz = ((tl * 16 + ln) ^ ((lk & 1) << 4)) - (tl * 16 + ln);
    const float* const Wt = Wst + swz;
    const float* const At = A_x + rt * kPanel * kFlLda;
    acc = tile_ksteps<1, kFlLda>(At, Wt, H, 1, tl, 0, 24, 24, lane, acc);
    M2_FC(3);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                      // the second half (the image may still be landing)
    __syncthreads();
    M2_FC(4);
    acc = tile_ksteps<1, kFlLda>(At, Wt, H, 1, tl, 24, 49, 49, lane, acc);
    constexpr int ngr = 2 * kPanel * H;            // granules one workgroup publishes: [32 rows][64 columns]
    {
      unsigned long long* xo = a.xfl + (((long long)sp * 4 + q) * 2 + hh) * ngr;
      // granule (wave, lane, r) at ((2 wave + r / 2) 64 + lane) 2 + r % 2: a wave's two stores are 1 KB each, contiguous
      granule2_publish(xo + ((wave * 2 + 0) * 64 + lane) * 2, epoch_fl, acc[0], acc[1]);
      granule2_publish(xo + ((wave * 2 + 1) * 64 + lane) * 2, epoch_fl, acc[2], acc[3]);
    }
    M2_FC(5);
    if (bid == 0 && tid == 0) a.step_dev[1] = step;       // the copy dw_adam reads (block 0 is never a phantom)
    if (nrow <= 0) {                               // a phantom panel's workgroup: its share of the pair's first layer is out
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (no LDS-DMA piece may still be landing when the LDS is handed on)
      M2_SPAN_END();
      return 0;
    }
    M2_WC(5);
    __syncthreads();                               // the staging area is dead: the panels it overlays may be written
    if (tid < qe) {
      st4(P_eps + tid * 4, make_float4(nz[0], nz[1], nz[2], nz[3]));
    } else if (tid < qe + qu) {
      const int li = tid - qe, row = li / qur, k0u = (li - row * qur) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0u + j < K) P_u[row * 16 + k0u + j] = nz[j];
    }
    {
      // the four quarters' partials (this workgroup's own included): all 16 granules of a lane in ONE sweep, re-read
      // until every tag carries this step's epoch; summed in quarter order, so every workgroup of the panel gets the same bits
      // output (row lk * 4 + r of THIS panel = row tile hh of the pair, column wave * 16 + ln of the 128): tensor wave >> 2,
      // its column tile wave & 3 -- from each of the four quarters' workgroups that hold that tensor
      const unsigned long long* const xp0 = a.xfl + ((long long)sp * 4 * 2 + (wave >> 2)) * ngr + (((hh * 4 + (wave & 3)) * 2) * 64 + lane) * 2;
      u32x4_t gv[8];                               // [quarter pq][pair]: {value r = 2 pair, epoch, value r = 2 pair + 1, epoch}
      unsigned spins = 0;
      for (;;) {
        // all 8 loads are in flight before the first tag is looked at
        if (M2_L2_SWEEPS > 0 && spins < M2_L2_SWEEPS && (wave >> 2) == hh) {        // (wave-uniform) this panel's own quarters: same XCD
#pragma unroll
          for (int pq = 0; pq < 4; ++pq) {
            gv[2 * pq] = granule2_load_l2(xp0 + (long long)pq * 2 * ngr);
            gv[2 * pq + 1] = granule2_load_l2(xp0 + (long long)pq * 2 * ngr + 128);
          }
        } else {
#pragma unroll
          for (int pq = 0; pq < 4; ++pq) {
            gv[2 * pq] = granule2_load(xp0 + (long long)pq * 2 * ngr);
            gv[2 * pq + 1] = granule2_load(xp0 + (long long)pq * 2 * ngr + 128);
          }
        }
        g2_wait(gv);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok = ok && gv[i][1] == epoch_fl && gv[i][3] == epoch_fl;
        if (__all(ok)) break;
        if (++spins > spin_limit) {
          if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
          for (int i = 0; i < 8; ++i) { gv[i][0] = 0x7fc00000u; gv[i][2] = 0x7fc00000u; }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                // summed in quarter order: every workgroup of the panel gets the same bits
        flt[r] = __uint_as_float(gv[0 + (r >> 1)][2 * (r & 1)]);
        flt[r] += __uint_as_float(gv[2 + (r >> 1)][2 * (r & 1)]);
        flt[r] += __uint_as_float(gv[4 + (r >> 1)][2 * (r & 1)]);
        flt[r] += __uint_as_float(gv[6 + (r >> 1)][2 * (r & 1)]);
      }
      if (a.dbg && a.fine >= 5 && tid == 0) {
        a.dbg[(size_t)blockIdx.x * 16 + 14] = spins;
        a.dbg[(size_t)blockIdx.x * 16 + 0] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID: the XCD this workgroup runs on
      }
    }
  }
  M2_WC(6);
  M2_FC(6);
  // ---- decoder operands of this wave's column tiles, straight into registers (used ~10 stages from here)
  // local tile lt = wave, wave + 8 of this workgroup's part; global tile t = 4 lt + q
  const int ntq = m2_dec_ntiles(q);                // 8 tiles for the lead (one per wave), 14 / 14 / 13 for the producers
  const bool two = wave + 8 < ntq;                 // (wave-uniform) this wave has a second tile
  const bool one = wave < ntq;                     //               ... a first one (the lead may own fewer tiles than waves)
  float4 wf[2][4], wb[2][4], bias4[2];
  unsigned xb[2] = {0u, 0u};
  {
    const float* const dfq = a.dimg2 + M2::dF + q * M2::dFq;
    const float* const dbq = a.dimg2 + M2::dB + q * M2::dBq;
    const float* const bq = a.dimg2 + M2::dbias + q * M2::dbq;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it == 0 ? one : two) {
        const int lt = wave + 8 * it;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) wf[it][kt] = ld4(dfq + (((kt * 4 + lk) * M2::DC + lt * 16 + ln) << 2));
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) wb[it][ht] = ld4(dbq + (((lt * 4 + lk) * 64 + ht * 16 + ln) << 2));
        bias4[it] = ld4(bq + lt * 16 + 4 * lk);
        const int c0 = m2_dec_tile(q, lt) * 16;
        xb[it] = *reinterpret_cast<const unsigned*>(a.x + (long long)min(r0 + ln, B - 1) * D + c0 + 4 * lk);
      }
    }
  }
  // vmcnt retires in order: everything OLDER than these 10 (20) register loads -- the whole operand image -- has landed
  // once at most that many are outstanding; the loads themselves stay in flight across the next stages.
  if (two) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (one) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  M2_FC(7);
  {                                                // bias + ReLU straight from the (row-major oriented) accumulator
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = lk * 4 + r, col = wave * 16 + ln;
      if (col < H) {
        const float h = fmaxf(flt[r] + img[M2::b_y0 + col], 0.f);
        P_h1[row * M2::ld128 + col] = h;
      } else {
        P_h1[row * M2::ld128 + col] = flt[r];
      }
    }
  }
  __syncthreads();
  GMVAE_STAMP(1);
  if (act && tid < 256) {                          // hy1 (kept for dWy1) leaves as 16-byte write-through stores
    const int row = tid >> 4, c = (tid & 15) << 2;
    if (row < nrow) st4o(a.hy1 + (long long)(r0 + row) * H + c, ld4(P_h1 + row * M2::ld128 + c));
  }
  // ======================================================================= F: forward chain
  // S1 logits: one 16-output tile, contraction 64 split over waves 0..3
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<1, 16>(img + M2::Wy1f, 0, P_h1, M2::ld128, wave, ln, lk, acc);
    st4(red + wave * 320 + ln * M2::ldk + 4 * lk, f4(acc));
  }
  __syncthreads();
  // S2 Gumbel-softmax + entropy: 16 lanes per row (scripts/gmvae.py:240,262-263)
  if (tid < 256) {
    const int row = tid >> 4, k = tid & 15;
    const bool ok = row < nrow, kv = k < K;
    float lg = -INFINITY, av = -INFINITY;
    if (kv) {
      lg = red[row * M2::ldk + k] + red[320 + row * M2::ldk + k] + red[640 + row * M2::ldk + k] + red[960 + row * M2::ldk + k] +
           img[M2::b_y1 + k];
      const float uu = ok ? P_u[row * 16 + k] : 0.5f;
      av = (lg - flog(-flog(uu))) * a.invT;
    }
    const float mx = row16_max(av);
    const float se = row16_sum(kv ? fexp(av - mx) : 0.f);
    const float lse = mx + flog(se);
    const float lga[1] = {lg};
    float lpa[1];
    cat_log_softmax<Row16, 1>(lga, lpa);           // log pi (gemm.hpp: accurate for a saturated q(y|x)); kept for phase B
    float yv = 0.f, ne = 0.f;
    if (kv) {
      yv = fexp(av - lse);
      const float lp = lpa[0];
      P_lg[row * M2::ldk + k] = lp;
      ne = fexp(lp) * lp;
      if (ok && act) st1o(a.y + (long long)(r0 + row) * K2 + k, yv);            // rows of pad4(K) floats
    }
    P_y[row * M2::ldk + k] = yv;                                            // columns K..15 are zero
    ne = row16_sum(ne);
    if (k == 0) { nllp[3 * kPanel + row] = ne; if (ok && lead) st1o(a.nent + r0 + row, ne); }
  }
  __syncthreads();
  GMVAE_STAMP(2);
  // S3 prior head (8 tiles) and encoder_gmm hidden (4 tiles), contraction = y (one tile of 16)
  // (Round 4, measured and reverted: the producers skipping the prior head and its half of S5 -- the launch got 0.5 us
  //  LONGER: the lead, which cannot skip them, then reaches its hand-off later relative to the producers' publish.)
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<1, 128>(img + M2::Wpf, wave, P_y, M2::ldk, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 bb = ld4(img + M2::b_p + c0);
    st4(P_pp + ln * M2::ld128 + c0, make_float4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w));
    if (wave < 4) {
      f32x4 ac2 = {0.f, 0.f, 0.f, 0.f};
      ac2 = m2_tile<1, 64>(img + M2::Wg0yf, wave, P_y, M2::ldk, 0, ln, lk, ac2);
      const float4 gx = ld4(P_h1 + ln * M2::ld128 + H + c0), bg = ld4(img + M2::b_g0 + c0);
      const float4 h = make_float4(fmaxf(ac2[0] + gx.x + bg.x, 0.f), fmaxf(ac2[1] + gx.y + bg.y, 0.f),
                                   fmaxf(ac2[2] + gx.z + bg.z, 0.f), fmaxf(ac2[3] + gx.w + bg.w, 0.f));
      st4(P_hg + ln * M2::ld64 + c0, h);
      if (act && ln < nrow) st4o(a.hg1 + (long long)(r0 + ln) * H + c0, h);
    }
  }
  __syncthreads();
  // S4 q head: 8 tiles, contraction 64
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, 128>(img + M2::Wg1f, wave, P_hg, M2::ld64, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 bb = ld4(img + M2::b_g1 + c0);
    st4(P_qp + ln * M2::ld128 + c0, make_float4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w));
  }
  __syncthreads();
  GMVAE_STAMP(3);
  // S5 z, log q, log p: 32 lanes per row; the panels are rewritten in place with the backward chain's inputs
  // (P_qp = [sigmoid(raw_q) | sigma_q], P_pp = [t | sigma_p], P_sp = sigmoid(raw_p)): phase B needs no transcendental
  {
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
    float aq = 0.f, ap = 0.f;
#pragma unroll
    for (int l = sub; l < L; l += 32) {
      float* const qr = P_qp + row * M2::ld128;
      float* const pr = P_pp + row * M2::ld128;
      const float mu = qr[l];
      const float vq = qr[L + l] + a.c;
      float sgq;                                             // sigmoid(raw_q + c): softplus' derivative, kept for phase B
      const float sg = fmaxf(softplus_sig(vq, sgq), a.smin);
      const float ee = ok ? P_eps[row * 64 + l] : 0.f;
      const float zz = mu + sg * ee;
      P_z[row * M2::ld64 + l] = zz;
      qr[l] = sgq;
      qr[L + l] = sg;
      aq += -0.5f * ee * ee - 0.5f * kLog2Pi - flog(sg);     // (z - mu) / sigma IS eps
      const float vp = pr[L + l] + a.c;                      // p(z|y): gmvae.py:258
      float sgp;
      const float sp = fmaxf(softplus_sig(vp, sgp), a.smin);
      const float t = (zz - pr[l]) * __builtin_amdgcn_rcpf(sp);
      ap += -0.5f * t * t - 0.5f * kLog2Pi - flog(sp);
      pr[l] = t;
      pr[L + l] = sp;
      P_sp[row * 64 + l] = sgp;
    }
    aq = row32_sum(aq); ap = row32_sum(ap);
    if (sub == 0) {
      nllp[row] = aq; nllp[kPanel + row] = ap;
      if (ok && lead) { st1o(a.logq + r0 + row, aq); st1o(a.logp + r0 + row, ap); }
    }
  }
  __syncthreads();
  if (act && tid >= 256) {                         // z (kept for dWd0): waves 4..7, which have no tile in S6
    const int t2 = tid - 256, row = t2 >> 4, c = (t2 & 15) << 2;
    if (row < nrow) st4o(a.z + (long long)(r0 + row) * L + c, ld4(P_z + row * M2::ld64 + c));
  }
  // S6 decoder hidden: 4 tiles, contraction 64
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, 64>(img + M2::Wd0f, wave, P_z, M2::ld64, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 bb = ld4(img + M2::b_d0 + c0);
    const float4 h = make_float4(fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
    st4(P_hd + ln * M2::ld64 + c0, h);
    if (act && ln < nrow) st4o(a.hd1 + (long long)(r0 + ln) * H + c0, h);
  }
  __syncthreads();                                 // the forward image is dead
  GMVAE_STAMP(4);
  if (lead) dma_copy_m(img, a.img2b, M2::imgB, wave, lane);       // the backward image lands while the decoder layer runs

  // ======================================================================= D: decoder output layer, in registers
  f32x4 dacc[4];
#pragma unroll
  for (int ht = 0; ht < 4; ++ht) dacc[ht] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  float4 gkeep[2];
  {
    float4 hb[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) hb[kt] = ld4(P_hd + ln * M2::ld64 + kt * 16 + 4 * lk);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int lt = wave + 8 * it;
      if (lt < ntq) {                               // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][kt].x, hb[kt].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][kt].y, hb[kt].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][kt].z, hb[kt].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][kt].w, hb[kt].w, acc, 0, 0, 0);
        }
        // lane = (row ln, columns c0 + 4 lk + r): Bernoulli term and g = sigmoid(lambda) - x, ONE exp, rcp and log each
        const float bv[4] = {bias4[it].x, bias4[it].y, bias4[it].z, bias4[it].w};
        const bool ok = ln < nrow;
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lam = acc[r] + bv[r] + a.gen_bias;
          const float e = fexp(-fabsf(lam));                 // (bare v_exp_f32 / v_log_f32: chain.hpp)
          const float rcp = __builtin_amdgcn_rcpf(1.f + e);
          const float sp = fmaxf(lam, 0.f) - flog(rcp);
          const float sgm = lam >= 0.f ? rcp : e * rcp;
          const float x_ = ok ? (float)((xb[it] >> (8 * r)) & 0xffu) : 0.f;
          rs += ok ? x_ * lam - sp : 0.f;
          g[r] = ok ? sgm - x_ : 0.f;
        }
        // quarter 0 keeps g in registers until its hand-off polls are through: vmcnt retires in order, so a poll's
        // data would otherwise wait for the acknowledgement of these stores (measured: 1.8 us per sweep)
        if (lead) gkeep[it] = make_float4(g[0], g[1], g[2], g[3]);
        else if (ok) st4o(a.g + (long long)(r0 + ln) * D + m2_dec_tile(q, lt) * 16 + 4 * lk, make_float4(g[0], g[1], g[2], g[3]));
        // dhd1 += g Wd1^T over this tile's 16 columns: the accumulator layout IS the B operand
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
          dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[it][ht].x, g[0], dacc[ht], 0, 0, 0);
          dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[it][ht].y, g[1], dacc[ht], 0, 0, 0);
          dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[it][ht].z, g[2], dacc[ht], 0, 0, 0);
          dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[it][ht].w, g[3], dacc[ht], 0, 0, 0);
        }
      }
    }
  }
  GMVAE_STAMP(5);
  // per-wave partials -> LDS: dhd1 tiles as [ht][lk][row][4] (a linear float4 per thread in the reduction below)
  rs += __shfl_xor(rs, 16, 64);
  rs += __shfl_xor(rs, 32, 64);
  if (lk == 0) rsum[wave * kPanel + ln] = rs;
#pragma unroll
  for (int ht = 0; ht < 4; ++ht) st4(dred + wave * 1024 + (((ht * 4 + lk) * 16 + ln) << 2), f4(dacc[ht]));
  __syncthreads();
  M2_WC(0);
  const unsigned epoch = epoch0;                   // (read at the kernel's start: the word changes between launches only)
  constexpr int ngr = kPanel * H + kPanel;         // granules one producer publishes: dhd1 partials + row sums
  // thread t < 256 owns dhd1[row = t & 15][4 (t >> 4) .. + 3]; its four granules are [4 t, 4 t + 4) of a producer's
  // block, so that a wave's hand-off stores and polls are contiguous in memory (granules laid out [row][col] cost every
  // lane a cache line of its own: 1.8 us per sweep, tools/handoff_clock.py)
  float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float rsn = 0.f;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < kMW; ++w) {
      const float4 o = ld4(dred + w * 1024 + 4 * tid);
      dsum.x += o.x; dsum.y += o.y; dsum.z += o.z; dsum.w += o.w;
    }
  } else if (tid < 256 + kPanel) {
#pragma unroll
    for (int w = 0; w < kMW; ++w) rsn += rsum[w * kPanel + (tid - 256)];
  }
  const int orow = tid & 15, ocol = (tid >> 4) << 2;     // for tid < 256
  if (!lead) {
    // ------------------------------------------------------------- producer: publish the partials and leave
    unsigned long long* xo = a.xchg + ((long long)pnl * (Q - 1) + (q - 1)) * ngr;
    if (tid < 256) {                               // granule (thread t, j) at ((2 (t / 64) + j / 2) 64 + t % 64) 2 + j % 2
      granule2_publish(xo + ((wave * 2 + 0) * 64 + lane) * 2, epoch, dsum.x, dsum.y);
      granule2_publish(xo + ((wave * 2 + 1) * 64 + lane) * 2, epoch, dsum.z, dsum.w);
    } else if (tid < 256 + kPanel) {
      granule_publish(xo + kPanel * H + (tid - 256), ((unsigned long long)epoch << 32) | __float_as_uint(rsn));
    }
    M2_WC(1);
    if (!FUSE && bid == 0 && tid == 0 && a.lr_t_out) {      // (a producer: off the launch's critical path)
      const double t = (double)(a.step_dev[0] + 1ull);
      *a.lr_t_out = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, t)) / (1.0 - pow((double)a.b1, t)));
    }
    M2_SPAN_END();
    return 1;
  }
  // ======================================================================= B: backward chain (quarter 0)
  M2_WC(1);
  {
    const unsigned long long* xi = a.xchg + (long long)pnl * (Q - 1) * ngr;
    const bool wd = tid < 256, wn = tid >= 256 && tid < 256 + kPanel;
    // all three producers' granules are requested in ONE sweep (a sweep is a memory round trip): waves 0..3 their threads'
    // four dhd1 granules per producer as two 16-byte loads, wave 4 (lanes 0..15) the row-sum granules; waves 5..7 own
    // nothing and stay out of the CU's memory queue (wave-uniform branches: the loads of a wave are still issued together)
    unsigned spins = 0;
    if (wave < 4) {
      u32x4_t v[6];
      const unsigned long long* const xs = xi + ((wave * 2) * 64 + lane) * 2;
      for (;;) {
        if (M2_L2_SWEEPS > 0 && spins < M2_L2_SWEEPS) {
#pragma unroll
          for (int pq = 0; pq < Q - 1; ++pq) {
            v[2 * pq] = granule2_load_l2(xs + (long long)pq * ngr);
            v[2 * pq + 1] = granule2_load_l2(xs + (long long)pq * ngr + 128);
          }
        } else {
#pragma unroll
          for (int pq = 0; pq < Q - 1; ++pq) {
            v[2 * pq] = granule2_load(xs + (long long)pq * ngr);
            v[2 * pq + 1] = granule2_load(xs + (long long)pq * ngr + 128);
          }
        }
        g2_wait(v);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 6; ++i) ok = ok && v[i][1] == epoch && v[i][3] == epoch;
        if (__all(ok)) break;
        if (++spins > spin_limit) {                          // a producer never ran; flag and go on
          if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
          for (int i = 0; i < 6; ++i) { v[i][0] = 0x7fc00000u; v[i][2] = 0x7fc00000u; }      // NaN: the step's loss and gradients say so loudly
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
#pragma unroll
      for (int pq = 0; pq < Q - 1; ++pq) {
        dsum.x += __uint_as_float(v[2 * pq][0]); dsum.y += __uint_as_float(v[2 * pq][2]);
        dsum.z += __uint_as_float(v[2 * pq + 1][0]); dsum.w += __uint_as_float(v[2 * pq + 1][2]);
      }
    } else if (wave == 4) {
      unsigned long long v[Q - 1];
      const unsigned long long* const xs = xi + kPanel * H + min(lane, kPanel - 1);
      for (;;) {
#pragma unroll
        for (int pq = 0; pq < Q - 1; ++pq)
          v[pq] = (M2_L2_SWEEPS > 0 && spins < M2_L2_SWEEPS) ? __hip_atomic_load(xs + (long long)pq * ngr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                                             : __hip_atomic_load(xs + (long long)pq * ngr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_sched_barrier(0);
        bool ok = true;
#pragma unroll
        for (int pq = 0; pq < Q - 1; ++pq) ok = ok && (unsigned)(v[pq] >> 32) == epoch;
        if (__all(ok)) break;
        if (++spins > spin_limit) {
          if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
          for (int pq = 0; pq < Q - 1; ++pq) v[pq] = 0x7fc00000ull;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
#pragma unroll
      for (int pq = 0; pq < Q - 1; ++pq)
        if (wn) rsn += __uint_as_float((unsigned)v[pq]);
    }
    if (a.dbg && a.fine >= 5 && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 15] = spins;       // sweeps that failed
    if (ln < nrow && one) {                        // now the decoder tiles' g = sigmoid(lambda) - x
      st4o(a.g + (long long)(r0 + ln) * D + m2_dec_tile(0, wave) * 16 + 4 * lk, gkeep[0]);
      if (two) st4o(a.g + (long long)(r0 + ln) * D + m2_dec_tile(0, wave + 8) * 16 + 4 * lk, gkeep[1]);
    }
    M2_WC(2);
    if (wd) {                                      // masked top gradient (+ saved for dWd0)
      const float4 hd = ld4(P_hd + orow * M2::ld64 + ocol);
      const bool ok = orow < nrow;
      const float4 d = make_float4((ok && hd.x > 0.f) ? dsum.x : 0.f, (ok && hd.y > 0.f) ? dsum.y : 0.f,
                                   (ok && hd.z > 0.f) ? dsum.z : 0.f, (ok && hd.w > 0.f) ? dsum.w : 0.f);
      st4(P_dhd + orow * M2::ld64 + ocol, d);
      if (ok) st4o(a.dhd1 + (long long)(r0 + orow) * H + ocol, d);
    }
    if (wn) {
      const int row = tid - 256;
      if (row < nrow) {
        st1o(a.logpx + r0 + row, rsn);
        st1o(a.logw + r0 + row, rsn + nllp[kPanel + row] - nllp[row] - nllp[3 * kPanel + row]);
      }
    }
  }
  M2_WC(3);
  // the backward image has landed: the hand-off polls of waves 0..4 returned data, and vmcnt retires in order -- their
  // pieces of the image were issued long before; waves 5..7 wait for everything but their one g store just issued.  (An
  // s_waitcnt vmcnt(0) here also waited for the write-through acknowledgement of the g stores: ~1 us in front of the
  // backward chain.)
  if (wave > 4) { if (one) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();
  M2_WC(4);
  GMVAE_STAMP(6);
  float *P_dz = sm + M2::P_dz, *P_dqp = sm + M2::P_dqp, *P_dpp = sm + M2::P_dpp, *P_dhg = sm + M2::P_dhg, *P_dl = sm + M2::P_dl;
  // B1 dz_dec = dhd1 * Wd0^T: 4 tiles, contraction 64
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, 64>(img + M2::Wd0b, wave, P_dhd, M2::ld64, 0, ln, lk, acc);
    st4(P_dz + ln * M2::ld64 + wave * 16 + 4 * lk, f4(acc));
  }
  __syncthreads();
  // B2 q / prior heads backward: 32 lanes per row; inputs prepared by S5
  {
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
#pragma unroll
    for (int l = sub; l < L; l += 32) {
      float dmu = 0.f, draw = 0.f, dmup = 0.f, drawp = 0.f;
      if (ok) {
        const float sg = P_qp[row * M2::ld128 + L + l];
        const float sp = P_pp[row * M2::ld128 + L + l], t = P_pp[row * M2::ld128 + l];
        const float isp = __builtin_amdgcn_rcpf(sp);
        const float pterm = t * isp;                   // d(-log p)/dz
        dmup = -pterm;
        drawp = (sp > a.smin) ? (1.f - t * t) * isp * P_sp[row * 64 + l] : 0.f;
        dmu = P_dz[row * M2::ld64 + l] + pterm;
        const float dsg = dmu * P_eps[row * 64 + l] - __builtin_amdgcn_rcpf(sg);
        draw = (sg > a.smin) ? dsg * P_qp[row * M2::ld128 + l] : 0.f;
      }
      P_dqp[row * M2::ld128 + l] = dmu;
      P_dqp[row * M2::ld128 + L + l] = draw;
      P_dpp[row * M2::ld128 + l] = dmup;
      P_dpp[row * M2::ld128 + L + l] = drawp;
    }
  }
  __syncthreads();
  {                                                // dqp, dpp (kept for dWg1, dWp): one 16-byte write-through store each per thread
    const int row = tid >> 5, c = (tid & 31) << 2;
    if (row < nrow) {
      st4o(a.dqp + (long long)(r0 + row) * L2 + c, ld4(P_dqp + row * M2::ld128 + c));
      st4o(a.dpp + (long long)(r0 + row) * L2 + c, ld4(P_dpp + row * M2::ld128 + c));
    }
  }
  // B3 dhg = (dqp * Wg1^T) [hg > 0]: 4 tiles, contraction 128 split in two over the wave halves
  {
    const int t = wave & 3, kh = wave >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, 64>(img + M2::Wg1b, t, P_dqp, M2::ld128, 4 * kh, ln, lk, acc);
    if (kh) st4(red + t * 256 + ((lk * 16 + ln) << 2), f4(acc));
    __syncthreads();
    if (!kh) {
      const float4 o = ld4(red + t * 256 + ((lk * 16 + ln) << 2));
      const int c0 = t * 16 + 4 * lk;
      const float4 hg = ld4(P_hg + ln * M2::ld64 + c0);
      const bool ok = ln < nrow;
      const float4 d = make_float4((ok && hg.x > 0.f) ? acc[0] + o.x : 0.f, (ok && hg.y > 0.f) ? acc[1] + o.y : 0.f,
                                   (ok && hg.z > 0.f) ? acc[2] + o.z : 0.f, (ok && hg.w > 0.f) ? acc[3] + o.w : 0.f);
      st4(P_dhg + ln * M2::ld64 + c0, d);
      if (ok) st4o(a.dhg1 + (long long)(r0 + ln) * H + c0, d);
    }
  }
  __syncthreads();
  // B4 dy = dhg * Wg0[D:]^T + dpp * Wp^T: one tile; the 4 + 8 contraction tiles are spread over the 8 waves
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) {
      acc = m2_tile<1, 16>(img + M2::Wg0yb, 0, P_dhg, M2::ld64, wave, ln, lk, acc);
      acc = m2_tile<1, 16>(img + M2::Wpb, 0, P_dpp, M2::ld128, 4 + wave, ln, lk, acc);
    } else {
      acc = m2_tile<1, 16>(img + M2::Wpb, 0, P_dpp, M2::ld128, wave - 4, ln, lk, acc);
    }
    st4(red + wave * 256 + ((lk * 16 + ln) << 2), f4(acc));
  }
  __syncthreads();
  // B5 softmax backward + entropy gradient
  if (tid < 256) {
    const int row = tid >> 4, k = tid & 15;
    const bool ok = row < nrow, kv = k < K && ok;
    float dy = 0.f;
#pragma unroll
    for (int w = 0; w < kMW; ++w) dy += red[w * 256 + (((k >> 2) * 16 + row) << 2) + (k & 3)];
    const float ya[1] = {kv ? P_y[row * M2::ldk + k] : 0.f}, dya[1] = {kv ? dy : 0.f};
    float daa[1];
    cat_softmax_bwd<Row16, 1>(ya, dya, daa);       // y (dy - y . dy), shifted (gemm.hpp)
    const float ne = nllp[3 * kPanel + row];
    float dl = 0.f;
    if (kv) {
      const float lp = P_lg[row * M2::ldk + k];    // log pi, left by S2
      dl = daa[0] * a.invT + fexp(lp) * (lp - ne);
      st1o(a.dlogits + (long long)(r0 + row) * K2 + k, dl);
    }
    P_dl[row * M2::ldk + k] = dl;
  }
  __syncthreads();
  // B6 dhy1 = (dlogits * Wy1^T) [hy1 > 0]: 4 tiles, contraction = k (one tile)
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<1, 64>(img + M2::Wy1b, wave, P_dl, M2::ldk, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 hy = ld4(P_h1 + ln * M2::ld128 + c0);
    if (ln < nrow)
      st4o(a.dhy1 + (long long)(r0 + ln) * H + c0,
           make_float4(hy.x > 0.f ? acc[0] : 0.f, hy.y > 0.f ? acc[1] : 0.f, hy.z > 0.f ? acc[2] : 0.f, hy.w > 0.f ? acc[3] : 0.f));
  }
  GMVAE_STAMP(7);
  M2_SPAN_END();
#undef M2_SPAN_END
  return 2;
}

__global__ __launch_bounds__(kMT) void mega2_fwd_bwd(const MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  mega2_body<0>(a, sm);
}

}  // namespace gmvae
