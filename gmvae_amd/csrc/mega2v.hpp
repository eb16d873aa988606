// mega2v_fwd_bwd: mega2_fwd_bwd's design (mega2.hpp: transposed products on row-major panels, the decoder layer in
// registers, 16-byte granule pairs in the hand-offs) for the VAE family at small batches -- the plain VAE
// (scripts/vae.py:167-185, standard-normal prior vae.py:247-250) and VAE_GMP (learned mixture prior, vae.py:231-244) at
// hidden 64, MNIST D = 784: BASELINE configs[0] (latent 2, batch 100) and configs[1] (latent 64, K = 10, batch 256), which
// ran round 1's mega_fwd_bwd ([k][17] LDS images, decoder layer through a 2-deep LDS ring) on 28 / 64 workgroups of a
// 256-CU chip.
//
// A 16-row panel is shared by Q = 7 workgroups (7 panels x 7 = 49, 16 x 7 = 112 workgroups), because 784 = 7 x 112 and
// 49 = 7 x 7: the first layer's contraction splits into seven 112-row slices (28 KB of weights per workgroup; the seven
// [16 x 64] partials meet through granules, every workgroup ending with identical bits) and the decoder's 49 column tiles
// into ONE tile per wave of seven waves per workgroup.  Quarter 0 (the lead) keeps the activations, receives the six
// producers' dhd1 and row-sum partials and runs the backward chain; VAE_GMP adds the mixture log-density (a 32-lane
// reduction per component), its share of dz and one partial of the prior variables' gradients per panel.
#pragma once
#include "mega2.hpp"

namespace gmvae {

template <int MODEL, int LT, int KT>
struct M2V {
  static constexpr int H = 64, D = 784, Q = 7, L = LT, K = KT;
  static constexpr int LP = (L + 15) & ~15, L2 = 2 * L, L2P = (L2 + 15) & ~15;
  static constexpr int KQ = D / Q;                  // 112 contraction rows per workgroup: 28 bursts of 4 rows, 28 k-steps
  static constexpr int DC = 16 * 7;                 // decoder columns of one workgroup's part: 7 tiles
  static constexpr int ldM = L + 1;
  // ---- forward operand image (floats)
  static constexpr int b_e0 = 0, b_g1 = 64, b_d0 = 64 + L2P;
  static constexpr int Wg1f = (b_d0 + 64 + 255) & ~255;           // [16][L2P][4]   qp = he * We1   (contraction h)
  static constexpr int Wd0f = Wg1f + 16 * L2P * 4;                 // [LP/4][64][4]  hd = z * Wd0    (contraction l, padded)
  static constexpr int M_loc = Wd0f + (LP / 4) * 64 * 4;           // VAE_GMP: loc [K][L+1], raw scale [K][L+1], mixture logits [16]
  static constexpr int M_raw = M_loc + (MODEL == 1 ? ((K * ldM + 3) & ~3) : 0);
  static constexpr int M_mix = M_raw + (MODEL == 1 ? ((K * ldM + 3) & ~3) : 0);
  static constexpr int imgF = (M_mix + (MODEL == 1 ? 16 : 0) + 255) & ~255;
  // ---- backward operand image (overlays the forward one; the mixture prior's arrays lie behind it and survive)
  static constexpr int Wd0b = 0;                                   // [16][LP][4]    dz  = dhd * Wd0^T (contraction h)
  static constexpr int Wg1b = Wd0b + 16 * LP * 4;                  // [L2P/4][64][4] dhe = dqp * We1^T (contraction 2L, padded)
  static constexpr int imgB = Wg1b + (L2P / 4) * 64 * 4;
  static_assert(MODEL != 1 || imgB <= M_loc, "the backward image must not reach the mixture prior's arrays");
  // ---- decoder operand images in global memory, per part q < 7: forward [16][DC][4], transposed [DC/4][64][4], bias [DC]
  static constexpr int dF = 0, dFq = 16 * DC * 4;
  static constexpr int dB = Q * dFq, dBq = (DC / 4) * 64 * 4;
  static constexpr int dbias = dB + Q * dBq, dbq = DC;
  static constexpr int dimg = dbias + Q * dbq;
  // ---- LDS map (floats)
  static constexpr int ld64 = 68, ldq = L2P + 4, ldz = LP + 4, kFlA = 130;      // (kFlA: x image row stride, = 2 mod 32, >= KQ)
  static constexpr int IMG = 0;
  static constexpr int P_h1 = imgF;                  // [16][68]   relu(he)
  static constexpr int P_qp = P_h1 + 16 * ld64;      // [16][ldq]  q head -> (sigmoid(raw_q) | sigma_q)
  static constexpr int P_z = P_qp + 16 * ldq;        // [16][ldz]  (columns >= L stay zero)
  static constexpr int P_hd = P_z + 16 * ldz;        // [16][68]
  static constexpr int P_eps = P_hd + 16 * ld64;     // [16][LP]
  static constexpr int nll = P_eps + 16 * LP;        // [4][16]: log q, log p per row
  static constexpr int rsum = nll + 64;              // [8][16]
  static constexpr int red = rsum + 128;             // [8][256]
  static constexpr int dred = red + 2048;            // [8][1024] per-wave partial dhd1 tiles; later the backward panels
  static constexpr int P_dhd = dred + 8192;          // [16][68]
  static constexpr int M_inv = P_dhd + 16 * ld64;    // VAE_GMP: 1/s [K][L+1], per-component constants [16], weights [16], resp [16][16]
  static constexpr int M_c = M_inv + (MODEL == 1 ? ((K * ldM + 3) & ~3) : 0);
  static constexpr int M_w = M_c + (MODEL == 1 ? 16 : 0);
  static constexpr int P_r = M_w + (MODEL == 1 ? 16 : 0);
  static constexpr int total = P_r + (MODEL == 1 ? 256 : 0);
  static constexpr int P_dz = dred;                  // [16][ldz]
  static constexpr int P_dqp = P_dz + 16 * ldz;      // [16][ldq]
  static_assert(P_dqp + 16 * ldq <= P_dhd, "backward panels must fit the dred region");
  // the first layer's staging ([112][64] weights + [16][130] x image) overlays the panels (nothing of them is live yet)
  static constexpr int fl_W = imgF, fl_A = fl_W + KQ * 64;
  static constexpr int fl_red = (fl_A + 16 * kFlA + 3) & ~3;       // [4][256]: where the two k halves of the first layer meet
  static_assert(fl_red + 1024 <= P_dhd, "first-layer staging and its scratch must fit below the gradient panel");
  static_assert(total * 4 <= 160 * 1024, "LDS budget");
  static constexpr int imgF_pieces = (imgF + 2047) / 2048, imgB_pieces = (imgB + 2047) / 2048;
  static_assert(K <= 16 && L <= 64, "mixture stage: 16 components, two latent dimensions per lane");
};

template <int N>
__device__ __forceinline__ void g2_wait_n(u32x4_t (&v)[N]) {
  static_assert(N == 12 || N == 14, "operand lists below");
  if constexpr (N == 14)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11])::"memory");
}

// The launch's body.  FUSE = 0: the whole of mega2v_fwd_bwd (below).  FUSE = 1: the per-row part of mega3v_step (mega3.hpp), whose
// workgroups go on to the weight-gradient tiles: every output another workgroup reads in the same launch leaves write-through,
// quarter 1 (off the critical path) stores the panel's forward activations, alpha_t and the closing span stamp are the
// caller's.  Returns 1 for a producer (partials published), 2 for the panel's lead (backward chain done).
template <int MODEL, int LT, int KT, int FUSE>
__device__ __forceinline__ int mega2v_body(const MegaArgs& a, float* const sm) {
  typedef M2V<MODEL, LT, KT> V;
  constexpr int H = V::H, L = V::L, K = V::K, D = V::D, L2 = V::L2, LP = V::LP, L2P = V::L2P, Q = V::Q, KQ = V::KQ;
  constexpr bool gmp = MODEL == 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, nP = (B + kPanel - 1) / kPanel;
  const int bid = blockIdx.x;
  // producers (quarters 1..6) take the LOWER block ids: a consumer can then never keep its producers off the chip
  const int q = bid < nP * (Q - 1) ? 1 + bid / nP : 0;
  const int pnl = bid < nP * (Q - 1) ? bid % nP : bid - nP * (Q - 1);
  const int r0 = pnl * kPanel;
  const int nrow = min(kPanel, B - r0);
  const bool lead = q == 0;
  const bool act = FUSE ? q == 1 : lead;           // who stores the forward activations (every quarter holds the same bits)
  if (a.span && tid == 0) a.span[2 * bid] = wall_clock64();
#define M2V_SPAN_END() if (!FUSE && a.span && threadIdx.x == 0) a.span[2 * blockIdx.x + 1] = wall_clock64()
  const unsigned spin_limit = *a.err_word ? 0u : (1u << 19);       // bounded spins (see mega.hpp)
  const unsigned epoch = *a.epoch_word;            // tag of this step's hand-offs
  float* const img = sm + V::IMG;
  float *P_h1 = sm + V::P_h1, *P_qp = sm + V::P_qp, *P_z = sm + V::P_z, *P_hd = sm + V::P_hd, *P_eps = sm + V::P_eps;
  float *nllp = sm + V::nll, *rsum = sm + V::rsum, *red = sm + V::red, *dred = sm + V::dred, *P_dhd = sm + V::P_dhd;
  float *M_inv = sm + V::M_inv, *M_c = sm + V::M_c, *M_w = sm + V::M_w, *P_r = sm + V::P_r;
  const float *M_loc = img + V::M_loc, *M_raw = img + V::M_raw, *M_mix = img + V::M_mix;
  GMVAE_STAMP(0);

  // ======================================================================= FL: first layer over this workgroup's 112 rows of We0
  float flt[4] = {0.f, 0.f, 0.f, 0.f};
  {
    constexpr int kq4 = KQ / 4;                    // 28 byte quads of x per row, 28 k-steps
    float* const Wst = sm + V::fl_W;
    float* const A_x = sm + V::fl_A;
    const int k0 = q * KQ;
    const float* const W0 = a.w0a + (long long)k0 * H;
    const unsigned long long step = a.step_dev[0];
    constexpr int qer = (L + 3) / 4, qe = kPanel * qer;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    const int row_ = lane >> 4, piece = lane & 15;
    const int src = row_ * 64 + ((piece ^ ((row_ & 1) << 2)) << 2);      // (mega.hpp dma_stage_w: odd rows swap 16-column halves)
    // 28 bursts of 4 weight rows (1 KB each): 4 per wave (the last four clamped), then this thread's 4 bytes of x, then
    // the forward operand image (it lands beside all of this: nothing overlays it)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = min(wave + 8 * j, KQ / 4 - 1);
      __builtin_amdgcn_global_load_lds(W0 + (idx << 8) + src, Wst + (idx << 8), 16, 0, 0);
    }
    unsigned xw;
    {
      const int i = min(tid, kPanel * kq4 - 1);
      const int row = i / kq4, k4 = (i - row * kq4) * 4;
      const unsigned wv = *reinterpret_cast<const unsigned*>(a.x + (long long)min(r0 + row, B - 1) * D + k0 + k4);
      xw = row < nrow ? wv : 0u;
    }
#pragma unroll
    for (int j = 0; j < V::imgF_pieces; ++j) {
      const int c = min(wave * 256 + j * 2048, V::imgF - 256);
      __builtin_amdgcn_global_load_lds(a.img2f + c + lane * 4, img + c, 16, 0, 0);
    }
    if (tid < qe) {                                // this panel's rows of the Philox stream (= gmvae_noise_fill's)
      const int row = tid / qer, quad = tid - row * qer;
      noise_vals(a.row0 + (unsigned long long)(r0 + row), (unsigned)quad, false, a.seed, step, nz);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V::imgF_pieces) : "memory");      // the weights and the x bytes are in
    if (tid < kPanel * kq4) {
      const int row = tid / kq4, k4 = (tid - row * kq4) * 4;
      float2* const dst = reinterpret_cast<float2*>(A_x + row * V::kFlA + k4);
      dst[0] = make_float2((float)(xw & 0xff), (float)((xw >> 8) & 0xff));
      dst[1] = make_float2((float)((xw >> 16) & 0xff), (float)(xw >> 24));
    }
    __syncthreads();
    // wave w: column tile w & 3, k-steps [14 (w >> 2), + 14); the two halves meet in LDS
    const int tl = wave & 3, kh = wave >> 2;
    const int swz = ((tl * 16 + ln) ^ ((lk & 1) << 4)) - (tl * 16 + ln);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_ksteps<1, V::kFlA>(A_x, Wst + swz, H, 1, tl, 14 * kh, 14 * kh + 14, 14, lane, acc);
    // (the halves meet in a scratch BEHIND the staging area: other waves are still reading the weights and the x image)
    float* const flred = sm + V::fl_red;
    if (kh) st4(flred + tl * 256 + ((lk * 16 + ln) << 2), f4(acc));
    __syncthreads();
    constexpr int ngr = kPanel * H;                // granules one workgroup publishes: [16 rows][64 columns]
    if (!kh) {
      const float4 o = ld4(flred + tl * 256 + ((lk * 16 + ln) << 2));
      unsigned long long* xo = a.xfl + ((long long)pnl * Q + q) * ngr;
      granule2_publish(xo + ((wave * 2 + 0) * 64 + lane) * 2, epoch, acc[0] + o.x, acc[1] + o.y);
      granule2_publish(xo + ((wave * 2 + 1) * 64 + lane) * 2, epoch, acc[2] + o.z, acc[3] + o.w);
    }
    if (bid == 0 && tid == 0) a.step_dev[1] = step;       // the copy dw_adam reads
    __syncthreads();                               // the staging area is dead: the panels it overlays may be written
    if (tid < qe) {
      const int row = tid / qer, l0 = (tid - row * qer) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (l0 + j < L) P_eps[row * LP + l0 + j] = nz[j];
    }
    if (wave < 4) {
      // the seven workgroups' partials (this one's own included), all 14 loads of a lane in ONE sweep, re-read until every
      // tag carries this step's epoch; summed in quarter order: every workgroup of the panel gets the same bits
      const unsigned long long* const xp0 = a.xfl + (long long)pnl * Q * ngr + ((wave * 2) * 64 + lane) * 2;
      u32x4_t gv[2 * Q];
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int pq = 0; pq < Q; ++pq) {
          gv[2 * pq] = granule2_load(xp0 + (long long)pq * ngr);
          gv[2 * pq + 1] = granule2_load(xp0 + (long long)pq * ngr + 128);
        }
        g2_wait_n(gv);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 2 * Q; ++i) ok = ok && gv[i][1] == epoch && gv[i][3] == epoch;
        if (__all(ok)) break;
        if (++spins > spin_limit) {
          if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
          for (int i = 0; i < 2 * Q; ++i) { gv[i][0] = 0x7fc00000u; gv[i][2] = 0x7fc00000u; }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        flt[r] = __uint_as_float(gv[r >> 1][2 * (r & 1)]);
#pragma unroll
        for (int pq = 1; pq < Q; ++pq) flt[r] += __uint_as_float(gv[2 * pq + (r >> 1)][2 * (r & 1)]);
      }
    }
  }
  // ---- decoder operands of this wave's column tile (waves 0..6: tile 7 wave' + q), straight into registers
  const bool has = wave < 7;                       // (wave-uniform)
  float4 wf[4], wb[4], bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned xb = 0u;
  const int dtile = 7 * wave + q;                  // global column tile of (part q, local tile wave)
  if (has) {
    const float* const dfq = a.dimg2 + V::dF + q * V::dFq;
    const float* const dbq = a.dimg2 + V::dB + q * V::dBq;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) wf[kt] = ld4(dfq + (((kt * 4 + lk) * V::DC + wave * 16 + ln) << 2));
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) wb[ht] = ld4(dbq + (((wave * 4 + lk) * 64 + ht * 16 + ln) << 2));
    bias4 = ld4(a.dimg2 + V::dbias + q * V::dbq + wave * 16 + 4 * lk);
    xb = *reinterpret_cast<const unsigned*>(a.x + (long long)min(r0 + ln, B - 1) * D + dtile * 16 + 4 * lk);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // everything OLDER than these 10 loads -- the operand image -- has landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (wave < 4) {                                  // bias + ReLU straight from the (row-major oriented) accumulator
#pragma unroll
    for (int r = 0; r < 4; ++r) P_h1[(lk * 4 + r) * V::ld64 + wave * 16 + ln] = fmaxf(flt[r] + img[V::b_e0 + wave * 16 + ln], 0.f);
  }
  if (gmp) {                                       // mixture constants: one wave per component (mega.hpp)
    for (int k = wave; k < K; k += kMW) {
      float ls = 0.f;
      for (int l = lane; l < L; l += 64) {
        const float iv = 1.f / fsoftplus(M_raw[k * V::ldM + l]);
        M_inv[k * V::ldM + l] = iv;
        ls += flog(iv);
      }
      ls = wave_sum(ls);
      if (lane == 0) M_c[k] = ls;
    }
  }
  __syncthreads();
  GMVAE_STAMP(1);
  if (gmp && tid < 64) {
    float mx = lane < K ? M_mix[lane] : -INFINITY;
    mx = wave_max(mx);
    float se = lane < K ? fexp(M_mix[lane] - mx) : 0.f;
    se = wave_sum(se);
    if (lane < K) {
      const float lw = M_mix[lane] - (mx + flog(se));
      M_w[lane] = fexp(lw);
      M_c[lane] = lw + M_c[lane] - 0.5f * kLog2Pi * (float)L;
    }
  }
  if (act && tid < 256) {                          // he (kept for dWe1) leaves as 16-byte write-through stores
    const int row = tid >> 4, c = (tid & 15) << 2;
    if (row < nrow) st4o(a.hy1 + (long long)(r0 + row) * H + c, ld4(P_h1 + row * V::ld64 + c));
  }
  // ======================================================================= F: forward chain
  // S4 q head: L2P / 16 tiles, contraction 64
  if (wave < L2P / 16) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, L2P>(img + V::Wg1f, wave, P_h1, V::ld64, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 bb = ld4(img + V::b_g1 + c0);
    st4(P_qp + ln * V::ldq + c0, make_float4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w));
  }
  __syncthreads();
  GMVAE_STAMP(2);
  // S5 z, log q, log p: 32 lanes per row; P_qp is rewritten in place with the backward chain's inputs
  // ([sigmoid(raw_q) | sigma_q]); the latent padding of P_z stays zero (it meets zero weight rows)
  {
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
    float aq = 0.f, ap = 0.f;
    float* const qr = P_qp + row * V::ldq;
#pragma unroll
    for (int l = sub; l < LP; l += 32) {
      float zz = 0.f;
      if (l < L) {
        const float mu = qr[l];
        const float vq = qr[L + l] + a.c;
        float sgq;
        const float sg = fmaxf(softplus_sig(vq, sgq), a.smin);
        const float ee = ok ? P_eps[row * LP + l] : 0.f;
        zz = mu + sg * ee;
        aq += -0.5f * ee * ee - 0.5f * kLog2Pi - flog(sg);     // (z - mu) / sigma IS eps
        if (!gmp) ap += -0.5f * zz * zz - 0.5f * kLog2Pi;      // standard-normal prior (vae.py:247-250)
        __builtin_amdgcn_sched_barrier(0);
        qr[l] = sgq;
        qr[L + l] = sg;
      }
      P_z[row * V::ldz + l] = zz;
    }
    aq = row32_sum(aq); ap = row32_sum(ap);
    if (sub == 0) {
      nllp[row] = aq;
      if (!gmp) nllp[kPanel + row] = ap;
      if (ok && lead) { st1o(a.logq + r0 + row, aq); if (!gmp) st1o(a.logp + r0 + row, ap); }
    }
  }
  __syncthreads();
  GMVAE_STAMP(3);
  if (act && tid >= 256 && tid < 256 + kPanel * ((L + 3) / 4)) {      // z (kept for dWd0): waves 4.., which have no tile in S6
    const int t2 = tid - 256, row = t2 / ((L + 3) / 4), c = (t2 - row * ((L + 3) / 4)) << 2;
    if (row < nrow) {
      if constexpr ((L & 3) == 0) {
        st4o(a.z + (long long)(r0 + row) * L + c, ld4(P_z + row * V::ldz + c));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < L) st1o(a.z + (long long)(r0 + row) * L + c + j, P_z[row * V::ldz + c + j]);
      }
    }
  }
  if constexpr (gmp) {
    // MixtureSameFamily.log_prob (vae.py:240-244,181): the 32 lanes of a row split the latent dimensions, every component's
    // squared distance is a 32-lane reduction; K-way logsumexp in lanes 0..15; responsibilities kept for the backward pass
    const int row = tid >> 5, sub = tid & 31;
    float zl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) zl[i] = sub + 32 * i < L ? P_z[row * V::ldz + sub + 32 * i] : 0.f;
    float comp = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int l = sub + 32 * i;
        if (l < L) {
          const float t = (zl[i] - M_loc[k * V::ldM + l]) * M_inv[k * V::ldM + l];
          acc += t * t;
        }
      }
      acc = row32_sum(acc);
      if (sub == k) comp = M_c[k] - 0.5f * acc;
    }
    const float mx = row32_max(comp);
    const float se = row32_sum(sub < K ? fexp(comp - mx) : 0.f);
    const float lse = mx + flog(se);
    if (sub < 16) P_r[row * 16 + sub] = sub < K ? fexp(comp - lse) : 0.f;
    if (sub == 0) {
      nllp[kPanel + row] = lse;
      if (lead && row < nrow) st1o(a.logp + r0 + row, lse);
    }
  }
  // S6 decoder hidden: 4 tiles, contraction LP
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<LP / 16, 64>(img + V::Wd0f, wave, P_z, V::ldz, 0, ln, lk, acc);
    const int c0 = wave * 16 + 4 * lk;
    const float4 bb = ld4(img + V::b_d0 + c0);
    const float4 h = make_float4(fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
    st4(P_hd + ln * V::ld64 + c0, h);
    if (act && ln < nrow) st4o(a.hd1 + (long long)(r0 + ln) * H + c0, h);
  }
  __syncthreads();                                 // the forward weights are dead (the mixture prior's arrays lie behind them)
  GMVAE_STAMP(4);
  if (lead) {                                      // the backward image lands while the decoder layer runs
#pragma unroll
    for (int j = 0; j < V::imgB_pieces; ++j) {
      const int c = min(wave * 256 + j * 2048, V::imgB - 256);
      __builtin_amdgcn_global_load_lds(a.img2b + c + lane * 4, img + c, 16, 0, 0);
    }
  }

  // ======================================================================= D: decoder output layer, in registers (one tile per wave)
  f32x4 dacc[4];
#pragma unroll
  for (int ht = 0; ht < 4; ++ht) dacc[ht] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  float4 gkeep = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has) {
    float4 hb[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) hb[kt] = ld4(P_hd + ln * V::ld64 + kt * 16 + 4 * lk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt].x, hb[kt].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt].y, hb[kt].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt].z, hb[kt].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt].w, hb[kt].w, acc, 0, 0, 0);
    }
    // lane = (row ln, columns 16 dtile + 4 lk + r): Bernoulli term and g = sigmoid(lambda) - x, ONE exp, rcp and log each
    const float bv[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
    const bool ok = ln < nrow;
    float g[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lam = acc[r] + bv[r] + a.gen_bias;
      const float e = fexp(-fabsf(lam));
      const float rcp = __builtin_amdgcn_rcpf(1.f + e);
      const float sp = fmaxf(lam, 0.f) - flog(rcp);
      const float sgm = lam >= 0.f ? rcp : e * rcp;
      const float x_ = ok ? (float)((xb >> (8 * r)) & 0xffu) : 0.f;
      rs += ok ? x_ * lam - sp : 0.f;
      g[r] = ok ? sgm - x_ : 0.f;
    }
    // the lead's waves 0..4 keep g in registers until their hand-off polls are through (vmcnt retires in order)
    if (lead && wave < 5) gkeep = make_float4(g[0], g[1], g[2], g[3]);
    else if (ok) st4o(a.g + (long long)(r0 + ln) * D + dtile * 16 + 4 * lk, make_float4(g[0], g[1], g[2], g[3]));
    // dhd1 += g Wd1^T over this tile's 16 columns: the accumulator layout IS the B operand
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
      dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ht].x, g[0], dacc[ht], 0, 0, 0);
      dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ht].y, g[1], dacc[ht], 0, 0, 0);
      dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ht].z, g[2], dacc[ht], 0, 0, 0);
      dacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ht].w, g[3], dacc[ht], 0, 0, 0);
    }
  }
  GMVAE_STAMP(5);
  rs += __shfl_xor(rs, 16, 64);
  rs += __shfl_xor(rs, 32, 64);
  if (lk == 0) rsum[wave * kPanel + ln] = rs;
#pragma unroll
  for (int ht = 0; ht < 4; ++ht) st4(dred + wave * 1024 + (((ht * 4 + lk) * 16 + ln) << 2), f4(dacc[ht]));
  __syncthreads();
  constexpr int ngr = kPanel * H + kPanel;         // granules one producer publishes: dhd1 partials + row sums
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
    if (tid < 256) {
      granule2_publish(xo + ((wave * 2 + 0) * 64 + lane) * 2, epoch, dsum.x, dsum.y);
      granule2_publish(xo + ((wave * 2 + 1) * 64 + lane) * 2, epoch, dsum.z, dsum.w);
    } else if (tid < 256 + kPanel) {
      granule_publish(xo + kPanel * H + (tid - 256), ((unsigned long long)epoch << 32) | __float_as_uint(rsn));
    }
    if (!FUSE && bid == 0 && tid == 0 && a.lr_t_out) {      // (a producer: off the launch's critical path)
      const double t = (double)(a.step_dev[0] + 1ull);
      *a.lr_t_out = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, t)) / (1.0 - pow((double)a.b1, t)));
    }
    M2V_SPAN_END();
    return 1;
  }
  // ======================================================================= B: backward chain (quarter 0)
  {
    const unsigned long long* xi = a.xchg + (long long)pnl * (Q - 1) * ngr;
    const bool wd = tid < 256, wn = tid >= 256 && tid < 256 + kPanel;
    unsigned spins = 0;
    if (wave < 4) {
      u32x4_t v[2 * (Q - 1)];
      const unsigned long long* const xs = xi + ((wave * 2) * 64 + lane) * 2;
      for (;;) {
#pragma unroll
        for (int pq = 0; pq < Q - 1; ++pq) {
          v[2 * pq] = granule2_load(xs + (long long)pq * ngr);
          v[2 * pq + 1] = granule2_load(xs + (long long)pq * ngr + 128);
        }
        g2_wait_n(v);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 2 * (Q - 1); ++i) ok = ok && v[i][1] == epoch && v[i][3] == epoch;
        if (__all(ok)) break;
        if (++spins > spin_limit) {                          // a producer never ran; flag and go on
          if (lane == 0) atomicExch(a.err_word, 1u);
#pragma unroll
          for (int i = 0; i < 2 * (Q - 1); ++i) { v[i][0] = 0x7fc00000u; v[i][2] = 0x7fc00000u; }
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
        for (int pq = 0; pq < Q - 1; ++pq) v[pq] = __hip_atomic_load(xs + (long long)pq * ngr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    if (wave < 5 && ln < nrow)                     // now these waves' g = sigmoid(lambda) - x
      st4o(a.g + (long long)(r0 + ln) * D + dtile * 16 + 4 * lk, gkeep);
    if (wd) {                                      // masked top gradient (+ saved for dWd0)
      const float4 hd = ld4(P_hd + orow * V::ld64 + ocol);
      const bool ok = orow < nrow;
      const float4 d = make_float4((ok && hd.x > 0.f) ? dsum.x : 0.f, (ok && hd.y > 0.f) ? dsum.y : 0.f,
                                   (ok && hd.z > 0.f) ? dsum.z : 0.f, (ok && hd.w > 0.f) ? dsum.w : 0.f);
      st4(P_dhd + orow * V::ld64 + ocol, d);
      if (ok) st4o(a.dhd1 + (long long)(r0 + orow) * H + ocol, d);
    }
    if (wn) {
      const int row = tid - 256;
      if (row < nrow) {
        st1o(a.logpx + r0 + row, rsn);
        st1o(a.logw + r0 + row, rsn + nllp[kPanel + row] - nllp[row]);
      }
    }
  }
  // (the backward image: waves 0..4 polled -- vmcnt retires in order -- waves 5..7 wait for all but their last store)
  if (wave == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (no decoder tile, no g store: its image pieces are its last requests)
  else if (wave > 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  __syncthreads();
  GMVAE_STAMP(6);
  float *P_dz = sm + V::P_dz, *P_dqp = sm + V::P_dqp;
  // B1 dz_dec = dhd1 * Wd0^T: LP / 16 tiles, contraction 64
  if (wave < LP / 16) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = m2_tile<4, LP>(img + V::Wd0b, wave, P_dhd, V::ld64, 0, ln, lk, acc);
    st4(P_dz + ln * V::ldz + wave * 16 + 4 * lk, f4(acc));
  }
  __syncthreads();
  // B2 q head backward + the prior's share of dz: 32 lanes per row; inputs prepared by S5
  {
    const int row = tid >> 5, sub = tid & 31;
    const bool ok = row < nrow;
#pragma unroll
    for (int l = sub; l < L2P / 2; l += 32) {       // (latent slots up to the padded head width: padding written as zeros)
      float dmu = 0.f, draw = 0.f;
      if (ok && l < L) {
        const float sg = P_qp[row * V::ldq + L + l];
        const float zz = P_z[row * V::ldz + l];
        float pterm;
        if constexpr (gmp) {                       // sum_k r_k (z - loc_k) / s_k^2
          pterm = 0.f;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float iv = M_inv[k * V::ldM + l];
            pterm += P_r[row * 16 + k] * (zz - M_loc[k * V::ldM + l]) * iv * iv;
          }
        } else {
          pterm = zz;                              // d(-log N(z; 0, 1)) / dz
        }
        dmu = P_dz[row * V::ldz + l] + pterm;
        const float dsg = dmu * P_eps[row * LP + l] - __builtin_amdgcn_rcpf(sg);
        draw = (sg > a.smin) ? dsg * P_qp[row * V::ldq + l] : 0.f;
      }
      if (l < L) { P_dqp[row * V::ldq + l] = dmu; P_dqp[row * V::ldq + L + l] = draw; }
    }
    // zero the head's padding columns [2L, L2P) (they meet zero weight rows, but must not be NaN)
    for (int c = L2 + sub; c < L2P; c += 32) P_dqp[row * V::ldq + c] = 0.f;
  }
  __syncthreads();
  {                                                // dqp (kept for dWe1)
    if constexpr ((L2 & 3) == 0 && L2 >= 128) {
      const int row = tid >> 5, c = (tid & 31) << 2;
      if (row < nrow && c < L2) st4o(a.dqp + (long long)(r0 + row) * L2 + c, ld4(P_dqp + row * V::ldq + c));
    } else {
      const int row = tid >> 5, c = tid & 31;
      if (row < nrow && c < L2) st1o(a.dqp + (long long)(r0 + row) * L2 + c, P_dqp[row * V::ldq + c]);
    }
  }
  // B3 dhe = (dqp * We1^T) [he > 0]: 4 tiles, contraction L2P (split in two over the wave halves where it is 128)
  {
    constexpr int KT2 = L2P / 16;                  // contraction tiles
    constexpr bool split = KT2 >= 8;
    const int t = wave & 3, kh = wave >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (split) {
      acc = m2_tile<KT2 / 2, 64>(img + V::Wg1b, t, P_dqp, V::ldq, (KT2 / 2) * kh, ln, lk, acc);
      if (kh) st4(red + t * 256 + ((lk * 16 + ln) << 2), f4(acc));
    } else if (!kh) {
      acc = m2_tile<KT2, 64>(img + V::Wg1b, t, P_dqp, V::ldq, 0, ln, lk, acc);
    }
    __syncthreads();
    if (!kh) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (split) o = ld4(red + t * 256 + ((lk * 16 + ln) << 2));
      const int c0 = t * 16 + 4 * lk;
      const float4 he = ld4(P_h1 + ln * V::ld64 + c0);
      const bool ok = ln < nrow;
      const float4 d = make_float4((ok && he.x > 0.f) ? acc[0] + o.x : 0.f, (ok && he.y > 0.f) ? acc[1] + o.y : 0.f,
                                   (ok && he.z > 0.f) ? acc[2] + o.z : 0.f, (ok && he.w > 0.f) ? acc[3] + o.w : 0.f);
      if (ok) st4o(a.dhg1 + (long long)(r0 + ln) * H + c0, d);
    }
  }
  if constexpr (gmp) {
    // gradients of the learned mixture prior over this panel's rows (A12): one partial per panel
    constexpr int KL = K * L, KLp = (KL + 3) & ~3;
    float* out = a.gmp_part + (long long)pnl * (2 * KLp + ((K + 3) & ~3));
    for (int i = tid; i < KL; i += kMT) {
      const int k = i / L, l = i - k * L;
      const float iv = M_inv[k * V::ldM + l], lc = M_loc[k * V::ldM + l];
      float ga = 0.f, gb = 0.f;
      float wr[kPanel], zz[kPanel];
#pragma unroll
      for (int row = 0; row < kPanel; ++row) { wr[row] = P_r[row * 16 + k]; zz[row] = P_z[row * V::ldz + l]; }
#pragma unroll
      for (int row = 0; row < kPanel; ++row) {
        const float w_ = row < nrow ? wr[row] : 0.f;
        const float t = row < nrow ? (zz[row] - lc) * iv : 0.f;
        ga -= w_ * t * iv;
        gb += w_ * (1.f - t * t) * iv;
      }
      st1o(out + i, ga);
      st1o(out + KLp + i, gb * sigmoidf_(M_raw[k * V::ldM + l]));
    }
    if (tid < K) {
      float ga = 0.f;
#pragma unroll
      for (int row = 0; row < kPanel; ++row) ga -= row < nrow ? P_r[row * 16 + tid] - M_w[tid] : 0.f;
      st1o(out + 2 * KLp + tid, ga);
    }
  }
  GMVAE_STAMP(7);
  M2V_SPAN_END();
#undef M2V_SPAN_END
  return 2;
}

template <int MODEL, int LT, int KT>
__global__ __launch_bounds__(kMT) void mega2v_fwd_bwd(const MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  mega2v_body<MODEL, LT, KT, 0>(a, sm);
}

}  // namespace gmvae
