// mega3_step: the WHOLE steady-state training step of the GMVAE at the reference's default sizes in ONE launch
// (scripts/runners.py:231-232 is one sess.run: forward, gradients, apply_gradients) -- mega2_fwd_bwd's per-row part
// (mega2.hpp: first layer, forward chain, decoder layer, backward chain) followed, IN THE SAME WORKGROUPS, by dw_adam's
// weight-gradient tiles with TF-Adam in their epilogue (dwadam.hpp; scripts/runners.py:181-183).
//
// Why: as two launches the step paid two kernel boundaries (~2 us each: dispatch + end-of-kernel write-back), dw_adam's cold
// start (kernel arguments, first loads: 1-3 us) and left the 192 producer workgroups of mega2_fwd_bwd idle for the ~6 us the
// 64 leads spend on the hand-off and the backward chain.  Here a workgroup that is done with its per-row role becomes a
// WORKER: it takes tile slots rank, rank + workers, ... of the step's tile list (240 tiles + the loss tail at B = 1024: one
// slot per workgroup), requests what does not depend on this launch (the tile's parameters and Adam moments, the uint8 batch
// operand) and then waits for the leads.
//
// Hand-off.  Everything a workgroup leaves for the tiles is stored write-through (st4o / st1o: sc1).  A workgroup ends its
// per-row role with: s_waitcnt vmcnt(0) in every wave (its stores are acknowledged by the memory side: 0.3-1.1 us measured) ->
// workgroup barrier -> ONE thread stores this step's epoch into the workgroup's flag (4 bytes, write-through):
//   rows 0..2 of the flag table: the producers (quarters 1..3) -- their decoder tiles' g; quarter 1 also stores the panel's
//                                forward activations (hy1, y, hg1, z, hd1: every quarter holds the same bits)
//   row 3:                       the lead -- g of its one decoder tile, dhd1, dqp, dpp, dhg1, dlogits, dhy1, per-row loss terms
// Tiles come in two phases: P needs the producers' flags only -- the decoder output layer's weight gradient (hd1^T g: 98
// tiles, the step's longest) over the 48 column tiles the producers own (kernels.hpp M2_LEAD_TILES = 1) -- and runs while the
// leads are still in their hand-off and backward chain; F needs the leads' flags too.  A worker polls the flag rows of its
// tile's phase (agent-scope loads, wave 0) until every panel's carry the epoch.  The flags are epoch tags, not counters:
// nothing resets them, and a stale flag can never equal the current epoch.  Polls are bounded (a timeout sets the error word:
// the step is not applied and the host falls back to the safe schedule, exactly as for the granule hand-offs).
// (Measured and dropped, profiles/round5_notes.md: flags in the middle of the backward chain -- behind counted vmcnt waits
// they cost the leads 0.6-0.9 us and arrive no earlier than the final flag + 1 us; polls every 0.1 us by 190 workgroups slowed
// the slowest lead by 1.5 us.)
//
// Operand loads behind a flag are PLAIN loads, served by the XCD's L2.  (Measured, tools/m3stamps.py: agent-scope sc1 loads
// of the operands -- 30 MB per step that no L2 may serve -- ran the fp32 tiles' contraction at the fabric's rate, 15 us against
// 4; an agent-scope acquire fence, buffer_inv sc1, in every worker cost 7 us and emptied the L2s.)  Why plain loads cannot
// see stale bytes here -- the hardware contract this kernel relies on, the same one every pair of dependent launches relies on:
//   1. the launch starts with the L2s invalidated (the dispatch's acquire), so a line of a hand-off buffer enters an L2 only by
//      a load or a store of THIS launch;
//   2. no workgroup loads from a hand-off buffer before it has seen the flags of the buffer's phase, and by then every byte of
//      the buffer is in memory (stores acknowledged before the flag);
//   3. buffers of different phases share no 128-byte line (the workspace carves 256-byte units); a panel's rows of one buffer
//      are written by that panel's workgroups only, and those run on ONE XCD when the panel count is a multiple of 8 (workgroups
//      are dealt round-robin over the XCDs; tools/handoff_clock.py prints HW_REG_XCC_ID per workgroup: 64 of 64 panels);
//   4. a write-through store of part of a line does not fetch the rest of the line into the writer's L2 (byte-masked write
//      allocation), so the writer's L2 holds no stale copy of bytes other workgroups write later.
// The per-row loss terms (two panels share a line) and the flags are read with agent-scope loads.
//
// What may be overwritten when: the optimizer epilogue rewrites parameters and operand images that THIS launch read -- but
// every such read (first-layer weights, the three operand images) was waited for before the reader's hand-off, i.e. happens
// before the last lead's flag, and no update happens before a worker has seen all flags.
#pragma once
#include "dwadam.hpp"
#include "mega2.hpp"
#include "mega2v.hpp"

namespace gmvae {

constexpr int kM3MaxSlots = 320;
constexpr int kM3MaxT = 8;
constexpr unsigned short kM3Tail = 0xffffu, kM3None = 0xfffeu, kM3PhaseF = 0x8000u, kM3Gmp = 0xff00u;      // kM3Gmp + i: block i of the mixture prior's variables
constexpr int kM3FlagLd = 64;                        // rows 0..2: producers q - 1; row 3: the leads
#ifndef M3_FLAG_REPLICAS
#define M3_FLAG_REPLICAS 8
#endif
// Every flag is stored M3_FLAG_REPLICAS times, 4 KB apart (one wave-level store), and a worker polls replica rank % replicas:
// ~200 workgroups polling the SAME two lines made them a hot spot of one memory channel (polls returned after 2-4 us instead of
// 1.2, and the slowest lead -- whose stores cross the same channel -- lost 1-1.5 us; profiles/round5_notes.md)
constexpr int kM3FlagReplicas = M3_FLAG_REPLICAS, kM3FlagRepLd = 1024;

struct M3Fin {                 // what the tiles' optimizer epilogue and the loss tail need (FinalArgs without slabs / maps)
  float *grads, *p, *m, *v;
  float lr, b1, b2, eps;
  int do_adam;                 // 0: gradients only (data parallel: the all-reduce and adam_tf_img follow)
  float count;
  const float *logw, *logpx, *logq, *logp, *nent;
  float* tail;
  int B;
  float* tail_log;
  float* img[kImgBufs];
  unsigned* epoch_word;        // bumped by the tail slot for the next step's hand-offs (null: a later launch does it)
  // VAE_GMP (mega3v_step): the learned mixture prior's variables have no matrix-product gradient -- the leads leave one partial
  // per panel (gmp_part); "gmp" slots sum them in panel order, apply the update and write the variables' LDS-image copies
  const float* gmp_part;
  int gmp_n, gmp_len;
  long long gmp_off;
};

struct M3Args {
  MegaArgs m;
  int ntens, total_slots;
  unsigned* flags;             // [kM3FlagReplicas][kM3FlagRepLd]: in each replica [4][kM3FlagLd] per-workgroup epoch tags
  unsigned* lr_next;           // [4]: {alpha_t bits, the Adam step t it is for, lr bits, alpha_key(b1, b2)}: left by the previous step's tail slot
  unsigned long long* dbg;     // diagnostic: [workgroup][8] wall-clock stamps of the worker phase (tools/m3stamps.py) or null
  int fault_pnl;               // test hook (GMVAE_DEBUG_LEAD_FAULT=<panel>, -1: none): that panel's lead sets the error word in front of
                               // its flag -- a hand-off that gave up DURING the launch, after phase P has started
  unsigned short perm[kM3MaxSlots];   // slot -> (tensor << 10) | tile inside the tensor (| kM3PhaseF); kM3Tail: the loss tail
  DwTensor t[kM3MaxT];
  M3Fin fa;
  int gmp_nmap;
  ImgMap gmp_map[3];
};
static_assert(sizeof(M3Args) <= 4096, "kernel arguments");

__device__ __forceinline__ float ld_sc(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// alpha_t = lr sqrt(1 - b2^t) / (1 - b1^t) (fp64, adam_tf's form)
__device__ __forceinline__ float m3_alpha(const M3Fin& a, const unsigned long long t) {
  return (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t)));
}

// the loss tail (kernels.hpp finalize_tail_block) over per-row terms the leads of THIS launch stored: agent-scope loads
// `poisoned`: a hand-off of this launch gave up waiting (the error word is set) -- the per-row terms may be a previous step's:
// the loss leaves as NaN, exactly what finalize_tail_block's inputs were in the two-launch form, so that the host sees the
// step as not applied (gmvae_amd/runners.py) and the data-parallel optimizer behind the all-reduce skips it on every rank
__device__ __forceinline__ void m3_tail(const M3Fin& a, unsigned long long* step_dev, const unsigned long long step, unsigned* lr_next,
                                        float* red, const bool poisoned) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int t = threadIdx.x;
  if (t < 256) {
    for (int b = t; b < a.B; b += 256) {
      a0 -= ld_sc(a.logw + b);
      a1 -= ld_sc(a.logpx + b);
      a2 += ld_sc(a.logq + b) - ld_sc(a.logp + b);
      a3 += a.nent ? ld_sc(a.nent + b) : 0.f;
    }
    red[t] = a0; red[256 + t] = a1; red[512 + t] = a2; red[768 + t] = a3;
  }
  if (t == 256) {
    // alpha_t for whoever applies the next update, tagged with its Adam step t (never 0, the workspace's initial value): the
    // NEXT step's tiles (single device), or -- gradients only -- THIS step's adam_tf_img behind the all-reduce
    unsigned long long tt = a.do_adam ? step + 2ull : step + 1ull;
    // (opaque to the optimizer: the expression is invariant over the slot loop, and hoisted in front of it the two fp64 pow cost
    //  EVERY workgroup 1.44 us before its first tile -- on the launch's critical path for the leads; tools/m3stamps.py)
    unsigned tt_lo = (unsigned)tt;
    asm volatile("" : "+v"(tt_lo));
    tt = (tt & ~0xffffffffull) | tt_lo;
    lr_next[0] = __float_as_uint(m3_alpha(a, tt));
    lr_next[1] = (unsigned)tt;
    lr_next[2] = __float_as_uint(a.lr);          // (a step counter may be rewound and the rate changed: the tag names both)
    lr_next[3] = alpha_key(a.b1, a.b2);
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o)
      for (int j = 0; j < 4; ++j) red[j * 256 + t] += red[j * 256 + t + o];
    __syncthreads();
  }
  if (t == 0) {
    if (poisoned) red[0] = red[256] = red[512] = red[768] = __builtin_nanf("");
    a.tail[0] = red[0]; a.tail[1] = red[256]; a.tail[2] = red[512]; a.tail[3] = red[768];
    a.tail[4] = (float)a.B; a.tail[5] = 0.f; a.tail[6] = 0.f; a.tail[7] = 0.f;
    if (a.tail_log) {
      a.tail_log[0] = red[0]; a.tail_log[1] = red[256]; a.tail_log[2] = red[512]; a.tail_log[3] = red[768];
      a.tail_log[4] = (float)a.B; a.tail_log[5] = 0.f; a.tail_log[6] = 0.f; a.tail_log[7] = 0.f;
    }
    // every workgroup read the counter and the epoch at its start, long before any lead's flag
    if (step_dev) { step_dev[1] = step; step_dev[0] = step + 1ull; }
    if (a.epoch_word) *a.epoch_word += 1u;
  }
}

// The worker phase of a one-launch step (mega3_step, mega3v_step): the calling workgroup -- `role` 1 a producer, 2 a lead, 3 a
// workgroup that had no per-row role -- stores its flag (NQ producer rows of the flag table, then the leads' row), takes the
// slots rank, rank + nW, ... of the tile list and runs them.
template <int NQ>
__device__ __forceinline__ void m3_worker_phase(const M3Args& aa, float* const sm, const int role, const int q, const int pnl,
                                                const int rank, const int nW, const unsigned epoch, const unsigned long long step,
                                                const unsigned spin_limit, const unsigned lr_bits, const bool lr_hit) {
  const MegaArgs& a = aa.m;
  const M3Fin& fa = aa.fa;
#define M3_END() if (a.span && threadIdx.x == 0) a.span[2 * blockIdx.x + 1] = wall_clock64()
#define M3_ST(i) if (aa.dbg && threadIdx.x == 0) aa.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64()
#define M3_ST2(i) if (aa.dbg && threadIdx.x == 0) aa.dbg[(size_t)(blockIdx.x + 256) * 8 + (i)] = wall_clock64()
  if (role == 0) { M3_END(); return; }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int B = a.B, nPr = (B + kPanel - 1) / kPanel;
  M3_ST(0);
  if (role != 3) {
    if (role == 2 && pnl == aa.fault_pnl && tid == 0) atomicExch(a.err_word, 1u);      // (vmcnt counts it: acknowledged before the flag)
    // the role's stores are acknowledged: the workgroup's flag goes out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    M3_ST2(0);
    __syncthreads();           // (also: the per-row role's LDS is dead in every wave)
    M3_ST2(1);
    if (tid < kM3FlagReplicas)
      __hip_atomic_store(aa.flags + tid * kM3FlagRepLd + (role == 2 ? NQ : q - 1) * kM3FlagLd + pnl, epoch, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  float* const red = sm;                         // [wave][mt * 4 + r][lane]
  float* const redcs = sm + kDwWaves * 64 * 16;  // [wave][lane]
  const bool upd = fa.do_adam != 0;
  // alpha_t: the previous step's tail slot left it (tagged with this step's index); the first step of a launch computes it
  float lr_t = __uint_as_float(lr_bits);
  if (upd && !lr_hit) lr_t = m3_alpha(fa, step + 1ull);
  M3_ST2(2);
  const float omb1 = 1.f - fa.b1, omb2 = 1.f - fa.b2, gs = 1.f / fa.count;
  unsigned seen = 0;                             // (uniform) bit 0: the producers' flags seen; bit 1: the leads'
  bool poisoned = false;
  // wait for the producers' rows (leads = false) or for all four rows
  const unsigned* const f0 = aa.flags + (rank % kM3FlagReplicas) * kM3FlagRepLd + min(lane, nPr - 1);
  const bool own = role == 2 && lane == pnl;       // a lead's own flag: it stored it itself (its visibility to itself is not waited for)
  // An EARLY poll (wave 0): the flag loads go out BEFORE the tile's operand prefetch -- vmcnt retires in order, so a poll issued
  // behind 44 cold prefetch loads returned with the last of them (a lead's first poll: 2.1 us instead of 1.2) -- and are looked at
  // after the prefetch has been issued.  early_ok: all four rows carried the epoch (then wait_flags below does not poll again).
  unsigned ef[NQ + 1];
#pragma unroll
  for (int r = 0; r <= NQ; ++r) ef[r] = 0u;
  auto early_issue = [&]() {
    if (wave == 0 && seen != 3u) {
#pragma unroll
      for (int r = 0; r <= NQ; ++r) ef[r] = __hip_atomic_load(f0 + r * kM3FlagLd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto early_ok = [&]() -> bool {                  // (meaningful in wave 0 only)
    bool ok = ef[NQ] == epoch || own;
#pragma unroll
    for (int r = 0; r < NQ; ++r) ok = ok && ef[r] == epoch;
    return wave == 0 && seen != 3u && __all(ok);
  };
  auto wait_flags = [&](const bool leads, const bool pre = false) {
    const unsigned want = leads ? 3u : 1u;
    if ((seen & want) == want) return;
    if (wave == 0 && !pre) {
      const bool np = !(seen & 1u);                // (uniform) the producers' rows are still to be seen
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
        if (np) {                                  // (all of the rows' loads in flight together)
          unsigned fr[NQ];
#pragma unroll
          for (int r = 0; r < NQ; ++r) fr[r] = __hip_atomic_load(f0 + r * kM3FlagLd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int r = 0; r < NQ; ++r) ok = ok && fr[r] == epoch;
        }
        if (leads) ok = ok && (own || __hip_atomic_load(f0 + NQ * kM3FlagLd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch);
        if (__all(ok)) break;
        if (++spins > spin_limit) {
          if (lane == 0) atomicExch(a.err_word, 1u);
          break;
        }
        if (aa.dbg && lane == 0) aa.dbg[(size_t)(blockIdx.x + 256) * 8 + 4] = wall_clock64();     // the last failed poll
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (aa.dbg && tid == 0) { aa.dbg[(size_t)(blockIdx.x + 256) * 8 + 3] = wall_clock64(); aa.dbg[(size_t)(blockIdx.x + 256) * 8 + 5] = (unsigned long long)leads; }
    __syncthreads();
    poisoned = __hip_atomic_load(a.err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    seen |= want;
  };
  for (int slot = rank; slot < aa.total_slots; slot += nW) {
    M3_ST2(6);
    const int pv_ = aa.perm[slot];
    if (pv_ == kM3None) continue;
    if (pv_ == kM3Tail) {
      wait_flags(true);
      M3_ST(2);
      m3_tail(fa, a.step_dev, step, aa.lr_next, red, poisoned);
      M3_ST(5);
      if (aa.dbg && tid == 0) aa.dbg[(size_t)blockIdx.x * 8 + 6] = 99ull;
      __syncthreads();
      continue;
    }
    if (pv_ >= kM3Gmp) {                           // the mixture prior's variables: partials -> gradient -> TF-Adam -> image
      wait_flags(true);
      const int e = (pv_ - kM3Gmp) * kMT + tid;
      if (e < fa.gmp_len) {
        const long long i = fa.gmp_off + e;
        float g = 0.f;
        for (int k = 0; k < fa.gmp_n; ++k) g += ld_sc(fa.gmp_part + (long long)k * fa.gmp_len + e);      // (panels share lines)
        fa.grads[i] = g;
        if (upd && !poisoned) {
          float pp = fa.p[i], pm = fa.m[i], pv = fa.v[i];
          adam_update(pp, pm, pv, g, gs, lr_t, omb1, omb2, fa.eps);
          fa.p[i] = pp; fa.m[i] = pm; fa.v[i] = pv;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < aa.gmp_nmap && i >= aa.gmp_map[k].begin && i < aa.gmp_map[k].end) {
              const ImgMap& mp = aa.gmp_map[k];
              const unsigned off = (unsigned)(i - mp.begin);
              const int r = (int)(((unsigned long long)off * mp.magic) >> 32);
              const int c = (int)off - r * mp.cols;
              fa.img[mp.which][img_dst(mp.kind, mp.base, mp.ld, mp.chunk, r, c)] = pp;
            }
        }
      }
      if (aa.dbg && tid == 0) aa.dbg[(size_t)blockIdx.x * 8 + 6] = 98ull;
      continue;
    }
    const bool phF = (pv_ & kM3PhaseF) != 0;
    const int ti = (pv_ & 0x7fff) >> 10, tl = pv_ & 1023;
    const DwTensor& T = aa.t[ti];
    const int M = T.M, N = T.N, lda = T.lda, ldy = T.ldy;
    if (aa.dbg) { asm volatile("" ::"s"(M), "s"(N), "s"(lda), "s"(ldy)); M3_ST2(7); }
    const int tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
    const int MUr = T.mu;
    const int m0 = tm * 16 * MUr, n0 = tn * 16;
    // ---- epilogue owners (dwadam.hpp): unit (lane slot el, eu) = dW[mb .. mb+3][en]
    const int el = tid & 63, eu = (tid >> 6) & 3;
    const int mb = m0 + 4 * MUr * (el >> 4) + 4 * eu, en = n0 + (el & 15);
    const bool eown = tid < 64 * MUr && mb < M && en < N;
    const bool bown = tm == 0 && T.b_off >= 0 && tid >= 256 && tid < 272 && n0 + (tid - 256) < N;
    float pp[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f};
    float bp = 0.f, bm = 0.f, bv = 0.f;
    const bool early = phF && T.a_u8 != 0;         // (uniform) tiles whose whole prefetch needs no flag: poll first, prefetch, look
    if (early) early_issue();
    // the optimizer's operands: the previous launch wrote them -- requested before any wait
    if (upd) {
      if (eown) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int mm = min(mb + j, M - 1);
          const long long i = (long long)T.w_off + (long long)mm * N + en;
          pp[j] = fa.p[i]; pm[j] = fa.m[i]; pv[j] = fa.v[i];
        }
      }
      if (bown) { const int i = T.b_off + n0 + tid - 256; bp = fa.p[i]; bm = fa.m[i]; bv = fa.v[i]; }
    }
    const int rows_w = (((B + kDwWaves - 1) / kDwWaves) + 3) & ~3;
    const int b_lo = wave * rows_w, b_hi = min(B, b_lo + rows_w);
    // The register-resident forms need 128 rows in EVERY wave: the choice must be workgroup-uniform, because the branches below
    // wait a different number of times and every wait holds a workgroup barrier.  (b_hi - b_lo == 128 alone is true for waves
    // 0..6 and false for wave 7 at B = 1000: wave 7 then passed ITS one barrier together with the others' first and read the
    // gradients before the leads' flags were checked -- a rare one-ulp-scale difference tools/fuse_check.py caught once in five runs.)
    // The uint8 tiles keep dw_adam's PER-WAVE choice (a wave with 128 rows multiplies bf16 pieces, a ragged last wave fp32:
    // same bits as the two-launch form at any batch) -- both of their branches wait exactly once.
    const bool full = B == 128 * kDwWaves, full_w = b_hi - b_lo == 128;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs = 0.f;
    M3_ST(1);
    if (T.a_u8 && full_w) {
      unsigned av[32];
      dw_u8x3_load_a(static_cast<const unsigned char*>(T.A), lda, M, m0, b_lo, ln, lk, av);     // the batch: final before the launch
      wait_flags(phF, early && early_ok());
      M3_ST(2);
      dw_u8x3_rest<false>(av, T.dY, ldy, M, N, m0, n0, b_lo, ln, lk, acc, cs);
    } else if (!T.a_u8 && full && phF && MUr <= 2) {
      // a forward activation times a pre-activation gradient: the activation (quarter 1 stored it) behind the producers'
      // flags, the gradient behind the leads'
      wait_flags(false);
      if (MUr == 2) {
        float2 av[32];
        dw_f32_load_a<2>(static_cast<const float*>(T.A), lda, M, m0, b_lo, ln, lk, av);
        wait_flags(true);
        M3_ST(2);
        // (the lead's g columns share 128-byte lines with producers' columns that phase P has already read: not from L2)
        if (T.dY == a.g) dw_f32_rest<2, true>(av, T.dY, ldy, M, N, m0, n0, b_lo, ln, lk, acc, cs);
        else dw_f32_rest<2>(av, T.dY, ldy, M, N, m0, n0, b_lo, ln, lk, acc, cs);
      } else {
        float av[32];
        dw_f32_load_a<1>(static_cast<const float*>(T.A), lda, M, m0, b_lo, ln, lk, av);
        wait_flags(true);
        M3_ST(2);
        dw_f32_rest<1>(av, T.dY, ldy, M, N, m0, n0, b_lo, ln, lk, acc, cs);
      }
    } else {
      wait_flags(phF);
      M3_ST(2);
      if (rows_w <= 32) {                          // small batch: 8 k-steps per batch of loads
        if (T.a_u8) dw_contract<true, 4, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
        else if (MUr == 2 && phF && T.dY == a.g) dw_contract<false, 2, 8, true>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
        else if (MUr == 2) dw_contract<false, 2, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
        else dw_contract<false, 1, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
      }
      else if (T.a_u8) dw_contract<true, 4>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
      else if (MUr == 2 && phF && T.dY == a.g) dw_contract<false, 2, 32, true>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);   // (the lead's g columns: not from L2, see above)
      else if (MUr == 2) dw_contract<false, 2>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
      else dw_contract<false, 1>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
    }
    M3_ST(3);
    // ---- the waves' partial tiles meet in LDS (fixed order: bit-reproducible)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * t + r) * 64 + lane] = acc[t][r];
    redcs[wave * 64 + lane] = cs;
    __syncthreads();
    M3_ST(4);
    // All or nothing: a phase-P tile contracted behind the producers' flags only, but no update may be applied before EVERY
    // hand-off of the launch is known to have arrived -- a lead that gives up later poisons the step, and the tiles that had
    // already applied theirs would be applied a second time when the host runs the step again.  (The leads' flags are long
    // there when a phase-P contraction ends: one poll.)
    if (upd && !phF) wait_flags(true);
    if (eown) {
      float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < kDwWaves; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = 4 * eu + j, r = MUr == 4 ? eu : (MUr == 2 ? o >> 1 : o), t = MUr == 4 ? j : (MUr == 2 ? o & 1 : 0);
          g[j] += red[(w * 16 + 4 * t + r) * 64 + el];
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (mb + j < M) {
          const long long i = (long long)T.w_off + (long long)(mb + j) * N + en;
          fa.grads[i] = g[j];
          if (upd && !poisoned) {
            adam_update(pp[j], pm[j], pv[j], g[j], gs, lr_t, omb1, omb2, fa.eps);
            fa.p[i] = pp[j]; fa.m[i] = pm[j]; fa.v[i] = pv[j];
          }
        }
      }
      if (upd && !poisoned) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (mb + j >= M) pp[j] = 0.f;              // rows past the tensor: the image's padding stays zero
        if (T.k1 == 2 || T.k1 == 4 || T.k1 == 7)
          *reinterpret_cast<float4*>(fa.img[T.which1] + img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb, en)) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        else if (T.k1 >= 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mb + j < M) fa.img[T.which1][img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb + j, en)] = pp[j];
        }
        if (T.k2 >= 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mb + j < M) fa.img[T.which2][img_dst(T.k2, T.base2, T.ld2, T.chunk2, mb + j, en)] = pp[j];
        }
      }
    }
    if (bown) {                                    // bias gradient = column sum of dY over the batch
      const int c = tid - 256;
      float g = 0.f;
#pragma unroll
      for (int w = 0; w < kDwWaves; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k) g += redcs[w * 64 + k * 16 + c];
      const int i = T.b_off + n0 + c;
      fa.grads[i] = g;
      if (upd && !poisoned) {
        adam_update(bp, bm, bv, g, gs, lr_t, omb1, omb2, fa.eps);
        fa.p[i] = bp; fa.m[i] = bm; fa.v[i] = bv;
        if (T.bk >= 0) fa.img[T.bwhich][img_dst(T.bk, T.bbase, 0, T.bchunk, 0, n0 + c)] = bp;
      }
    }
    M3_ST(5);
    if (aa.dbg && tid == 0) aa.dbg[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)ti;
    __syncthreads();                               // the meeting area is free for the next slot
  }
  M3_END();
#undef M3_END
#undef M3_ST
#undef M3_ST2
}


__global__ __launch_bounds__(kMT) void mega3_step(const M3Args aa) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const MegaArgs& a = aa.m;
  const M3Fin& fa = aa.fa;
  // what the worker phase needs of the launch's start state (read before anything can have changed it)
  const unsigned epoch = *a.epoch_word;
  const unsigned long long step = a.step_dev[0];
  const unsigned spin_limit = *a.err_word ? 0u : (1u << 19);
  const unsigned lr_bits = aa.lr_next[0];
  const bool lr_hit = aa.lr_next[1] == (unsigned)(a.step_dev[0] + 1ull) && aa.lr_next[2] == __float_as_uint(aa.fa.lr) &&
                      aa.lr_next[3] == alpha_key(aa.fa.b1, aa.fa.b2);
  if (aa.dbg && threadIdx.x == 0) aa.dbg[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
  // (measured and dropped: pulling the worker phase's kernel-argument lines into the scalar cache here -- the dependent scalar
  //  loads delay the first layer, the launch's critical path: 32.4 against 31.9 us)
  const int role = mega2_body<1>(a, sm);
  const int nPr_ = (a.B + kPanel - 1) / kPanel, nP_ = (nPr_ + 1) & ~1, bid_ = blockIdx.x;
  const int q_ = bid_ < nP_ * 3 ? 1 + bid_ / nP_ : 0, pnl_ = bid_ < nP_ * 3 ? bid_ % nP_ : bid_ - nP_ * 3;
  m3_worker_phase<3>(aa, sm, role, q_, pnl_, (q_ == 0 ? 3 * nPr_ : (q_ - 1) * nPr_) + pnl_, 4 * nPr_, epoch, step, spin_limit, lr_bits, lr_hit);
}

// mega3v_step: the same for the VAE family at small batches (mega2v.hpp: seven workgroups per panel; BASELINE configs[0] / [1]).
// The per-row part needs only 49 / 112 of the chip's 256 CUs: the grid is padded with workgroups that have NO per-row role
// and are workers from the launch's first cycle -- they request their tile's parameters, moments and batch operand at once and
// poll; the leads take no tile at all.  Flag table: rows 0..5 the producers, row 6 the leads.
template <int MODEL, int LT, int KT>
__global__ __launch_bounds__(kMT) void mega3v_step(const M3Args aa) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const MegaArgs& a = aa.m;
  const unsigned epoch = *a.epoch_word;
  const unsigned long long step = a.step_dev[0];
  const unsigned spin_limit = *a.err_word ? 0u : (1u << 19);
  const unsigned lr_bits = aa.lr_next[0];
  const bool lr_hit = aa.lr_next[1] == (unsigned)(a.step_dev[0] + 1ull) && aa.lr_next[2] == __float_as_uint(aa.fa.lr) &&
                      aa.lr_next[3] == alpha_key(aa.fa.b1, aa.fa.b2);
  if (aa.dbg && threadIdx.x == 0) aa.dbg[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
  const int nP = (a.B + kPanel - 1) / kPanel, bid = blockIdx.x;
  int role = 3, q = 0, pnl = 0, rank = bid;
  if (bid < 7 * nP) {
    role = mega2v_body<MODEL, LT, KT, 1>(a, sm);
    q = bid < nP * 6 ? 1 + bid / nP : 0;
    pnl = bid < nP * 6 ? bid % nP : bid - nP * 6;
    rank = (q == 0 ? 6 * nP : (q - 1) * nP) + pnl;
  } else if (a.span && threadIdx.x == 0) {
    a.span[2 * bid] = wall_clock64();
  }
  m3_worker_phase<6>(aa, sm, role, q, pnl, rank, (int)gridDim.x, epoch, step, spin_limit, lr_bits, lr_hit);
}

// adam_tiles: the data-parallel step's optimizer launch behind the all-reduce (scripts/runners.py:183 apply_gradients, scaled by
// the all-reduced row count) in the TILE shape of mega3_step's epilogue: a workgroup owns one [16 MU x 16] tile of a weight
// tensor (+ its bias columns on the first tile row); gradients, parameters and moments move as lane-contiguous accesses and
// the updated values leave for the next step's operand images as 16-byte units -- the generic adam_tf_img (kernels.hpp) walks
// the flat buffer and scatters element by element through a 15-entry map: 7.5 us in-kernel against this launch's ~3.
// Same update statement (adam_update), same alpha_t, same scale: bit-identical parameters.
struct AdamTilesArgs {
  int ntens, total_tiles;
  unsigned short perm[kM3MaxSlots];
  DwTensor t[kM3MaxT];
  float *grads, *p, *m, *v;
  float lr, b1, b2, eps;
  const unsigned long long* t_dev;     // Adam's t of this step (the gradient launch's tail slot already advanced the counter)
  const float* gscale_dev;             // the all-reduced row count
  const float* loss_sum_dev;           // the all-reduced loss sum: non-finite = a poisoned step on some rank, nothing is applied
  float* tail_log;
  float* img[kImgBufs];
  unsigned* epoch_word;
  const unsigned* lr_dev;
  unsigned long long* span;
};
static_assert(sizeof(AdamTilesArgs) <= 4096, "kernel arguments");

__global__ __launch_bounds__(256) void adam_tiles(const AdamTilesArgs a) {
  if (a.span && threadIdx.x == 0) atomicMax(a.span, (1ull << 62) - wall_clock64());
  struct SpanEnd {
    unsigned long long* p;
    __device__ ~SpanEnd() { if (p && threadIdx.x == 0) atomicMax(p + 1, wall_clock64()); }
  } span_end{a.span};
  const int tid = threadIdx.x, bid = blockIdx.x;
  if (bid == 0 && tid == 0 && a.epoch_word) *a.epoch_word += 1u;
  if (bid == 0 && tid < 8 && a.tail_log) a.tail_log[tid] = a.loss_sum_dev[tid];       // the all-reduced tail
  if (bid >= a.total_tiles) return;
  const int pv_ = a.perm[bid];
  const int ti = (pv_ & 0x7fff) >> 10, tl = pv_ & 1023;
  const DwTensor& T = a.t[ti];
  const int M = T.M, N = T.N, MUr = T.mu;
  const int tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  const int m0 = tm * 16 * MUr, n0 = tn * 16;
  const int el = tid & 63, eu = (tid >> 6) & 3;
  const int mb = m0 + 4 * MUr * (el >> 4) + 4 * eu, en = n0 + (el & 15);
  const bool eown = tid < 64 * MUr && mb < M && en < N;
  // (the bias columns: threads 240..255 -- free at every tile height)
  const bool bown = tm == 0 && T.b_off >= 0 && tid >= 240 && n0 + (tid - 240) < N;
  float pp[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  float bp = 0.f, bm = 0.f, bv = 0.f, bg = 0.f;
  if (eown) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mm = min(mb + j, M - 1);
      const long long i = (long long)T.w_off + (long long)mm * N + en;
      pp[j] = a.p[i]; pm[j] = a.m[i]; pv[j] = a.v[i]; g[j] = a.grads[i];
    }
  }
  if (bown) { const int i = T.b_off + n0 + tid - 240; bp = a.p[i]; bm = a.m[i]; bv = a.v[i]; bg = a.grads[i]; }
  if (!__builtin_isfinite(*a.loss_sum_dev)) return;            // poisoned step: keep params, m, v and the images
  const unsigned long long t = *a.t_dev;
  const float gs = 1.f / *a.gscale_dev;
  float lr_t;
  if (a.lr_dev && a.lr_dev[1] == (unsigned)t && a.lr_dev[2] == __float_as_uint(a.lr) && a.lr_dev[3] == alpha_key(a.b1, a.b2))
    lr_t = __uint_as_float(a.lr_dev[0]);
  else lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t)));
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
  if (eown) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (mb + j < M) {
        const long long i = (long long)T.w_off + (long long)(mb + j) * N + en;
        adam_update(pp[j], pm[j], pv[j], g[j], gs, lr_t, omb1, omb2, a.eps);
        a.p[i] = pp[j]; a.m[i] = pm[j]; a.v[i] = pv[j];
      } else {
        pp[j] = 0.f;                               // rows past the tensor: the image's padding stays zero
      }
    }
    if (T.k1 == 2 || T.k1 == 4 || T.k1 == 7)
      *reinterpret_cast<float4*>(a.img[T.which1] + img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb, en)) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    else if (T.k1 >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (mb + j < M) a.img[T.which1][img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb + j, en)] = pp[j];
    }
    if (T.k2 >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (mb + j < M) a.img[T.which2][img_dst(T.k2, T.base2, T.ld2, T.chunk2, mb + j, en)] = pp[j];
    }
  }
  if (bown) {
    const int c = tid - 240, i = T.b_off + n0 + c;
    adam_update(bp, bm, bv, bg, gs, lr_t, omb1, omb2, a.eps);
    a.p[i] = bp; a.m[i] = bm; a.v[i] = bv;
    if (T.bk >= 0) a.img[T.bwhich][img_dst(T.bk, T.bbase, 0, T.bchunk, 0, n0 + c)] = bp;
  }
}

}  // namespace gmvae
