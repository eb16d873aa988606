// libgmvae_hip.so -- C ABI (include/gmvae_hip.h) and the host-side schedule of
// the VAE / VAE_GMP / GMVAE ELBO training step on gfx950.
//
// The step is a fixed sequence of "levels"; every level is ONE kernel launch:
// either a grouped fp32-MFMA GEMM (gemm.hpp) that executes all independent
// matrix products of that dependency level, or a row-local kernel
// (kernels.hpp).  Nothing here allocates or synchronises; everything is
// enqueued on the caller's stream and is hipGraph-capturable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <chrono>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gmvae_hip.h"
#include "gemm.hpp"
#include "kernels.hpp"
#include "chain.hpp"
#include "mega.hpp"
#include "mega2.hpp"
#include "mega2v.hpp"
#include "dwadam.hpp"
#include "mega3.hpp"
#include "skinny.hpp"
#include "rowsws.hpp"
#include "evalf.hpp"

using namespace gmvae;

namespace {

constexpr int MAXH = GMVAE_MAX_HIDDEN;
constexpr int NS_MAX = 16;
constexpr int MAX_LEVELS = 96;
constexpr int GMP_PARTS = 64;
constexpr int kMegaQMax = 4;
constexpr int kSkNs1 = 4;        // slabs of the skinny schedule's first layer (contraction split); <= skinny.hpp kSkNs1x

// Workgroups that share one 16-row panel of mega_fwd_bwd (they split its decoder chunks): as many as keep the
// whole grid co-resident on the chip's 256 CUs (one 150 KB-LDS workgroup per CU).
// GmvaeDims::sched_flags & GMVAE_SCHED_SAFE: the schedules in which no workgroup waits for another of its own launch
// (one workgroup per panel, the first layer as a launch of its own) -- what a caller degrades to after a hand-off timeout
static bool sched_safe(const GmvaeDims& d) { return (d.sched_flags & GMVAE_SCHED_SAFE) != 0; }
// compute units of the CURRENT device (cached per device id; 256 on an unpartitioned MI355X): the hand-offs inside a launch
// need every workgroup of the grid resident at once, one per CU
static int device_cus() {
  static int cu_of[64];
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev < 0 || dev >= 64) { hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }
  if (!cu_of[dev]) { hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cu_of[dev] = n > 0 ? n : 256; }
  return cu_of[dev];
}
// The one-launch steps (mega3.hpp) read hand-off buffers with PLAIN loads behind flags: correct under the cache behaviour of the
// architecture it was verified on (gfx950: L2s invalidated at dispatch, byte-masked write-through allocation) -- by the HIP memory
// model it is a data race, and a stale line would give silently wrong gradients.  Anything else takes the two-launch form.
static bool device_is_gfx950() {
  static signed char is_of[64];              // 0 unknown, 1 yes, -1 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!is_of[dev]) {
    hipDeviceProp_t pr;
    is_of[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && !strncmp(pr.gcnArchName, "gfx950", 6)) ? 1 : -1;
  }
  return is_of[dev] > 0;
}
// one-shot per DEVICE (hipFuncSetAttribute belongs to the device's code object; a process may drive several devices)
static bool first_on_device(bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (seen[dev]) return false;
  seen[dev] = true;
  return true;
}
static int mega_q(const GmvaeDims& d) {
  const int panels = (d.B + 15) / 16, cus = device_cus();
  int q = panels * 4 <= cus ? 4 : panels * 2 <= cus ? 2 : 1;
  const char* e = getenv("GMVAE_MEGA_Q");          // (tests / diagnostics)
  if (e && atoi(e) >= 1 && atoi(e) <= kMegaQMax) q = atoi(e);
  if (sched_safe(d)) q = 1;
  return q;
}

// ------------------------------------------------------------------ layout
struct NetL {
  int nl = 0;                 // number of Linear layers
  int dim[MAXH + 2] = {0};    // dim[i] -> dim[i+1]
  uint64_t w[MAXH + 1] = {0}, b[MAXH + 1] = {0};
};
struct Layout {
  NetL ency, prior, encg, dec, enc;
  uint64_t loc = 0, rawscale = 0, mixlog = 0, P_pad = 0, P_real = 0;
  int n = 0;
  GmvaeParamEntry e[64];
};

static uint64_t pad4(uint64_t n) { return (n + 3) / 4 * 4; }

static void add_entry(Layout& L, const char* name, int rows, int cols, uint64_t& off_out) {
  GmvaeParamEntry& e = L.e[L.n++];
  snprintf(e.name, sizeof(e.name), "%s", name);
  e.rows = rows;
  e.cols = cols;
  e.offset = L.P_pad;
  off_out = L.P_pad;
  L.P_real += (uint64_t)rows * cols;
  L.P_pad += pad4((uint64_t)rows * cols);
}

static void add_net(Layout& L, NetL& net, const char* name, int n_in, const int* hidden, int nh, int n_out) {
  net.nl = nh + 1;
  net.dim[0] = n_in;
  for (int i = 0; i < nh; ++i) net.dim[i + 1] = hidden[i];
  net.dim[nh + 1] = n_out;
  char buf[64];
  for (int i = 0; i < net.nl; ++i) {
    snprintf(buf, sizeof(buf), "%s_fcnet/linear_%d/w", name, i);
    add_entry(L, buf, net.dim[i], net.dim[i + 1], net.w[i]);
    snprintf(buf, sizeof(buf), "%s_fcnet/linear_%d/b", name, i);
    add_entry(L, buf, 1, net.dim[i + 1], net.b[i]);
  }
}

static int check_dims(const GmvaeDims* d, int model) {
  if (!d) return GMVAE_E_NULL;
  if (model < 0 || model > 2) return GMVAE_E_MODEL;
  if (d->B < 1 || d->D < 1 || d->L < 1 || d->K < 1 || d->S < 1) return GMVAE_E_DIMS;
  if (d->n_hidden < 0 || d->n_hidden > MAXH) return GMVAE_E_DIMS;
  for (int i = 0; i < d->n_hidden; ++i)
    if (d->hidden[i] < 1) return GMVAE_E_DIMS;
  if (d->gen_bias_vec ? d->gen_bias_len != d->D : d->gen_bias_len != 0) return GMVAE_E_DIMS;
  if (d->gen_bias_vec && (reinterpret_cast<uintptr_t>(d->gen_bias_vec) & 3)) return GMVAE_E_ALIGN;
  if ((long long)d->B * d->S > (1LL << 30)) return GMVAE_E_DIMS;
  if (d->hidden_act < GMVAE_ACT_RELU || d->hidden_act > GMVAE_ACT_ELU) return GMVAE_E_DIMS;
  return 0;
}

// Variable-creation order of the reference: GMVAE: encoder_y,
// prior_gmm, encoder_gmm, decoder (gmvae.py:238,243,246,251); VAE: prior
// variables (vae.py:233-238), encoder, decoder.
static void build_layout(const GmvaeDims& d, int model, Layout& L) {
  if (model == GMVAE_MODEL_GMVAE) {
    add_net(L, L.ency, "encoder_y", d.D, d.hidden, d.n_hidden, d.K);
    add_net(L, L.prior, "prior_gmm", d.K, nullptr, 0, 2 * d.L);
    add_net(L, L.encg, "encoder_gmm", d.D + d.K, d.hidden, d.n_hidden, 2 * d.L);
    add_net(L, L.dec, "decoder", d.L, d.hidden, d.n_hidden, d.D);
  } else {
    if (model == GMVAE_MODEL_VAE_GMP) {
      add_entry(L, "loc", d.K, d.L, L.loc);
      add_entry(L, "raw_scale_diag", d.K, d.L, L.rawscale);
      add_entry(L, "mixture_logits", 1, d.K, L.mixlog);
    }
    add_net(L, L.enc, "encoder", d.D, d.hidden, d.n_hidden, 2 * d.L);
    add_net(L, L.dec, "decoder", d.L, d.hidden, d.n_hidden, d.D);
  }
}

// --------------------------------------------------------------- workspace
struct WS {
  float *eps, *u;
  float* he[MAXH + 2];   // enc_y (GMVAE) / encoder (VAE) activations, he[i] = input of layer i (i>=1), [B, dim[i]]
  float* hg[MAXH + 2];   // encoder_gmm activations [R, dim[i]]
  float* hd[MAXH + 2];   // decoder activations [R, dim[i]]
  float *gx, *logits, *y, *nent, *pp, *qp, *z, *logq, *logp, *logpx, *logw, *rw, *resp, *g, *part;
  float *dbuf[3], *dz, *dqp, *dpp, *dy, *dlogits, *dqb, *slabs, *gmp_part;
  unsigned* sk_cnt;                  // skinny schedule, sk_dwc: arrived batch shares per weight-gradient tile
  float *sk_s1, *sk_lqp, *sk_part;   // skinny schedule: first-layer slabs [ns1][B][2H]; log q / log p partials [2][L/16][B]; logpx partials [B][D/16]
  float *gmp_inv, *gmp_cst;     // tiled mixture log-prob (any K, L): 1 / softplus(raw_scale_diag) [K][L], per-component constants [K]
  double* lw64;            // S > 1: log w per row in fp64 (kernels.hpp row_terms / iwae_rows)
  float *pb, *dsum;        // S > 1: per-row IWAE partials [B][4]; sum over s of encoder_gmm's first-layer gradient [B][H]
  float *s1, *s4;          // split-K slabs of the fused schedule: [NSF][B][2H], [NSF][B][H]
  unsigned long long* stamps;   // diagnostic stamps of the chain kernels: [2][grid][16]
  unsigned long long* xchg;     // mega_fwd_bwd's in-launch hand-off granules: [panels][Q-1][16*H + 16]
  unsigned long long* xfl;      // mega_fwd_bwd's first-layer exchange granules: [panels][4][16 * H2]
  unsigned long long* spans;    // measurement: [2 slots][3 kernels][2048 blocks][2] wall-clock stamps (mega_fwd_bwd, finalize_adam / dw_adam, fl_split)
  unsigned long long* gstamps;  // diagnostic stamps of the grouped-GEMM launches: [4 slots][2048 blocks][8]
  unsigned* sync;               // [0] = per-step epoch of the hand-off, [1] = hand-off timeout flag
  unsigned* m3flags;            // mega3_step's per-workgroup epoch flags, replicated (mega3.hpp)
  float *img_f, *img_b;         // per-step LDS weight images of chain_fwd / chain_bwd (prepared by aux blocks)
  float *img_m, *dimg;          // mega kernel: small-weight image (odd leading dimensions) + decoder chunk images
  float *img2f, *img2b, *dimg2; // mega2 kernel: forward / backward operand images, decoder operand images
  float* pscale;
  float* pscale_f;                 // forward-only pairs (fwd_pairs_ok): the same words for hd2f / w2f
  unsigned short *hd2f, *w2f;      // ... the top decoder layer's input [R x Ht] and weight [Ht x pad128(D)] as f16 pairs
  unsigned short *hd3, *g3, *w3;   // general schedule, large top decoder layer: planes of 16-bit pieces (gemm.hpp plane_rounds) of its
                                   // input activation [3][R][H], of (sigmoid - x) [3][R][D] and of its weight [3][H][D]
  int32_t* cl_pred;
  unsigned long long* ev_dbg;           // ... its diagnostic stamps [1024][16]
  float *ev_img, *ev_rows, *ev_slots;   // evalf.hpp (forward-only evaluation at the reference's default sizes): operand images, [R][4] row terms,
                                        // per-workgroup sums [1024][4] + the arrival counter
  uint64_t bytes;
};

// ---- fused schedule for the launch-bound default sizes (chain.hpp) ---------------------------
// measured (tools/sweep.sh, B=1024 D=784 H=64): 4 forward splits of 256 and 8 batch splits with 64x64 tiles for the
// weight-gradient launch are the fastest combination (re-checked after every change of the launch: 4, 6, 12 and 16
// splits all measured slower; the fp32 problems of the mega schedule take twice this, see run_step_mega)
static int dw_splits(long long B) {
  long long ns = B / 128;
  if (ns < 1) ns = 1;
  if (ns > NS_MAX) ns = NS_MAX;
  return (int)ns;
}
static int fwd_splits(int D) {
  const int ns = (D + 255) / 256;
  return ns < 1 ? 1 : (ns > NS_MAX ? NS_MAX : ns);
}
// the single-launch per-row kernel (mega.hpp), all three models: one hidden layer <= 64, S = 1, 16-byte aligned
// x rows, its LDS budget, and (VAE_GMP) one prior-gradient partial per workgroup
// (the *_shape predicates read the dims only: carve() sizes the workspace with them, so that its layout never depends on an
//  environment switch; the *_ok forms add the switches and choose the schedule)
static bool mega_shape(const GmvaeDims& d, int model) {
  if (d.n_hidden != 1 || d.S != 1 || d.D % 16 || d.hidden_act != GMVAE_ACT_RELU) return false;
  if (d.gen_bias_vec) return false;            // the vector bias_init is applied by the grouped GEMM's epilogue (Problem::bias2)
  const int H = d.hidden[0];
  if (H % 16 || H > 64 || d.L % 2 || d.L > 128 || d.K > 64) return false;      // L even: k-steps of 4 over [mu | raw]
  if (model == GMVAE_MODEL_VAE_GMP && (d.B + kPanel - 1) / kPanel > GMP_PARTS) return false;
  return (size_t)mega_lay(H, d.L, d.K, d.D, model).total * 4 <= 160 * 1024;
}
static bool mega_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_MEGA");
  if (e && atoi(e)) return false;
  const char* e2 = getenv("GMVAE_NO_FUSED");
  if (e2 && atoi(e2)) return false;
  return mega_shape(d, model);
}
// mega2_fwd_bwd (mega2.hpp): the steady-state launch specialised for the reference's default sizes
static bool mega2_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_MEGA2");
  if (e && atoi(e)) return false;
  return model == GMVAE_MODEL_GMVAE && mega_ok(d, model) && d.hidden[0] == M2::H && d.L == M2::L && d.K == M2::K &&
         d.D == M2::D && d.B <= 1024;
}
// mega2v_fwd_bwd (mega2v.hpp): the same design for the VAE family at small batches, SEVEN workgroups per panel -- the plain
// VAE at latent 2 (BASELINE configs[0]) and VAE_GMP at latent 64, K = 10 (configs[1]), hidden 64, D = 784
static int mega2v_kind(const GmvaeDims& d, int model) {       // 0: not these sizes; 1: VAE L = 2; 2: VAE_GMP L = 64 K = 10
  if (d.n_hidden != 1 || d.hidden[0] != 64 || d.D != 784 || d.S != 1 || d.gen_bias_vec) return 0;
  if ((d.B + kPanel - 1) / kPanel * 7 > 256) return 0;
  if (model == GMVAE_MODEL_VAE && d.L == 2) return 1;
  if (model == GMVAE_MODEL_VAE_GMP && d.L == 64 && d.K == 10) return 2;
  return 0;
}
typedef M2V<0, 2, 1> MV0;
typedef M2V<1, 64, 10> MV1;
static bool mega2v_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_MEGA2");
  if (e && atoi(e)) return false;
  return mega_ok(d, model) && mega2v_kind(d, model) != 0 && (d.B + kPanel - 1) / kPanel * 7 <= device_cus() && !sched_safe(d) &&
         !getenv("GMVAE_MEGA_Q");
}
// the skinny schedule (skinny.hpp): GMVAE, one WIDE hidden layer, a SMALL batch -- bin/run_train.sh's sizes
constexpr int kSkMaxB = 4096;       // hard bound of the skinny schedule's batch (its buffers are carved up to here)
static bool skinny_shape(const GmvaeDims& d, int model) {
  // GMVAE; the VAE with the standard-normal prior (no y path: eight launches); VAE_GMP (the learned mixture prior is not
  // column-local: its log-density, its share of dz and its variables' gradients stay three row kernels: eleven launches)
  if (d.n_hidden != 1 || d.S != 1 || d.hidden_act != GMVAE_ACT_RELU) return false;
  const int H = d.hidden[0];
  // measured against the general schedule at H = 128 / 256 / 512, L = 128 (tools/sk_sweep.py, one box): 2.9x faster at B = 32..64,
  // 2.4 - 2.6x at 256, 1.9 - 2.1x at 512, 1.8 - 1.9x at 1024, 1.5 - 1.6x at 2048, 1.2 - 1.4x at 4096 (round 4: the forms for
  // more than 128 rows -- row-tile groups, 64-column tiles, [64 x 64] W tiles)
  // (L a multiple of 4: 16-byte loads along latent rows; a ragged last tile of 16 latent dimensions is masked)
  // (K <= 16: the y path's per-row softmax in 16 lanes -- GMVAE only; the mixture prior's K is the row kernels' business)
  return H % 64 == 0 && H <= 1024 && d.D % 16 == 0 && d.L % 4 == 0 && d.L >= 4 && d.L <= 256 &&
         (model != GMVAE_MODEL_GMVAE || d.K <= 16) && d.B <= kSkMaxB;
}
static bool skinny_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_SKINNY");
  if (e && atoi(e)) return false;
  int maxb = kSkMaxB;
  if (const char* mb = getenv("GMVAE_SKINNY_MAXB")) maxb = atoi(mb) < kSkMaxB ? atoi(mb) : kSkMaxB;     // (tools/sk_sweep.py)
  return skinny_shape(d, model) && d.B <= maxb;
}
static bool fused_shape(const GmvaeDims& d, int model) {
  if (model != GMVAE_MODEL_GMVAE || d.n_hidden != 1 || d.S != 1 || d.hidden_act != GMVAE_ACT_RELU) return false;
  const int H = d.hidden[0];
  if (H % 16 || H > 64 || d.L % 8 || d.L > 128 || d.K > 64) return false;
  const int f = fwd_lay(H, d.L, d.K).total, b = bwd_lay(H, d.L, d.K).total;
  return (size_t)(f > b ? f : b) * 4 <= 156 * 1024;
}
static bool fused_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_FUSED");
  if (e && atoi(e)) return false;
  return fused_shape(d, model);
}

// evalf_rows (evalf.hpp): the forward-only pass of the GMVAE at the reference's default sizes, any batch, any number of samples
static bool evalf_shape(const GmvaeDims& d, int model) {
  if (d.n_hidden != 1 || d.hidden[0] != EV::H || d.D != EV::D || d.hidden_act != GMVAE_ACT_RELU || d.gen_bias_vec) return false;
  if (model == GMVAE_MODEL_GMVAE) return d.L == EV::L && d.K == EV::K;
  if (model == GMVAE_MODEL_VAE) return d.L == 2 || d.L == 64;          // (evalf_rows_v: BASELINE configs[0]'s latent size, and 64)
  return d.L == 64 && d.K == 10;                                       // VAE_GMP: configs[1]
}
static bool evalf_ok(const GmvaeDims& d, int model) {
  const char* e = getenv("GMVAE_NO_EVALF");
  if (e && atoi(e)) return false;
  if (!evalf_shape(d, model)) return false;
  // a workgroup stages 8 batch rows per table pass: with ONE sample per row and more than 8 batch rows per workgroup a pass is half
  // a panel on one wave (measured, tools/eval_time.py: B = 8192, S = 1: 110 / 139 us against 115 / 81 on the chain / general
  // schedules; every other shape tried is 1.0 - 2.4x faster here)
  int grid = device_cus();
  if (grid > d.B) grid = d.B;
  if (grid > 1024) grid = 1024;
  return !(d.S == 1 && (d.B + grid - 1) / grid > EV::NB);
}

static int num_splits(long long R) {
  long long ns = R / 256;
  if (ns < 1) ns = 1;
  if (ns > NS_MAX) ns = NS_MAX;
  return (int)ns;
}

// splits of the weight gradients with few outputs (general schedule): more than NS_MAX at large R
constexpr int NS_SMALL_MAX = 64;
static int num_splits_small(long long R) {
  long long ns = R / 400;
  if (ns < 1) ns = 1;
  if (ns > NS_SMALL_MAX) ns = NS_SMALL_MAX;
  const char* e = getenv("GMVAE_NSPLIT_SMALL");      // (tests: the many-splits path at sizes the oracle covers)
  if (e && atoi(e) >= 1 && atoi(e) <= NS_SMALL_MAX) ns = atoi(e);
  return (int)ns;
}

// The top decoder layer's three GEMMs (logits, data gradient, weight gradient) on the bf16 matrix cores from operands split
// once by their producers (gemm.hpp plane_rounds): interior 128-tiles in every orientation, and enough rows to pay for the
// split launches (measured at the config-5 shard, tools/gemm_planes.py: 1.45 - 1.65x the fp32 MFMA instance per GEMM).
static bool planes_ok(const GmvaeDims& d, const Layout& L) {
  const char* e = getenv("GMVAE_NO_PLANES");
  if (e && atoi(e)) return false;
  if (d.hidden_act != GMVAE_ACT_RELU) return false;     // (the plane producers' epilogues are the ReLU ones)
  const long long R = (long long)d.B * d.S;
  const int Ht = L.dec.dim[L.dec.nl - 1];
  long long minr = 4096;
  if (const char* m = getenv("GMVAE_PLANES_MINROWS")) minr = atoll(m);     // (tests: the plane path at sizes the oracle covers)
  // (below ~4096 rows, or with a contraction of a few rounds, the split launches and the tile prologues eat the gain: measured
  //  only at the config-5 shard's 25600 x 3072 x 512; the forced test sizes set GMVAE_PLANES_MINROWS)
  const bool big = getenv("GMVAE_PLANES_MINROWS") != nullptr || (Ht >= 256 && d.D >= 512);
  return L.dec.nl >= 2 && big && R >= minr && R % 128 == 0 && d.D % 128 == 0 && Ht % 128 == 0 && R * d.D < (1ll << 32);
}

static void carve(const GmvaeDims& d, int model, const Layout& L, void* base, WS& w) {
  uint64_t off = 0;
  auto take = [&](uint64_t nfloats) -> float* {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off += (nfloats * 4 + 255) / 256 * 256;
    return p;
  };
  const uint64_t B = d.B, R = (uint64_t)d.B * d.S, K = d.K, Lz = d.L, D = d.D;
  int maxh = 1;
  for (int i = 0; i < d.n_hidden; ++i) maxh = d.hidden[i] > maxh ? d.hidden[i] : maxh;
  memset(&w, 0, sizeof(w));
  w.eps = take(pad4(R * Lz));
  w.u = take(pad4(R * K));
  const NetL& e = (model == GMVAE_MODEL_GMVAE) ? L.ency : L.enc;
  for (int i = 1; i < e.nl; ++i) w.he[i] = take(B * e.dim[i]);
  if (model == GMVAE_MODEL_GMVAE) {
    for (int i = 1; i < L.encg.nl; ++i) w.hg[i] = take(R * L.encg.dim[i]);
    w.gx = take(B * L.encg.dim[1]);
    w.logits = take(B * K);
    w.y = take(R * pad4(K));                 // the mega schedule stores rows of pad4(K): 16-byte loads in the dW launch
    w.nent = take(B);
    w.pp = take(R * 2 * Lz);
    w.dpp = take(R * 2 * Lz);
    w.dy = take(R * K);
    w.dlogits = take(B * pad4(K));
    w.qp = take(R * 2 * Lz);
  } else {
    w.qp = take(B * 2 * Lz);
    w.dqb = take(B * 2 * Lz);
    if (model == GMVAE_MODEL_VAE_GMP) {
      w.resp = take(R * K);
      w.gmp_part = take((uint64_t)GMP_PARTS * (2 * pad4(K * Lz) + pad4(K)));
      w.gmp_inv = take(pad4(K * Lz));
      w.gmp_cst = take(pad4(K));
    }
  }
  for (int i = 1; i < L.dec.nl; ++i) w.hd[i] = take(R * L.dec.dim[i]);
  w.z = take(R * Lz);
  w.logq = take(R); w.logp = take(R); w.logpx = take(R); w.logw = take(R); w.rw = take(R);
  w.g = take(R * D);
  w.part = take(R * ((D + 31) / 32));
  w.dbuf[0] = take(R * maxh);
  w.dbuf[1] = take(R * maxh);
  w.dbuf[2] = take(R * maxh);
  if (fused_shape(d, model)) {
    w.s1 = take((uint64_t)fwd_splits(d.D) * B * 2 * d.hidden[0]);
    w.s4 = take((uint64_t)fwd_splits(d.D) * B * d.hidden[0]);
    w.stamps = reinterpret_cast<unsigned long long*>(take(2ull * ((B + 15) / 16) * 16 * 2));
    w.img_f = take((uint64_t)fwd_lay(d.hidden[0], d.L, d.K).img);
    w.img_b = take((uint64_t)bwd_lay(d.hidden[0], d.L, d.K).img);
  }
  if (mega_shape(d, model)) {
    const MegaLay ml = mega_lay(d.hidden[0], d.L, d.K, d.D, model);
    w.img_m = take((uint64_t)ml.img);
    w.dimg = take((uint64_t)ml.nch * ml.chunk);
    if (!w.s1) w.s1 = take((uint64_t)fwd_splits(d.D) * B * 2 * d.hidden[0]);
    w.stamps = reinterpret_cast<unsigned long long*>(take(2ull * ((B + 15) / 16) * kMegaQMax * 16 * 2));   // one slot per workgroup
    // (mega2v_fwd_bwd: six producers per panel)
    w.xchg = reinterpret_cast<unsigned long long*>(
        take(2ull * ((B + 15) / 16) * (mega2v_kind(d, model) ? 6 : kMegaQMax - 1) * (kPanel * d.hidden[0] + kPanel)));
    w.sync = reinterpret_cast<unsigned*>(take(256));          // [0] epoch, [1] timeout flag, [2] alpha_t; [64, 128): mega3_step's per-panel flags
    if (ml.fl_ok)
      w.xfl = reinterpret_cast<unsigned long long*>(take(2ull * ((B + 15) / 16 + 1) * 4 * kPanel * 2 * d.hidden[0]));      // (+ 1: mega2 pairs panels)
    w.gstamps = reinterpret_cast<unsigned long long*>(take(2ull * 4 * 2048 * 8));
    w.spans = reinterpret_cast<unsigned long long*>(take(2ull * 2 * 3 * 2048 * 2));
    if (model == GMVAE_MODEL_GMVAE && d.hidden[0] == M2::H && d.L == M2::L && d.K == M2::K && d.D == M2::D && d.B <= 1024) {
      w.m3flags = reinterpret_cast<unsigned*>(take((uint64_t)kM3FlagReplicas * kM3FlagRepLd));      // mega3_step's flag replicas
      w.img2f = take(M2::imgF);                  // (not gated by GMVAE_NO_MEGA2: the workspace layout must not depend on a switch)
      w.img2b = take(M2::imgB);
      w.dimg2 = take(M2::dimg);
    }
    if (const int vk = mega2v_kind(d, model)) {
      w.m3flags = reinterpret_cast<unsigned*>(take((uint64_t)kM3FlagReplicas * kM3FlagRepLd));      // mega3v_step's flag replicas
      w.img2f = take(vk == 1 ? MV0::imgF : MV1::imgF);
      w.img2b = take(vk == 1 ? MV0::imgB : MV1::imgB);
      w.dimg2 = take(vk == 1 ? MV0::dimg : MV1::dimg);
    }
  }
  if (skinny_shape(d, model)) {       // (sized by the dims alone: no switch, no batch bound below kSkMaxB)
    w.sk_s1 = take((uint64_t)kSkNs1 * B * 2 * d.hidden[0]);
    w.sk_lqp = take(2ull * ((Lz + 15) / 16) * B);
    w.sk_cnt = reinterpret_cast<unsigned*>(take(kSkDwcMaxTiles));
    w.sk_part = take(B * ((D + 15) / 16));
  }
  w.dz = take(R * Lz);
  w.dqp = take(R * 2 * Lz);
  if (d.S > 1) {
    w.pb = take(B * 4);
    w.lw64 = reinterpret_cast<double*>(take(2 * R));
    if (model == GMVAE_MODEL_GMVAE) w.dsum = take(B * maxh);
  }
  {
    int ns = num_splits(R);
    if (num_splits_small(R) > ns) ns = num_splits_small(R);       // (the extra slabs are only touched in the small tensors' ranges)
    if ((fused_shape(d, model) || mega_shape(d, model)) && dw_splits(d.B) > ns) ns = dw_splits(d.B);
    if (mega_shape(d, model) && 2 * dw_splits(d.B) <= NS_MAX && 2 * dw_splits(d.B) > ns) ns = 2 * dw_splits(d.B);
    if (skinny_shape(d, model) && ns < kSkDwShares) ns = kSkDwShares;       // (sk_dwc's partial gradients)
    w.slabs = take((uint64_t)ns * L.P_pad);
  }
  w.cl_pred = reinterpret_cast<int32_t*>(take(B));
  {
    // (sized by the dims alone, not by GMVAE_NO_PLANES / GMVAE_PLANES_MINROWS: the workspace layout must not depend on a switch)
    const uint64_t Ht = L.dec.dim[L.dec.nl - 1];
    if (L.dec.nl >= 2 && R % 128 == 0 && D % 128 == 0 && Ht % 128 == 0 && R * D < (1ull << 32)) {
      w.hd3 = reinterpret_cast<unsigned short*>(take((3 * R * Ht + 1) / 2));
      w.g3 = reinterpret_cast<unsigned short*>(take((3 * R * D + 1) / 2));
      w.w3 = reinterpret_cast<unsigned short*>(take((3 * Ht * D + 1) / 2));
      // f16 pairs: [0], [1] bits of max |h|, max |W| of the step; [2], [3] 1 / scale of either; [16..] partial maxima: kAmaxBlocks
      // words for W, then one per wave of the launch that produces h (rows_nn_bf6: a wave per 16 rows x 64 columns at least)
      {
        uint64_t units = ((R + 15) / 16) * ((Ht + 63) / 64);            // (or kAmaxBlocks partials when amax_abs reduces h,
        if (units < (uint64_t)kRwsMaxWaves) units = kRwsMaxWaves;       //  or one per wave of rows_ws)
        w.pscale = take(16 + kAmaxBlocks + (units > (uint64_t)kAmaxBlocks ? units : (uint64_t)kAmaxBlocks));
      }
    }
  }
  {
    // forward-only evaluation at thousands of rows (fwd_pairs_ok): the logits GEMM on f16 pairs whatever D (the weight's planes
    // are zero-padded to whole 128-column tiles) and from a 32-wide hidden layer up
    const uint64_t Ht = L.dec.dim[L.dec.nl - 1], Dp = (D + 127) / 128 * 128;
    if (L.dec.nl >= 2 && R % 128 == 0 && Ht % 32 == 0 && D % 8 == 0 && R * Dp < (1ull << 32)) {
      w.hd2f = reinterpret_cast<unsigned short*>(take(R * Ht));
      w.w2f = reinterpret_cast<unsigned short*>(take(Ht * Dp));
      uint64_t units = ((R + 15) / 16) * ((Ht + 63) / 64);              // (or kAmaxBlocks partials when amax_abs reduces h,
      if (units < (uint64_t)kRwsMaxWaves) units = kRwsMaxWaves;         //  or one per wave of rows_ws)
      w.pscale_f = take(16 + kAmaxBlocks + (units > (uint64_t)kAmaxBlocks ? units : (uint64_t)kAmaxBlocks));
    }
  }
  if (evalf_shape(d, model)) {                   // (at the end: no earlier offset moves)
    w.ev_img = take(EV::total);
    w.ev_rows = take(R * 4);
    w.ev_slots = take(4 * 1024 + 64);
    w.ev_dbg = reinterpret_cast<unsigned long long*>(take(2ull * 1024 * 16));
  }
  w.bytes = off;
}

// forward-only steps (evaluation: S importance samples per row) whose logits GEMM is worth running on f16 pairs: the plane path
// proper (planes_ok) needs D % 128 == 0 and a wide hidden layer because its BACKWARD GEMMs read (sigmoid - x) planes; forward
// only, the one GEMM takes any D % 8 == 0 (weight planes zero-padded to the tile) and any hidden width % 32.  (eval_iwae:
// 51200 x 784 x 64 -- as fp32 MFMA the launch is half matrix time, half Bernoulli epilogue.)
static bool fwd_pairs_ok(const GmvaeDims& d, const Layout& L) {
  const char* e = getenv("GMVAE_NO_PLANES");
  if (e && atoi(e)) return false;
  const char* x = getenv("GMVAE_PLANES_EXACT");
  if (x && atoi(x)) return false;
  if (d.hidden_act != GMVAE_ACT_RELU || L.dec.nl < 2) return false;
  const long long R = (long long)d.B * d.S;
  const int Ht = L.dec.dim[L.dec.nl - 1];
  long long minr = 8192;
  if (const char* m = getenv("GMVAE_PLANES_MINROWS")) minr = atoll(m);
  return R >= minr && R % 128 == 0 && Ht % 32 == 0 && d.D % 8 == 0 && R * ((d.D + 127) / 128 * 128) < (1ll << 32);
}

// ----------------------------------------------------------- GEMM building
static Operand opnd(const void* p, int ld, int n_mn, bool u8, bool kc, int div = 1) {
  Operand o;
  o.ptr = p; o.ld = ld; o.n_mn = n_mn; o.row_div = div;
  o.is_u8 = u8; o.k_contig = kc; o.pad_ = 0;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  o.vec_ok = (ld % 4 == 0) && (a % (u8 ? 4 : 16) == 0);
  return o;
}
static Problem blank() {
  Problem p;
  memset(&p, 0, sizeof(p));
  p.nseg = 1; p.splits = 1; p.add_div = 1; p.x_div = 1; p.epi = EPI_STORE;
  p.seg[0].a.row_div = p.seg[0].b.row_div = p.seg[1].a.row_div = p.seg[1].b.row_div = 1;
  return p;
}
// C[M,N] = act(A[M,K] W[K,N] + bias)
// hidden_activation_fn of the step being enqueued on this host thread (Problem::relu / mask_act kinds: GMVAE_ACT_* + 1);
// set by run_step / gmvae_mlp_forward from GmvaeDims::hidden_act
static thread_local int tl_hact = 1;
static Problem p_nn(const void* A, bool u8, int lda, const float* W, int ldw, int M, int N, int K, float* C,
                    int ldc, const float* bias, bool relu) {
  Problem p = blank();
  p.M = M; p.N = N;
  p.seg[0].a = opnd(A, lda, M, u8, true);
  p.seg[0].b = opnd(W, ldw, N, false, false);
  p.seg[0].K = K;
  p.C = C; p.ldc = ldc; p.bias = bias; p.relu = relu ? tl_hact : 0;
  return p;
}
// dX[M,N] = dY[M,K] W[N,K]^T  (W row-major [N rows, ldw])
static Problem p_nt(const float* dY, int ldy, const float* W, int ldw, int M, int N, int K, float* C, int ldc,
                    const float* mask, int ld_mask) {
  Problem p = blank();
  p.M = M; p.N = N;
  p.seg[0].a = opnd(dY, ldy, M, false, true);
  p.seg[0].b = opnd(W, ldw, N, false, true);
  p.seg[0].K = K;
  p.C = C; p.ldc = ldc; p.mask = mask; p.ld_mask = ld_mask; p.mask_act = tl_hact;
  return p;
}
// dW[in,out] = Act[rows,in]^T dY[rows,out] (+ db = column sums of dY), split-K over rows into slabs
static Problem p_tn(const void* Act, bool u8, int lda, int a_div, const float* dY, int ldy, int n_in, int n_out,
                    int rows, float* dW, float* db, int ns, long long slab_stride, const float* kscale) {
  Problem p = blank();
  p.M = n_in; p.N = n_out;
  p.seg[0].a = opnd(Act, lda, n_in, u8, false, a_div);
  p.seg[0].b = opnd(dY, ldy, n_out, false, false);
  p.seg[0].K = rows;
  p.seg[0].kscale = kscale;
  p.C = dW; p.ldc = n_out; p.colsum_out = db;
  p.splits = ns; p.split_stride = slab_stride;
  // uint8 activations with plain 4-aligned extents take the bf16 matrix-core path of the medium tile configuration
  p.xbf16 = (u8 && a_div == 1 && !kscale && p.seg[0].a.vec_ok && p.seg[0].b.vec_ok && n_in % 4 == 0 && n_out % 4 == 0) ? 1 : 0;
  return p;
}

struct Prof {
  int n = 0;
  hipEvent_t ev[MAX_LEVELS + 1];
  char name[MAX_LEVELS][48];
  double flops[MAX_LEVELS];
  bool active = false;
  bool events = true;      // false: only names and FLOPs are collected (the launches time themselves)
};

struct Ctx {
  hipStream_t st;
  Prof* prof = nullptr;
  int err = 0;
  int force_cfg = -1;
  Ctx() { (void)hipGetLastError(); }   // drop any stale error another library left on this thread
  void mark(const char* name, double flops) {
    if (!prof || !prof->active || prof->n >= MAX_LEVELS) return;
    snprintf(prof->name[prof->n], sizeof(prof->name[0]), "%s", name);
    prof->flops[prof->n] = flops;
    prof->n++;
    if (prof->events) hipEventRecord(prof->ev[prof->n], st);
  }
  void check() {
    if (!err) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) err = (int)e;
    }
  }
};

struct Group {
  Launch L;
  Group() { memset(&L, 0, sizeof(L)); }
  void add(const Problem& p) { if (L.nprob < MAXP) L.p[L.nprob++] = p; }
};

template <class C>
static int tile_up(Launch& L) {
  int t = 0;
  for (int i = 0; i < L.nprob; ++i) {
    Problem& p = L.p[i];
    p.tiles_m = (p.M + C::BM - 1) / C::BM;
    p.tiles_n = (p.N + C::BN - 1) / C::BN;
    // XCD-aware tile order (gemm.hpp): only where it matters -- many tiles and one operand much larger than the other
    p.xorder = 0;
    if (C::BM >= 128 && (long long)p.tiles_m * p.tiles_n * p.splits >= (C::BM == 256 ? 256 : 512)) {
      // (4: a split-K weight gradient with a multiple of 8 splits: ALL units of one split on one XCD -- the XCD then fetches only
      //  its splits' k slices of BOTH operands; with order 2 every XCD fetched the whole A operand: at config 5 the activation
      //  planes, 78 MB x 8)
      static const bool x4 = !(getenv("GMVAE_NO_XORDER4") && atoi(getenv("GMVAE_NO_XORDER4")));
      if (x4 && p.splits >= 8 && (p.splits & 7) == 0 && p.tiles_m * p.tiles_n >= 32 && p.tiles_m * p.tiles_n <= 512) p.xorder = 4;
      else if (p.tiles_m >= 16 * p.tiles_n && p.tiles_n * p.splits <= 16) p.xorder = 1;
      else if (p.tiles_n >= 4 * p.tiles_m && p.tiles_m <= 16) p.xorder = 2;
      else if (p.tiles_m >= 64 && p.tiles_n >= 16 && (p.tiles_n & 7) == 0 && p.splits == 1) p.xorder = 3;
    }
    p.tile_begin = t;
    L.tile_begin[i] = t;
    p.nparts = p.tiles_n;
    t += p.tiles_m * p.tiles_n * p.splits;
  }
  return t;
}

// every tile of the launch can take gemm.hpp's big rounds: interior tiles of fp32 operands read 16 bytes at a time, k
// ranges made of whole 32-deep rounds, per-k scale / column sums only beside an mn-contiguous operand b
static bool big_eligible(const Launch& L) {
  for (int i = 0; i < L.nprob; ++i) {
    const Problem& p = L.p[i];
    if (p.xbf16 || p.M % 128 || p.N % 128) return false;
    for (int s = 0; s < p.nseg; ++s) {
      const Segment& sg = p.seg[s];
      if (sg.a.is_u8 || sg.b.is_u8 || !sg.a.vec_ok || !sg.b.vec_ok || sg.a.row_div != 1 || sg.b.row_div != 1) return false;
      if (sg.a.n_mn < p.M || sg.b.n_mn < p.N || sg.K % 32) return false;
      if ((sg.kscale || p.colsum_out) && sg.b.k_contig) return false;
      const unsigned long long ext_a = sg.a.n_mn > sg.K ? sg.a.n_mn : sg.K, ext_b = sg.b.n_mn > sg.K ? sg.b.n_mn : sg.K;
      if (ext_a * sg.a.ld >= (1ull << 32) || ext_b * sg.b.ld >= (1ull << 32)) return false;      // 32-bit element offsets
    }
  }
  return true;
}

static int grid_for(long long items, int per_block, int cap = 4096);
// fp32 [rows][ld] (x rowscale[row]) -> three planes of 16-bit pieces in plane_rounds3's blocked-by-16 layout
static void launch_split(hipStream_t st, const float* src, const float* rowscale, int ld, long long n, unsigned short* dst) {
  hipLaunchKernelGGL(split_planes_b16, dim3(grid_for(n / ld / 16 * ((ld + 31) / 32), 4, 16384)), dim3(256), 0, st, src, rowscale, ld, (int)(n / ld), dst, n);
}

// the f16-pair form (gemm.hpp plane_rounds2): per-workgroup partial maxima of a tensor (launch_amax; or left by the producing
// launch's waves), folded into the bits of its largest magnitude and 1 / scale (launch_amax_final), then the split into two
// planes scaled by the power of two that brings that magnitude into [2^14, 2^15)
static void launch_amax(hipStream_t st, const float* src, long long n, unsigned* part) {
  hipLaunchKernelGGL(amax_abs, dim3(kAmaxBlocks), dim3(256), 0, st, src, n / 4, part);
}
static void launch_amax_final(hipStream_t st, const unsigned* pa, int na, const unsigned* pb, int nb, unsigned* bits, float* uns) {
  AmaxFinalArgs f;
  f.part[0] = pa; f.n[0] = na; f.part[1] = pb; f.n[1] = nb; f.bits = bits; f.uns = uns;
  hipLaunchKernelGGL(amax_final, dim3(2), dim3(256), 0, st, f);
}
static void launch_split_pairs(hipStream_t st, const float* src, const float* rowscale, int ld, long long n, unsigned short* dst,
                               const unsigned* amax, int ldp = 0) {
  if (ldp < ld) ldp = ld;                         // (ldp > ld: zero columns up to ldp; the plane stride follows)
  const long long rows = n / ld;
  hipLaunchKernelGGL(split_pairs_b16, dim3(grid_for(rows / 16 * ((ldp + 31) / 32), 4, 16384)), dim3(256), 0, st, src, rowscale, ld,
                     (int)rows, dst, rows * ldp, amax, ldp);
}

// every problem of the launch reads pre-split operands (gemm.hpp plane_rounds): interior 128 x 128 tiles, whole 32-deep
// rounds, 16-byte chunks of eight 16-bit pieces
static bool planes_eligible(const Launch& L) {
  for (int i = 0; i < L.nprob; ++i) {
    const Problem& p = L.p[i];
    if (!p.planes || p.xbf16 || p.nseg != 1 || p.M % 128 || (p.N % 128 && !(p.planes == 2 && p.n_padded && p.N % 4 == 0))) return false;
    const Segment& sg = p.seg[0];
    if (sg.a.row_div != 1 || sg.b.row_div != 1 || sg.a.n_mn < p.M || sg.b.n_mn < p.N || sg.K % 32) return false;
    if (sg.a.ld % 16 || sg.b.ld % 16) return false;
    if (sg.a.ld % 8 || sg.b.ld % 8 || (reinterpret_cast<uintptr_t>(sg.a.ptr) & 15) || (reinterpret_cast<uintptr_t>(sg.b.ptr) & 15)) return false;
    if ((p.a_pstride & 7) || (p.b_pstride & 7)) return false;
    if (p.colsum_out && sg.b.k_contig) return false;
    const int kper = ((sg.K + p.splits - 1) / p.splits + 31) / 32 * 32;
    if (sg.K % kper && (sg.K % kper) % 32) return false;
    const unsigned long long ext_a = sg.a.n_mn > sg.K ? sg.a.n_mn : sg.K, ext_b = sg.b.n_mn > sg.K ? sg.b.n_mn : sg.K;
    if (ext_a * sg.a.ld >= (1ull << 32) || ext_b * sg.b.ld >= (1ull << 32)) return false;
  }
  return L.nprob > 0;
}

// Estimated duration (us) of a grouped launch under tile configuration C: the larger of its longest tile running alone
// (rounds x the round time of a workgroup that has its CU to itself -- latency-bound) and of all tiles sharing the 256 CUs
// (rounds x the round time of a busy CU -- matrix-pipe-bound, padding of partial tiles included through the tile counts;
// plus a tail when the tiles do not all fit the chip at once).
// Round times measured with tools/gemm_bench.py / gemm_c5.py (profiles/round2_notes.md): a 128x128x32 round is 4096 MFMA
// cycles per wave and runs at 0.8 of that busy on the big-round instance (0.64 on the general loop), a 64x64x64 round
// 2048 cycles at 0.55, a 32x32x128 round 1024 cycles at 0.3.  (Round 1 picked the largest tile that gave >= 192 tiles:
// a 64 x 512 weight gradient over 25600 rows then ran as 64 tiles of 50 rounds -- 214 us for 3.3 GFLOP.)
template <class C>
static double launch_cost(Launch& t, bool big) {
  const bool isL = C::BM == 128, isM = C::BM == 64;
  const double busy = isL ? (big ? 2.2 : 2.8) : (isM ? 1.6 : 1.5);        // us per tile-round, CU shared
  const double alone = isL ? (big ? 2.5 : 3.7) : (isM ? 1.9 : 1.36);      // us per round, one workgroup per CU
  const double tile_alone = isL ? 4.0 : (isM ? 2.5 : 2.0), tile_busy = isL ? 1.0 : (isM ? 0.4 : 0.2);   // prologue + epilogue
  tile_up<C>(t);
  double lat = 0, thr = 0;
  for (int i = 0; i < t.nprob; ++i) {
    const Problem& p = t.p[i];
    double rounds = 0;
    for (int s = 0; s < p.nseg; ++s) {
      const int kper = (p.seg[s].K + p.splits - 1) / p.splits;
      rounds += (kper + C::BK - 1) / C::BK;
    }
    const double tiles = (double)p.tiles_m * p.tiles_n * p.splits;
    const double l = rounds * alone + tile_alone;
    lat = l > lat ? l : lat;
    thr += tiles * (rounds * busy + tile_busy);
  }
  thr /= 256.0;
  double tiles_all = 0;
  for (int i = 0; i < t.nprob; ++i) tiles_all += (double)t.p[i].tiles_m * t.p[i].tiles_n * t.p[i].splits;
  const double slots = 256.0 * (isL ? 2 : 3);
  // more tiles than resident workgroups: the last, partly filled wave of tiles adds about half a lone tile
  if (tiles_all > slots) return thr + 0.5 * lat;
  return lat > thr ? lat : thr;
}

// returns the chosen tile configuration (0 small, 1 medium, 2 large)
static int launch_group(Ctx& cx, Group& g, const char* name, int cfg = -1, unsigned long long* dbg = nullptr) {
  if (g.L.nprob == 0) return 0;
  if (g.L.nprob > 1) {
    // Longest tiles first: tiles are dispatched in index order and a launch ends with its last tiles, so the problems
    // go in descending order of the k extent one tile walks (config 5's decoder backward launch: the data gradient's
    // tiles run 96 rounds, a split of the weight gradient's 50 -- 1443 -> 1368 us with the data gradient first).
    auto klen = [](const Problem& p) {
      long long k = 0;
      for (int s = 0; s < p.nseg; ++s) k += (p.seg[s].K + p.splits - 1) / p.splits;
      return k;
    };
    for (int i = 1; i < g.L.nprob; ++i)            // insertion sort (stable; <= 8 problems)
      for (int j = i; j > 0 && klen(g.L.p[j]) > klen(g.L.p[j - 1]); --j) { const Problem t = g.L.p[j]; g.L.p[j] = g.L.p[j - 1]; g.L.p[j - 1] = t; }
  }
  g.L.dbg = dbg;
  g.L.aux_nblocks = g.L.aux.nblocks;
  bool planes = g.L.p[0].planes != 0;
  for (int i = 1; i < g.L.nprob; ++i) planes = planes && g.L.p[i].planes == g.L.p[0].planes;      // (one piece form per launch)
  const bool pairs = planes && g.L.p[0].planes == 2;
  if (planes) {
    if (!planes_eligible(g.L)) { cx.err = cx.err ? cx.err : GMVAE_E_DIMS; return 2; }      // (the host only marks problems it checked)
    cfg = 2;
  }
  if (cfg < 0) cfg = cx.force_cfg;
  if (cfg < 0) {
    Launch t = g.L;
    const double tl = launch_cost<CfgL>(t, big_eligible(t)), tm = launch_cost<CfgM>(t, false), ts = launch_cost<CfgS>(t, false);
    cfg = (tl <= tm && tl <= ts) ? 2 : (tm <= ts ? 1 : 0);
  }
  double fl = 0;
  for (int i = 0; i < g.L.nprob; ++i)
    for (int s = 0; s < g.L.p[i].nseg; ++s) fl += 2.0 * g.L.p[i].M * g.L.p[i].N * g.L.p[i].seg[s].K;
  unsigned long long* gs_buf = nullptr;
  if (const char* gn = getenv("GMVAE_GSTAMP_LAUNCH")) {
    if (!strcmp(gn, name) && !dbg) {
      static unsigned long long* buf = nullptr;
      if (!buf) hipMalloc(&buf, (size_t)(1 << 18) * 8 * 8);
      hipMemsetAsync(buf, 0, (size_t)(1 << 18) * 8 * 8, cx.st);
      gs_buf = buf;
      g.L.dbg = buf;
    }
  }
  int tiles;
  if (cfg == 2) {
    tiles = g.L.total_tiles = tile_up<CfgL>(g.L);
    const bool no_big = getenv("GMVAE_NO_BIG") != nullptr;      // diagnostic / A-B: the general loop
    // f16 pairs: 256-row tiles (one 8-wave workgroup per CU, 0.75 of the LDS-DMA bytes per product) where every problem's rows
    // allow and no tile ends in the Bernoulli epilogue -- with one workgroup per CU nothing multiplies while a tile's epilogue
    // runs: config 5's backward launch (tiles of 100 - 192 rounds) 538 -> 513 us, its forward launch (32 rounds + the
    // Bernoulli epilogue) 309 -> 347.  GMVAE_PAIRS_BM=128 / 256 forces either.
    bool x256 = pairs, store_only = true;
    for (int i = 0; i < g.L.nprob && x256; ++i) {
      x256 = g.L.p[i].M % 256 == 0 && g.L.aux.nblocks == 0;
      store_only = store_only && g.L.p[i].epi == EPI_STORE;
    }
    if (const char* e = getenv("GMVAE_PAIRS_BM")) x256 = x256 && atoi(e) == 256;
    else x256 = x256 && store_only;
    if (x256) {
      tiles = g.L.total_tiles = tile_up<CfgX>(g.L);
      hipLaunchKernelGGL((gemm_grouped<CfgX, 3>), dim3(tiles), dim3(CfgX::THREADS), 0, cx.st, g.L);
    } else if (pairs) hipLaunchKernelGGL((gemm_grouped<CfgL, 3>), dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
    else if (planes) hipLaunchKernelGGL((gemm_grouped<CfgL, 2>), dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
    else if (!no_big && big_eligible(g.L)) hipLaunchKernelGGL((gemm_grouped<CfgL, 1>), dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
    else hipLaunchKernelGGL(gemm_grouped<CfgL>, dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
  } else if (cfg == 3) {
    tiles = g.L.total_tiles = tile_up<CfgM1>(g.L);
    hipLaunchKernelGGL(gemm_grouped<CfgM1>, dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
  } else if (cfg == 1) {
    tiles = g.L.total_tiles = tile_up<CfgM>(g.L);
    hipLaunchKernelGGL(gemm_grouped<CfgM>, dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
  } else {
    tiles = g.L.total_tiles = tile_up<CfgS>(g.L);
    hipLaunchKernelGGL(gemm_grouped<CfgS>, dim3(tiles + g.L.aux.nblocks), dim3(kThreads), 0, cx.st, g.L);
  }
  cx.check();
  cx.mark(name, fl);
  if (gs_buf) {                                   // GMVAE_GSTAMP_LAUNCH=<name>: per-tile phase medians of this launch (eager only)
    hipStreamSynchronize(cx.st);
    const int nb = tiles + g.L.aux.nblocks;
    std::vector<unsigned long long> h((size_t)nb * 8);
    hipMemcpy(h.data(), gs_buf, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ph[7];
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = g.L.aux.nblocks; b < nb; ++b) {
      const unsigned long long* r = &h[(size_t)b * 8];
      if (!r[0] || !r[4]) continue;
      t0 = r[0] < t0 ? r[0] : t0; t1 = r[4] > t1 ? r[4] : t1;
      ph[0].push_back((double)(r[1] - r[0])); ph[1].push_back((double)(r[2] - r[1])); ph[2].push_back((double)(r[3] - r[2]));
      ph[3].push_back((double)(r[4] - r[3])); ph[4].push_back((double)(r[4] - r[0]));
      if (r[6] > r[3] && r[7] > r[6]) { ph[5].push_back((double)(r[6] - r[3])); ph[6].push_back((double)(r[7] - r[6])); }
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2] / 100.0; };
    fprintf(stderr, "[gstamp] %s cfg %d tiles %d: span %.1f us | per tile (median us): prologue %.2f loop %.2f stage %.2f epilogue %.2f total %.2f\n",
            name, cfg, tiles, (double)(t1 - t0) / 100.0, med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]));
    if (!ph[5].empty()) fprintf(stderr, "[gstamp]    epilogue: option loads landed +%.2f us, first half of the passes +%.2f us\n", med(ph[5]), med(ph[6]));
  }
  return cfg;
}
static int cfg_bn(int cfg) { return cfg == 2 ? CfgL::BN : ((cfg == 1 || cfg == 3) ? CfgM::BN : CfgS::BN); }

static int grid_for(long long items, int per_block, int cap) {
  long long g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// -------------------------------------------------------------- the step
struct StepArgs {
  const GmvaeDims* d;
  int model;
  const uint8_t* x;
  const float *eps, *u, *params;
  float* grads;        // [P_pad + TAIL] (backward) or nullptr
  float* tail;         // forward-only tail destination
  float *row_terms, *z_out, *y_out, *logits_out;
  void* workspace;
  uint64_t seed, step;
  uint64_t* step_dev;
  bool backward;
  // when non-null the fused schedule ends with ONE kernel doing slab reduce + loss tail + TF-Adam
  float *adam_p = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  float lr = 0.f, beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-8f;
  // the previous step of the same graph ran finalize_adam with the image scatter on this workspace and nothing
  // touched the parameters since: the step may skip its first launch (mega_fwd_bwd runs the first layer itself)
  bool imgs_ready = false;
  bool want_spans = false;     // measurement: every launch of the step records per-workgroup wall-clock stamps
  int span_slot = 0;           // ... into this slot of WS::spans (two consecutive steps can be stamped)
  bool dp_images = false;      // data-parallel graph: the Adam launch after the all-reduce scatters the weight images
  float* tail_log = nullptr;   // train graph: this step's slot of the per-step tail log (may be null)
};

static void rowk(Ctx& cx, const char* name) {
  cx.check();
  cx.mark(name, 0);
}

// Where every parameter that mega_fwd_bwd reads from an LDS image lives: as copy tasks (for the auxiliary
// workgroups of the first launch) and as a parameter-index map (for finalize_adam's scatter of the updated values).
struct ImgPlan {
  ImgTask task[kMaxImgTasks];
  int nt = 0;
  ImgMap map[kMaxImgMap];
  int nmap = 0, lo = 0, hi = 0;
  bool map_ok = true;
};
static void plan_images(const GmvaeDims& d, int model, const Layout& L, const WS& w, const MegaLay& ml, const float* P,
                        ImgPlan& pl) {
  const bool gm = model == GMVAE_MODEL_GMVAE, gmp = model == GMVAE_MODEL_VAE_GMP;
  const int K = d.K, Lz = d.L, D = d.D, H = d.hidden[0];
  const NetL &E = gm ? L.ency : L.enc, &G = L.encg, &Dn = L.dec;
  float* im = w.img_m;
  auto add_map = [&](long long begin, long long n, int cols, int kind, int base, int ld, int which, int chunk = -1) {
    if (pl.nmap >= kMaxImgMap || (unsigned long long)n * (unsigned long long)cols >= (1ull << 32)) { pl.map_ok = false; return; }
    ImgMap& m = pl.map[pl.nmap++];
    m.begin = (int)begin; m.end = (int)(begin + n); m.cols = cols; m.kind = kind; m.base = base; m.ld = ld;
    m.cw = kCW; m.chunk = chunk < 0 ? ml.chunk : chunk; m.magic = (unsigned)((1ull << 32) / (unsigned)cols) + 1u; m.which = which;
  };
  auto task = [&](float* dst, int ld, const float* src, int rows, int cols) {      // dense source rows
    ImgTask& t = pl.task[pl.nt++];
    t.dst = dst; t.ld = ld; t.src = src; t.rows = rows; t.cols = cols; t.src_ld = cols; t.trans = 0;
    add_map(src - P, (long long)rows * cols, cols, 0, (int)(dst - im), ld, 0);
  };
  if (gm) {
    task(im + ml.W_y1, ml.ldY1, P + E.w[1], H, K);
    task(im + ml.W_g0y, ml.ldG0, P + G.w[0] + (uint64_t)D * H, K, H);
    task(im + ml.W_p, ml.ldP, P + L.prior.w[0], K, 2 * Lz);
    task(im + ml.W_g1, ml.ldG1, P + G.w[1], H, 2 * Lz);
    task(im + ml.b_y1, K, P + E.b[1], 1, K);
    task(im + ml.b_g0, H, P + G.b[0], 1, H);
    task(im + ml.b_p, 2 * Lz, P + L.prior.b[0], 1, 2 * Lz);
    task(im + ml.b_g1, 2 * Lz, P + G.b[1], 1, 2 * Lz);
  } else {                       // the encoder's second layer takes the q-head slot
    task(im + ml.W_g1, ml.ldG1, P + E.w[1], H, 2 * Lz);
    task(im + ml.b_g1, 2 * Lz, P + E.b[1], 1, 2 * Lz);
    if (gmp) {
      task(im + ml.M_loc, ml.ldM, P + L.loc, K, Lz);
      task(im + ml.M_raw, ml.ldM, P + L.rawscale, K, Lz);
      task(im + ml.M_mix, K, P + L.mixlog, 1, K);
    }
  }
  task(im + ml.b_y0, H, P + E.b[0], 1, H);
  task(im + ml.W_d0, ml.ldD0, P + Dn.w[0], Lz, H);
  task(im + ml.b_d0, H, P + Dn.b[0], 1, H);
  for (int c = 0; c < ml.nch; ++c) {             // decoder chunk images: [H rows of Wd1 | bias row]
    const int nc = (D - c * kCW) < kCW ? (D - c * kCW) : kCW;
    ImgTask& t0 = pl.task[pl.nt++];
    t0.dst = w.dimg + (uint64_t)c * ml.chunk; t0.ld = ml.ldc; t0.src = P + Dn.w[1] + c * kCW; t0.rows = H; t0.cols = nc;
    t0.src_ld = D; t0.trans = 0;
    ImgTask& t1 = pl.task[pl.nt++];
    t1.dst = w.dimg + (uint64_t)c * ml.chunk + H * ml.ldc; t1.ld = nc; t1.src = P + Dn.b[1] + c * kCW; t1.rows = 1;
    t1.cols = nc; t1.src_ld = nc; t1.trans = 0;
  }
  add_map(Dn.w[1], (long long)H * D, D, 1, 0, ml.ldc, 1);
  add_map(Dn.b[1], D, D, 1, H * ml.ldc, ml.ldc, 1);
  if (mega2_ok(d, model) && w.img2f) {
    // the steady-state launch is mega2_fwd_bwd: the optimiser scatters into ITS operand images instead (every weight of a
    // matrix product twice: forward and transposed orientation; kernels.hpp img_dst kinds 2..6)
    pl.nmap = 0; pl.map_ok = true;
    const long long Wg0y = (long long)G.w[0] + (long long)D * H;
    add_map(E.b[0], H, H, 0, M2::b_y0, H, 2);
    add_map(E.b[1], K, K, 0, M2::b_y1, K, 2);
    add_map(G.b[0], H, H, 0, M2::b_g0, H, 2);
    add_map(L.prior.b[0], 2 * Lz, 2 * Lz, 0, M2::b_p, 2 * Lz, 2);
    add_map(G.b[1], 2 * Lz, 2 * Lz, 0, M2::b_g1, 2 * Lz, 2);
    add_map(Dn.b[0], H, H, 0, M2::b_d0, H, 2);
    add_map(E.w[1], (long long)H * K, K, 2, M2::Wy1f, 16, 2);
    add_map(E.w[1], (long long)H * K, K, 3, M2::Wy1b, 64, 3);
    add_map(Wg0y, (long long)K * H, H, 2, M2::Wg0yf, 64, 2);
    add_map(Wg0y, (long long)K * H, H, 3, M2::Wg0yb, 16, 3);
    add_map(L.prior.w[0], (long long)K * 2 * Lz, 2 * Lz, 2, M2::Wpf, 128, 2);
    add_map(L.prior.w[0], (long long)K * 2 * Lz, 2 * Lz, 3, M2::Wpb, 16, 3);
    add_map(G.w[1], (long long)H * 2 * Lz, 2 * Lz, 2, M2::Wg1f, 128, 2);
    add_map(G.w[1], (long long)H * 2 * Lz, 2 * Lz, 3, M2::Wg1b, 64, 3);
    add_map(Dn.w[0], (long long)Lz * H, H, 2, M2::Wd0f, 64, 2);
    add_map(Dn.w[0], (long long)Lz * H, H, 3, M2::Wd0b, 64, 3);
    add_map(Dn.w[1], (long long)H * D, D, 4, M2::dF, M2::DC, 4, M2::dFq);
    add_map(Dn.w[1], (long long)H * D, D, 5, M2::dB, 64, 4, M2::dBq);
    add_map(Dn.b[1], D, D, 6, M2::dbias, 0, 4, M2::dbq);
  }
  if (mega2v_ok(d, model) && w.img2f) {
    // mega2v_fwd_bwd's operand images (mega2v.hpp M2V): the same kinds; the decoder layer dealt to seven workgroups (7..9)
    const int vk = mega2v_kind(d, model);
    pl.nmap = 0; pl.map_ok = true;
#define GMVAE_MV(f) (vk == 1 ? MV0::f : MV1::f)
    add_map(E.b[0], H, H, 0, GMVAE_MV(b_e0), H, 2);
    add_map(E.b[1], 2 * Lz, 2 * Lz, 0, GMVAE_MV(b_g1), 2 * Lz, 2);
    add_map(Dn.b[0], H, H, 0, GMVAE_MV(b_d0), H, 2);
    add_map(E.w[1], (long long)H * 2 * Lz, 2 * Lz, 2, GMVAE_MV(Wg1f), GMVAE_MV(L2P), 2);
    add_map(E.w[1], (long long)H * 2 * Lz, 2 * Lz, 3, GMVAE_MV(Wg1b), 64, 3);
    add_map(Dn.w[0], (long long)Lz * H, H, 2, GMVAE_MV(Wd0f), 64, 2);
    add_map(Dn.w[0], (long long)Lz * H, H, 3, GMVAE_MV(Wd0b), GMVAE_MV(LP), 3);
    add_map(Dn.w[1], (long long)H * D, D, 7, GMVAE_MV(dF), GMVAE_MV(DC), 4, GMVAE_MV(dFq));
    add_map(Dn.w[1], (long long)H * D, D, 8, GMVAE_MV(dB), 64, 4, GMVAE_MV(dBq));
    add_map(Dn.b[1], D, D, 9, GMVAE_MV(dbias), 0, 4, GMVAE_MV(dbq));
    if (gmp) {
      add_map(L.loc, (long long)K * Lz, Lz, 0, MV1::M_loc, MV1::ldM, 2);
      add_map(L.rawscale, (long long)K * Lz, Lz, 0, MV1::M_raw, MV1::ldM, 2);
      add_map(L.mixlog, K, K, 0, MV1::M_mix, K, 2);
    }
#undef GMVAE_MV
  }
  pl.lo = 1 << 30; pl.hi = 0;
  for (int i = 0; i < pl.nmap; ++i) {
    pl.lo = pl.map[i].begin < pl.lo ? pl.map[i].begin : pl.lo;
    pl.hi = pl.map[i].end > pl.hi ? pl.map[i].end : pl.hi;
  }
}

// end of the fused schedules: slab reduce + loss tail (+ TF-Adam in the graph path)
static int finish_fused(Ctx& cx, const StepArgs& a, const Layout& L, WS& w, float* tail, int NS, int B,
                        const SlabX* sxp = nullptr) {
  SlabX sx;
  memset(&sx, 0, sizeof(sx));
  if (sxp) sx = *sxp;
  hipStream_t st = cx.st;
  const GmvaeDims& d = *a.d;
  const long long PP = (long long)L.P_pad;
  float* sl = w.slabs;
  const bool gmp = a.model == GMVAE_MODEL_VAE_GMP;
  const float* nent = a.model == GMVAE_MODEL_GMVAE ? w.nent : nullptr;
  if ((!gmp || mega_ok(d, a.model)) && (!a.adam_p || a.step_dev)) {   // slab reduce + loss tail (+ TF-Adam) in one launch
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.slabs = sl; fa.nslab = NS; fa.P = PP; fa.grads = a.grads; fa.p = a.adam_p; fa.m = a.adam_m; fa.v = a.adam_v;
    fa.lr = a.lr; fa.b1 = a.beta1; fa.b2 = a.beta2; fa.eps = a.epsilon; fa.do_adam = a.adam_p ? 1 : 0; fa.count = (float)B;
    fa.logw = w.logw; fa.logpx = w.logpx; fa.logq = w.logq; fa.logp = w.logp; fa.nent = nent;
    fa.tail = tail; fa.B = B; fa.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev); fa.tail_log = a.tail_log;
    fa.nmap = 0; fa.map_lo = fa.map_hi = 0; fa.epoch_word = nullptr;
    fa.err_word = (mega_ok(d, a.model) && w.sync) ? w.sync + 1 : nullptr;
    fa.sx = sx;
    if (gmp) {                               // (mega schedule only: one partial per panel)
      const int KLp = (int)pad4((uint64_t)d.K * d.L);
      fa.gmp_part = w.gmp_part; fa.gmp_n = (B + kPanel - 1) / kPanel; fa.gmp_len = 2 * KLp + (int)pad4(d.K); fa.gmp_off = (long long)L.loc;
    }
    fa.span = (a.want_spans && w.spans) ? w.spans + (size_t)a.span_slot * 3 * 2048 * 2 + 2048 * 2 : nullptr;
    if (mega_ok(d, a.model) && a.adam_p && a.adam_p == a.params) {      // the next step's weight images ride on the update
      const MegaLay ml = mega_lay(d.hidden[0], d.L, d.K, d.D, a.model);
      ImgPlan pl;
      plan_images(d, a.model, L, w, ml, a.params, pl);
      fa.epoch_word = w.sync;
      if (pl.map_ok && ml.fl_ok) {
        fa.nmap = pl.nmap; fa.map_lo = pl.lo; fa.map_hi = pl.hi;
        fa.img[0] = w.img_m; fa.img[1] = w.dimg; fa.img[2] = w.img2f; fa.img[3] = w.img2b; fa.img[4] = w.dimg2;
        int n = 0;
        for (int i = 0; i < pl.nmap; ++i) {
          const ImgMap& mp = pl.map[i];
          const int rows = mp.cols > 0 ? (mp.end - mp.begin) / mp.cols : 0;
          if (mp.kind == 4 && !fa.quad_blocks && rows % 4 == 0) {   // -> quad blocks (kernels.hpp)
            fa.q_begin = mp.begin; fa.q_end = mp.end; fa.q_rows = rows; fa.q_cols = mp.cols; fa.q_base = mp.base; fa.q_ld = mp.ld;
            fa.q_chunk = mp.chunk; fa.q_which = mp.which;
            fa.quad_blocks = ((rows / 4) * mp.cols + 255) / 256;
            continue;
          }
          if (fa.quad_blocks && !fa.q2_kind && mp.begin == fa.q_begin && mp.end == fa.q_end) {       // the same tensor's other image
            fa.q2_kind = mp.kind; fa.q2_base = mp.base; fa.q2_ld = mp.ld; fa.q2_chunk = mp.chunk; fa.q2_which = mp.which;
            continue;
          }
          fa.map[n] = mp; fa.mbegin[n] = mp.begin; fa.mend[n] = mp.end; ++n;
        }
        fa.nmap = n;
      }
    }
    hipLaunchKernelGGL(finalize_adam, dim3((unsigned)((PP / 4 + 255) / 256) + 1 + fa.quad_blocks), dim3(256), 0, st, fa);
    rowk(cx, a.adam_p ? "finalize_adam" : "finalize_grads+loss_tail");
    return cx.err;
  }
  hipLaunchKernelGGL(loss_tail, dim3(1), dim3(1024), 0, st, w.logw, w.logpx, w.logq, w.logp, nent, (float*)nullptr, tail,
                     B, 1, a.step_dev);
  rowk(cx, "loss_tail");
  const int KLp = (int)pad4((uint64_t)d.K * d.L);
  hipLaunchKernelGGL(finalize_grads, dim3((unsigned)((PP / 4 + 255) / 256)), dim3(256), 0, st, sl, NS, PP, a.grads,
                     gmp ? w.gmp_part : (const float*)nullptr, (B + kPanel - 1) / kPanel,
                     gmp ? 2 * KLp + (int)pad4(d.K) : 0, (long long)L.loc, sx, AdamTail{});
  rowk(cx, "finalize_grads");
  if (a.adam_p && a.step_dev) {           // VAE_GMP in the graph path: its prior partials need finalize_grads first
    hipLaunchKernelGGL(adam_tf, dim3((unsigned)((PP / 4 + 255) / 256)), dim3(256), 0, st, a.adam_p, a.adam_m, a.adam_v,
                       a.grads, PP, a.lr, a.beta1, a.beta2, a.epsilon, (uint64_t)0, a.step_dev, 1.f / (float)B,
                       (const float*)nullptr, (const float*)tail);
    rowk(cx, "adam_tf");
  }
  if (a.tail_log) hipMemcpyAsync(a.tail_log, tail, GMVAE_TAIL * sizeof(float), hipMemcpyDeviceToDevice, st);
  return cx.err;
}

// The weight tensors of the single-hidden-layer step as dw_adam / mega3_step / adam_tiles tile them (dwadam.hpp DwTensor): operand
// pointers, flat parameter offsets, where the optimizer also leaves the updated values (the operand images of plan_images)
static void dw_tensors(const GmvaeDims& d, const int model, const Layout& L, const WS& w, const uint8_t* x, const ImgPlan& pl, DwArgs& da) {
  const bool gm = model == GMVAE_MODEL_GMVAE;
  const int B = d.B, K = d.K, Lz = d.L, D = d.D, H = d.hidden[0];
  (void)B;
  const NetL &E = gm ? L.ency : L.enc, &G = L.encg, &Dn = L.dec;
  struct { const uint8_t* x; } a = {x};
  auto add = [&](const void* A, bool u8, int lda, const float* dY, int ldy, int M, int N, uint64_t w_off, long long b_off,
                 int mu = 4) {
    DwTensor& T = da.t[da.ntens];
    T.A = A; T.a_u8 = u8 ? 1 : 0; T.lda = lda; T.dY = dY; T.ldy = ldy; T.M = M; T.N = N; T.w_off = (int)w_off; T.b_off = (int)b_off;
    T.mu = u8 ? 4 : mu;
    T.tiles_n = (N + 15) / 16; T.tile_begin = da.total_tiles;
    da.total_tiles += ((M + 16 * T.mu - 1) / (16 * T.mu)) * T.tiles_n;
    T.bk = T.k1 = T.k2 = -1;
    for (int i = 0; i < pl.nmap; ++i) {      // where the optimizer also has to leave the updated values (mega2's operand images)
      const ImgMap& mp = pl.map[i];
      if (mp.begin == (int)w_off && mp.end == (int)(w_off + (uint64_t)M * N)) {
        if (T.k1 < 0) { T.k1 = mp.kind; T.base1 = mp.base; T.ld1 = mp.ld; T.chunk1 = mp.chunk; T.which1 = mp.which; }
        else { T.k2 = mp.kind; T.base2 = mp.base; T.ld2 = mp.ld; T.chunk2 = mp.chunk; T.which2 = mp.which; }
      }
      if (b_off >= 0 && mp.begin == (int)b_off && mp.end == (int)(b_off + N)) { T.bk = mp.kind; T.bbase = mp.base; T.bchunk = mp.chunk; T.bwhich = mp.which; }
    }
    // the row-interleaved image (kinds 2 / 4) first: it leaves as one 16-byte store per thread
    if (T.k2 == 2 || T.k2 == 4 || T.k2 == 7) {
      const int k = T.k1, b_ = T.base1, l_ = T.ld1, c_ = T.chunk1, w_ = T.which1;
      T.k1 = T.k2; T.base1 = T.base2; T.ld1 = T.ld2; T.chunk1 = T.chunk2; T.which1 = T.which2;
      T.k2 = k; T.base2 = b_; T.ld2 = l_; T.chunk2 = c_; T.which2 = w_;
    }
    da.ntens++;
  };
  const int K4 = (int)pad4(K);
  const int mu = 2;
  constexpr int kDwd1Mu = 2;       // (the decoder output layer as [16 x 16] tiles: measured slower, dwadam.hpp)
  // (32-row tiles for the fp32 problems: more, lighter workgroups -- 240 <= 256 CUs at the default sizes -- and the
  //  decoder output layer's longer epilogue, two operand images, no longer ends the launch)
  if (gm) {
    add(a.x, true, D, w.dbuf[2], H, D, H, E.w[0], (long long)E.b[0]);                                 // dWy0 (+ dby0)
    add(a.x, true, D, w.dbuf[1], H, D, H, G.w[0], (long long)G.b[0]);                                 // dWg0[x] (+ dbg0)
    add(w.hd[1], false, H, w.g, D, H, D, Dn.w[1], (long long)Dn.b[1], kDwd1Mu);                            // dWd1 (+ dbd1)
    add(w.y, false, K4, w.dbuf[1], H, K, H, G.w[0] + (uint64_t)D * H, -1, mu);                        // dWg0[y]
    add(w.he[1], false, H, w.dlogits, K4, H, K, E.w[1], (long long)E.b[1], mu);                       // dWy1
    add(w.y, false, K4, w.dpp, 2 * Lz, K, 2 * Lz, L.prior.w[0], (long long)L.prior.b[0], mu);         // dWp
    add(w.hg[1], false, H, w.dqp, 2 * Lz, H, 2 * Lz, G.w[1], (long long)G.b[1], mu);                  // dWg1
  } else {
    add(a.x, true, D, w.dbuf[1], H, D, H, E.w[0], (long long)E.b[0]);                                 // dWe0 (+ dbe0)
    add(w.hd[1], false, H, w.g, D, H, D, Dn.w[1], (long long)Dn.b[1], kDwd1Mu);                            // dWd1 (+ dbd1)
    add(w.he[1], false, H, w.dqp, 2 * Lz, H, 2 * Lz, E.w[1], (long long)E.b[1], mu);                  // dWe1
  }
  add(w.z, false, Lz, w.dbuf[0], H, Lz, H, Dn.w[0], (long long)Dn.b[0], mu);                          // dWd0
}

// ---- the mega schedule (all three models): first-layer split-K GEMM (+ aux) -> mega_fwd_bwd -> all dW -> finish
static int run_step_mega(Ctx& cx, const StepArgs& a, const Layout& L, WS& w, const float* eps, const float* u,
                         float* gen_eps, float* gen_u) {
  const GmvaeDims& d = *a.d;
  const int model = a.model;
  const bool gm = model == GMVAE_MODEL_GMVAE, gmp = model == GMVAE_MODEL_VAE_GMP;
  const int B = d.B, K = d.K, Lz = d.L, D = d.D, H = d.hidden[0];
  const float* P = a.params;
  hipStream_t st = cx.st;
  const int NSF = fwd_splits(D), NS = dw_splits(B);
  const long long PP = (long long)L.P_pad;
  float* sl = w.slabs;
  float* tail = a.grads + L.P_pad;
  const NetL &E = gm ? L.ency : L.enc, &G = L.encg, &Dn = L.dec;
  const MegaLay ml = mega_lay(H, Lz, K, D, model);
  const int H2 = gm ? 2 * H : H;
  // The first launch (first layer as split-K partials + noise + weight images) is skipped when the previous step
  // of the same graph left the images behind (finalize_adam) and the launch below can run the first layer itself:
  // a panel's 4 workgroups must all be resident for their exchange, i.e. one workgroup per CU.
  const int n_cu = device_cus();
  const int vk = (mega2v_ok(d, model) && w.img2f) ? mega2v_kind(d, model) : 0;
  const int Qm = vk ? 7 : mega_q(d);
  const int np_grid = (!vk && mega2_ok(d, model)) ? (((B + kPanel - 1) / kPanel + 1) & ~1) : (B + kPanel - 1) / kPanel;   // (mega2 pairs panels)
  bool fl = ml.fl_ok && (Qm == 4 || vk) && np_grid * Qm <= n_cu &&
            (a.adam_p == a.params || a.dp_images) && a.step_dev && gen_eps && w.xfl && !getenv("GMVAE_NO_FL") && !sched_safe(d);
  if (fl && !a.imgs_ready) {
    // first step of a train graph / an eager step: the weight images straight from the parameters (kernels.hpp img_build),
    // then the same launches as every later step
    ImgPlan pl;
    plan_images(d, model, L, w, ml, P, pl);
    if (!pl.map_ok) fl = false;
    else {
      ImgScatter sc;
      memset(&sc, 0, sizeof(sc));
      sc.nmap = pl.nmap; sc.lo = pl.lo; sc.hi = pl.hi; sc.epoch_word = w.sync;
      sc.img[0] = w.img_m; sc.img[1] = w.dimg; sc.img[2] = w.img2f; sc.img[3] = w.img2b; sc.img[4] = w.dimg2;
      for (int i = 0; i < pl.nmap; ++i) { sc.map[i] = pl.map[i]; sc.mbegin[i] = pl.map[i].begin; sc.mend[i] = pl.map[i].end; }
      hipLaunchKernelGGL(img_build, dim3((unsigned)((L.P_pad / 4 + 255) / 256)), dim3(256), 0, st, P, (long long)L.P_pad, sc);
      cx.check();
      cx.mark("img_build", 0);
    }
  }
  if (!fl) {  // P1: first layer(s) over the uint8 batch as single-round split-K partials + auxiliary workgroups
    Group g;
    Problem p0 = p_nn(a.x, true, D, P + E.w[0], H, B, H, D, w.s1, H2, nullptr, false);
    p0.splits = NSF; p0.split_stride = (long long)B * H2;
    g.add(p0);
    if (gm) {
      Problem p1 = p_nn(a.x, true, D, P + G.w[0], H, B, H, D, w.s1 + H, H2, nullptr, false);
      p1.splits = NSF; p1.split_stride = (long long)B * H2;
      g.add(p1);
    }
    Aux& ax = g.L.aux;
    ax.eps = gen_eps; ax.u = gen_u; ax.n_rows = (unsigned long long)B * d.S; ax.nL = Lz; ax.nK = K; ax.row_base = d.row0 * d.S;
    ax.seed = a.seed; ax.step = a.step;
    ax.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev);
    ax.epoch_word = w.sync;
    ax.noise_blocks = (int)((noise_items(gen_eps, gen_u, ax.n_rows, Lz, K) + kThreads - 1) / kThreads);
    ImgPlan pl;
    plan_images(d, model, L, w, ml, P, pl);
    const int nt = pl.nt;
    for (int i = 0; i < nt; ++i) ax.task[i] = pl.task[i];
    ax.ntasks = nt;
    ax.nblocks = ax.noise_blocks + nt;
    launch_group(cx, g, "fwd_x_first_layers_splitk+aux", 0, getenv("GMVAE_STAMPS") ? w.gstamps : nullptr);
  }
  bool m2_ran = false;
  // mega3_step (mega3.hpp): at mega2_fwd_bwd's sizes the weight-gradient tiles + TF-Adam run in the SAME launch -- its
  // arguments are kept until the tile list below is built
  MegaArgs c3;
  bool fuse_pending = false, fusev_pending = false;
  int vkind = 0;
  double m2_flops = 0;
  const bool dw_upd_ = a.adam_p && a.adam_p == a.params;
  {  // the whole per-row forward + backward in one launch
    MegaArgs c;
    memset(&c, 0, sizeof(c));
    c.model = model;
    c.B = B; c.H = H; c.L = Lz; c.K = K; c.D = D; c.NS = NSF;
    c.c = d.raw_sigma_bias; c.smin = d.sigma_min; c.invT = 1.f / d.temperature; c.gen_bias = d.gen_bias_init;
    c.s1 = w.s1; c.img = w.img_m; c.dimg = w.dimg; c.x = a.x; c.eps = eps; c.u = u;
    c.hy1 = w.he[1]; c.y = w.y; c.hg1 = w.hg[1]; c.z = w.z; c.hd1 = w.hd[1]; c.g = w.g;
    c.dhd1 = w.dbuf[0]; c.dqp = w.dqp; c.dpp = w.dpp; c.dhg1 = w.dbuf[1]; c.dlogits = w.dlogits; c.dhy1 = w.dbuf[2];
    c.nent = w.nent; c.logq = w.logq; c.logp = w.logp; c.logpx = w.logpx; c.logw = w.logw;
    c.gmp_part = w.gmp_part;
    c.lay = ml;
    c.fl = fl ? 1 : 0;
    c.w0a = P + E.w[0]; c.w0b = gm ? P + G.w[0] : nullptr; c.xfl = w.xfl; c.seed = a.seed; c.row0 = d.row0;
    c.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev);
    c.Q = Qm; c.xchg = w.xchg; c.epoch_word = w.sync; c.err_word = w.sync + 1;
    c.dbg = getenv("GMVAE_STAMPS") ? w.stamps : nullptr;
    c.span = a.want_spans ? w.spans + (size_t)a.span_slot * 3 * 2048 * 2 : nullptr;
    c.fine = getenv("GMVAE_STAMPS") ? atoi(getenv("GMVAE_STAMPS")) : 0;
    // the reference's default sizes (run_gmvae.py: latent 64, hidden 64, K 10; MNIST D 784) run a specialised instance
    typedef void (*MegaFn)(const MegaArgs);
    const bool spec = H == 64 && Lz == 64 && K == 10 && D == 784 && (gm || gmp);
    // BASELINE configs[0] (scripts/vae.py defaults at latent_size 2 on MNIST): the plain VAE's instance (round 4; it ran the
    // generic one before: ~1100 instructions of index set-up and ~100 spilled scalars per launch)
    const bool spec_vae = model == GMVAE_MODEL_VAE && H == 64 && Lz == 2 && K == 1 && D == 784;
    const MegaFn fns[8] = {mega_fwd_bwd<0, 0, 0, 0, -1, 0>,      mega_fwd_bwd<0, 0, 0, 0, -1, 1>,
                           mega_fwd_bwd<64, 64, 10, 784, 2, 0>, mega_fwd_bwd<64, 64, 10, 784, 2, 1>,
                           mega_fwd_bwd<64, 64, 10, 784, 1, 0>, mega_fwd_bwd<64, 64, 10, 784, 1, 1>,
                           mega_fwd_bwd<64, 2, 1, 784, 0, 0>,   mega_fwd_bwd<64, 2, 1, 784, 0, 1>};
    const MegaFn fn = fns[(spec ? (gm ? 2 : 4) : (spec_vae ? 6 : 0)) + (fl ? 1 : 0)];
    if (getenv("GMVAE_TRACE"))
      fprintf(stderr, "[gmvae] mega_fwd_bwd: model %d B %d first_layer_inside %d workgroups_per_panel %d specialised %d\n", model, B,
              (int)fl, Qm, (int)(spec || spec_vae));
    static bool mattr[64];
    if (first_on_device(mattr)) {
      for (int i = 0; i < 8; ++i)
        hipFuncSetAttribute(reinterpret_cast<const void*>(fns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const bool m2 = fl && mega2_ok(d, model) && w.img2f;
    const bool m2v = fl && vk != 0;
    m2_ran = m2 || m2v;
    if (m2) {
      c.img2f = w.img2f; c.img2b = w.img2b; c.dimg2 = w.dimg2;
      c.lr = a.lr; c.b1 = a.beta1; c.b2 = a.beta2;
      c.lr_t_out = (a.adam_p && a.adam_p == a.params && a.step_dev) ? reinterpret_cast<float*>(w.sync + 2) : nullptr;
      static bool m2attr[64];
      if (first_on_device(m2attr)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mega2_fwd_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
      // (B = 1024: 256 workgroups for the 241 tile slots and the register-resident contraction forms; measured at smaller
      //  batches -- fewer workgroups than slots -- the one-launch form loses: B = 512 43.2 against 33.8 us, profiles/round5_notes.md.
      //  GMVAE_FUSE=1 forces it for any batch: tools/fuse_check.py)
      fuse_pending = (dw_upd_ || a.dp_images) && a.step_dev && !getenv("GMVAE_NO_FUSE") && (B == 1024 || getenv("GMVAE_FUSE")) &&
                     device_is_gfx950();
      c3 = c;
      // (an even number of panels: the first layer works on pairs of them, mega2.hpp)
      if (!fuse_pending)
        hipLaunchKernelGGL(mega2_fwd_bwd, dim3((((B + kPanel - 1) / kPanel + 1) & ~1) * 4), dim3(kMT), (size_t)M2::total * sizeof(float), st, c);
    } else if (m2v) {
      c.img2f = w.img2f; c.img2b = w.img2b; c.dimg2 = w.dimg2;
      c.lr = a.lr; c.b1 = a.beta1; c.b2 = a.beta2;
      c.lr_t_out = (a.adam_p && a.adam_p == a.params && a.step_dev) ? reinterpret_cast<float*>(w.sync + 2) : nullptr;
      static bool m2vattr[64];
      if (first_on_device(m2vattr)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mega2v_fwd_bwd<0, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(mega2v_fwd_bwd<1, 64, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
      const unsigned grid = (unsigned)((B + kPanel - 1) / kPanel * 7);
      // (mega3v_step: the step as ONE launch where the optimizer runs on this device; the data-parallel graph keeps two launches)
      // (its grid is 256 workgroups whatever the batch -- the role-less workers --, one per CU: all of them must be resident)
      fusev_pending = dw_upd_ && a.step_dev && !getenv("GMVAE_NO_FUSE") && grid <= 256 && n_cu >= 256 && device_is_gfx950();
      vkind = vk;
      c3 = c;
      if (fusev_pending) { /* launched below, with the tile list */ }
      else if (vk == 1) hipLaunchKernelGGL((mega2v_fwd_bwd<0, 2, 1>), dim3(grid), dim3(kMT), (size_t)MV0::total * sizeof(float), st, c);
      else hipLaunchKernelGGL((mega2v_fwd_bwd<1, 64, 10>), dim3(grid), dim3(kMT), (size_t)MV1::total * sizeof(float), st, c);
    } else {
      hipLaunchKernelGGL(fn, dim3((B + kPanel - 1) / kPanel * c.Q), dim3(kMT), (size_t)ml.total * sizeof(float), st, c);
    }
    cx.check();
    // algorithmic MFMA FLOPs of the launch: forward chain + decoder layer (lambda and its data gradient) + backward chain
    double macs = (double)H * 2 * Lz + (double)Lz * H + 2.0 * H * D + (double)H * Lz + 2.0 * Lz * H;
    if (fl) macs += (double)D * H2;        // the first layer rides in the launch
    if (gm) macs += (double)H * K + (double)K * H + (double)K * 2 * Lz + (double)(H + 2 * Lz) * K + (double)K * H;
    if (m2) { m2_flops = 2.0 * B * macs; if (!fuse_pending) cx.mark("mega2_fwd_bwd", m2_flops); goto mega_done; }
    if (m2v) { m2_flops = 2.0 * B * macs; if (!fusev_pending) cx.mark("mega2v_fwd_bwd", m2_flops); goto mega_done; }
    cx.mark("mega_fwd_bwd", 2.0 * B * macs);
  mega_done:;
  }
  // Single device, steady state of a train graph at the specialised sizes: weight gradients AND the optimizer in one
  // launch (dwadam.hpp) -- no split-K slabs, no finalize_adam.
  const bool dw_upd = a.adam_p && a.adam_p == a.params;     // single device: the optimizer runs in the same launch
  // (every model of the mega schedule whose gradients are all matrix products: the learned mixture prior's variables of
  //  VAE_GMP come as per-panel partials and keep the split-K launch + finalize_adam)
  if (fl && (dw_upd || a.dp_images) && a.step_dev) {
    ImgPlan pl;
    plan_images(d, model, L, w, ml, a.params, pl);
    if (pl.map_ok) {
      std::vector<DwArgs> da_store(1);           // (4 KB of host scratch per call: engines may step from several host threads)
      DwArgs& da = da_store[0];
      memset(&da, 0, sizeof(da));
      da.B = B;
      da.u8x3 = 1;
      da.dbg = getenv("GMVAE_STAMPS") ? w.gstamps + 3 * 2048 * 8 : nullptr;
      da.lr_t = m2_ran ? reinterpret_cast<const float*>(w.sync + 2) : nullptr;
      da.ln_b1 = (float)log((double)a.beta1); da.ln_b2 = (float)log((double)a.beta2);
      dw_tensors(d, model, L, w, a.x, pl, da);
      {  // XCD-aware order: slot b runs on XCD b % 8 (observed round-robin placement; speed only)
        std::vector<int> cls(kDwMaxTiles);       // the XCD a tile would like: the one that shares its larger operand
        bool used[kDwMaxTiles];
        const int nt = da.total_tiles <= kDwMaxTiles ? da.total_tiles : 0;
        for (int i = 0; i < da.ntens; ++i) {
          const DwTensor& T = da.t[i];
          const int tiles_m = ((T.M + 16 * T.mu - 1) / (16 * T.mu));
          for (int tm = 0; tm < tiles_m; ++tm)
            for (int tn = 0; tn < T.tiles_n; ++tn) {
              const int t = T.tile_begin + tm * T.tiles_n + tn;
              if (t < kDwMaxTiles) cls[t] = (tiles_m >= T.tiles_n ? tm : tn) & 7;      // share the operand with more blocks
            }
        }
        for (int t = 0; t < nt; ++t) used[t] = false;
        const bool xcd = nt > 0;
        int next_any = 0;
        for (int b = 0; b < da.total_tiles && b < kDwMaxTiles; ++b) {
          int pick = -1;
          if (xcd) {
            for (int t = 0; t < nt; ++t)
              if (!used[t] && cls[t] == (b & 7)) { pick = t; break; }
            if (pick < 0) { while (next_any < nt && used[next_any]) ++next_any; pick = next_any; }
            used[pick] = true;
          } else {
            pick = b;
          }
          int pt = 0;                             // (tensor << 10) | tile inside the tensor (dwadam.hpp)
          for (int i = 1; i < da.ntens; ++i)
            if (pick >= da.t[i].tile_begin) pt = i;
          da.perm[b] = (unsigned short)((pt << 10) | (pick - da.t[pt].tile_begin));
        }
        if (da.total_tiles > kDwMaxTiles) { /* cannot happen at these sizes; the launch below is guarded */ }
      }
      FinalArgs& fa = da.fa;
      fa.P = (long long)L.P_pad; fa.grads = a.grads; fa.p = a.adam_p; fa.m = a.adam_m; fa.v = a.adam_v;
      fa.lr = a.lr; fa.b1 = a.beta1; fa.b2 = a.beta2; fa.eps = a.epsilon; fa.do_adam = dw_upd ? 1 : 0; fa.count = (float)B;
      fa.logw = w.logw; fa.logpx = w.logpx; fa.logq = w.logq; fa.logp = w.logp; fa.nent = gm ? w.nent : nullptr;
      fa.tail = tail; fa.B = B; fa.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev); fa.tail_log = a.tail_log;
      fa.epoch_word = dw_upd ? w.sync : nullptr;   // (data parallel: adam_tf_img, after the all-reduce, bumps the hand-off tag)
      fa.err_word = w.sync + 1;
      fa.img[0] = w.img_m; fa.img[1] = w.dimg; fa.img[2] = w.img2f; fa.img[3] = w.img2b; fa.img[4] = w.dimg2;
      fa.span = (a.want_spans && w.spans) ? w.spans + (size_t)a.span_slot * 3 * 2048 * 2 + 2048 * 2 : nullptr;
      if (gmp) {                                 // the prior variables: per-panel partials of mega_fwd_bwd
        const int KLp = (int)pad4((uint64_t)K * Lz);
        fa.gmp_part = w.gmp_part; fa.gmp_n = (B + kPanel - 1) / kPanel; fa.gmp_len = 2 * KLp + (int)pad4(K); fa.gmp_off = (long long)L.loc;
        da.gmp_blocks = (fa.gmp_len + kDwThreads - 1) / kDwThreads;
        for (int i = 0; i < pl.nmap && da.gmp_nmap < 3; ++i)
          if (pl.map[i].begin >= fa.gmp_off && pl.map[i].end <= fa.gmp_off + fa.gmp_len) da.gmp_map[da.gmp_nmap++] = pl.map[i];
      }
      // the padding words of the flat gradient buffer are never written by the tiles: the buffer is all-reduced / read whole
      if (da.total_tiles > kDwMaxTiles) return GMVAE_E_DIMS;
      double fl_ = 0;
      for (int i = 0; i < da.ntens; ++i) fl_ += 2.0 * da.t[i].M * da.t[i].N * B;
      const unsigned m2_grid = (unsigned)((((B + kPanel - 1) / kPanel + 1) & ~1) * 4);
      if ((fuse_pending || fusev_pending) && da.ntens <= kM3MaxT) {
        const bool vfam = fusev_pending;           // the VAE family: mega3v_step (seven workgroups per panel + role-less workers)
        std::vector<M3Args> m3_store(1);
        M3Args& m3 = m3_store[0];
        memset(&m3, 0, sizeof(m3));
        m3.m = c3;
        m3.m.lr_t_out = nullptr;
        m3.ntens = da.ntens;
        m3.flags = w.m3flags;
        m3.lr_next = w.sync + 4;
        m3.dbg = getenv("GMVAE_M3_STAMPS") ? w.gstamps + 3 * 2048 * 8 : nullptr;
        m3.fault_pnl = getenv("GMVAE_DEBUG_LEAD_FAULT") ? atoi(getenv("GMVAE_DEBUG_LEAD_FAULT")) : -1;
        const int nPr = (B + kPanel - 1) / kPanel;
        const unsigned grid3 = vfam ? 256u : m2_grid;
        {
          // The slot list (mega3.hpp): workgroup `rank` takes slots rank, rank + workers, ...  Phase P first -- the decoder
          // output layer's gradient over the PRODUCERS' column tiles: it waits for the producers' flags only and runs under the
          // leads' hand-off and backward chain --, then phase F: the fp32 tiles (longer), the uint8-batch tiles, the mixture
          // prior's blocks, the loss tail.  GMVAE at B = 1024: ranks 0..191 are the producers (done first), 192..255 the leads:
          // P on producers 0..95, F on producers 96..191 and on the leads.  VAE family: ranks below 6 panels are producers,
          // the next `panels` ranks the leads -- they get NO slot --, the rest of the 256 workgroups have no per-row role and
          // wait from the launch's start.  Inside a group, slot b (XCD b % 8 under round-robin placement) prefers a tile of
          // its operand class (speed only).
          const int nt = da.total_tiles;
          std::vector<int> tile_cls(nt), tile_grp(nt), tile_pt(nt);
          for (int i = 0; i < da.ntens; ++i) {
            const DwTensor& T = da.t[i];
            const int tiles_m = ((T.M + 16 * T.mu - 1) / (16 * T.mu));
            for (int tm = 0; tm < tiles_m; ++tm)
              for (int tn = 0; tn < T.tiles_n; ++tn) {
                const int t = T.tile_begin + tm * T.tiles_n + tn;
                int q_ = 1, lt_ = 0;
                if (T.dY == w.g) {                                                          // whose g columns: a producer's or the lead's
                  if (vfam) q_ = tn % 7;                                                     // (mega2v.hpp: column tile 7 wave + quarter)
                  else m2_dec_part(tn, q_, lt_);
                }
                const bool phP = T.dY == w.g && q_ != 0;
                tile_cls[t] = (tiles_m >= T.tiles_n ? tm : tn) & 7;
                tile_grp[t] = phP ? 0 : (T.a_u8 ? 2 : 1);
                tile_pt[t] = (i << 10) | (t - T.tile_begin) | (phP ? 0 : kM3PhaseF);
              }
          }
          const int lead_lo = vfam ? 6 * nPr : -1, lead_hi = vfam ? 7 * nPr : -1;          // ranks that take no slot
          std::vector<char> used(nt, 0);
          int slot = 0;
          auto skip_leads = [&]() {
            while (slot < kM3MaxSlots && (int)(slot % grid3) >= lead_lo && (int)(slot % grid3) < lead_hi) m3.perm[slot++] = kM3None;
          };
          for (int grp = 0; grp < 3; ++grp) {
            int left = 0;
            for (int t = 0; t < nt; ++t) left += tile_grp[t] == grp;
            for (; left > 0; --left, ++slot) {
              skip_leads();
              int pick = -1, any = -1;
              for (int t = 0; t < nt && pick < 0; ++t)
                if (!used[t] && tile_grp[t] == grp) {
                  if (any < 0) any = t;
                  if (tile_cls[t] == (slot & 7)) pick = t;
                }
              if (pick < 0) pick = any;
              used[pick] = 1;
              if (slot < kM3MaxSlots) m3.perm[slot] = (unsigned short)tile_pt[pick];
            }
          }
          for (int gb = 0; gb < da.gmp_blocks; ++gb, ++slot) {
            skip_leads();
            if (slot < kM3MaxSlots) m3.perm[slot] = (unsigned short)(kM3Gmp + gb);
          }
          skip_leads();
          if (slot < kM3MaxSlots) m3.perm[slot] = kM3Tail;
          m3.total_slots = ++slot;
        }
        for (int i = 0; i < da.ntens; ++i) m3.t[i] = da.t[i];
        M3Fin& f3 = m3.fa;
        f3.grads = fa.grads; f3.p = fa.p; f3.m = fa.m; f3.v = fa.v; f3.lr = fa.lr; f3.b1 = fa.b1; f3.b2 = fa.b2; f3.eps = fa.eps;
        f3.do_adam = fa.do_adam; f3.count = fa.count; f3.logw = fa.logw; f3.logpx = fa.logpx; f3.logq = fa.logq; f3.logp = fa.logp;
        f3.nent = fa.nent; f3.tail = fa.tail; f3.B = fa.B; f3.tail_log = fa.tail_log; f3.epoch_word = fa.epoch_word;
        f3.gmp_part = fa.gmp_part; f3.gmp_n = fa.gmp_n; f3.gmp_len = fa.gmp_len; f3.gmp_off = fa.gmp_off;
        m3.gmp_nmap = da.gmp_nmap;
        for (int i = 0; i < 3; ++i) m3.gmp_map[i] = da.gmp_map[i];
        for (int i = 0; i < kImgBufs; ++i) f3.img[i] = fa.img[i];
        static bool m3attr[64];
        if (first_on_device(m3attr)) {
          hipFuncSetAttribute(reinterpret_cast<const void*>(mega3_step), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipFuncSetAttribute(reinterpret_cast<const void*>(mega3v_step<0, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipFuncSetAttribute(reinterpret_cast<const void*>(mega3v_step<1, 64, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        // (the meeting area of a worker's tile: 34 KB at the bottom of the dynamic LDS, whatever the per-row part's map)
        if (m3.total_slots <= kM3MaxSlots && (!gmp || da.gmp_blocks < 0xfe)) {
          if (!vfam) hipLaunchKernelGGL(mega3_step, dim3(grid3), dim3(kMT), (size_t)M2::total * sizeof(float), st, m3);
          else if (vkind == 1) hipLaunchKernelGGL((mega3v_step<0, 2, 1>), dim3(grid3), dim3(kMT), (size_t)MV0::total * sizeof(float), st, m3);
          else hipLaunchKernelGGL((mega3v_step<1, 64, 10>), dim3(grid3), dim3(kMT), (size_t)MV1::total * sizeof(float), st, m3);
          cx.check();
          cx.mark(vfam ? (dw_upd ? "mega3v_step" : "mega3v_grads") : (dw_upd ? "mega3_step" : "mega3_grads"), m2_flops + fl_);
          return cx.err;
        }
      }
      if (fusev_pending) {                       // the two-launch form after all
        const unsigned gridv = (unsigned)((B + kPanel - 1) / kPanel * 7);
        if (vkind == 1) hipLaunchKernelGGL((mega2v_fwd_bwd<0, 2, 1>), dim3(gridv), dim3(kMT), (size_t)MV0::total * sizeof(float), st, c3);
        else hipLaunchKernelGGL((mega2v_fwd_bwd<1, 64, 10>), dim3(gridv), dim3(kMT), (size_t)MV1::total * sizeof(float), st, c3);
        cx.check();
        cx.mark("mega2v_fwd_bwd", m2_flops);
        fusev_pending = false;
      }
      if (fuse_pending) {                        // (cannot happen at mega2's sizes: the tile list fits) the two-launch form
        hipLaunchKernelGGL(mega2_fwd_bwd, dim3(m2_grid), dim3(kMT), (size_t)M2::total * sizeof(float), st, c3);
        cx.check();
        cx.mark("mega2_fwd_bwd", m2_flops);
        fuse_pending = false;
      }
      hipLaunchKernelGGL(dw_adam, dim3(da.total_tiles + 1 + da.gmp_blocks), dim3(kDwThreads), 0, st, da);
      cx.check();
      cx.mark(dw_upd ? "dw_adam" : "dw_grads", fl_);
      return cx.err;
    }
  }
  if (fusev_pending) {
    const unsigned gridv = (unsigned)((B + kPanel - 1) / kPanel * 7);
    if (vkind == 1) hipLaunchKernelGGL((mega2v_fwd_bwd<0, 2, 1>), dim3(gridv), dim3(kMT), (size_t)MV0::total * sizeof(float), st, c3);
    else hipLaunchKernelGGL((mega2v_fwd_bwd<1, 64, 10>), dim3(gridv), dim3(kMT), (size_t)MV1::total * sizeof(float), st, c3);
    cx.check();
    cx.mark("mega2v_fwd_bwd", m2_flops);
    fusev_pending = false;
  }
  if (fuse_pending) {                            // the tile list could not be built: mega2_fwd_bwd as a launch of its own
    hipLaunchKernelGGL(mega2_fwd_bwd, dim3((unsigned)((((B + kPanel - 1) / kPanel + 1) & ~1) * 4)), dim3(kMT), (size_t)M2::total * sizeof(float), st, c3);
    cx.check();
    cx.mark("mega2_fwd_bwd", m2_flops);
    fuse_pending = false;
  }
  // Every weight gradient in one grouped launch.  The uint8-activation problems (bf16 matrix cores) take NS splits
  // of two 64-row staging rounds; the fp32 problems take 2 NS splits of ONE round each: their workgroups, the
  // launch's critical path, are latency chains, and the single-buffered tile configuration leaves room for all
  // of them on the chip at once (3 per CU).
  const int NS2 = (2 * NS <= NS_MAX && B / (2 * NS) >= 64) ? 2 * NS : NS;
  const int NSX = NS;            // splits of the uint8-activation problems
  SlabX sx;
  memset(&sx, 0, sizeof(sx));
  {
    Group g;
    auto xrange = [&](uint64_t b, uint64_t n) { sx.b[sx.n] = (int)b; sx.e[sx.n] = (int)(b + pad4(n)); sx.ns[sx.n] = NSX; sx.n++; };
    if (gm) {
      g.add(p_tn(a.x, true, D, 1, w.dbuf[2], H, D, H, B, sl + E.w[0], sl + E.b[0], NSX, PP, nullptr));           // dWy0
      g.add(p_tn(a.x, true, D, 1, w.dbuf[1], H, D, H, B, sl + G.w[0], sl + G.b[0], NSX, PP, nullptr));           // dWg0[x]
      xrange(E.w[0], (uint64_t)D * H); xrange(E.b[0], H); xrange(G.w[0], (uint64_t)D * H); xrange(G.b[0], H);
      g.add(p_tn(w.hd[1], false, H, 1, w.g, D, H, D, B, sl + Dn.w[1], sl + Dn.b[1], NS2, PP, nullptr));          // dWd1
      const int K4 = (int)pad4(K);       // row stride of y and dlogits as mega_fwd_bwd stores them
      g.add(p_tn(w.y, false, K4, 1, w.dbuf[1], H, K, H, B, sl + G.w[0] + (uint64_t)D * H, nullptr, NS2, PP, nullptr));
      g.add(p_tn(w.he[1], false, H, 1, w.dlogits, K4, H, K, B, sl + E.w[1], sl + E.b[1], NS2, PP, nullptr));     // dWy1
      g.add(p_tn(w.y, false, K4, 1, w.dpp, 2 * Lz, K, 2 * Lz, B, sl + L.prior.w[0], sl + L.prior.b[0], NS2, PP, nullptr));
      g.add(p_tn(w.hg[1], false, H, 1, w.dqp, 2 * Lz, H, 2 * Lz, B, sl + G.w[1], sl + G.b[1], NS2, PP, nullptr)); // dWg1
    } else {
      g.add(p_tn(a.x, true, D, 1, w.dbuf[1], H, D, H, B, sl + E.w[0], sl + E.b[0], NSX, PP, nullptr));           // dWe0
      xrange(E.w[0], (uint64_t)D * H); xrange(E.b[0], H);
      g.add(p_tn(w.hd[1], false, H, 1, w.g, D, H, D, B, sl + Dn.w[1], sl + Dn.b[1], NS2, PP, nullptr));          // dWd1
      g.add(p_tn(w.he[1], false, H, 1, w.dqp, 2 * Lz, H, 2 * Lz, B, sl + E.w[1], sl + E.b[1], NS2, PP, nullptr)); // dWe1
    }
    g.add(p_tn(w.z, false, Lz, 1, w.dbuf[0], H, Lz, H, B, sl + Dn.w[0], sl + Dn.b[0], NS2, PP, nullptr));         // dWd0
    if (NS2 == NSX) sx.n = 0;
    launch_group(cx, g, "bwd_dw_all", B >= 512 ? (NS2 != NS ? 3 : 1) : 0,
                 (getenv("GMVAE_STAMPS") || a.want_spans) ? w.gstamps + 2048 * 8 : nullptr);
  }
  return finish_fused(cx, a, L, w, tail, NS2, B, &sx);
}

// The fused schedule: 9 launches instead of 21 for GMVAE with one hidden layer at sizes whose
// small-layer weights fit in LDS.  Every GEMM launch is a single staging round per workgroup
// (split-K into slabs that the consumer reduces), so the chip is filled and no workgroup
// waits on more than ~2 dependent memory round trips.
static int run_step_fused(Ctx& cx, const StepArgs& a, const Layout& L, WS& w, const float* eps, const float* u,
                          float* gen_eps, float* gen_u) {
  const GmvaeDims& d = *a.d;
  const int B = d.B, K = d.K, Lz = d.L, D = d.D, H = d.hidden[0];
  const float* P = a.params;
  hipStream_t st = cx.st;
  const int NSF = fwd_splits(D);
  static bool attr_done[64];
  if (first_on_device(attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const NetL &E = L.ency, &G = L.encg, &Dn = L.dec;
  {  // P1: X * [Wy0 | Wg0x] as single-round split-K partials
    Group g;
    Problem p0 = p_nn(a.x, true, D, P + E.w[0], H, B, H, D, w.s1, 2 * H, nullptr, false);
    p0.splits = NSF; p0.split_stride = (long long)B * 2 * H;
    Problem p1 = p_nn(a.x, true, D, P + G.w[0], H, B, H, D, w.s1 + H, 2 * H, nullptr, false);
    p1.splits = NSF; p1.split_stride = (long long)B * 2 * H;
    g.add(p0);
    g.add(p1);
    // auxiliary workgroups on the same launch: Philox noise + this step's LDS weight images
    Aux& ax = g.L.aux;
    ax.eps = gen_eps; ax.u = gen_u; ax.n_rows = (unsigned long long)B * d.S; ax.nL = Lz; ax.nK = K; ax.row_base = d.row0 * d.S;
    ax.seed = a.seed; ax.step = a.step;
    ax.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev);
    ax.noise_blocks = (int)((noise_items(gen_eps, gen_u, ax.n_rows, Lz, K) + kThreads - 1) / kThreads);
    const FwdLay fl = fwd_lay(H, Lz, K);
    const BwdLay bl = bwd_lay(H, Lz, K);
    const float* Wy1 = P + E.w[1];
    const float* Wg0y = P + G.w[0] + (uint64_t)D * H;
    const float* Wp = P + L.prior.w[0];
    const float* Wg1 = P + G.w[1];
    const float* Wd0 = P + Dn.w[0];
    int nt = 0;
    auto task = [&](float* dst, int ld, const float* src, int rows, int cols, int src_ld, int trans) {
      ImgTask& t = ax.task[nt++];
      t.dst = dst; t.ld = ld; t.src = src; t.rows = rows; t.cols = cols; t.src_ld = src_ld; t.trans = trans;
    };
    {
      task(w.img_f + fl.W_y1, fl.KP, Wy1, H, K, K, 0);
      task(w.img_f + fl.W_g0y, H, Wg0y, K, H, H, 0);
      task(w.img_f + fl.W_p, 2 * Lz, Wp, K, 2 * Lz, 2 * Lz, 0);
      task(w.img_f + fl.W_g1, 2 * Lz, Wg1, H, 2 * Lz, 2 * Lz, 0);
      task(w.img_f + fl.W_d0, H, Wd0, Lz, H, H, 0);
      task(w.img_f + fl.b_y0, H, P + E.b[0], 1, H, H, 0);
      task(w.img_f + fl.b_y1, K, P + E.b[1], 1, K, K, 0);
      task(w.img_f + fl.b_g0, H, P + G.b[0], 1, H, H, 0);
      task(w.img_f + fl.b_p, 2 * Lz, P + L.prior.b[0], 1, 2 * Lz, 2 * Lz, 0);
      task(w.img_f + fl.b_g1, 2 * Lz, P + G.b[1], 1, 2 * Lz, 2 * Lz, 0);
      task(w.img_f + fl.b_d0, H, P + Dn.b[0], 1, H, H, 0);
      if (a.backward) {
        task(w.img_b + bl.W_d0T, bl.ldD, Wd0, Lz, H, H, 1);
        task(w.img_b + bl.W_g1T, bl.ldG, Wg1, H, 2 * Lz, 2 * Lz, 1);
        task(w.img_b + bl.W_cT, bl.ldC, Wg0y, K, H, H, 1);
        task(w.img_b + bl.W_cT + H * bl.ldC, bl.ldC, Wp, K, 2 * Lz, 2 * Lz, 1);
        task(w.img_b + bl.W_y1T, bl.ldY, Wy1, H, K, K, 1);
      }
    }
    ax.ntasks = nt;
    ax.nblocks = ax.noise_blocks + nt;
    launch_group(cx, g, "fwd_x_first_layers_splitk+aux", 0);
  }
  const int NS = dw_splits(B);
  const long long PP = (long long)L.P_pad;
  float* sl = w.slabs;
  float* tail = a.backward ? a.grads + L.P_pad : a.tail;
  {  // P2: the whole row-local forward chain
    ChainFwdArgs c;
    c.B = B; c.H = H; c.L = Lz; c.K = K; c.NS = NSF;
    c.c = d.raw_sigma_bias; c.smin = d.sigma_min; c.invT = 1.f / d.temperature;
    c.s1 = w.s1; c.img = w.img_f;
    c.eps = eps; c.u = u;
    c.hy1 = w.he[1]; c.logits = w.logits; c.y = w.y; c.nent = w.nent; c.hg1 = w.hg[1]; c.pp = w.pp; c.qp = w.qp;
    c.z = w.z; c.logq = w.logq; c.logp = w.logp; c.hd1 = w.hd[1];
    c.dbg = getenv("GMVAE_STAMPS") ? w.stamps : nullptr;
    const size_t sh = (size_t)fwd_lay(H, Lz, K).total * sizeof(float);
    hipLaunchKernelGGL(chain_fwd, dim3((B + kPanel - 1) / kPanel), dim3(kThreads), sh, st, c);
    rowk(cx, "chain_fwd");
  }
  int nparts = 1;
  {  // P3: decoder output layer + Bernoulli log-likelihood
    Group g;
    Problem p = p_nn(w.hd[1], false, H, P + Dn.w[1], D, B, D, H, a.backward ? w.g : nullptr, D, P + Dn.b[1], false);
    p.epi = EPI_BERNOULLI;
    p.addconst = d.gen_bias_init;
    p.bias2 = d.gen_bias_vec;
    p.x = a.x; p.ldx = D; p.x_div = 1; p.part = w.part;
    g.add(p);
    const int bn = cfg_bn(launch_group(cx, g, "fwd_dec_bernoulli"));
    nparts = (D + bn - 1) / bn;
  }
  if (!a.backward) {
    hipLaunchKernelGGL(row_terms, dim3(grid_for(B, 256, 1 << 22)), dim3(256), 0, st, w.part, nparts, w.logq, w.logp,
                       w.nent, 1, w.logpx, w.logw, a.row_terms, B);
    rowk(cx, "row_terms");
    hipLaunchKernelGGL(loss_tail, dim3(1), dim3(1024), 0, st, w.logw, w.logpx, w.logq, w.logp, w.nent, (float*)nullptr,
                       tail, B, 1, a.step_dev);
    rowk(cx, "loss_tail");
    return cx.err;
  }
  {  // P4: top of the backward pass: dWd1 (+db) and the split-K partials of (sigmoid - x) * Wd1^T
    Group g;
    g.add(p_tn(w.hd[1], false, H, 1, w.g, D, H, D, B, sl + Dn.w[1], sl + Dn.b[1], NS, PP, nullptr));
    Problem p = p_nt(w.g, D, P + Dn.w[1], D, B, H, D, w.s4, H, nullptr, 0);
    p.splits = NSF; p.split_stride = (long long)B * H;
    g.add(p);
    launch_group(cx, g, "bwd_dec_top_splitk", 0);
  }
  {  // P5: the whole row-local backward chain
    ChainBwdArgs c;
    c.B = B; c.H = H; c.L = Lz; c.K = K; c.NS = NSF; c.nparts = nparts;
    c.c = d.raw_sigma_bias; c.smin = d.sigma_min; c.invT = 1.f / d.temperature;
    c.s4 = w.s4; c.img = w.img_b;
    c.hd1 = w.hd[1]; c.hg1 = w.hg[1]; c.hy1 = w.he[1]; c.qp = w.qp; c.pp = w.pp; c.z = w.z; c.eps = eps; c.y = w.y;
    c.logits = w.logits; c.nent = w.nent; c.part = w.part; c.logq = w.logq; c.logp = w.logp;
    c.dhd1 = w.dbuf[0]; c.dqp = w.dqp; c.dpp = w.dpp; c.dhg1 = w.dbuf[1]; c.dlogits = w.dlogits; c.dhy1 = w.dbuf[2];
    c.logpx = w.logpx; c.logw = w.logw;
    c.dbg = getenv("GMVAE_STAMPS") ? w.stamps + (size_t)((B + 15) / 16) * 16 : nullptr;
    const size_t sh = (size_t)bwd_lay(H, Lz, K).total * sizeof(float);
    hipLaunchKernelGGL(chain_bwd, dim3((B + kPanel - 1) / kPanel), dim3(kThreads), sh, st, c);
    rowk(cx, "chain_bwd");
  }
  {  // P6: every remaining weight gradient in one grouped launch
    Group g;
    g.add(p_tn(a.x, true, D, 1, w.dbuf[2], H, D, H, B, sl + E.w[0], sl + E.b[0], NS, PP, nullptr));            // dWy0
    g.add(p_tn(a.x, true, D, 1, w.dbuf[1], H, D, H, B, sl + G.w[0], sl + G.b[0], NS, PP, nullptr));            // dWg0[x]
    g.add(p_tn(w.y, false, K, 1, w.dbuf[1], H, K, H, B, sl + G.w[0] + (uint64_t)D * H, nullptr, NS, PP, nullptr));  // dWg0[y]
    g.add(p_tn(w.he[1], false, H, 1, w.dlogits, K, H, K, B, sl + E.w[1], sl + E.b[1], NS, PP, nullptr));       // dWy1
    g.add(p_tn(w.y, false, K, 1, w.dpp, 2 * Lz, K, 2 * Lz, B, sl + L.prior.w[0], sl + L.prior.b[0], NS, PP, nullptr));
    g.add(p_tn(w.hg[1], false, H, 1, w.dqp, 2 * Lz, H, 2 * Lz, B, sl + G.w[1], sl + G.b[1], NS, PP, nullptr)); // dWg1
    g.add(p_tn(w.z, false, Lz, 1, w.dbuf[0], H, Lz, H, B, sl + Dn.w[0], sl + Dn.b[0], NS, PP, nullptr));        // dWd0
    launch_group(cx, g, "bwd_dw_all", 0);
  }
  return finish_fused(cx, a, L, w, tail, NS, B);
}

static unsigned long long* g_sk_dbg = nullptr;
// ---- the skinny schedule (skinny.hpp): 10 launches, every weight matrix crosses the fabric once per pass
// log p(z) under the learned mixture prior (scripts/vae.py:231-244) and the responsibilities: z -> w.logp, w.resp
static void launch_mixture_logprob(Ctx& cx, WS& w, const float* P, const Layout& L, int R, int Lz, int K) {
  hipStream_t st = cx.st;
  int Kp = 1;
  while (Kp < K) Kp <<= 1;
  const int nw = 4, rpw = Kp <= 64 ? 64 / Kp : 0;
  const size_t sh = (size_t)(2 * K * (Lz | 1) + 64 + nw * rpw * Lz) * sizeof(float);
  // the LDS-resident form needs K <= 64 and the (loc, 1/s) image inside the default 64 KB of dynamic LDS; any other
  // size (scripts/vae.py:231-244 bounds neither K nor L) takes the tiled form
  if (K > 64 || sh > 64 * 1024 || getenv("GMVAE_GMP_TILED")) {
    hipLaunchKernelGGL(gmp_consts, dim3(K), dim3(256), 0, st, P + L.rawscale, P + L.mixlog, w.gmp_inv, w.gmp_cst, Lz, K);
    rowk(cx, "gmp_consts");
    hipLaunchKernelGGL(mixture_logprob_tiled, dim3(grid_for(R, 16, 2048)), dim3(256), 0, st, w.z, P + L.loc, w.gmp_inv,
                       w.gmp_cst, w.logp, w.resp, R, Lz, K);
    rowk(cx, "mixture_logprob_tiled");
  } else {
    hipLaunchKernelGGL(mixture_logprob_lse, dim3(grid_for(R, nw * rpw, 1024)), dim3(64 * nw), sh, st, w.z,
                       P + L.loc, P + L.rawscale, P + L.mixlog, w.logp, w.resp, R, Lz, K, Kp);
    rowk(cx, "mixture_logprob_lse");
  }
}

static int run_step_skinny(Ctx& cx, const StepArgs& a, const Layout& L, WS& w, const float* eps, const float* u,
                           float* gen_eps, float* gen_u) {
  const GmvaeDims& d = *a.d;
  const int B = d.B, K = d.K, Lz = d.L, D = d.D, H = d.hidden[0];
  hipStream_t st = cx.st;
  SkArgs s;                                      // (a stack local: two host threads may step at once; passed by value to the kernels)
  memset(&s, 0, sizeof(s));
  const bool vae = a.model != GMVAE_MODEL_GMVAE, gmp = a.model == GMVAE_MODEL_VAE_GMP;
  // (VAE: the one encoder stands in for both of the GMVAE's: SkArgs::model)
  const NetL &E = vae ? L.enc : L.ency, &G = vae ? L.enc : L.encg, &Dn = L.dec;
  s.model = a.model;
  if (gmp) {
    s.dz = w.dz; s.gmp_part = w.gmp_part; s.resp = w.resp;
    s.gmp_raw = (long long)L.rawscale; s.gmp_mix = (long long)L.mixlog;
    s.gmp_n = B / 4 < 16 ? 16 : (B / 4 > GMP_PARTS ? GMP_PARTS : B / 4);       // strips of ~4 rows (gmp_param_bwd walks a strip serially), 16 .. GMP_PARTS
    s.gmp_len = 2 * (int)pad4((uint64_t)K * Lz) + (int)pad4(K); s.gmp_off = (long long)L.loc;
  }
  s.B = B; s.D = D; s.H = H; s.L = Lz; s.K = K; s.K4 = (int)pad4(K);
  // first-layer slabs: the contraction over D splits only while the unsplit launch would leave compute units idle (B = 64, H = 512:
  // 64 workgroups -> 256 with 4 slabs); at B = 1024 the 1024 unsplit workgroups already fill the chip and each slab is a second
  // pass over the staging and the slab sum
  {
    const int wgs1 = (2 * H / 64) * ((B + 15) / 16), ns1 = 256 / (wgs1 > 0 ? wgs1 : 1);
    s.ns1 = vae ? 1 : (ns1 < 1 ? 1 : (ns1 > kSkNs1 ? kSkNs1 : ns1));
  }
  // the two D-wide layers (F5, B1) in 64-column tiles once 16-row x 64-column tiles alone give the chip its workgroups
  const bool wide = ((D + 63) / 64) * ((B + 15) / 16) >= 256;
  s.nparts = wide ? (D + 63) / 64 : D / 16;

  s.c = d.raw_sigma_bias; s.smin = d.sigma_min; s.invT = 1.f / d.temperature; s.gen_bias = d.gen_bias_init;
  s.gen_bias_vec = d.gen_bias_vec;
  s.x = a.x; s.P = a.params;
  s.Wy0 = (long long)E.w[0]; s.by0 = (long long)E.b[0]; s.Wy1 = (long long)E.w[1]; s.by1 = (long long)E.b[1];
  if (!vae) { s.Wp = (long long)L.prior.w[0]; s.bp = (long long)L.prior.b[0]; }
  s.Wg0 = (long long)G.w[0]; s.bg0 = (long long)G.b[0]; s.Wg1 = (long long)G.w[1]; s.bg1 = (long long)G.b[1];
  s.Wd0 = (long long)Dn.w[0]; s.bd0 = (long long)Dn.b[0]; s.Wd1 = (long long)Dn.w[1]; s.bd1 = (long long)Dn.b[1];
  s.s1 = w.sk_s1; s.hy = w.he[1]; s.hg = vae ? w.he[1] : w.hg[1]; s.y = w.y; s.logits = w.logits; s.nent = w.nent; s.pp = w.pp; s.qp = w.qp;
  s.z = w.z; s.hd = w.hd[1]; s.g = w.g; s.part = w.sk_part; s.lqp = w.sk_lqp;
  s.dhd = w.dbuf[0]; s.dqp = w.dqp; s.dpp = w.dpp; s.dhg = w.dbuf[1]; s.dlogits = w.dlogits; s.dhy = w.dbuf[2];
  s.eps = eps; s.u = u; s.eps_w = gen_eps; s.u_w = gen_u; s.gen_eps = gen_eps ? 1 : 0; s.gen_u = gen_u ? 1 : 0;
  s.seed = a.seed; s.step = a.step; s.row0 = d.row0; s.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev);
  s.grads = a.grads; s.ap = a.adam_p; s.am = a.adam_m; s.av = a.adam_v;
  s.lr = a.lr; s.b1 = a.beta1; s.b2 = a.beta2; s.aeps = a.epsilon;
  s.ln_b1 = (float)log((double)a.beta1); s.ln_b2 = (float)log((double)a.beta2);
  s.tail = a.grads + L.P_pad; s.tail_log = a.tail_log;
  s.dw_cnt = w.sk_cnt;
  s.logpx = w.logpx; s.logq = w.logq; s.logp = w.logp; s.logw = w.logw;
  // the W stage's form: up to 128 batch rows sk_dw (one-wave tiles); above, [64 x 64] tiles: one workgroup each (sk_dwb<1>: bf16
  // pieces, one per CU at its 170 registers; more tiles than CUs: sk_dwb<0>, fp32 matrix instructions, two per CU) or -- more
  // tiles than CUs and at least 512 rows -- two batch shares each, the last-arriving share running the optimizer (sk_dwc)
  const bool dw_big = B > 128;
  auto add = [&](const void* A, bool u8, int lda, const float* dY, int ldy, int M, int N, uint64_t w_off, long long b_off) {
    SkTensor& T = s.t[s.ntens++];
    T.A = A; T.a_u8 = u8 ? 1 : 0; T.lda = lda; T.dY = dY; T.ldy = ldy; T.M = M; T.N = N; T.w_off = (int)w_off; T.b_off = (int)b_off;
    T.vec = (N % 4 == 0 && ldy % 4 == 0) ? 1 : 0;            // 16-byte optimizer accesses ([16 x 64] tiles) where rows allow
    T.tile_begin = s.total_tiles;
    if (dw_big) {                                            // sk_dwb / sk_dwc: [64 x 64] tiles
      T.tiles_n = (N + 63) / 64;
      s.total_tiles += ((M + 63) / 64) * T.tiles_n;
    } else {
      T.tiles_n = T.vec ? (N + 63) / 64 : (N + 15) / 16;
      s.total_tiles += (T.vec ? (M + 15) / 16 : (M + 63) / 64) * T.tiles_n;
    }
  };
  const int nrt = (B + 15) / 16;
  // row tiles per workgroup of a matrix-product launch with nct column tiles: as many (4, 2, 1) as still leave 256 workgroups
  auto rt_of = [&](int nct, int min_wgs = 256) {
    int rt = 4;
    while (rt > 1 && nct * ((nrt + rt - 1) / rt) < min_wgs) rt >>= 1;
    return rt;
  };
  const double fB = 2.0 * B;
  s.dbg = g_sk_dbg;                                // diagnostic: tools/skstamps.py (null unless gmvae_debug_sk_stamps(NULL) ran)
  auto launch = [&](auto kern, int grid, int threads, size_t sh, const char* name, double fl) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), sh, st, s);
    cx.check();
    cx.mark(name, fl);
  };
  auto gemm = [&](auto stage, int nct, int slabs, int extra, const char* name, double fl) {
    constexpr int ST = decltype(stage)::value;
    // (F5W: 13 column tiles at D = 784 -- four row tiles already from 192 workgroups: one round of them beats two of 416)
    const int rt = rt_of(nct * slabs, ST == SK_F5W ? 192 : 256), grid = nct * ((nrt + rt - 1) / rt) * slabs + extra;
    launch(rt == 4 ? sk_gemm<ST, 4> : rt == 2 ? sk_gemm<ST, 2> : sk_gemm<ST, 1>, grid, kSkThreads, 0, name, fl);
  };
  // rows per workgroup of the y path: one while the chip holds every row's workgroup at once (3 per CU), then 2, 4 (H <= 512)
  const int yr = (H > 512 || B <= 768) ? 1 : (B <= 1536 ? 2 : 4);
  if (vae) {
    const int eps_blocks = gen_eps ? (int)(((long long)B * ((Lz + 3) / 4) + kSkThreads - 1) / kSkThreads) : 0;
    gemm(std::integral_constant<int, SK_F1>{}, H / 64, 1, eps_blocks, "sk_first_layer", fB * D * H);
  } else {
  gemm(std::integral_constant<int, SK_F1>{}, 2 * H / 64, s.ns1, 0, "sk_first_layers", fB * D * 2 * H);
  const int eps_blocks = gen_eps ? (int)(((long long)B * ((Lz + 3) / 4) + 255) / 256) : 0;       // extra workgroups: the eps rows
  if (yr == 1) launch(H <= 512 ? sk_ypath<2> : sk_ypath<4>, B + eps_blocks, 256, 0, "sk_y_path", fB * ((double)H * K + K * H + K * 2.0 * Lz));
  else launch(yr == 2 ? sk_ypath_r<2, 2> : sk_ypath_r<2, 4>, (B + yr - 1) / yr + eps_blocks, 256, 0, "sk_y_path", fB * ((double)H * K + K * H + K * 2.0 * Lz));
  }
  gemm(std::integral_constant<int, SK_F3>{}, (Lz + 15) / 16, 1, 0, "sk_q_head_z", fB * H * 2 * Lz);
  // (VAE_GMP: the mixture log-density rides on sk_gmp_bwd behind B2 where its LDS form fits; else the row kernel here)
  const size_t gmp_sh = (size_t)(2 * K * (Lz | 1) + 128 + 64 * kGmpStripRows) * sizeof(float);
  const bool gmp_fused = gmp && K <= 64 && gmp_sh <= 64 * 1024 && (B + s.gmp_n - 1) / (s.gmp_n > 0 ? s.gmp_n : 1) <= kGmpStripRows;
  if (gmp && !gmp_fused) launch_mixture_logprob(cx, w, a.params, L, B, Lz, K);
  gemm(std::integral_constant<int, SK_F4>{}, H / 64, 1, 0, "sk_dec_hidden", fB * Lz * H);
  if (wide) gemm(std::integral_constant<int, SK_F5W>{}, (D + 63) / 64, 1, 0, "sk_dec_bernoulli", fB * H * D);
  else gemm(std::integral_constant<int, SK_F5>{}, D / 16, 1, 0, "sk_dec_bernoulli", fB * H * D);
  // (Measured and removed in round 4: the W launch in three parts on a forked graph branch beside B3 / B4 -- a fork / join
  //  pair inside a hipGraph cost far more than it hid on this stack: 94.3 vs 60.8 us per step, profiles/round3_notes.md.)
  auto launch_dw = [&]() {
    s.ntens = 0; s.total_tiles = 0; s.has_tail = 1;
    add(s.hd, false, H, s.g, D, H, D, Dn.w[1], (long long)Dn.b[1]);                                      // dWd1 (+ dbd1)
    add(s.z, false, Lz, s.dhd, H, Lz, H, Dn.w[0], (long long)Dn.b[0]);                                   // dWd0
    add(s.hg, false, H, s.dqp, 2 * Lz, H, 2 * Lz, G.w[1], (long long)G.b[1]);                            // dWg1
    if (vae) {
      add(a.x, true, D, s.dhg, H, D, H, E.w[0], (long long)E.b[0]);                                      // dWe0 (+ dbe0)
    } else {
      add(a.x, true, D, s.dhy, H, D, H, E.w[0], (long long)E.b[0]);                                      // dWy0 (+ dby0)
      add(a.x, true, D, s.dhg, H, D, H, G.w[0], (long long)G.b[0]);                                      // dWg0[x] (+ dbg0)
      add(s.y, false, s.K4, s.dhg, H, K, H, G.w[0] + (uint64_t)D * H, -1);                               // dWg0[y]
      add(s.hy, false, H, s.dlogits, s.K4, H, K, E.w[1], (long long)E.b[1]);                             // dWy1
      add(s.y, false, s.K4, s.dpp, 2 * Lz, K, 2 * Lz, L.prior.w[0], (long long)L.prior.b[0]);            // dWp
    }
    double fw = 0;
    for (int i = 0; i < s.ntens; ++i) fw += 2.0 * s.t[i].M * s.t[i].N * B;
    const int gmp_wgs = s.gmp_part ? (s.gmp_len + kSkThreads - 1) / kSkThreads : 0;      // (sk_dw derives the same count)
    if (dw_big && s.total_tiles > 256 && B >= 512 && s.total_tiles <= kSkDwcMaxTiles) {
      s.dwp = w.slabs; s.dwp_stride = (long long)L.P_pad; s.dw_ks = kSkDwShares;
      const int gmp256 = s.gmp_part ? (s.gmp_len + kDwcThreads - 1) / kDwcThreads : 0;
      hipLaunchKernelGGL(sk_dwc, dim3(16 * ((s.total_tiles * s.dw_ks + 15) / 16) + 1 + gmp256), dim3(kDwcThreads), 0, st, s);
    } else if (dw_big) hipLaunchKernelGGL(s.total_tiles > 256 ? sk_dwb<0> : sk_dwb<1>, dim3(8 * ((s.total_tiles + 7) / 8) + 1 + gmp_wgs), dim3(kSkThreads), 0, st, s);
    else hipLaunchKernelGGL(sk_dw, dim3((s.total_tiles + kSkWaves - 1) / kSkWaves + 1 + gmp_wgs), dim3(kSkThreads), 0, st, s);
    cx.check();
    cx.mark(s.ap ? "sk_dw_adam" : "sk_dw", fw);
  };
  if (wide) gemm(std::integral_constant<int, SK_B1W>{}, H / 64, 1, 0, "sk_bwd_dhd", fB * D * H);
  else gemm(std::integral_constant<int, SK_B1>{}, H / 16, 1, 0, "sk_bwd_dhd", fB * D * H);
  gemm(std::integral_constant<int, SK_B2>{}, (Lz + 15) / 16, 1, 0, "sk_bwd_dz_heads", fB * H * Lz);
  if (gmp) {                                     // the mixture prior's share of dz and the q head's reverse; its variables' gradients
    if (gmp_fused) {                             // one launch, (loc, 1 / s) staged in LDS (skinny.hpp sk_gmp_bwd)
      const int nz = (B + 3) / 4;
      hipLaunchKernelGGL(sk_gmp_bwd, dim3(nz + s.gmp_n), dim3(256), gmp_sh, st, s, nz);
      rowk(cx, "sk_gmp_bwd");
    } else {                                     // any K, L: the general schedule's row kernels
      hipLaunchKernelGGL(z_head_bwd, dim3(grid_for(B, 4)), dim3(256), 0, st, w.dz, w.qp, 1, (const float*)nullptr, eps, w.z, (const float*)nullptr,
                         w.resp, a.params + L.loc, a.params + L.rawscale, w.dqp, (float*)nullptr, B, Lz, K, (int)PRIOR_GMP, d.raw_sigma_bias, d.sigma_min);
      rowk(cx, "z_head_bwd");
      hipLaunchKernelGGL(gmp_param_bwd, dim3(s.gmp_n), dim3(256), 0, st, w.z, w.resp, (const float*)nullptr, a.params + L.loc,
                         a.params + L.rawscale, a.params + L.mixlog, w.gmp_part, B, Lz, K, (int)pad4((uint64_t)K * Lz));
      rowk(cx, "gmp_param_bwd");
    }
  }
  gemm(std::integral_constant<int, SK_B3>{}, H / 32, 1, 0, "sk_bwd_dhg", fB * 2 * Lz * H);
  if (!vae) {

    if (yr == 1) launch(H <= 512 ? sk_ybwd<2> : sk_ybwd<4>, B, 256, 0, "sk_y_path_bwd", fB * ((double)(H + 2 * Lz) * K + K * H));
    else launch(yr == 2 ? sk_ybwd_r<2, 2> : sk_ybwd_r<2, 4>, (B + yr - 1) / yr, 256, 0, "sk_y_path_bwd", fB * ((double)(H + 2 * Lz) * K + K * H));
  }
  launch_dw();
  return cx.err;
}

// Forward-only evaluation at the reference's default sizes (evalf.hpp): first layers -> operand images -> ONE launch for the whole
// per-row chain, the Bernoulli term, the IWAE bound and the batch sums.  scripts/runners.py:324-333.
static int run_eval_fused(Ctx& cx, const StepArgs& a, const Layout& L, WS& w) {
  const GmvaeDims& d = *a.d;
  const int B = d.B, S = d.S, D = d.D;
  const float* P = a.params;
  hipStream_t st = cx.st;
  const bool gm = a.model == GMVAE_MODEL_GMVAE;
  const NetL &E = gm ? L.ency : L.enc, &G = L.encg, &Dn = L.dec;
  {
    FlxArgs f;
    f.x = a.x; f.W0 = P + E.w[0]; f.b0 = P + E.b[0]; f.out0 = w.he[1]; f.H0 = EV::H; f.relu0 = 1;
    f.W1 = gm ? P + G.w[0] : nullptr; f.out1 = gm ? w.gx : nullptr; f.H1 = gm ? EV::H : 0;
    f.B = B; f.D = D;
    const int nct = gm ? 2 : 1, nrt = (B + 15) / 16;
    int rt = 4;
    while (rt > 1 && nct * ((nrt + rt - 1) / rt) < 256) rt >>= 1;
    const int grid = nct * ((nrt + rt - 1) / rt);
    if (rt == 4) hipLaunchKernelGGL(first_layers_u8bf<4>, dim3(grid), dim3(kSkThreads), 0, st, f);
    else if (rt == 2) hipLaunchKernelGGL(first_layers_u8bf<2>, dim3(grid), dim3(kSkThreads), 0, st, f);
    else hipLaunchKernelGGL(first_layers_u8bf<1>, dim3(grid), dim3(kSkThreads), 0, st, f);
    cx.check();
    cx.mark("fwd_x_first_layers", 2.0 * B * D * nct * EV::H);
  }
  unsigned* const counter = reinterpret_cast<unsigned*>(w.ev_slots + 4 * 1024);
  if (!(d.sched_flags & GMVAE_SCHED_EVAL_IMAGES_VALID)) {      // (else: a previous pass on fixed parameters left the images; its last workgroup reset the counter)
    EvalPrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    if (gm) { pa.Wp = P + L.prior.w[0]; pa.bp = P + L.prior.b[0]; pa.Wg0 = P + G.w[0]; pa.bg0 = P + G.b[0]; pa.Wg1 = P + G.w[1]; pa.bg1 = P + G.b[1]; }
    pa.family = gm ? 0 : (d.L == 2 ? 2 : 1);
    pa.Wd0 = P + Dn.w[0]; pa.bd0 = P + Dn.b[0]; pa.Wd1 = P + Dn.w[1]; pa.bd1 = P + Dn.b[1];
    pa.gen_bias = d.gen_bias_init; pa.img = w.ev_img; pa.counter = counter;
    hipLaunchKernelGGL(evalf_prep, dim3((EV::total + 255) / 256), dim3(256), 0, st, pa);
    cx.check();
    cx.mark("evalf_prep", 0);
  }
  EvalArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.B = B; ea.S = S; ea.x = a.x; ea.he1 = w.he[1]; ea.gx = w.gx; ea.img = w.ev_img;
  if (gm) { ea.Wy1 = P + E.w[1]; ea.by1 = P + E.b[1]; }
  else {
    ea.We1 = P + E.w[1]; ea.be1 = P + E.b[1];
    if (a.model == GMVAE_MODEL_VAE_GMP) { ea.loc = P + L.loc; ea.raw_scale = P + L.rawscale; ea.mixlog = P + L.mixlog; }
  }
  ea.eps = a.eps; ea.u = a.u; ea.seed = a.seed; ea.step = a.step; ea.row_base = (unsigned long long)d.row0 * S;
  ea.c = d.raw_sigma_bias; ea.smin = d.sigma_min; ea.invT = 1.f / d.temperature;
  ea.rows4 = a.row_terms; ea.z_out = a.z_out; ea.y_out = a.y_out; ea.logits_out = a.logits_out;
  ea.rows_ws = a.row_terms ? a.row_terms : w.ev_rows;
  ea.slots = w.ev_slots; ea.counter = counter; ea.tail = a.tail;
  ea.dbg = getenv("GMVAE_EV_STAMPS") ? w.ev_dbg : nullptr;
  static bool eattr[64];
  if (first_on_device(eattr)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows<0>), hipFuncAttributeMaxDynamicSharedMemorySize, EV::lds * (int)sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows<1>), hipFuncAttributeMaxDynamicSharedMemorySize, EV::lds * (int)sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows<2>), hipFuncAttributeMaxDynamicSharedMemorySize, EV::lds * (int)sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows_v<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, EVV::lds * (int)sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows_v<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, EVV::lds * (int)sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(evalf_rows_v<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, EVV::lds * (int)sizeof(float));
  }
  int grid = device_cus();
  if (grid > B) grid = B;
  if (grid > 1024) grid = 1024;
  const int ev_mode = getenv("GMVAE_EV_MODE") ? atoi(getenv("GMVAE_EV_MODE")) : 0;      // (timing experiments: wrong results)
  const size_t shv = (size_t)EVV::lds * sizeof(float);
  if (!gm) {
    if (a.model == GMVAE_MODEL_VAE_GMP) hipLaunchKernelGGL((evalf_rows_v<1, 64>), dim3(grid), dim3(kMT), shv, st, ea);
    else if (d.L == 2) hipLaunchKernelGGL((evalf_rows_v<0, 2>), dim3(grid), dim3(kMT), shv, st, ea);
    else hipLaunchKernelGGL((evalf_rows_v<0, 64>), dim3(grid), dim3(kMT), shv, st, ea);
    cx.check();
    const double Rv = (double)B * S;
    cx.mark("evalf_rows_v", 2.0 * B * EV::H * 2 * d.L + 2.0 * Rv * ((double)d.L * EV::H + (double)EV::H * D));
    return cx.err;
  }
  if (ev_mode == 1) hipLaunchKernelGGL(evalf_rows<1>, dim3(grid), dim3(kMT), (size_t)EV::lds * sizeof(float), st, ea);
  else if (ev_mode == 2) hipLaunchKernelGGL(evalf_rows<2>, dim3(grid), dim3(kMT), (size_t)EV::lds * sizeof(float), st, ea);
  else hipLaunchKernelGGL(evalf_rows<0>, dim3(grid), dim3(kMT), (size_t)EV::lds * sizeof(float), st, ea);
  cx.check();
  const double R = (double)B * S;
  cx.mark("evalf_rows", 2.0 * B * EV::H * EV::K + 2.0 * R * (EV::K * (EV::H + EV::L2) + EV::H * EV::L2 + EV::L * EV::H + (double)EV::H * D));
  return cx.err;
}

static int run_step(Ctx& cx, const StepArgs& a) {
  const GmvaeDims& d = *a.d;
  const int model = a.model;
  Layout L;
  build_layout(d, model, L);
  WS w;
  carve(d, model, L, a.workspace, w);
  const int B = d.B, S = d.S, R = B * S, K = d.K, Lz = d.L, D = d.D;
  const float* P = a.params;
  const bool gm = model == GMVAE_MODEL_GMVAE;
  const float c = d.raw_sigma_bias, smin = d.sigma_min;
  hipStream_t st = cx.st;
  const bool planes = planes_ok(d, L) && w.hd3 != nullptr;
  // the piece form of the plane GEMMs: f16 pairs (3 piece products, <= 3 x 2^-22 per product) unless GMVAE_PLANES_EXACT=1 asks for
  // the bf16 triples (6 piece products, every product exact)
  const char* const exact_env = getenv("GMVAE_PLANES_EXACT");
  const bool pairs = planes && !(exact_env && atoi(exact_env));
  const bool fwdp = !planes && !a.backward && w.hd2f != nullptr && fwd_pairs_ok(d, L);      // forward only: the logits GEMM on pairs
  constexpr float kGScale = 32768.f;            // (sigmoid - x) in [-1, 1]: a fixed scale for its pairs

  // ---- noise (fast mode): Philox for eps and u -- its own launch in the general schedule, auxiliary
  // workgroups of the first GEMM launch in the fused one
  const float* eps = a.eps;
  const float* u = a.u;
  float* ge = eps ? nullptr : w.eps;
  float* gu = (gm && !u) ? w.u : nullptr;
  if (ge) eps = ge;
  if (gu) u = gu;
  tl_hact = 1 + d.hidden_act;                  // (every Problem built below for this step: its epilogue's activation kind)
  if (a.backward && mega_ok(d, model)) return run_step_mega(cx, a, L, w, eps, u, ge, gu);
  if (a.backward && skinny_ok(d, model)) return run_step_skinny(cx, a, L, w, eps, u, ge, gu);
  if (!a.backward && evalf_ok(d, model) && w.ev_img) return run_eval_fused(cx, a, L, w);
  if (fused_ok(d, model) && !a.z_out && !a.y_out && !a.logits_out)
    return run_step_fused(cx, a, L, w, eps, u, ge, gu);
  // (general schedule: the Philox fill rides as auxiliary workgroups of the first GEMM launch below)
  const bool noise_aux = ge || gu;
  const bool relu_act = d.hidden_act == GMVAE_ACT_RELU;      // (tanh / sigmoid / ELU: the grouped GEMM's epilogues only)

  // row-panel layers over thousands of rows as register-direct bf16 piece products (skinny.hpp rows_nn_bf6): K % 32 = 0, widths
  // % 64 = 0, R >= 2048 -- else the grouped GEMM
  auto rows_ok = [&](int Kd, int N0, int N1) { return relu_act && R >= 2048 && Kd % 32 == 0 && N0 % 64 == 0 && N1 % 64 == 0; };
  int rows_units = 0;                            // waves with work of the last launch_rows (one partial maximum each: RowsProb::amax)
  auto launch_rows = [&](RowsArgs& ra, const char* name) {
    const int nct = (ra.p[0].N + (ra.np > 1 ? ra.p[1].N : 0)) / 64;
    int rt = 4;
    // (row tiles per wave: as many as leave 1024 units -- every wave splits its W fragments itself, so fewer, fatter units
    //  win while the chip stays full: fwd_enc_gmm at the config-5 sizes 43.1 us with 3200 units, 37.1 with 1600, 48.3 with 800)
    while (rt > 1 && (long long)((R + 16 * rt - 1) / (16 * rt)) * nct < 1024) rt >>= 1;
    const long long units = (long long)((R + 16 * rt - 1) / (16 * rt)) * nct;
    rows_units = (int)units;
    const dim3 grid((unsigned)((units + kSkWaves - 1) / kSkWaves));
    if (rt == 4) hipLaunchKernelGGL(rows_nn_bf6<4>, grid, dim3(kSkThreads), 0, st, ra);
    else if (rt == 2) hipLaunchKernelGGL(rows_nn_bf6<2>, grid, dim3(kSkThreads), 0, st, ra);
    else hipLaunchKernelGGL(rows_nn_bf6<1>, grid, dim3(kSkThreads), 0, st, ra);
    cx.check();
    double fl = 0;
    for (int i = 0; i < ra.np; ++i) fl += 2.0 * R * ra.K * ra.p[i].N;
    cx.mark(name, fl);
  };
  // the same layers with the weight stationary (rowsws.hpp) where a wave's slice of it fits 96 registers: K = 64 (NN) / 128 (NT)
  // in a wave, K = 512 x N = 64 (NT) over the eight waves of a workgroup; one resident wave of workgroups
  const bool rws_on = relu_act && R >= 2048 && !getenv("GMVAE_NO_RWS");
  auto rws_grid = [&]() { const int cu = device_cus(); return cu * kSkWaves > kRwsMaxWaves ? kRwsMaxWaves / kSkWaves : cu; };
  auto rws_fits = [&](long long slices) { return slices >= 1 && slices <= (long long)rws_grid() * kSkWaves; };      // (a wave per column slice at least)
  auto rws_prob = [&](const float* W, int ldw, const float* bias, float* out, int N, bool relu) {
    RwsProb q;
    memset(&q, 0, sizeof(q));
    q.W = W; q.ldw = ldw; q.bias = bias; q.out = out; q.N = N; q.relu = relu ? 1 : 0; q.add_div = 1;
    return q;
  };
  auto launch_rws = [&](RwsArgs& ra, const int form, const char* name, const int K) {      // form 0: NN K = 64; 1: NT K = 128; 2: NT K = 512, N = 64
    const int grid = rws_grid();
    if (form == 0) {
      ra.ns0 = ra.p[0].N / 64; ra.ns = ra.ns0 + (ra.np > 1 ? ra.p[1].N / 64 : 0);
      hipLaunchKernelGGL((rows_ws<2, 4, false>), dim3(grid), dim3(kSkThreads), 0, st, ra);
    } else if (form == 1) {
      ra.ns0 = ra.p[0].N / 32; ra.ns = ra.ns0 + (ra.np > 1 ? ra.p[1].N / 32 : 0);
      hipLaunchKernelGGL((rows_ws<4, 2, true>), dim3(grid), dim3(kSkThreads), 0, st, ra);
    } else {
      ra.ns0 = ra.ns = 1;
      hipLaunchKernelGGL(rows_ws_k8, dim3(grid), dim3(kSkThreads), 0, st, ra);
    }
    rows_units = grid * kSkWaves;
    cx.check();
    double fl = 0;
    for (int i = 0; i < ra.np; ++i) fl += 2.0 * R * K * ra.p[i].N;
    cx.mark(name, fl);
  };
  auto rows_prob = [&](const float* W, const float* bias, float* out, int N, bool relu) {
    RowsProb q;
    memset(&q, 0, sizeof(q));
    q.W = W; q.bias = bias; q.out = out; q.N = N; q.relu = relu ? 1 : 0; q.add_div = 1;
    return q;
  };

  // ================================ forward ================================
  const NetL& E = gm ? L.ency : L.enc;
  const int flx_h0 = E.dim[1], flx_h1 = gm ? L.encg.dim[1] : 0;
  if (relu_act && D % 16 == 0 && flx_h0 % 64 == 0 && flx_h1 % 64 == 0 && E.nl > 1) {
    // first layers over the uint8 batch as exact bf16 piece products (skinny.hpp first_layers_u8bf); the Philox fill in a
    // launch of its own
    FlxArgs f;
    f.x = a.x; f.W0 = P + E.w[0]; f.b0 = P + E.b[0]; f.out0 = w.he[1]; f.H0 = flx_h0; f.relu0 = 1;
    f.W1 = gm ? P + L.encg.w[0] : nullptr; f.out1 = gm ? w.gx : nullptr; f.H1 = flx_h1;
    f.B = B; f.D = D;
    const int nct = (flx_h0 + flx_h1) / 64, nrt = (B + 15) / 16;
    int rt = 4;
    while (rt > 1 && nct * ((nrt + rt - 1) / rt) < 256) rt >>= 1;
    const int grid = nct * ((nrt + rt - 1) / rt);
    if (rt == 4) hipLaunchKernelGGL(first_layers_u8bf<4>, dim3(grid), dim3(kSkThreads), 0, st, f);
    else if (rt == 2) hipLaunchKernelGGL(first_layers_u8bf<2>, dim3(grid), dim3(kSkThreads), 0, st, f);
    else hipLaunchKernelGGL(first_layers_u8bf<1>, dim3(grid), dim3(kSkThreads), 0, st, f);
    cx.check();
    cx.mark("fwd_x_first_layers", 2.0 * B * D * (flx_h0 + flx_h1));
    if (noise_aux) {
      const uint64_t q = noise_items(ge, gu, (uint64_t)R, Lz, K);
      hipLaunchKernelGGL(noise_fill, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, ge, gu, (uint64_t)R, Lz, K,
                         (uint64_t)d.row0 * S, (uint64_t)a.seed, (uint64_t)a.step, reinterpret_cast<const uint64_t*>(a.step_dev));
      rowk(cx, "noise_fill");
    }
  } else
  {  // first layers over the uint8 batch: enc_y layer 0 and the x-part of enc_gmm layer 0
    Group g;
    float* out = (E.nl == 1) ? (gm ? w.logits : w.qp) : w.he[1];
    g.add(p_nn(a.x, true, D, P + E.w[0], E.dim[1], B, E.dim[1], D, out, E.dim[1], P + E.b[0], E.nl > 1));
    if (gm)
      g.add(p_nn(a.x, true, D, P + L.encg.w[0], L.encg.dim[1], B, L.encg.dim[1], D, w.gx, L.encg.dim[1], nullptr,
                 false));
    if (noise_aux) {
      Aux& ax = g.L.aux;
      ax.eps = ge; ax.u = gu; ax.n_rows = (unsigned long long)R; ax.nL = Lz; ax.nK = K; ax.row_base = (unsigned long long)d.row0 * S;
      ax.seed = a.seed; ax.step = a.step;
      ax.step_dev = reinterpret_cast<unsigned long long*>(a.step_dev);
      ax.epoch_word = nullptr;
      ax.noise_blocks = (int)((noise_items(ge, gu, (uint64_t)R, Lz, K) + kThreads - 1) / kThreads);
      ax.ntasks = 0;
      ax.nblocks = ax.noise_blocks;
      g.L.aux_last = 1;
    }
    launch_group(cx, g, noise_aux ? "fwd_x_first_layers+noise" : "fwd_x_first_layers");
  }
  for (int i = 1; i < E.nl; ++i) {
    Group g;
    float* out = (i == E.nl - 1) ? (gm ? w.logits : w.qp) : w.he[i + 1];
    g.add(p_nn(w.he[i], false, E.dim[i], P + E.w[i], E.dim[i + 1], B, E.dim[i + 1], E.dim[i], out, E.dim[i + 1],
               P + E.b[i], i < E.nl - 1));
    launch_group(cx, g, gm ? "fwd_enc_y" : "fwd_enc");
  }
  if (gm) {
    hipLaunchKernelGGL(y_head_fwd, dim3(grid_for(R, 4)), dim3(256), 0, st, w.logits, u, w.y, w.nent, R, S, K,
                       1.f / d.temperature);
    rowk(cx, "y_head_fwd");
    const NetL& G = L.encg;
    if (rws_on && K == 64 && G.dim[1] % 64 == 0 && (2 * Lz) % 64 == 0 && rws_fits((G.dim[1] + 2 * Lz) / 64)) {
      RwsArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.A = w.y; ra.lda = K; ra.R = R; ra.np = 2;
      ra.p[0] = rws_prob(P + G.w[0] + (uint64_t)D * G.dim[1], G.dim[1], P + G.b[0], (G.nl == 1) ? w.qp : w.hg[1], G.dim[1], G.nl > 1);
      ra.p[0].addsrc = w.gx; ra.p[0].ld_add = G.dim[1]; ra.p[0].add_div = S;
      ra.p[1] = rws_prob(P + L.prior.w[0], 2 * Lz, P + L.prior.b[0], w.pp, 2 * Lz, false);
      launch_rws(ra, 0, "fwd_y_layers", K);
    } else if (rows_ok(K, G.dim[1], 2 * Lz)) {
      RowsArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.A = w.y; ra.R = R; ra.K = K; ra.np = 2;
      ra.p[0] = rows_prob(P + G.w[0] + (uint64_t)D * G.dim[1], P + G.b[0], (G.nl == 1) ? w.qp : w.hg[1], G.dim[1], G.nl > 1);
      ra.p[0].addsrc = w.gx; ra.p[0].ld_add = G.dim[1]; ra.p[0].add_div = S;
      ra.p[1] = rows_prob(P + L.prior.w[0], P + L.prior.b[0], w.pp, 2 * Lz, false);
      launch_rows(ra, "fwd_y_layers");
    } else if (K <= kSmallKMax && G.dim[1] % 4 == 0 && Lz % 2 == 0 && G.dim[1] + 2 * Lz <= kSmallKCols && (relu_act || G.nl == 1) &&
               (long long)R * (G.dim[1] + 2 * Lz) < (1ll << 32)) {
      // the reference's K = 10: a 10-deep contraction is no GEMM (kernels.hpp rows_small_k)
      SmallKArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.A = w.y; sa.R = R; sa.K = K; sa.np = 2;
      sa.p[0].W = P + G.w[0] + (uint64_t)D * G.dim[1]; sa.p[0].bias = P + G.b[0]; sa.p[0].out = (G.nl == 1) ? w.qp : w.hg[1];
      sa.p[0].N = G.dim[1]; sa.p[0].relu = G.nl > 1; sa.p[0].addsrc = w.gx; sa.p[0].ld_add = G.dim[1]; sa.p[0].add_div = S;
      sa.p[1].W = P + L.prior.w[0]; sa.p[1].bias = P + L.prior.b[0]; sa.p[1].out = w.pp; sa.p[1].N = 2 * Lz; sa.p[1].add_div = 1;
      // (a grid whose thread count is a multiple of the row's quads: a thread keeps its column quad and its weights in registers)
      const int qn = (G.dim[1] + 2 * Lz) / 4;
      // (about one wave of resident workgroups: each stages the weights once)
      int nblk = grid_for((long long)R * qn, 256 * 4, 4 * device_cus());
      nblk = nblk >= qn ? nblk / qn * qn : qn;
      if (K == 10) hipLaunchKernelGGL(rows_small_k<10>, dim3(nblk), dim3(256), 0, st, sa);
      else hipLaunchKernelGGL(rows_small_k<0>, dim3(nblk), dim3(256), 0, st, sa);
      cx.check();
      cx.mark("fwd_y_layers", 2.0 * R * K * (G.dim[1] + 2 * Lz));
    } else {
      Group g;
      Problem p = p_nn(w.y, false, K, P + G.w[0] + (uint64_t)D * G.dim[1], G.dim[1], R, G.dim[1], K,
                       (G.nl == 1) ? w.qp : w.hg[1], G.dim[1], P + G.b[0], G.nl > 1);
      p.addsrc = w.gx; p.ld_add = G.dim[1]; p.add_div = S;
      g.add(p);
      g.add(p_nn(w.y, false, K, P + L.prior.w[0], 2 * Lz, R, 2 * Lz, K, w.pp, 2 * Lz, P + L.prior.b[0], false));
      launch_group(cx, g, "fwd_y_layers");
    }
    for (int i = 1; i < G.nl; ++i) {
      float* const out = (i == G.nl - 1) ? w.qp : w.hg[i + 1];
      if (rows_ok(G.dim[i], G.dim[i + 1], 64)) {
        RowsArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.A = w.hg[i]; ra.R = R; ra.K = G.dim[i]; ra.np = 1;
        ra.p[0] = rows_prob(P + G.w[i], P + G.b[i], out, G.dim[i + 1], i < G.nl - 1);
        launch_rows(ra, "fwd_enc_gmm");
        continue;
      }
      Group g;
      g.add(p_nn(w.hg[i], false, G.dim[i], P + G.w[i], G.dim[i + 1], R, G.dim[i + 1], G.dim[i], out, G.dim[i + 1],
                 P + G.b[i], i < G.nl - 1));
      launch_group(cx, g, "fwd_enc_gmm");
    }
  }
  const int prior = gm ? PRIOR_COND : (model == GMVAE_MODEL_VAE ? PRIOR_STD : PRIOR_GMP);
  const int qp_div = gm ? 1 : S;
  hipLaunchKernelGGL(z_head_fwd, dim3(grid_for(R, 4)), dim3(256), 0, st, w.qp, qp_div, w.pp, eps, w.z, w.logq,
                     w.logp, R, Lz, prior, c, smin);
  rowk(cx, "z_head_fwd");
  if (prior == PRIOR_GMP) launch_mixture_logprob(cx, w, P, L, R, Lz, K);
  int nparts = 1;
  const NetL& Dn = L.dec;
  bool hd3_fused = false;
  int hmax_n = 0;                                // partial maxima the launch producing the top layer's input left (f16 pairs)
  for (int i = 0; i < Dn.nl; ++i) {
    Group g;
    const float* in = (i == 0) ? w.z : w.hd[i];
    if (i < Dn.nl - 1 && rws_on && Dn.dim[i] == 64 && Dn.dim[i + 1] % 64 == 0 && rws_fits(Dn.dim[i + 1] / 64) && !(planes && !pairs && i == Dn.nl - 2)) {
      RwsArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.A = in; ra.lda = Dn.dim[i]; ra.R = R; ra.np = 1;
      ra.p[0] = rws_prob(P + Dn.w[i], Dn.dim[i + 1], P + Dn.b[i], w.hd[i + 1], Dn.dim[i + 1], true);
      if ((pairs || fwdp) && i == Dn.nl - 2) ra.p[0].amax = reinterpret_cast<unsigned*>(pairs ? w.pscale : w.pscale_f) + 16 + kAmaxBlocks;
      launch_rws(ra, 0, "fwd_dec", Dn.dim[i]);
      if (ra.p[0].amax) hmax_n = rows_units;
    } else if (i < Dn.nl - 1 && rows_ok(Dn.dim[i], Dn.dim[i + 1], 64)) {
      RowsArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.A = in; ra.R = R; ra.K = Dn.dim[i]; ra.np = 1;
      ra.p[0] = rows_prob(P + Dn.w[i], P + Dn.b[i], w.hd[i + 1], Dn.dim[i + 1], true);
      if (planes && !pairs && i == Dn.nl - 2) {    // the top layer's input activation also as planes (no split launch over R x H)
        ra.p[0].C3 = w.hd3; ra.p[0].c3_stride = (long long)R * Dn.dim[i + 1];
        hd3_fused = true;
      }
      if ((pairs || fwdp) && i == Dn.nl - 2) {     // f16 pairs: the activation's largest magnitude rides on this launch (a word per wave)
        ra.p[0].amax = reinterpret_cast<unsigned*>(pairs ? w.pscale : w.pscale_f) + 16 + kAmaxBlocks;
      }
      launch_rows(ra, "fwd_dec");
      if (ra.p[0].amax) hmax_n = rows_units;
    } else if (i < Dn.nl - 1) {
      Problem ph = p_nn(in, false, Dn.dim[i], P + Dn.w[i], Dn.dim[i + 1], R, Dn.dim[i + 1], Dn.dim[i], w.hd[i + 1],
                        Dn.dim[i + 1], P + Dn.b[i], true);
      if (planes && !pairs && i == Dn.nl - 2) {
        // the top layer's input activation leaves this launch's epilogue as planes too (no split launch over R x H)
        ph.C3 = w.hd3; ph.c3_stride = (long long)R * Dn.dim[i + 1];
        hd3_fused = true;
      }
      g.add(ph);
      launch_group(cx, g, "fwd_dec");
    } else {  // logits -> Bernoulli log-prob partials + (sigmoid(logit) - x)
      Problem p = p_nn(in, false, Dn.dim[i], P + Dn.w[i], D, R, D, Dn.dim[i], a.backward ? w.g : nullptr, D,
                       P + Dn.b[i], false);
      p.epi = EPI_BERNOULLI;
      p.addconst = d.gen_bias_init;
      p.bias2 = d.gen_bias_vec;
      p.x = a.x; p.ldx = D; p.x_div = S; p.part = w.part;
      if (fwdp) {
        // forward only: both operands as f16 pairs, the weight's planes zero-padded to whole 128-column tiles
        const int Dp = (D + 127) / 128 * 128;
        const long long nh = (long long)R * Dn.dim[i], nw = (long long)Dn.dim[i] * D;
        unsigned* const am = reinterpret_cast<unsigned*>(w.pscale_f);
        unsigned* const hp = am + 16 + kAmaxBlocks;
        if (!hmax_n) { launch_amax(st, in, nh, hp); hmax_n = kAmaxBlocks; }
        launch_amax(st, P + Dn.w[i], nw, am + 16);
        launch_amax_final(st, hp, hmax_n, am + 16, kAmaxBlocks, am, w.pscale_f + 2);
        launch_split_pairs(st, in, nullptr, Dn.dim[i], nh, w.hd2f, am);
        launch_split_pairs(st, P + Dn.w[i], nullptr, D, nw, w.w2f, am + 1, Dp);
        rowk(cx, "split_planes");
        p.planes = 2; p.n_padded = 1;
        p.a_uns = w.pscale_f + 2; p.b_uns = w.pscale_f + 3; p.uns_c = 1.f; p.uns_cb = 1.f;
        p.seg[0].a.ptr = w.hd2f; p.seg[0].b.ptr = w.w2f;
        p.seg[0].b.ld = Dp; p.seg[0].b.n_mn = Dp;      // (the planes' own extents: padded)
        p.a_pstride = nh; p.b_pstride = (long long)Dn.dim[i] * Dp;
        p.C = nullptr;
      } else if (planes) {
        // both operands as planes of 16-bit pieces (the weight's are shared with the data gradient below), (sigmoid - x)
        // leaves as planes only: its two consumers are plane GEMMs
        const long long nh = (long long)R * Dn.dim[i], nw = (long long)Dn.dim[i] * D;
        if (pairs) {
          // f16 pairs need the tensors' scales first: largest magnitudes (two small reductions), then the scaled splits
          unsigned* const am = reinterpret_cast<unsigned*>(w.pscale);
          unsigned* const hp = am + 16 + kAmaxBlocks;
          if (!hmax_n) { launch_amax(st, in, nh, hp); hmax_n = kAmaxBlocks; }
          launch_amax(st, P + Dn.w[i], nw, am + 16);
          launch_amax_final(st, hp, hmax_n, am + 16, kAmaxBlocks, am, w.pscale + 2);
          launch_split_pairs(st, in, nullptr, Dn.dim[i], nh, w.hd3, am);
          launch_split_pairs(st, P + Dn.w[i], nullptr, D, nw, w.w3, am + 1);
          p.planes = 2;
          p.a_uns = w.pscale + 2; p.b_uns = w.pscale + 3; p.uns_c = 1.f; p.uns_cb = 1.f;
          p.c3_scale = kGScale;
        } else {
          if (!hd3_fused) launch_split(st, in, nullptr, Dn.dim[i], nh, w.hd3);
          launch_split(st, P + Dn.w[i], nullptr, D, nw, w.w3);
          p.planes = 1;
        }
        rowk(cx, "split_planes");
        p.seg[0].a.ptr = w.hd3; p.seg[0].b.ptr = w.w3;
        p.a_pstride = nh; p.b_pstride = nw;
        p.C = nullptr;
        p.C3 = a.backward ? w.g3 : nullptr; p.c3_stride = (long long)R * D;
      }
      g.add(p);
      const int bn = cfg_bn(launch_group(cx, g, "fwd_dec_bernoulli"));
      nparts = (D + bn - 1) / bn;
    }
  }
  float* tail = a.backward ? a.grads + L.P_pad : a.tail;
  const float* rwS = (S > 1 && a.backward) ? w.rw : nullptr;
  if (S > 1 && S <= 64) {       // the row terms and the IWAE groups in one launch (a wave per batch row)
    hipLaunchKernelGGL(iwae_rows_terms, dim3((B + 3) / 4), dim3(256), 0, st, w.part, nparts, w.logq, w.logp,
                       gm ? w.nent : (const float*)nullptr, w.logpx, w.logw, a.row_terms, a.backward ? w.rw : (float*)nullptr, w.pb, B, S);
    rowk(cx, "iwae_rows_terms");
  } else {
    hipLaunchKernelGGL(row_terms, dim3(grid_for(R, 256, 1 << 22)), dim3(256), 0, st, w.part, nparts, w.logq, w.logp,
                       gm ? w.nent : (const float*)nullptr, S, w.logpx, w.logw, a.row_terms, R, S > 1 ? w.lw64 : (double*)nullptr);
    rowk(cx, "row_terms");
  }
  if (S > 64) {
    hipLaunchKernelGGL(iwae_rows, dim3((B + 3) / 4), dim3(256), 0, st, w.lw64, w.logpx, w.logq, w.logp,
                       a.backward ? w.rw : (float*)nullptr, w.pb, B, S);
    rowk(cx, "iwae_rows");
  }
  hipLaunchKernelGGL(loss_tail, dim3(1), dim3(1024), 0, st, w.logw, w.logpx, w.logq, w.logp,
                     gm ? w.nent : (const float*)nullptr, (float*)nullptr, tail, B, S, a.step_dev,
                     S > 1 ? w.pb : (const float*)nullptr);
  rowk(cx, "loss_tail");
  if (a.z_out) hipMemcpyAsync(a.z_out, w.z, (size_t)R * Lz * 4, hipMemcpyDeviceToDevice, st);
  if (a.y_out && gm) hipMemcpyAsync(a.y_out, w.y, (size_t)R * K * 4, hipMemcpyDeviceToDevice, st);
  if (a.logits_out && gm) hipMemcpyAsync(a.logits_out, w.logits, (size_t)B * K * 4, hipMemcpyDeviceToDevice, st);
  if (!a.backward) return cx.err;

  // ================================ backward ===============================
  const int NS = num_splits(R);
  // S > 1: weight gradients whose contraction runs over the B batch rows (the encoder of x, and the x rows of enc_gmm's
  // first layer after sum_over_s) need only num_splits(B) slabs: finalize_grads then reads 2 instead of 16 slabs for the
  // two largest tensors of the config-5 sizes, and their launches write as many fewer
  const int NSB = S > 1 ? (num_splits(B) < NS ? num_splits(B) : NS) : NS;
  // Weight gradients with few outputs and a contraction over all R rows (decoder layer 0, the layers of enc_gmm after the
  // first, the prior, the y rows of enc_gmm's first layer) take MORE splits than NS -- up to 64 -- at large R: as 64 tiles of
  // 50 rounds each such launch was a 125 us latency chain at the config-5 sizes.  (The slab buffer is sized for it: carve.)
  const int NSS = num_splits_small(R) > NS ? num_splits_small(R) : NS;
  SlabX sxb;
  memset(&sxb, 0, sizeof(sxb));
  auto brange = [&](uint64_t b, uint64_t n, int ns) {
    if (ns != NS && sxb.n < kSlabRanges) { sxb.b[sxb.n] = (int)b; sxb.e[sxb.n] = (int)(b + pad4(n)); sxb.ns[sxb.n] = ns; sxb.n++; }
  };
  auto small_ns = [&](long long outs) { return (NSS != NS && outs <= 131072 && sxb.n + 2 <= kSlabRanges) ? NSS : NS; };
  const long long PP = (long long)L.P_pad;
  float* sl = w.slabs;
  int pb = 0;
  const float* dcur = w.g;
  for (int i = Dn.nl - 1; i >= 0; --i) {   // decoder: dW_i (+db_i) and dX_i in one launch
    Group g;
    const bool top = (i == Dn.nl - 1);
    const float* act = (i == 0) ? w.z : w.hd[i];
    int nsd = small_ns((long long)Dn.dim[i] * Dn.dim[i + 1]);
    if (top && planes && sxb.n + 2 <= kSlabRanges) {
      // The plane launch's weight gradient: as many slabs as put its tiles on the chip ONCE with a few CUs to spare -- tiles per
      // slab x slabs just under the resident workgroups (256-row tiles: one per CU).  Measured at the config-5 shard (48 tiles per
      // slab, tools/ab/ns_top.sh, us per step on one box): 16 slabs 1315, 12 1283, 8 1308, 7 1314, 6 1283, **5 1228**, 4 1252,
      // 3 1273, 2 1469 -- bwd_dec_top 526 -> 454 (the slabs it writes 100 -> 31 MB, and no second round of weight-gradient tiles
      // behind the data gradient's), finalize_grads 40 -> 27; D = 2048 (32 per slab): 7 slabs; hidden 1024 (96): 2; B = 1024: 5.
      const bool x256 = pairs && Dn.dim[i] % 256 == 0 && R % 256 == 0;
      const int tps = (Dn.dim[i] / (x256 ? 256 : 128)) * (int)(D / 128), cap = device_cus() * (x256 ? 1 : 2);
      int nst = (cap - cap / 16) / (tps > 0 ? tps : 1);
      nst = nst < 1 ? 1 : (nst > NS ? NS : nst);
      const char* e = getenv("GMVAE_NSPLIT_TOP");      // (A/B)
      if (e && atoi(e) >= 1 && atoi(e) <= NS) nst = atoi(e);
      nsd = nst;
    }
    if (nsd != NS) { brange(Dn.w[i], (uint64_t)Dn.dim[i] * Dn.dim[i + 1], nsd); brange(Dn.b[i], Dn.dim[i + 1], nsd); }
    const Problem pw = p_tn(act, false, Dn.dim[i], 1, dcur, Dn.dim[i + 1], Dn.dim[i], Dn.dim[i + 1], R, sl + Dn.w[i], sl + Dn.b[i],
                            nsd, PP, top ? rwS : nullptr);
    float* out = (i == 0) ? w.dz : w.dbuf[pb];
    Problem p = p_nt(dcur, Dn.dim[i + 1], P + Dn.w[i], Dn.dim[i + 1], R, Dn.dim[i], Dn.dim[i + 1], out, Dn.dim[i],
                     (i > 0) ? w.hd[i] : nullptr, Dn.dim[i]);
    p.rowscale = top ? rwS : nullptr;
    if (top && planes) {
      const long long nh = (long long)R * Dn.dim[i], nw = (long long)Dn.dim[i] * D, ng = (long long)R * D;
      if (rwS) {           // IWAE: the row weights ride on the activation's pieces (they cannot scale pieces inside the loop)
        // (pairs: under the scale of the unweighted activation -- the weights are <= 1)
        if (pairs) launch_split_pairs(st, act, rwS, Dn.dim[i], nh, w.hd3, reinterpret_cast<unsigned*>(w.pscale));
        else launch_split(st, act, rwS, Dn.dim[i], nh, w.hd3);
        rowk(cx, "split_planes_rw");
      }
      Problem pw3 = pw;
      pw3.seg[0].a.ptr = w.hd3; pw3.seg[0].b.ptr = w.g3;
      pw3.planes = pairs ? 2 : 1; pw3.a_pstride = nh; pw3.b_pstride = ng;      // (seg[0].kscale = rwS now only weighs the column sums)
      p.seg[0].a.ptr = w.g3; p.seg[0].b.ptr = w.w3;
      p.planes = pairs ? 2 : 1; p.a_pstride = ng; p.b_pstride = nw;
      if (pairs) {
        pw3.a_uns = w.pscale + 2; pw3.uns_c = 1.f / kGScale; pw3.uns_cb = 1.f / kGScale;
        p.b_uns = w.pscale + 3; p.uns_c = 1.f / kGScale; p.uns_cb = 1.f;
      }
      g.add(pw3);
    } else {
      g.add(pw);
    }
    // a thin data gradient behind a wide layer (dz = dhd Wd0^T: 512 -> 64) with the weight stationary, in a launch of its own
    const bool rws_dz = !top && rws_on && i == 0 && Dn.dim[1] == 512 && Dn.dim[0] == 64;
    if (!rws_dz) g.add(p);
    launch_group(cx, g, top ? "bwd_dec_top" : "bwd_dec");
    if (rws_dz) {
      RwsArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.A = dcur; ra.lda = Dn.dim[1]; ra.R = R; ra.np = 1;
      ra.p[0] = rws_prob(P + Dn.w[0], Dn.dim[1], nullptr, out, Dn.dim[0], false);
      launch_rws(ra, 2, "bwd_dec_dz", Dn.dim[1]);
    }
    dcur = out;
    pb ^= 1;
  }
  hipLaunchKernelGGL(z_head_bwd, dim3(grid_for(R, 4)), dim3(256), 0, st, w.dz, w.qp, qp_div, w.pp, eps, w.z, rwS,
                     w.resp, P + L.loc, P + L.rawscale, w.dqp, w.dpp, R, Lz, K, prior, c, smin);
  rowk(cx, "z_head_bwd");

  if (gm) {
    const NetL& G = L.encg;
    bool prior_done = false;
    auto prior_dw = [&]() {
      const int nsp = small_ns((long long)K * 2 * Lz);
      if (nsp != NS) { brange(L.prior.w[0], (uint64_t)K * 2 * Lz, nsp); brange(L.prior.b[0], 2 * Lz, nsp); }
      return p_tn(w.y, false, K, 1, w.dpp, 2 * Lz, K, 2 * Lz, R, sl + L.prior.w[0], sl + L.prior.b[0], nsp, PP, nullptr);
    };
    dcur = w.dqp;
    for (int i = G.nl - 1; i >= 1; --i) {
      Group g;
      const int nsg = small_ns((long long)G.dim[i] * G.dim[i + 1]);
      if (nsg != NS) { brange(G.w[i], (uint64_t)G.dim[i] * G.dim[i + 1], nsg); brange(G.b[i], G.dim[i + 1], nsg); }
      g.add(p_tn(w.hg[i], false, G.dim[i], 1, dcur, G.dim[i + 1], G.dim[i], G.dim[i + 1], R, sl + G.w[i], sl + G.b[i],
                 nsg, PP, nullptr));
      float* out = w.dbuf[pb];
      // (dhg = dqp Wg1^T under the ReLU mask, K = 128: the weight stationary, in a launch of its own)
      const bool rws_dh = rws_on && G.dim[i + 1] == 128 && G.dim[i] % 32 == 0 && rws_fits(G.dim[i] / 32);
      if (!rws_dh)
        g.add(p_nt(dcur, G.dim[i + 1], P + G.w[i], G.dim[i + 1], R, G.dim[i], G.dim[i + 1], out, G.dim[i], w.hg[i],
                   G.dim[i]));
      // (the prior's small weight gradient rides on the layer-0 launch below: here it would be the one problem that keeps
      // a launch of 128-aligned problems off the big-round GEMM instance)
      launch_group(cx, g, "bwd_enc_gmm");
      if (rws_dh) {
        RwsArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.A = dcur; ra.lda = G.dim[i + 1]; ra.R = R; ra.np = 1;
        ra.p[0] = rws_prob(P + G.w[i], G.dim[i + 1], nullptr, out, G.dim[i], false);
        ra.p[0].mask = w.hg[i]; ra.p[0].ld_mask = G.dim[i];
        launch_rws(ra, 1, "bwd_enc_gmm_dh", G.dim[i + 1]);
      }
      dcur = out;
      pb ^= 1;
    }
    {  // layer 0 of enc_gmm: input is concat(x, y) -- x rows need no gradient
      // S > 1: the S samples of a batch row share their x, so dW[x part] = x^T (sum_s d_g0[b, s]): the gradient rows are
      // summed over s first and the product runs over B rows instead of B*S (and on the bf16 matrix cores: the byte
      // operand is then a plain row-major matrix).  Its bias gradient is the column sum of the same B rows.
      const float* dg = dcur;
      if (S > 1) {
        hipLaunchKernelGGL(sum_over_s, dim3(grid_for((long long)B * G.dim[1], 256, 1 << 22)), dim3(256), 0, st, dcur,
                           w.dsum, B, S, G.dim[1]);
        rowk(cx, "sum_over_s");
        dg = w.dsum;
      }
      Group g;
      const float* Wy = P + G.w[0] + (uint64_t)D * G.dim[1];
      const int nsx = (S > 1 && sxb.n + 2 <= kSlabRanges) ? NSB : NS;
      if (nsx != NS) { brange(G.w[0], (uint64_t)D * G.dim[1], nsx); brange(G.b[0], G.dim[1], nsx); }
      g.add(p_tn(a.x, true, D, 1, dg, G.dim[1], D, G.dim[1], B, sl + G.w[0], sl + G.b[0], nsx, PP, nullptr));
      const int nsy = small_ns((long long)K * G.dim[1]);
      if (nsy != NS) brange(G.w[0] + (uint64_t)D * G.dim[1], (uint64_t)K * G.dim[1], nsy);
      g.add(p_tn(w.y, false, K, 1, dcur, G.dim[1], K, G.dim[1], R, sl + G.w[0] + (uint64_t)D * G.dim[1], nullptr, nsy,
                 PP, nullptr));
      Problem p = p_nt(dcur, G.dim[1], Wy, G.dim[1], R, K, G.dim[1], w.dy, K, nullptr, 0);
      p.nseg = 2;                                   // dy = d_g0 Wg0[D:,:]^T + d_prior Wp^T
      p.seg[1].a = opnd(w.dpp, 2 * Lz, R, false, true);
      p.seg[1].b = opnd(P + L.prior.w[0], 2 * Lz, K, false, true);
      p.seg[1].K = 2 * Lz;
      p.seg[1].kscale = nullptr;
      // (dy = dhg Wg0[D:,:]^T + dpp Wp^T, 512 + 128 -> 64, with both weights stationary in one launch of its own: rows_ws_k8's
      //  second segment)
      const bool rws_dy = rws_on && G.dim[1] == 512 && K == 64 && 2 * Lz == 128;      // (interleaved on one box: 1219.4 -> 1214.1 us per step)
      if (!rws_dy) g.add(p);
      if (!prior_done) { g.add(prior_dw()); prior_done = true; }
      launch_group(cx, g, "bwd_enc_gmm_l0");
      if (rws_dy) {
        RwsArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.A = dcur; ra.lda = G.dim[1]; ra.R = R; ra.np = 1;
        ra.A2 = w.dpp; ra.lda2 = 2 * Lz;
        ra.p[0] = rws_prob(Wy, G.dim[1], nullptr, w.dy, K, false);
        ra.p[0].W2 = P + L.prior.w[0]; ra.p[0].ldw2 = 2 * Lz;
        launch_rws(ra, 2, "bwd_dy", G.dim[1] + 2 * Lz);
      }
    }
    hipLaunchKernelGGL(y_head_bwd, dim3(grid_for(B, 1)), dim3(512), 0, st, w.logits, w.y, w.dy, w.nent, w.dlogits, B,
                       S, K, 1.f / d.temperature);
    rowk(cx, "y_head_bwd");
    dcur = w.dlogits;
  } else {
    if (model == GMVAE_MODEL_VAE_GMP) {
      hipLaunchKernelGGL(gmp_param_bwd, dim3(GMP_PARTS), dim3(256), 0, st, w.z, w.resp, rwS, P + L.loc,
                         P + L.rawscale, P + L.mixlog, w.gmp_part, R, Lz, K, (int)pad4((uint64_t)K * Lz));
      rowk(cx, "gmp_param_bwd");
    }
    if (S > 1) {
      hipLaunchKernelGGL(sum_over_s, dim3(grid_for((long long)B * 2 * Lz, 256, 1 << 22)), dim3(256), 0, st, w.dqp,
                         w.dqb, B, S, 2 * Lz);
      rowk(cx, "sum_over_s");
      dcur = w.dqb;
    } else {
      dcur = w.dqp;
    }
  }
  for (int i = E.nl - 1; i >= 0; --i) {   // enc_y (GMVAE) / encoder (VAE): rows = B
    Group g;
    const void* act = (i == 0) ? (const void*)a.x : (const void*)w.he[i];
    const int nse = (S > 1 && sxb.n + 2 <= kSlabRanges) ? NSB : NS;
    if (nse != NS) { brange(E.w[i], (uint64_t)E.dim[i] * E.dim[i + 1], nse); brange(E.b[i], E.dim[i + 1], nse); }
    g.add(p_tn(act, i == 0, E.dim[i], 1, dcur, E.dim[i + 1], E.dim[i], E.dim[i + 1], B, sl + E.w[i], sl + E.b[i], nse,
               PP, nullptr));
    float* out = nullptr;
    if (i > 0) {
      out = w.dbuf[pb];
      g.add(p_nt(dcur, E.dim[i + 1], P + E.w[i], E.dim[i + 1], B, E.dim[i], E.dim[i + 1], out, E.dim[i], w.he[i],
                 E.dim[i]));
    }
    launch_group(cx, g, i == 0 ? "bwd_enc_l0" : "bwd_enc");
    dcur = out;
    pb ^= 1;
  }
  {
    const bool gmp = model == GMVAE_MODEL_VAE_GMP;
    const int KLp = (int)pad4((uint64_t)K * Lz);
    AdamTail adt;
    memset(&adt, 0, sizeof(adt));
    if (a.adam_p && a.step_dev) {                 // train graph / gmvae_train_step: the optimizer in the same launch
      adt.p = a.adam_p; adt.m = a.adam_m; adt.v = a.adam_v; adt.lr = a.lr; adt.b1 = a.beta1; adt.b2 = a.beta2; adt.eps = a.epsilon;
      adt.t_dev = a.step_dev; adt.tail_log = a.tail_log;
    }
    hipLaunchKernelGGL(finalize_grads, dim3((unsigned)((PP / 4 + 255) / 256)), dim3(256), 0, st, sl, NS, PP, a.grads,
                       gmp ? w.gmp_part : (const float*)nullptr, GMP_PARTS, gmp ? 2 * KLp + (int)pad4(K) : 0,
                       (long long)L.loc, sxb, adt);
    rowk(cx, adt.p ? "finalize_grads_adam" : "finalize_grads");
  }
  return cx.err;
}

static int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// ================================ C ABI ====================================
extern "C" {

int gmvae_abi_version(void) { return GMVAE_ABI_VERSION; }

int gmvae_param_count(const GmvaeDims* dims, int model, uint64_t* P_padded, uint64_t* P_real) {
  if (int e = check_dims(dims, model)) return e;
  Layout L;
  build_layout(*dims, model, L);
  if (P_padded) *P_padded = L.P_pad;
  if (P_real) *P_real = L.P_real;
  return 0;
}

int gmvae_param_layout(const GmvaeDims* dims, int model, GmvaeParamEntry* out, int max_entries, int* n_entries) {
  if (int e = check_dims(dims, model)) return e;
  if (!n_entries) return GMVAE_E_NULL;
  Layout L;
  build_layout(*dims, model, L);
  *n_entries = L.n;
  if (out) {
    if (max_entries < L.n) return GMVAE_E_SMALL;
    memcpy(out, L.e, sizeof(GmvaeParamEntry) * L.n);
  }
  return 0;
}

int gmvae_workspace_bytes(const GmvaeDims* dims, int model, uint64_t* bytes) {
  if (int e = check_dims(dims, model)) return e;
  if (!bytes) return GMVAE_E_NULL;
  Layout L;
  build_layout(*dims, model, L);
  WS w;
  carve(*dims, model, L, nullptr, w);
  *bytes = w.bytes;
  return 0;
}

int gmvae_step(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
               const float* params, float* grads, void* workspace, uint64_t seed, uint64_t step, uint64_t* step_dev,
               void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !grads || !workspace) return GMVAE_E_NULL;
  if (!aligned16(params) || !aligned16(grads) || !aligned16(workspace) || (eps && !aligned16(eps)) ||
      (u && !aligned16(u)))
    return GMVAE_E_ALIGN;
  Ctx cx;
  cx.st = static_cast<hipStream_t>(stream);
  const char* fc = getenv("GMVAE_FORCE_CFG");
  if (fc) cx.force_cfg = atoi(fc);
  StepArgs a = {dims, model, x, eps, u, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, seed, step,
                step_dev, true};
  return run_step(cx, a);
}

/* internal: gmvae_step with the end-of-step Adam fused in (single device; used by the train graph) */
static int step_with_adam(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                          float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float b1,
                          float b2, float eps_, hipStream_t st, bool imgs_ready = false, float* tail_log = nullptr) {
  Ctx cx;
  cx.st = st;
  StepArgs a = {dims, model, x, nullptr, nullptr, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                seed, 0, step_dev, true};
  a.adam_p = params; a.adam_m = m; a.adam_v = v; a.lr = lr; a.beta1 = b1; a.beta2 = b2; a.epsilon = eps_;
  a.imgs_ready = imgs_ready;
  a.tail_log = tail_log;
  return run_step(cx, a);
}

int gmvae_forward(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
                  const float* params, float* tail, float* row_terms, float* z_out, float* y_out, float* logits_out,
                  void* workspace, uint64_t seed, uint64_t step, void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !tail || !workspace) return GMVAE_E_NULL;
  if (!aligned16(params) || !aligned16(workspace) || (eps && !aligned16(eps)) || (u && !aligned16(u)))
    return GMVAE_E_ALIGN;
  Ctx cx;
  cx.st = static_cast<hipStream_t>(stream);
  StepArgs a = {dims, model, x, eps, u, params, nullptr, tail, row_terms, z_out, y_out, logits_out, workspace, seed,
                step, nullptr, false};
  return run_step(cx, a);
}

int adam_tf_step(float* params, float* m, float* v, const float* grads, uint64_t P, float lr, float beta1,
                 float beta2, float epsilon, uint64_t t, const uint64_t* t_dev, float grad_scale,
                 const float* grad_scale_dev, const float* loss_sum_dev, void* stream) {
  if (!params || !m || !v || !grads) return GMVAE_E_NULL;
  if (P == 0) return GMVAE_E_DIMS;
  if (!aligned16(params) || !aligned16(m) || !aligned16(v) || !aligned16(grads)) return GMVAE_E_ALIGN;
  (void)hipGetLastError();
  hipLaunchKernelGGL(adam_tf, dim3((unsigned)(((P + 3) / 4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), params, m, v, grads, (long long)P, lr, beta1, beta2, epsilon, t,
                     t_dev, grad_scale, grad_scale_dev, loss_sum_dev);
  return (int)hipGetLastError();
}

int gmvae_noise_fill(float* eps, float* u, uint64_t rows, int L, int K, uint64_t row_base, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, void* stream) {
  if ((eps && L < 1) || (u && K < 1)) return GMVAE_E_DIMS;
  if ((eps && !aligned16(eps)) || (u && !aligned16(u))) return GMVAE_E_ALIGN;
  const uint64_t q = noise_items(eps, u, rows, L, K);
  if (q == 0) return 0;
  (void)hipGetLastError();
  hipLaunchKernelGGL(noise_fill, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     eps, u, rows, L, K, row_base, seed, step, step_dev);
  return (int)hipGetLastError();
}

int gmvae_binarize(const uint8_t* pixels, uint64_t n_rows, const int32_t* idx, uint64_t row0, int B, int D,
                   uint64_t seed, uint64_t step, const uint64_t* step_dev, uint8_t* x_out, uint64_t out_row0,
                   void* stream) {
  if (!pixels || !x_out) return GMVAE_E_NULL;
  if (B < 1 || D < 4 || (D & 3) || n_rows < 1 || (!idx && row0 + (uint64_t)B > n_rows)) return GMVAE_E_DIMS;
  if ((reinterpret_cast<uintptr_t>(pixels) & 3) || (reinterpret_cast<uintptr_t>(x_out) & 3)) return GMVAE_E_ALIGN;
  (void)hipGetLastError();
  const uint64_t q = (uint64_t)B * (D >> 2);
  hipLaunchKernelGGL(binarize_rows, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pixels,
                     idx, row0, n_rows, B, D, seed, step, step_dev, x_out, out_row0);
  return (int)hipGetLastError();
}

int gmvae_mlp_forward(const GmvaeDims* dims, int model, int net, const void* in, int in_is_u8, const float* in2,
                      int rows, const float* params, float* out, void* workspace, void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!in || !params || !out || !workspace) return GMVAE_E_NULL;
  if (rows < 1) return GMVAE_E_DIMS;
  GmvaeDims d = *dims;
  d.B = rows;
  d.S = 1;
  Layout L;
  build_layout(d, model, L);
  WS w;
  carve(d, model, L, workspace, w);
  const NetL* N = nullptr;
  const bool gm = model == GMVAE_MODEL_GMVAE;
  switch (net) {
    case GMVAE_NET_ENCODER_Y: N = gm ? &L.ency : nullptr; break;
    case GMVAE_NET_PRIOR_GMM: N = gm ? &L.prior : nullptr; break;
    case GMVAE_NET_ENCODER_GMM: N = gm ? &L.encg : nullptr; break;
    case GMVAE_NET_DECODER: N = &L.dec; break;
    case GMVAE_NET_ENCODER: N = gm ? nullptr : &L.enc; break;
    default: return GMVAE_E_NET;
  }
  if (!N) return GMVAE_E_NET;
  if (net == GMVAE_NET_ENCODER_GMM && !in2) return GMVAE_E_NULL;
  Ctx cx;
  cx.st = static_cast<hipStream_t>(stream);
  tl_hact = 1 + d.hidden_act;
  float** hbuf = (net == GMVAE_NET_DECODER) ? w.hd : (net == GMVAE_NET_ENCODER_GMM ? w.hg : w.he);
  const float* P = params;
  for (int i = 0; i < N->nl; ++i) {
    const bool last = i == N->nl - 1;
    float* o = last ? out : hbuf[i + 1];
    if (i == 0 && net == GMVAE_NET_ENCODER_GMM) {
      const int D = d.D, K = d.K, H = N->dim[1];
      Group g0;
      g0.add(p_nn(in, in_is_u8 != 0, D, P + N->w[0], H, rows, H, D, w.gx, H, nullptr, false));
      launch_group(cx, g0, "mlp_x");
      Group g1;
      Problem p = p_nn(in2, false, K, P + N->w[0] + (uint64_t)D * H, H, rows, H, K, o, H, P + N->b[0], !last);
      p.addsrc = w.gx; p.ld_add = H; p.add_div = 1;
      g1.add(p);
      launch_group(cx, g1, "mlp_y");
    } else {
      Group g;
      const void* a = (i == 0) ? in : (const void*)hbuf[i];
      Problem p = p_nn(a, i == 0 && in_is_u8, N->dim[i], P + N->w[i], N->dim[i + 1], rows, N->dim[i + 1], N->dim[i],
                       o, N->dim[i + 1], P + N->b[i], !last);
      if (last && net == GMVAE_NET_DECODER) { p.addconst = d.gen_bias_init; p.bias2 = d.gen_bias_vec; }
      g.add(p);
      launch_group(cx, g, "mlp");
    }
  }
  return cx.err;
}

int gmvae_cluster_acc(const float* logits, const int64_t* labels, int B, int K, int n_labels, int32_t* scratch,
                      float* acc_out, void* stream) {
  if (!logits || !labels || !scratch || !acc_out) return GMVAE_E_NULL;
  if (B < 1 || K < 1 || K > 256 || n_labels < 1) return GMVAE_E_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t* hist = scratch;
  int32_t* pred = scratch + (size_t)K * n_labels;
  (void)hipGetLastError();
  hipError_t e = hipMemsetAsync(hist, 0, sizeof(int32_t) * K * n_labels, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(cluster_hist, dim3((B + 255) / 256), dim3(256), 0, st, logits, labels, B, K, n_labels, hist, pred);
  hipLaunchKernelGGL(cluster_match, dim3(1), dim3(1024), 0, st, hist, pred, labels, B, K, n_labels, acc_out);
  return (int)hipGetLastError();
}

int gmvae_gemm_test(const void* A, int a_is_u8, const float* W, const float* bias, float* C, int M, int N, int K,
                    int trans, int relu, int cfg, int splitk, void* stream) {
  if (!A || !W || !C) return GMVAE_E_NULL;
  if (M < 1 || N < 1 || K < 1 || splitk < 1 || splitk > NS_MAX) return GMVAE_E_DIMS;
  Ctx cx;
  cx.st = static_cast<hipStream_t>(stream);
  Group g;
  if (cfg == 6 || cfg == 7) {
    // the f16-pair instance: cfg 6 scales and splits both operands first, cfg 7 reuses the pairs of the previous cfg-6 call
    static unsigned short *pa = nullptr, *pb = nullptr;
    static float* sc = nullptr;                     // [0], [1] amax bits; [2], [3] 1 / scale; [16..], [16 + kAmaxBlocks..] partial maxima
    static size_t cap_a = 0, cap_b = 0;
    if (a_is_u8) return GMVAE_E_DIMS;
    const size_t na = (size_t)M * K, nb = (size_t)N * K;
    if (na % 8 || nb % 8) return GMVAE_E_DIMS;
    if (!sc && hipMalloc(&sc, (16 + 2 * kAmaxBlocks) * 4) != hipSuccess) return GMVAE_E_ALIGN;
    if (na > cap_a) { if (pa) hipFree(pa); if (hipMalloc(&pa, 2 * na * 2) != hipSuccess) return GMVAE_E_ALIGN; cap_a = na; }
    if (nb > cap_b) { if (pb) hipFree(pb); if (hipMalloc(&pb, 2 * nb * 2) != hipSuccess) return GMVAE_E_ALIGN; cap_b = nb; }
    if (cfg == 6) {
      unsigned* const u = reinterpret_cast<unsigned*>(sc);
      launch_amax(cx.st, static_cast<const float*>(A), (long long)na, u + 16);
      launch_amax(cx.st, W, (long long)nb, u + 16 + kAmaxBlocks);
      launch_amax_final(cx.st, u + 16, kAmaxBlocks, u + 16 + kAmaxBlocks, kAmaxBlocks, u, sc + 2);
      launch_split_pairs(cx.st, static_cast<const float*>(A), nullptr, trans == 2 ? M : K, (long long)na, pa, u);
      launch_split_pairs(cx.st, W, nullptr, trans == 1 ? K : N, (long long)nb, pb, u + 1);
    }
    Problem p;
    const float* fa = reinterpret_cast<const float*>(pa);
    const float* fb = reinterpret_cast<const float*>(pb);
    tl_hact = 1;
    if (trans == 0) p = p_nn(fa, false, K, fb, N, M, N, K, C, N, bias, relu != 0);
    else if (trans == 1) p = p_nt(fa, K, fb, K, M, N, K, C, N, nullptr, 0);
    else p = p_tn(fa, false, M, 1, fb, N, M, N, K, C, bias ? C + (size_t)M * N : nullptr, splitk, (long long)(M + 1) * N, nullptr);
    p.planes = 2;
    p.a_pstride = (long long)na; p.b_pstride = (long long)nb;
    p.a_uns = sc + 2; p.b_uns = sc + 3; p.uns_c = 1.f; p.uns_cb = 1.f;
    g.add(p);
    launch_group(cx, g, "gemm_test_pairs", 2);
    return cx.err;
  }
  if (cfg == 4 || cfg == 5) {
    // the pre-split ("planes") instance: cfg 4 splits both operands into scratch planes first, cfg 5 reuses the planes of
    // the previous cfg-4 call (timing the GEMM alone).  M, N multiples of 128, K of 32 (TN: rows K of 32 x splitk).
    static unsigned short *pa = nullptr, *pb = nullptr;
    static size_t cap_a = 0, cap_b = 0;
    if (a_is_u8) return GMVAE_E_DIMS;
    const size_t na = (size_t)M * K, nb = (size_t)N * K;
    if (na % 8 || nb % 8) return GMVAE_E_DIMS;
    if (na > cap_a) { if (pa) hipFree(pa); if (hipMalloc(&pa, 3 * na * 2) != hipSuccess) return GMVAE_E_ALIGN; cap_a = na; }
    if (nb > cap_b) { if (pb) hipFree(pb); if (hipMalloc(&pb, 3 * nb * 2) != hipSuccess) return GMVAE_E_ALIGN; cap_b = nb; }
      if (cfg == 4) {
      launch_split(cx.st, static_cast<const float*>(A), nullptr, trans == 2 ? M : K, (long long)na, pa);
      launch_split(cx.st, W, nullptr, trans == 1 ? K : N, (long long)nb, pb);
    }
    Problem p;
    const float* fa = reinterpret_cast<const float*>(pa);
    const float* fb = reinterpret_cast<const float*>(pb);
    tl_hact = 1;
    if (trans == 0) p = p_nn(fa, false, K, fb, N, M, N, K, C, N, bias, relu != 0);
    else if (trans == 1) p = p_nt(fa, K, fb, K, M, N, K, C, N, nullptr, 0);
    else p = p_tn(fa, false, M, 1, fb, N, M, N, K, C, bias ? C + (size_t)M * N : nullptr, splitk, (long long)(M + 1) * N, nullptr);
    p.planes = 1;
    p.a_pstride = (long long)na; p.b_pstride = (long long)nb;
    g.add(p);
    launch_group(cx, g, "gemm_test_planes", 2);
    return cx.err;
  }
  if (cfg == 8) {
    // the weight-stationary row kernels (rowsws.hpp): NN K = 64 (N % 64 = 0); NT K = 128 (N % 32 = 0; `bias` = a ReLU mask [M][N]
    // or NULL) and K = 512 or 512 + 128, N = 64 (`bias` = an addend [M][N] or NULL)
    if (a_is_u8 || trans > 1) return GMVAE_E_DIMS;
    RwsArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.A = static_cast<const float*>(A); ra.lda = K; ra.R = M; ra.np = 1;
    RwsProb& q = ra.p[0];
    q.W = W; q.out = C; q.N = N; q.add_div = 1;
    const int cu = device_cus(), grid = cu * kSkWaves > kRwsMaxWaves ? kRwsMaxWaves / kSkWaves : cu;
    if (trans == 0 && K == 64 && N % 64 == 0 && N / 64 <= grid * kSkWaves) {
      q.ldw = N; q.bias = bias; q.relu = relu != 0;
      ra.ns0 = ra.ns = N / 64;
      hipLaunchKernelGGL((rows_ws<2, 4, false>), dim3(grid), dim3(kSkThreads), 0, cx.st, ra);
    } else if (trans == 1 && K == 128 && N % 32 == 0 && N / 32 <= grid * kSkWaves) {
      q.ldw = K; q.mask = bias; q.ld_mask = N;
      ra.ns0 = ra.ns = N / 32;
      hipLaunchKernelGGL((rows_ws<4, 2, true>), dim3(grid), dim3(kSkThreads), 0, cx.st, ra);
    } else if (trans == 1 && K == 512 && N == 64) {
      q.ldw = K; q.addsrc = bias; q.ld_add = N;
      ra.ns0 = ra.ns = 1;
      hipLaunchKernelGGL(rows_ws_k8, dim3(grid), dim3(kSkThreads), 0, cx.st, ra);
    } else if (trans == 1 && K == 640 && N == 64) {      // the two-segment form: columns 0..511 and 512..639 of A [M][640] against W [N][640]
      ra.lda = 640; q.ldw = 640; q.addsrc = bias; q.ld_add = N;
      ra.A2 = static_cast<const float*>(A) + 512; ra.lda2 = 640;
      q.W2 = W + 512; q.ldw2 = 640;
      ra.ns0 = ra.ns = 1;
      hipLaunchKernelGGL(rows_ws_k8, dim3(grid), dim3(kSkThreads), 0, cx.st, ra);
    } else {
      return GMVAE_E_DIMS;
    }
    cx.check();
    return cx.err;
  }
  if (trans == 0) {
    tl_hact = 1;
    g.add(p_nn(A, a_is_u8 != 0, K, W, N, M, N, K, C, N, bias, relu != 0));
  } else if (trans == 1) {
    g.add(p_nt(static_cast<const float*>(A), K, W, K, M, N, K, C, N, nullptr, 0));
  } else {  // TN: A given as [K rows, M], W as dY [K rows, N]; output [splitk][M+1][N] when bias (any non-null) requested
    Problem p = p_tn(A, a_is_u8 != 0, M, 1, W, N, M, N, K, C, bias ? C + (size_t)M * N : nullptr, splitk,
                     (long long)(M + 1) * N, nullptr);
    g.add(p);
  }
  launch_group(cx, g, "gemm_test", cfg);
  return cx.err;
}

int gmvae_step_profile(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
                       const float* params, float* grads, void* workspace, uint64_t seed, int iters, int max_levels,
                       int* n_levels, char* names, float* usec, double* flops, void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !grads || !workspace || !n_levels || !names || !usec || !flops) return GMVAE_E_NULL;
  if (iters < 1) return GMVAE_E_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Prof* pr = new Prof();
  for (int i = 0; i <= MAX_LEVELS; ++i) hipEventCreate(&pr->ev[i]);
  double acc[MAX_LEVELS] = {0};
  int rc = 0;
  for (int it = 0; it < iters && rc == 0; ++it) {
    Ctx cx;
    cx.st = st;
    cx.prof = pr;
    pr->n = 0;
    pr->active = true;
    hipEventRecord(pr->ev[0], st);
    StepArgs a = {dims, model, x, eps, u, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, seed,
                  (uint64_t)it, nullptr, true};
    rc = run_step(cx, a);
    hipStreamSynchronize(st);
    for (int i = 0; i < pr->n; ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, pr->ev[i], pr->ev[i + 1]);
      acc[i] += ms * 1000.0;
    }
  }
  const int n = pr->n < max_levels ? pr->n : max_levels;
  *n_levels = n;
  for (int i = 0; i < n; ++i) {
    memcpy(names + (size_t)i * 48, pr->name[i], 48);
    usec[i] = (float)(acc[i] / iters);
    flops[i] = pr->flops[i];
  }
  for (int i = 0; i <= MAX_LEVELS; ++i) hipEventDestroy(pr->ev[i]);
  delete pr;
  return rc;
}

/* measurement hook (bench.py --config eval_iwae): the FORWARD-ONLY evaluation (gmvae_forward with in-kernel Philox noise: the
 * -log p(x) importance-weighted bound of scripts/runners.py:324-333 at dims->S samples).  Pass 1: `iters` eager forwards with
 * hipEvents around every launch -> names, mean microseconds and algorithmic FLOPs per launch.  Pass 2: ONE forward captured
 * into a hipGraph and replayed `iters` times back to back -> *usec_total = mean microseconds per forward (events around the
 * whole train of replays). */
int gmvae_forward_profile(const GmvaeDims* dims, int model, const uint8_t* x, const float* params, float* tail, void* workspace,
                          uint64_t seed, int iters, int max_levels, int* n_levels, char* names, float* usec, double* flops,
                          float* usec_total, void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !tail || !workspace || !n_levels || !names || !usec || !flops || !usec_total) return GMVAE_E_NULL;
  if (iters < 1) return GMVAE_E_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Prof* pr = new Prof();
  for (int i = 0; i <= MAX_LEVELS; ++i) hipEventCreate(&pr->ev[i]);
  double acc[MAX_LEVELS] = {0};
  int rc = 0;
  for (int it = 0; it < iters && rc == 0; ++it) {
    Ctx cx;
    cx.st = st;
    cx.prof = pr;
    pr->n = 0;
    pr->active = true;
    hipEventRecord(pr->ev[0], st);
    StepArgs a = {dims, model, x, nullptr, nullptr, params, nullptr, tail, nullptr, nullptr, nullptr, nullptr, workspace, seed,
                  (uint64_t)it, nullptr, false};
    rc = run_step(cx, a);
    hipStreamSynchronize(st);
    for (int i = 0; i < pr->n; ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, pr->ev[i], pr->ev[i + 1]);
      acc[i] += ms * 1000.0;
    }
  }
  const int n = pr->n < max_levels ? pr->n : max_levels;
  *n_levels = n;
  for (int i = 0; i < n; ++i) {
    memcpy(names + (size_t)i * 48, pr->name[i], 48);
    usec[i] = (float)(acc[i] / iters);
    flops[i] = pr->flops[i];
  }
  *usec_total = 0.f;
  if (rc == 0) {                                 // pass 2: the same launches back to back
    hipStream_t cs = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess &&
              hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      Ctx cx;
      cx.st = cs;
      StepArgs a = {dims, model, x, nullptr, nullptr, params, nullptr, tail, nullptr, nullptr, nullptr, nullptr, workspace, seed,
                    0, nullptr, false};
      // G consecutive passes per graph launch (an evaluation walks a split batch by batch, scripts/runners.py:320-333: the ~8 us
      // the device idles between two graph launches is paid once per G batches, as in the train graphs); GMVAE_EVAL_GRAPH_PASSES=1:
      // one pass per launch (rounds 4-5's figure)
      int G = 8;
      if (const char* e = getenv("GMVAE_EVAL_GRAPH_PASSES")) G = atoi(e) >= 1 && atoi(e) <= 64 ? atoi(e) : G;
      if (G > iters) G = iters;
      int r2 = 0;
      GmvaeDims d2 = *dims;
      d2.sched_flags |= GMVAE_SCHED_EVAL_IMAGES_VALID;      // passes 2..G of a launch: the parameters cannot have changed in between
      StepArgs a2 = a;
      a2.d = &d2;
      for (int g = 0; g < G && r2 == 0; ++g) r2 = run_step(cx, g == 0 ? a : a2);
      ok = hipStreamEndCapture(cs, &graph) == hipSuccess && r2 == 0 && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (ok) {
        const int nl = (iters + G - 1) / G;
        for (int w_ = 0; w_ < 3; ++w_) hipGraphLaunch(exec, st);
        hipEventRecord(pr->ev[0], st);
        for (int it = 0; it < nl; ++it) hipGraphLaunch(exec, st);
        hipEventRecord(pr->ev[1], st);
        hipStreamSynchronize(st);
        float ms = 0.f;
        hipEventElapsedTime(&ms, pr->ev[0], pr->ev[1]);
        *usec_total = ms * 1000.f / (float)(nl * G);
      }
    }
    if (exec) hipGraphExecDestroy(exec);
    if (graph) hipGraphDestroy(graph);
    if (cs) hipStreamDestroy(cs);
    (void)hipGetLastError();
  }
  for (int i = 0; i <= MAX_LEVELS; ++i) hipEventDestroy(pr->ev[i]);
  delete pr;
  return rc;
}

int gmvae_train_profile(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                        float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, int iters,
                        int max_levels, int* n_levels, char* names, float* usec, float* usec_timeline, double* flops,
                        void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !m || !v || !grads || !workspace || !step_dev || !n_levels || !names || !usec || !flops)
    return GMVAE_E_NULL;
  if (iters < 1) return GMVAE_E_DIMS;
  if (!mega_ok(*dims, model)) return GMVAE_E_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Prof* pr = new Prof();
  double acc[MAX_LEVELS] = {0}, acc_tl[MAX_LEVELS] = {0};
  int rc = 0;
  // Three steps in ONE captured graph: an untimed one that leaves the weight images behind, then two steady-state
  // steps whose launches record per-workgroup wall-clock stamps (s_memrealtime, 100 MHz, one clock for the whole
  // device) into two slots.  Per launch of the FIRST stamped step two durations come out:
  //   usec          = last workgroup end - first workgroup start (the in-kernel span);
  //   usec_timeline = first workgroup start of the NEXT launch (the following step's first launch for the last one)
  //                   - its own first workgroup start: the launch's share of the step's timeline, dispatch and
  //                   end-of-kernel write-back included -- the interval rocprofv3 reports, and the shares add up to the step.
  // Replayed `iters` times, so the kernels run back to back as in the train graph.  (hipEventRecord nodes inside a
  // captured graph return no elapsed time on this stack, and eager launches with events in between add ~10 us of idle
  // per launch.)
  pr->events = false;
  Layout L;
  build_layout(*dims, model, L);
  WS w;
  carve(*dims, model, L, workspace, w);
  hipStream_t cs = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  auto steps3 = [&](hipStream_t s) {
    for (int it = 0; it < 3 && rc == 0; ++it) {
      Ctx cx;
      cx.st = s;
      cx.prof = pr;
      if (it < 2) pr->n = 0;
      pr->active = it == 1;
      StepArgs a = {dims, model, x, nullptr, nullptr, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                    seed, 0, step_dev, true};
      a.adam_p = params; a.adam_m = m; a.adam_v = v; a.lr = lr; a.imgs_ready = it > 0; a.want_spans = it > 0;
      a.span_slot = it == 2 ? 1 : 0;
      // (diagnostic: every stamped step starts with cold instruction caches -- kernels.hpp icache_flush, tools/icache_cold.py)
      if (it > 0 && getenv("GMVAE_ICACHE_FLUSH")) hipLaunchKernelGGL(icache_flush, dim3(2 * device_cus()), dim3(64), 0, s, w.spans ? reinterpret_cast<float*>(w.gstamps) : nullptr);
      rc = run_step(cx, a);
    }
  };
  bool graphed = false;
  if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess) {
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      steps3(cs);
      const hipError_t he = hipStreamEndCapture(cs, &graph);
      if (rc == 0 && he == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) graphed = true;
    }
    (void)hipGetLastError();
  }
  if (!graphed) rc = 0;
  const size_t nsp = 2 * 3 * 2048 * 2, ngs = 2048 * 8, slot = 3 * 2048 * 2;
  unsigned long long* hsp = new unsigned long long[nsp + ngs];
  for (int it = 0; it < iters && rc == 0; ++it) {
    hipMemsetAsync(w.spans, 0, nsp * 8, st);
    hipMemsetAsync(w.gstamps + 2048 * 8, 0, ngs * 8, st);
    if (graphed) {
      if (hipGraphLaunch(exec, st) != hipSuccess) { rc = (int)hipGetLastError(); break; }
    } else {
      steps3(st);
    }
    hipStreamSynchronize(st);
    hipMemcpy(hsp, w.spans, nsp * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hsp + nsp, w.gstamps + 2048 * 8, ngs * 8, hipMemcpyDeviceToHost);
    auto first_last = [&](const unsigned long long* p, int stride, int e_off, unsigned long long& lo, unsigned long long& hi) {
      lo = ~0ull; hi = 0;
      for (int b = 0; b < 2048; ++b) {
        const unsigned long long s0 = p[(size_t)b * stride], s1 = p[(size_t)b * stride + e_off];
        if (!s0 || !s1) continue;
        lo = s0 < lo ? s0 : lo;
        hi = s1 > hi ? s1 : hi;
      }
    };
    unsigned long long start[MAX_LEVELS + 1] = {0};
    for (int i = 0; i < pr->n; ++i) {
      unsigned long long lo = ~0ull, hi = 0;
      if (!strncmp(pr->name[i], "mega", 4)) first_last(hsp, 2, 1, lo, hi);
      else if (!strncmp(pr->name[i], "bwd_dw_all", 10)) first_last(hsp + nsp, 8, 4, lo, hi);
      else if (!strncmp(pr->name[i], "finalize_adam", 13) || !strncmp(pr->name[i], "dw_adam", 7)) first_last(hsp + 2048 * 2, 2, 1, lo, hi);
      if (hi > lo) { acc[i] += (double)(hi - lo) * 0.01; start[i] = lo; }       // 100 MHz ticks -> microseconds
    }
    {  // the next step's first stamped launch closes the last launch's share (steady state: the same launch sequence)
      unsigned long long lo = ~0ull, hi = 0;
      if (pr->n > 0 && !strncmp(pr->name[0], "mega", 4)) first_last(hsp + slot, 2, 1, lo, hi);
      start[pr->n] = hi > lo ? lo : 0;
    }
    for (int i = 0; i < pr->n; ++i)
      if (start[i] && start[i + 1] > start[i]) acc_tl[i] += (double)(start[i + 1] - start[i]) * 0.01;
  }
  delete[] hsp;
  if (exec) hipGraphExecDestroy(exec);
  if (graph) hipGraphDestroy(graph);
  if (cs) hipStreamDestroy(cs);
  const int n = pr->n < max_levels ? pr->n : max_levels;
  *n_levels = n;
  for (int i = 0; i < n; ++i) {
    memcpy(names + (size_t)i * 48, pr->name[i], 48);
    usec[i] = (float)(acc[i] / iters);
    if (usec_timeline) usec_timeline[i] = (float)(acc_tl[i] / iters);
    flops[i] = pr->flops[i];
  }
  delete pr;
  return rc;
}

/* measurement hook: `iters` full training steps (gmvae_step + adam_tf_step) issued from C on `stream`,
 * timed with hipEvents.  mode 0: eager launches; mode 1: one hipGraph (captured once here) replayed. */
int gmvae_bench_loop(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                     float* grads, void* workspace, uint64_t* step_dev, int iters, int mode, float* usec_per_step,
                     void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !m || !v || !grads || !workspace || !step_dev || !usec_per_step) return GMVAE_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Layout L;
  build_layout(*dims, model, L);
  auto one = [&](hipStream_t s) -> int {
    int rc = gmvae_step(dims, model, x, nullptr, nullptr, params, grads, workspace, 1234, 0, step_dev, s);
    if (rc) return rc;
    return adam_tf_step(params, m, v, grads, L.P_pad, 1e-3f, 0.9f, 0.999f, 1e-8f, 0, step_dev, 1.f,
                        grads + L.P_pad + 4, grads + L.P_pad, s);
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int rc = one(st);
  hipStreamSynchronize(st);
  if (rc) return rc;
  if (mode == 0) {
    hipEventRecord(e0, st);
    for (int i = 0; i < iters && rc == 0; ++i) rc = one(st);
    hipEventRecord(e1, st);
  } else {
    hipStream_t cs;
    hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    rc = one(cs);
    hipError_t he = hipStreamEndCapture(cs, &g);
    if (rc == 0 && he != hipSuccess) rc = (int)he;
    if (rc == 0) {
      he = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      if (he != hipSuccess) rc = (int)he;
    }
    if (rc == 0) {
      hipGraphLaunch(ge, cs);
      hipStreamSynchronize(cs);
      hipEventRecord(e0, cs);
      for (int i = 0; i < iters; ++i) hipGraphLaunch(ge, cs);
      hipEventRecord(e1, cs);
      hipStreamSynchronize(cs);
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
    hipStreamDestroy(cs);
  }
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  *usec_per_step = ms * 1000.f / (float)iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

/* ---- hipGraph of one full training step (owned by the library: PyTorch's CUDAGraph.replay() costs
 * about twice the GPU time of this step on ROCm 7) ------------------------------------------------ */
struct GmvaeTrainGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

static int train_graph_create(const GmvaeDims* dims, int model, const uint8_t* pixels, uint64_t n_rows, const int32_t* idx,
                              uint8_t* x, int n_steps, float* params, float* m, float* v, float* grads, void* workspace,
                              uint64_t seed, uint64_t* step_dev, float lr, float beta1, float beta2, float epsilon,
                              float* tail_log, void** graph_out) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !m || !v || !grads || !workspace || !step_dev || !graph_out) return GMVAE_E_NULL;
  if (n_steps < 1 || n_steps > 1024) return GMVAE_E_DIMS;
  if (pixels && (!idx || n_rows < 1 || (dims->D & 3))) return GMVAE_E_DIMS;
  const size_t xstride = (size_t)dims->B * dims->D;
  Layout L;
  build_layout(*dims, model, L);
  hipStream_t cs;
  hipError_t he = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (he != hipSuccess) return (int)he;
  GmvaeTrainGraph* tg = new GmvaeTrainGraph();
  int rc = 0;
  he = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (he != hipSuccess) rc = (int)he;
  // the launch's batches: ONE kernel in front of the steps (kernels.hpp binarize_batches); the uniforms are keyed by
  // (seed, the consuming step's index, global quad), none of which depends on the training state.  (Round 4 removed the
  // per-step forms -- a launch per batch, or extra workgroups of the optimizer launch -- measured 3-4 us per step slower.)
  if (rc == 0 && pixels) {
    const uint64_t q = (uint64_t)n_steps * dims->B * (uint64_t)(dims->D >> 2);
    (void)hipGetLastError();
    hipLaunchKernelGGL(binarize_batches, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, cs, pixels, idx, n_rows, dims->B, dims->D,
                       n_steps, seed ^ 0x62696e6172697a65ull, step_dev, x, dims->row0);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) {
    // every schedule ends with a launch that applies TF-Adam and logs the tail; every step after the first finds its
    // weight images written by the step before it (same graph, nothing in between)
    for (int s = 0; s < n_steps && rc == 0; ++s)
      rc = step_with_adam(dims, model, x + s * xstride, params, m, v, grads, workspace, seed, step_dev, lr, beta1, beta2, epsilon, cs,
                          s > 0, tail_log ? tail_log + (size_t)s * GMVAE_TAIL : nullptr);
    he = hipStreamEndCapture(cs, &tg->graph);
    if (rc == 0 && he != hipSuccess) rc = (int)he;
  }
  if (rc == 0) {
    he = hipGraphInstantiate(&tg->exec, tg->graph, nullptr, nullptr, 0);
    if (he != hipSuccess) rc = (int)he;
  }
  hipStreamDestroy(cs);
  if (rc != 0) {
    if (tg->graph) hipGraphDestroy(tg->graph);
    delete tg;
    return rc;
  }
  *graph_out = tg;
  return 0;
}

int gmvae_train_graph_create(const GmvaeDims* dims, int model, const uint8_t* x, int n_steps, float* params, float* m,
                             float* v, float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr,
                             float beta1, float beta2, float epsilon, float* tail_log, void** graph_out) {
  return train_graph_create(dims, model, nullptr, 0, nullptr, const_cast<uint8_t*>(x), n_steps, params, m, v, grads, workspace,
                            seed, step_dev, lr, beta1, beta2, epsilon, tail_log, graph_out);
}

int gmvae_train_graph_create_pipeline(const GmvaeDims* dims, int model, const uint8_t* pixels, uint64_t n_rows,
                                      const int32_t* idx, uint8_t* x_scratch, int n_steps, float* params, float* m, float* v,
                                      float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr,
                                      float beta1, float beta2, float epsilon, float* tail_log, void** graph_out) {
  if (!pixels || !idx) return GMVAE_E_NULL;
  return train_graph_create(dims, model, pixels, n_rows, idx, x_scratch, n_steps, params, m, v, grads, workspace, seed, step_dev,
                            lr, beta1, beta2, epsilon, tail_log, graph_out);
}

int gmvae_train_graph_launch(void* graph, void* stream) {
  if (!graph) return GMVAE_E_NULL;
  return (int)hipGraphLaunch(static_cast<GmvaeTrainGraph*>(graph)->exec, static_cast<hipStream_t>(stream));
}

int gmvae_train_graph_destroy(void* graph) {
  if (!graph) return GMVAE_E_NULL;
  GmvaeTrainGraph* tg = static_cast<GmvaeTrainGraph*>(graph);
  hipGraphExecDestroy(tg->exec);
  hipGraphDestroy(tg->graph);
  delete tg;
  return 0;
}

/* debugging aid: copies the skinny schedule's stamp buffer ([10][256][8] uint64, GMVAE_SK_STAMPS=1) to the host */
int gmvae_debug_sk_stamps(unsigned long long* host_out) {
  if (!host_out) {                                 // arm: allocate the buffer (outside any stream capture); later captures stamp into it
    if (!g_sk_dbg) {
      if (hipMalloc(&g_sk_dbg, (size_t)10 * kSkDbgWgs * 8 * 8) != hipSuccess) return GMVAE_E_NULL;
      hipMemset(g_sk_dbg, 0, (size_t)10 * kSkDbgWgs * 8 * 8);
    }
    return 0;
  }
  if (!g_sk_dbg) return GMVAE_E_NULL;
  return (int)hipMemcpy(host_out, g_sk_dbg, (size_t)10 * kSkDbgWgs * 8 * 8, hipMemcpyDeviceToHost);
}

/* disarm: later steps and captures no longer stamp; frees the buffer.  The caller must first destroy every train graph
 * captured while the buffer was armed (their kernel arguments hold its address). */
int gmvae_debug_sk_stamps_free(void) {
  if (!g_sk_dbg) return 0;
  hipDeviceSynchronize();
  const hipError_t e = hipFree(g_sk_dbg);
  g_sk_dbg = nullptr;
  return (int)e;
}

/* debugging aid: resident workgroups per CU as the runtime computes them */
int gmvae_step_schedule(const GmvaeDims* dims, int model, char* out48) {
  if (int e = check_dims(dims, model)) return e;
  if (!out48) return GMVAE_E_NULL;
  const GmvaeDims& d = *dims;
  Layout L;
  build_layout(d, model, L);
  const char* nm = "general";
  if (mega_ok(d, model)) nm = mega2_ok(d, model) ? "mega2" : (mega2v_ok(d, model) ? "mega2v" : "mega");
  else if (skinny_ok(d, model)) nm = "skinny";
  else if (fused_ok(d, model)) nm = "fused";
  const bool gen = !strcmp(nm, "general");
  snprintf(out48, 48, "%s%s", nm, (gen && planes_ok(d, L)) ? "+planes" : "");
  return 0;
}

int gmvae_kernel_occupancy(int which, int* blocks_per_cu) {
  if (!blocks_per_cu) return GMVAE_E_NULL;
  hipError_t e;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, gemm_grouped<CfgS>, kThreads, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, gemm_grouped<CfgM>, kThreads, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, gemm_grouped<CfgL>, kThreads, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mega_fwd_bwd<0, 0, 0, 0, -1, 0>, kMT, 150 * 1024); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, finalize_adam, 256, 0); break;
    default: return GMVAE_E_DIMS;
  }
  return (int)e;
}

/* debugging aid: byte offset of a named workspace buffer (tests compare intermediates with the oracle) */
int gmvae_workspace_offset(const GmvaeDims* dims, int model, const char* name, uint64_t* byte_offset) {
  if (int e = check_dims(dims, model)) return e;
  if (!name || !byte_offset) return GMVAE_E_NULL;
  Layout L;
  build_layout(*dims, model, L);
  WS w;
  char base[16];
  carve(*dims, model, L, base, w);
  struct { const char* n; float* p; } tab[] = {
      {"hy1", w.he[1]}, {"hg1", w.hg[1]}, {"hd1", w.hd[1]}, {"gx", w.gx}, {"logits", w.logits}, {"y", w.y},
      {"nent", w.nent}, {"pp", w.pp}, {"qp", w.qp}, {"z", w.z}, {"logq", w.logq}, {"logp", w.logp},
      {"logpx", w.logpx}, {"logw", w.logw}, {"g", w.g}, {"dz", w.dz}, {"dqp", w.dqp}, {"dpp", w.dpp},
      {"dy", w.dy}, {"dlogits", w.dlogits}, {"dbuf0", w.dbuf[0]}, {"dbuf1", w.dbuf[1]}, {"dbuf2", w.dbuf[2]},
      {"slabs", w.slabs}, {"s1", w.s1}, {"s4", w.s4}, {"eps", w.eps}, {"u", w.u},
      {"stamps", reinterpret_cast<float*>(w.stamps)}, {"gstamps", reinterpret_cast<float*>(w.gstamps)},
      {"sync", reinterpret_cast<float*>(w.sync)}, {"ev_dbg", reinterpret_cast<float*>(w.ev_dbg)}};
  for (auto& t : tab)
    if (!strcmp(t.n, name)) {
      if (!t.p) return GMVAE_E_NET;
      *byte_offset = (uint64_t)(reinterpret_cast<char*>(t.p) - base);
      return 0;
    }
  if (name[0] == 'h' && (name[1] == 'e' || name[1] == 'g' || name[1] == 'd') && name[2] >= '1' && name[2] <= '9' && !name[3]) {
    const int i = name[2] - '0';                   // "he<i>" / "hg<i>" / "hd<i>": the kept input activation of layer i
    float* const* const tabh = name[1] == 'e' ? w.he : (name[1] == 'g' ? w.hg : w.hd);
    if (i > MAXH || !tabh[i]) return GMVAE_E_NET;
    *byte_offset = (uint64_t)(reinterpret_cast<char*>(tabh[i]) - base);
    return 0;
  }
  return GMVAE_E_NET;
}

/* ---- data parallel: RCCL is bound at run time (dlopen of the copy the host process already uses), so the
 * library itself has no link-time dependency on it.  One all-reduce(SUM) of the flat [P + TAIL] buffer per
 * step, on the caller's stream, between the gradient half and the optimiser half of the step. ---------- */
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ...) = nullptr;   // (ncclComm_t*, int nranks, ncclUniqueId by value, int rank)
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
};
static RcclApi g_rccl;
struct UniqueId128 { char b[128]; };

static int rccl_load(const char* path) {
  if (g_rccl.h) return 0;
  void* h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return GMVAE_E_NULL;
  g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, ...)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.CommCount = reinterpret_cast<int (*)(void*, int*)>(dlsym(h, "ncclCommCount"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) return GMVAE_E_NULL;
  g_rccl.h = h;
  return 0;
}

int gmvae_comm_unique_id(const char* rccl_path, char* out128) {
  if (!out128) return GMVAE_E_NULL;
  if (int e = rccl_load(rccl_path)) return e;
  return g_rccl.GetUniqueId(out128) ? 1000 : 0;
}

int gmvae_comm_init(const char* rccl_path, const char* id128, int rank, int world, void** comm) {
  if (!id128 || !comm) return GMVAE_E_NULL;
  if (world < 1 || rank < 0 || rank >= world) return GMVAE_E_DIMS;
  if (int e = rccl_load(rccl_path)) return e;
  UniqueId128 id;
  memcpy(id.b, id128, 128);
  typedef int (*InitFn)(void**, int, UniqueId128, int);
  // ncclCommInitRank blocks until EVERY rank has joined: a rank that never arrives (a crashed peer, a mismatched world size)
  // would hang the others for ever.  Bounded: the call runs on a helper thread; after GMVAE_COMM_INIT_TIMEOUT seconds (default
  // 180) this returns GMVAE_E_TIMEOUT and the caller exits non-zero (the helper thread stays blocked inside RCCL: the process
  // is expected to end; nothing is re-executed).
  double limit = 180.0;
  if (const char* e = getenv("GMVAE_COMM_INIT_TIMEOUT")) limit = atof(e) > 0 ? atof(e) : limit;
  struct State { std::mutex mu; std::condition_variable cv; bool done = false; int rc = 0; void* comm = nullptr; };
  auto stp = std::make_shared<State>();
  int dev = 0;
  hipGetDevice(&dev);
  const InitFn fn = reinterpret_cast<InitFn>(g_rccl.CommInitRank);
  std::thread([stp, fn, world, id, rank, dev]() {
    hipSetDevice(dev);                             // (the device is per thread)
    void* c = nullptr;
    const int rc = fn(&c, world, id, rank);
    std::lock_guard<std::mutex> lk(stp->mu);
    stp->rc = rc; stp->comm = c; stp->done = true;
    stp->cv.notify_all();
  }).detach();
  std::unique_lock<std::mutex> lk(stp->mu);
  if (!stp->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return stp->done; })) return GMVAE_E_TIMEOUT;
  *comm = stp->comm;
  return stp->rc ? 1000 + stp->rc : 0;
}

int gmvae_comm_count(void* comm, int* nranks) {
  if (!comm || !nranks || !g_rccl.h) return GMVAE_E_NULL;
  if (!g_rccl.CommCount) return GMVAE_E_NULL;
  return g_rccl.CommCount(comm, nranks) ? 1000 : 0;
}

int gmvae_comm_destroy(void* comm) {
  if (!comm || !g_rccl.h) return GMVAE_E_NULL;
  return g_rccl.CommDestroy(comm) ? 1000 : 0;
}

/* one data-parallel training step on `stream`: gradient sums -> ONE RCCL all-reduce -> TF-Adam scaled by 1/count.
 * in_graph: the Adam launch also scatters the next step's weight images (kernels.hpp adam_tf_img); imgs_ready: the
 * previous step of the same graph did so, and this step may run its first layer inside mega_fwd_bwd. */
static int dp_step_impl(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v, float* grads,
                        void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float beta1, float beta2,
                        float epsilon, void* comm, hipStream_t st, bool in_graph, bool imgs_ready, float* tail_log = nullptr,
                        int span_slot = -1, Prof* prof = nullptr) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !m || !v || !grads || !workspace || !comm || !g_rccl.h || !step_dev) return GMVAE_E_NULL;
  if (!aligned16(params) || !aligned16(grads) || !aligned16(workspace)) return GMVAE_E_ALIGN;
  Layout L;
  build_layout(*dims, model, L);
  WS w;
  carve(*dims, model, L, workspace, w);
  const bool mega = mega_ok(*dims, model) && model != GMVAE_MODEL_VAE_GMP;
  ImgPlan pl;
  MegaLay ml;
  bool scatter = false;
  if (in_graph && mega) {
    ml = mega_lay(dims->hidden[0], dims->L, dims->K, dims->D, model);
    plan_images(*dims, model, L, w, ml, params, pl);
    scatter = pl.map_ok && ml.fl_ok && !getenv("GMVAE_NO_FL") && !sched_safe(*dims);
  }
  Ctx cx;
  cx.st = st;
  StepArgs a = {dims, model, x, nullptr, nullptr, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                seed, 0, step_dev, true};
  a.dp_images = scatter;
  a.imgs_ready = scatter && imgs_ready;
  if (span_slot >= 0) { a.want_spans = true; a.span_slot = span_slot; }
  cx.prof = prof;
  int rc = run_step(cx, a);
  if (rc) return rc;
  const int nrc = g_rccl.AllReduce(grads, grads, (size_t)L.P_pad + GMVAE_TAIL, /*ncclFloat*/ 7, /*ncclSum*/ 0, comm, st);
  if (nrc) return 1000 + nrc;
  if (!scatter) {
    if (tail_log) hipMemcpyAsync(tail_log, grads + L.P_pad, GMVAE_TAIL * sizeof(float), hipMemcpyDeviceToDevice, st);
    return adam_tf_step(params, m, v, grads, L.P_pad, lr, beta1, beta2, epsilon, 0, step_dev, 1.f, grads + L.P_pad + 4,
                        grads + L.P_pad, st);
  }
  if (mega2_ok(*dims, model) && w.img2f && !getenv("GMVAE_NO_ADAM_TILES")) {
    // at mega2's sizes every parameter is a tile of one of the step's weight tensors: the optimizer in tile shape (mega3.hpp)
    std::vector<DwArgs> da_store(1);
    DwArgs& da = da_store[0];
    memset(&da, 0, sizeof(da));
    dw_tensors(*dims, model, L, w, x, pl, da);
    if (da.ntens <= kM3MaxT && da.total_tiles <= kM3MaxSlots) {
      std::vector<AdamTilesArgs> at_store(1);
      AdamTilesArgs& at = at_store[0];
      memset(&at, 0, sizeof(at));
      at.ntens = da.ntens; at.total_tiles = da.total_tiles;
      for (int i = 0; i < da.ntens; ++i) {
        at.t[i] = da.t[i];
        const int nt_i = (i + 1 < da.ntens ? da.t[i + 1].tile_begin : da.total_tiles) - da.t[i].tile_begin;
        for (int t = 0; t < nt_i; ++t) at.perm[da.t[i].tile_begin + t] = (unsigned short)((i << 10) | t);
      }
      at.grads = grads; at.p = params; at.m = m; at.v = v; at.lr = lr; at.b1 = beta1; at.b2 = beta2; at.eps = epsilon;
      at.t_dev = reinterpret_cast<const unsigned long long*>(step_dev);
      at.gscale_dev = grads + L.P_pad + 4; at.loss_sum_dev = grads + L.P_pad; at.tail_log = tail_log;
      at.img[0] = w.img_m; at.img[1] = w.dimg; at.img[2] = w.img2f; at.img[3] = w.img2b; at.img[4] = w.dimg2;
      at.epoch_word = w.sync; at.lr_dev = w.sync + 4;
      if (span_slot >= 0 && w.spans) at.span = w.spans + (size_t)span_slot * 3 * 2048 * 2 + 2 * 2048 * 2;
      (void)hipGetLastError();
      hipLaunchKernelGGL(adam_tiles, dim3((unsigned)da.total_tiles), dim3(256), 0, st, at);
      return (int)hipGetLastError();
    }
  }
  ImgScatter sc;
  memset(&sc, 0, sizeof(sc));
  sc.nmap = pl.nmap; sc.lo = pl.lo; sc.hi = pl.hi; sc.epoch_word = w.sync;
  sc.img[0] = w.img_m; sc.img[1] = w.dimg; sc.img[2] = w.img2f; sc.img[3] = w.img2b; sc.img[4] = w.dimg2;
  for (int i = 0; i < pl.nmap; ++i) { sc.map[i] = pl.map[i]; sc.mbegin[i] = pl.map[i].begin; sc.mend[i] = pl.map[i].end; }
  if (span_slot >= 0 && w.spans) sc.span = w.spans + (size_t)span_slot * 3 * 2048 * 2 + 2 * 2048 * 2;     // (the third kernel's region of the slot)
  sc.lr_dev = w.sync + 4;
  (void)hipGetLastError();
  hipLaunchKernelGGL(adam_tf_img, dim3((unsigned)((L.P_pad / 4 + 255) / 256)), dim3(256), 0, st, params, m, v, grads,
                     (long long)L.P_pad, lr, beta1, beta2, epsilon, step_dev, grads + L.P_pad + 4, grads + L.P_pad, tail_log, sc);
  return (int)hipGetLastError();
}

int gmvae_dp_step(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                  float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float beta1,
                  float beta2, float epsilon, void* comm, void* stream) {
  return dp_step_impl(dims, model, x, params, m, v, grads, workspace, seed, step_dev, lr, beta1, beta2, epsilon, comm,
                      static_cast<hipStream_t>(stream), false, false);
}

/* measurement hook (bench.py --config configs3_dp1, and every rank of bench.py --gpus N): the data-parallel step's timeline.
 * Three consecutive steps in ONE captured graph (RCCL node included), replayed `iters` times; the launches of steps 2 and 3
 * stamp the device wall clock (100 MHz, one clock for the device).  out[0] in-kernel span of the gradient launch(es) of a
 * step (first workgroup start -> last workgroup end); out[1] = that end -> first block start of the Adam launch: the
 * all-reduce window (the RCCL node and the two launch boundaries around it); out[2] span of the Adam launch; out[3] its end ->
 * the NEXT step's first workgroup start; out[4] the step: first workgroup start -> the next step's.  Microseconds, means.
 * COLLECTIVE: every rank of the communicator must call it with the same `iters`. */
int gmvae_dp_profile(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v, float* grads,
                     void* workspace, uint64_t seed, uint64_t* step_dev, float lr, void* comm, int iters, float* out,
                     int max_levels, int* n_levels, char* names, void* stream) {
  if (int e = check_dims(dims, model)) return e;
  if (!x || !params || !m || !v || !grads || !workspace || !step_dev || !comm || !out) return GMVAE_E_NULL;
  if (iters < 1) return GMVAE_E_DIMS;
  if (!mega_ok(*dims, model)) return GMVAE_E_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Layout L;
  build_layout(*dims, model, L);
  WS w;
  carve(*dims, model, L, workspace, w);
  if (!w.spans) return GMVAE_E_DIMS;
  hipStream_t cs = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  Prof* pr = new Prof();
  pr->events = false;
  int rc = 0;
  if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { delete pr; return (int)hipGetLastError(); }
  rc = g_rccl.AllReduce(grads, grads, (size_t)L.P_pad + GMVAE_TAIL, 7, 0, comm, cs) ? 1000 : 0;      // (channel set-up outside capture)
  hipStreamSynchronize(cs);
  if (rc == 0 && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = (int)hipGetLastError();
  if (rc == 0) {
    for (int it = 0; it < 3 && rc == 0; ++it) {
      if (it < 2) pr->n = 0;
      pr->active = it == 1;
      rc = dp_step_impl(dims, model, x, params, m, v, grads, workspace, seed, step_dev, lr, 0.9f, 0.999f, 1e-8f, comm, cs, true,
                        it > 0, nullptr, it == 0 ? -1 : it - 1, pr);
    }
    const hipError_t he = hipStreamEndCapture(cs, &graph);
    if (rc == 0 && he != hipSuccess) rc = (int)he;
  }
  if (rc == 0 && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = (int)hipGetLastError();
  const size_t nsp = 2 * 3 * 2048 * 2, slot = 3 * 2048 * 2;
  unsigned long long* hsp = new unsigned long long[nsp];
  double acc[5] = {0, 0, 0, 0, 0};
  int good = 0;
  for (int it = 0; it < iters && rc == 0; ++it) {
    hipMemsetAsync(w.spans, 0, nsp * 8, st);
    if (hipGraphLaunch(exec, st) != hipSuccess) { rc = (int)hipGetLastError(); break; }
    hipStreamSynchronize(st);
    hipMemcpy(hsp, w.spans, nsp * 8, hipMemcpyDeviceToHost);
    unsigned long long lo[2] = {~0ull, ~0ull}, hi[2] = {0, 0};
    for (int sl = 0; sl < 2; ++sl)
      for (int k = 0; k < 2; ++k)                 // the per-row launch and (two-launch form) the weight-gradient launch
        for (int b = 0; b < 2048; ++b) {
          const unsigned long long s0 = hsp[sl * slot + (size_t)k * 2048 * 2 + 2 * b], s1 = hsp[sl * slot + (size_t)k * 2048 * 2 + 2 * b + 1];
          if (!s0 || !s1) continue;
          lo[sl] = s0 < lo[sl] ? s0 : lo[sl];
          hi[sl] = s1 > hi[sl] ? s1 : hi[sl];
        }
    const unsigned long long a0r = hsp[2 * 2048 * 2], a1 = hsp[2 * 2048 * 2 + 1];
    if (!a0r || !a1 || hi[0] <= lo[0] || lo[1] == ~0ull) continue;
    const unsigned long long a0 = (1ull << 62) - a0r;
    acc[0] += (double)(hi[0] - lo[0]) * 0.01;
    acc[1] += (double)(a0 - hi[0]) * 0.01;
    acc[2] += (double)(a1 - a0) * 0.01;
    acc[3] += (double)(lo[1] - a1) * 0.01;
    acc[4] += (double)(lo[1] - lo[0]) * 0.01;
    ++good;
  }
  delete[] hsp;
  for (int i = 0; i < 5; ++i) out[i] = good ? (float)(acc[i] / good) : 0.f;
  if (n_levels && names) {
    const int n = pr->n < max_levels ? pr->n : max_levels;
    *n_levels = n;
    for (int i = 0; i < n; ++i) memcpy(names + (size_t)i * 48, pr->name[i], 48);
  }
  if (exec) hipGraphExecDestroy(exec);
  if (graph) hipGraphDestroy(graph);
  if (cs) hipStreamDestroy(cs);
  delete pr;
  (void)hipGetLastError();
  return rc ? rc : (good ? 0 : GMVAE_E_DIMS);
}

/* the same step captured once into a hipGraph (RCCL kernels included); replay with gmvae_train_graph_launch */
int gmvae_dp_graph_create(const GmvaeDims* dims, int model, const uint8_t* x, int n_steps, float* params, float* m, float* v,
                          float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float beta1,
                          float beta2, float epsilon, void* comm, float* tail_log, void** graph_out) {
  if (int e = check_dims(dims, model)) return e;
  if (!graph_out || !comm) return GMVAE_E_NULL;
  if (n_steps < 1 || n_steps > 1024) return GMVAE_E_DIMS;
  hipStream_t cs;
  hipError_t he = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (he != hipSuccess) return (int)he;
  // warm-up outside capture: RCCL sets up its channels on first use of a communicator.  The gradient buffer is
  // scratch at this point (gmvae_step overwrites it), so reducing it changes no training state.
  Layout L;
  build_layout(*dims, model, L);
  int rc = g_rccl.AllReduce(grads, grads, (size_t)L.P_pad + GMVAE_TAIL, 7, 0, comm, cs) ? 1000 : 0;
  hipStreamSynchronize(cs);
  GmvaeTrainGraph* tg = new GmvaeTrainGraph();
  if (rc == 0) {
    he = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (he != hipSuccess) rc = (int)he;
  }
  if (rc == 0) {
    for (int s = 0; s < n_steps && rc == 0; ++s)
      rc = dp_step_impl(dims, model, x + (size_t)s * dims->B * dims->D, params, m, v, grads, workspace, seed, step_dev, lr,
                        beta1, beta2, epsilon, comm, cs, true, s > 0, tail_log ? tail_log + (size_t)s * GMVAE_TAIL : nullptr);
    he = hipStreamEndCapture(cs, &tg->graph);
    if (rc == 0 && he != hipSuccess) rc = (int)he;
  }
  if (rc == 0) {
    he = hipGraphInstantiate(&tg->exec, tg->graph, nullptr, nullptr, 0);
    if (he != hipSuccess) rc = (int)he;
  }
  hipStreamDestroy(cs);
  if (rc != 0) {
    if (tg->graph) hipGraphDestroy(tg->graph);
    delete tg;
    (void)hipGetLastError();
    return rc;
  }
  *graph_out = tg;
  return 0;
}

}  // extern "C"
