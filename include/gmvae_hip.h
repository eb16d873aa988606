/*
 * gmvae_hip.h -- C ABI of libgmvae_hip.so: the MI355X (gfx950) implementation
 * of the one hot path of mazrk7/gmvae: the VAE / VAE_GMP / GMVAE single-sample
 * ELBO training step (+ the IWAE n_samples extension).
 *
 * The reference has NO FFI/plugin API (it is pure TF1 graph Python); the
 * boundary it exposes is the Python object protocol of scripts/vae.py and
 * scripts/gmvae.py.  gmvae_amd/{base,vae,gmvae}.py mirror that protocol and
 * bind the entry points below with ctypes.  Each entry point cites the
 * reference call site(s) whose arithmetic it replaces (paths relative to the
 * upstream repo).
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t on a HIP
 *     failure, or a negative GMVAE_E_* code on a bad argument; nothing throws;
 *   - no allocation and no synchronisation inside: the caller owns every
 *     buffer (device memory, 16-byte aligned) including the workspace, whose
 *     size is queried with gmvae_workspace_bytes() and which must be zeroed
 *     ONCE after allocation and then used with ONE set of dims (padding words
 *     are never written again; it also holds the epoch counter and the error
 *     word of the in-launch hand-offs of the fused schedule: a hand-off that
 *     times out -- the schedule needs the whole device to itself -- poisons
 *     that step's loss and gradients with NaN and sets the error word, after
 *     which waits no longer block; re-zero the workspace to clear it);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), is
 *     asynchronous and graph-capturable; no global state;
 *   - all float tensors are fp32 row-major; x is uint8/bool {0,1} [B,D];
 *   - sample-dependent tensors have R = B*S rows, row r = b*S + s.
 */
#ifndef GMVAE_HIP_H_
#define GMVAE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMVAE_MAX_HIDDEN 8
#define GMVAE_TAIL 8          /* floats appended to the gradient buffer */
#define GMVAE_ABI_VERSION 7   /* 5: + gmvae_dp_profile, gmvae_forward_profile; 6: GmvaeDims.hidden_act appended; 7: + gmvae_comm_count, GMVAE_E_TIMEOUT */

enum { GMVAE_MODEL_VAE = 0, GMVAE_MODEL_VAE_GMP = 1, GMVAE_MODEL_GMVAE = 2 };

enum {
  GMVAE_E_NULL = -1,      /* required pointer is NULL            */
  GMVAE_E_DIMS = -2,      /* non-positive / unsupported sizes    */
  GMVAE_E_MODEL = -3,     /* unknown model id                    */
  GMVAE_E_ALIGN = -4,     /* pointer not 16-byte aligned         */
  GMVAE_E_NET = -5,       /* unknown sub-network id              */
  GMVAE_E_SMALL = -6,     /* caller array too small              */
  GMVAE_E_TIMEOUT = -7    /* gmvae_comm_init: a rank did not join within GMVAE_COMM_INIT_TIMEOUT seconds */
};

/* Sizes + the hyper-parameters scripts/runners.py:78-101 binds
 * (sigma_min=0, raw_sigma_bias=0.5, temperature=1; gen_bias_init=0 from
 * scripts/gmvae.py:285).  hidden[] = fcnet_hidden_sizes. */
typedef struct GmvaeDims {
  int32_t B;                          /* rows of x on this device            */
  int32_t D;                          /* data_size                           */
  int32_t L;                          /* latent_size                         */
  int32_t K;                          /* mixture_components (1 for plain VAE)*/
  int32_t S;                          /* IWAE samples; 1 == the reference    */
  int32_t n_hidden;
  int32_t hidden[GMVAE_MAX_HIDDEN];
  float sigma_min;
  float raw_sigma_bias;
  float temperature;
  float gen_bias_init;
  /* Data parallel (no reference counterpart; scripts/runners.py:193 pins one device): global index of this
   * device's first batch row, rank * B for equal shards, 0 on a single device.  It enters ONLY the Philox counters
   * of the in-kernel noise (eps, u) and of gmvae_binarize: row b of this device draws what row row0 + b of the
   * single-device step on the whole global batch draws, so G shards reproduce the 1-device step on G*B rows.
   * Sizes, layouts and the workspace do not depend on it. */
  uint64_t row0;
  /* ABI v3 -- vector bias_init of ConditionalBernoulli (scripts/base.py:102-103 "a scalar or vector Tensor that is added
   * to the output of the fully connected network", e.g. the logit of the training-set mean; added at base.py:135):
   * device pointer to gen_bias_len == D fp32 values, or NULL / 0 for the scalar form alone.  The decoder logits are
   * MLP(z) + gen_bias_init + gen_bias_vec[j].  It is a constant of the model (no gradient), read by the decoder
   * output layer's epilogue; steps with a vector run the schedules whose decoder layer is a grouped-GEMM launch. */
  const float* gen_bias_vec;
  int32_t gen_bias_len;
  /* ABI v4 -- schedule flags (the v3 `reserved_` word; 0 = the default schedules).  GMVAE_SCHED_SAFE: only schedules in which
   * no workgroup waits for another workgroup of its own launch (one workgroup per 16-row panel, the first layer as a launch
   * of its own): slower (4 launches per step instead of 2), never stalling when something else holds part of the device.  A
   * caller sets it after a hand-off timeout (the workspace's error word) -- per call, not per process.  It changes neither
   * sizes, layouts nor the workspace. */
  int32_t sched_flags;
  /* ABI v6 -- hidden_activation_fn of every conditional's MLP (scripts/base.py:19,90,153 take any callable; gmvae.py:282 and
   * vae.py:196 pass ONE to all networks; default tf.nn.relu): GMVAE_ACT_*.  Steps with an activation other than ReLU run the
   * general schedule (one grouped-GEMM launch per dependency level: the activation in the forward epilogue, its derivative --
   * a function of the kept activation -- in the data-gradient epilogue). */
  int32_t hidden_act;
} GmvaeDims;
enum { GMVAE_SCHED_SAFE = 1,
       /* forward-only calls (gmvae_forward): the operand images a previous gmvae_forward left in THIS workspace were built from
        * the parameters as they still are (an evaluation walks a split batch by batch on fixed parameters): evalf_prep is skipped.
        * The caller vouches for it; gmvae_amd.Engine tracks every writer of its parameter buffer. */
       GMVAE_SCHED_EVAL_IMAGES_VALID = 2 };
enum { GMVAE_ACT_RELU = 0, GMVAE_ACT_TANH = 1, GMVAE_ACT_SIGMOID = 2, GMVAE_ACT_ELU = 3 };

/* One tensor of the flat parameter buffer.  Names are the reference's TF
 * variable names (scripts/base.py:53,60 '<name>_fcnet/linear_<i>/{w,b}';
 * scripts/vae.py:233-238 'loc','raw_scale_diag','mixture_logits'), in the
 * reference's variable-creation order. */
typedef struct GmvaeParamEntry {
  char name[64];
  int32_t rows;                       /* w: in  ; b: 1 ; loc: K              */
  int32_t cols;                       /* w: out ; b: out                     */
  uint64_t offset;                    /* in floats, multiple of 4            */
} GmvaeParamEntry;

/* sub-network ids for gmvae_mlp_forward */
enum {
  GMVAE_NET_ENCODER_Y = 0,            /* scripts/gmvae.py:340-345            */
  GMVAE_NET_PRIOR_GMM = 1,            /* scripts/gmvae.py:321-327            */
  GMVAE_NET_ENCODER_GMM = 2,          /* scripts/gmvae.py:347-353            */
  GMVAE_NET_DECODER = 3,              /* scripts/gmvae.py:331-336, vae.py:254-259 */
  GMVAE_NET_ENCODER = 4               /* scripts/vae.py:262-268              */
};

int gmvae_abi_version(void);

/* Replaces variable creation in create_vae / create_gmvae
 * (scripts/vae.py:191-271, scripts/gmvae.py:277-356): sizes of the flat
 * buffer.  P_padded counts the 16-byte alignment padding, P_real does not
 * (166,618 for GMVAE D=784 H=64 L=64 K=10). */
int gmvae_param_count(const GmvaeDims* dims, int model, uint64_t* P_padded, uint64_t* P_real);
int gmvae_param_layout(const GmvaeDims* dims, int model, GmvaeParamEntry* out, int max_entries, int* n_entries);

int gmvae_workspace_bytes(const GmvaeDims* dims, int model, uint64_t* bytes);

/* TrainableGMVAE.run_model (scripts/gmvae.py:223-274) /
 * TrainableVAE.run_model (scripts/vae.py:153-188) PLUS the reverse-mode pass
 * of opt.compute_gradients (scripts/runners.py:182).
 *   x     uint8 [B,D]
 *   eps   fp32 [B*S,L]  N(0,1) noise     (MultivariateNormalDiag.sample)
 *   u     fp32 [B*S,K]  U[tiny,1) noise  (RelaxedOneHotCategorical.sample; GMVAE only)
 *         eps/u NULL => generated in-kernel by Philox4x32-10(seed, step)
 *   grads fp32 [P_padded + GMVAE_TAIL], OUT: SUMS over this device's rows of
 *         d loss_b / d theta (NOT divided by B), then the tail
 *         [0] sum_b loss_b  [1] sum nll  [2] sum kl  [3] sum nent  [4] B
 *         (nll/kl are averaged over S inside a row group) -- one RCCL
 *         all-reduce(SUM) of this buffer makes it global; adam_tf_step's
 *         grad_scale = 1/tail[4] turns sums into the reference's batch means.
 *   step_dev (may be NULL): device-resident step counter for hipGraph replay: TWO uint64 words, [0] the counter,
 *         [1] a scratch copy the step's launches hand to each other (every entry point that takes step_dev).
 *         When given it overrides `step` for the Philox stream and is
 *         incremented once per call (after the noise is drawn), so that
 *         adam_tf_step(t_dev = step_dev) later on the stream sees t = step+1.
 */
int gmvae_step(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
               const float* params, float* grads, void* workspace, uint64_t seed, uint64_t step,
               uint64_t* step_dev, void* stream);

/* Forward only (eval: scripts/runners.py:324-333).  tail: float[GMVAE_TAIL]
 * as above.  row_terms (may be NULL): fp32 [B*S,4] = logpx, logq, logp, logw.
 * z_out (may be NULL) [B*S,L]; y_out (may be NULL, GMVAE) [B*S,K];
 * logits_out (may be NULL, GMVAE) [B,K] = q_y.distribution.logits. */
int gmvae_forward(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
                  const float* params, float* tail, float* row_terms, float* z_out, float* y_out,
                  float* logits_out, void* workspace, uint64_t seed, uint64_t step, void* stream);

/* tf.compat.v1.train.AdamOptimizer.apply_gradients (scripts/runners.py:181-183):
 * epsilon is added to the UN-corrected sqrt(v).  t = 1-based step count.
 * t_dev (may be NULL): device pointer overriding t (graph replay).
 * g = grads[i] * grad_scale.  grad_scale_dev (may be NULL): device pointer to
 * a count; when given, grad_scale = 1 / (*grad_scale_dev) overrides.
 * loss_sum_dev (may be NULL): device pointer to the step's loss sum (tail[0] of the gradient buffer).  When it
 * holds a non-finite value -- a hand-off of the fused schedule timed out and poisoned the step, on this or (after
 * the all-reduce) on any rank -- the update is SKIPPED: params, m and v keep their values (the step counter still
 * advances; the caller sees the NaN loss and the workspace error word). */
int adam_tf_step(float* params, float* m, float* v, const float* grads, uint64_t P, float lr, float beta1,
                 float beta2, float epsilon, uint64_t t, const uint64_t* t_dev, float grad_scale,
                 const float* grad_scale_dev, const float* loss_sum_dev, void* stream);

/* One conditional network's MLP (scripts/base.py:66-67,133-135,196-198):
 * out[rows, out_dim] = MLP(concat(in, in2)) (+ gen_bias_init for the decoder).
 * `in` is uint8 when in_is_u8 else fp32.  in2 is y for ENCODER_GMM, else NULL.
 * The distribution heads (softplus, sigmoid, softmax) are applied by the
 * Python distribution objects on top of this. `B` in dims is ignored; rows is used. */
int gmvae_mlp_forward(const GmvaeDims* dims, int model, int net, const void* in, int in_is_u8,
                      const float* in2, int rows, const float* params, float* out, void* workspace,
                      void* stream);

/* Dynamic binarisation of the reference's input pipeline on the device (scripts/runners.py:44-47 `_preprocess`:
 * image = pixel / 255.; x = image < uniform -- note P[x = 1] = 1 - pixel/255 -- re-drawn on every pass).
 * pixels: uint8 [n_rows][D] resident in HBM (MNIST: 60000 x 784 = 47 MB); idx: int32 [B] source rows of this batch
 * (an epoch permutation kept on the device) or NULL for rows row0 .. row0+B-1; x_out: uint8 [B][D] of 0/1, the
 * layout gmvae_step takes.  The uniforms are Philox4x32-10 keyed by (seed, step or *step_dev) and the element's
 * position in the GLOBAL batch (row out_row0 + b, column), so a step is reproducible and a sharded batch draws what
 * the whole batch would (out_row0 = GmvaeDims::row0).  D % 4 == 0; pixels and x_out 4-byte aligned. */
int gmvae_binarize(const uint8_t* pixels, uint64_t n_rows, const int32_t* idx, uint64_t row0, int B, int D,
                   uint64_t seed, uint64_t step, const uint64_t* step_dev, uint8_t* x_out, uint64_t out_row0,
                   void* stream);

/* The noise gmvae_step draws when eps/u are NULL, as arrays: eps ~ N(0,1) [rows][L], u ~ U[tiny,1) [rows][K]
 * (either pointer may be NULL).  Philox4x32-10, counter = (global row row_base + r, quad of the row, stream, step or
 * *step_dev), key = seed: gmvae_step(dims, ..., seed, step) with dims->row0 = row_base and rows = B*S draws exactly
 * these values, in every schedule (tests feed them to the CPU oracle). */
int gmvae_noise_fill(float* eps, float* u, uint64_t rows, int L, int K, uint64_t row_base, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, void* stream);

/* utils.cluster_acc (scripts/utils.py:173-191) on device.  scratch: int32
 * [K*n_labels + B] (histogram, zeroed by the callee, then per-row argmax);
 * acc_out float[1].  Mode ties resolve to the smallest label. */
int gmvae_cluster_acc(const float* logits, const int64_t* labels, int B, int K, int n_labels,
                      int32_t* scratch, float* acc_out, void* stream);

/* ---- measurement / test hooks (not used by the reference-facing API) ---- */

/* One GEMM through the same grouped fp32-MFMA kernel the step uses.
 * cfg: tile configuration (0 small 32x32, 1 medium 64x64, 2 large 128x128, -1 auto); 4 / 5: the pre-split bf16-triple planes
 *      (5 reuses the planes of the previous cfg-4 call); 6 / 7: the f16-pair planes likewise; 8: the weight-stationary row
 *      kernels (NN K = 64; NT K = 128 with `bias` as a ReLU mask [M][N]; NT K = 512 or 640 = 512 + 128, N = 64, with `bias`
 *      as an addend [M][N]); shapes a form does not take are refused (GMVAE_E_DIMS).
 * trans 0 (NN): C[M,N] = act(A[M,K] W[K,N] + bias)
 * trans 1 (NT): C[M,N] = A[M,K] W[N,K]^T
 * trans 2 (TN): C[s][M(+1),N] = A[K,M]^T W[K,N] split over K into `splitk` slabs
 *               of (M+1)*N floats; bias != NULL requests the bias gradient
 *               (column sums of W over the slab's K range) at row M.
 * a_is_u8: A holds uint8. */
int gmvae_gemm_test(const void* A, int a_is_u8, const float* W, const float* bias, float* C, int M, int N,
                    int K, int trans, int relu, int cfg, int splitk, void* stream);

/* Runs the full step `iters` times with hipEvents around EVERY launch (on
 * `stream`) and returns, per launch ("level"), its name [48 chars each], mean
 * microseconds and the algorithmic FLOPs (2*M*N*K summed over its GEMMs,
 * 0 for row-local kernels).  Synchronises the stream: measurement only.
 * gmvae_train_profile does the same for the steady-state TRAINING step of a train graph (Philox noise, TF-Adam
 * fused into the last launch, first layer inside mega_fwd_bwd where that schedule applies): a three-step graph (an
 * untimed step, then two steps whose launches stamp the device wall clock per workgroup) replayed `iters` times, so
 * the kernels run back to back as in a train graph; it advances params / m / v / *step_dev like 3 * iters real steps.
 * usec = in-kernel span (last workgroup end - first workgroup start); usec_timeline (may be NULL) = the launch's share
 * of the step's timeline: first workgroup start of the NEXT launch - its own (dispatch and end-of-kernel write-back
 * included: the interval rocprofv3 --kernel-trace reports; the shares of a step's launches add up to the step). */
int gmvae_train_profile(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                        float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, int iters,
                        int max_levels, int* n_levels, char* names, float* usec, float* usec_timeline, double* flops,
                        void* stream);
/* The data-parallel step's timeline (scripts/runners.py:231-232 has no counterpart: the reference is single-device; SURVEY.md
 * 8(e)): three consecutive steps of gmvae_dp_graph_create's graph (RCCL all-reduce node included) in ONE graph, replayed
 * `iters` times; the launches of steps 2 and 3 stamp the device wall clock.  out[5], microseconds, means: [0] in-kernel span of
 * a step's gradient launch(es); [1] their last workgroup's end -> first block of the Adam launch (the all-reduce window: the
 * RCCL node and the launch boundaries around it); [2] span of the Adam launch; [3] its end -> the next step's first workgroup;
 * [4] the step.  names / n_levels (may be NULL): the gradient launches' names.  COLLECTIVE over `comm`: same iters on every rank. */
int gmvae_dp_profile(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v, float* grads,
                     void* workspace, uint64_t seed, uint64_t* step_dev, float lr, void* comm, int iters, float* out,
                     int max_levels, int* n_levels, char* names, void* stream);
/* The forward-only evaluation (gmvae_forward with in-kernel Philox noise; scripts/runners.py:324-333 reuses the model's loss
 * for the -log p(x) bound) timed: per launch with hipEvents (names / usec / flops as gmvae_step_profile), and *usec_total =
 * microseconds per forward when ONE captured forward is replayed `iters` times back to back. */
int gmvae_forward_profile(const GmvaeDims* dims, int model, const uint8_t* x, const float* params, float* tail, void* workspace,
                          uint64_t seed, int iters, int max_levels, int* n_levels, char* names, float* usec, double* flops,
                          float* usec_total, void* stream);
int gmvae_step_profile(const GmvaeDims* dims, int model, const uint8_t* x, const float* eps, const float* u,
                       const float* params, float* grads, void* workspace, uint64_t seed, int iters,
                       int max_levels, int* n_levels, char* names, float* usec, double* flops, void* stream);

/* `iters` full training steps (gmvae_step in Philox mode + adam_tf_step) issued from C on `stream` and
 * timed with hipEvents: mode 0 = eager launches, mode 1 = one hipGraph captured here and replayed.
 * Synchronises: measurement only. */
int gmvae_bench_loop(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                     float* grads, void* workspace, uint64_t* step_dev, int iters, int mode, float* usec_per_step,
                     void* stream);

/* One full training step -- Philox noise + gmvae_step + adam_tf_step(t = *step_dev, grad_scale =
 * 1/count from the tail) -- captured ONCE into a hipGraph owned by the library, for replay with a single
 * call per step (the sess.run([train_op, global_step]) of scripts/runners.py:231-232).  All pointers
 * are baked into the graph: copy each new batch into `x` before launching.
 * n_steps >= 1 consecutive steps go into the one graph; `x` then holds n_steps batches back to back
 * ([n_steps][B][D]: the next n_steps batches of the input pipeline).  One launch per n_steps steps amortises
 * the ~6 us the GPU idles between two graph launches (measured, profiles/): the kernels inside a graph run
 * back to back.
 * tail_log (may be NULL): fp32 [n_steps][GMVAE_TAIL]; step s of a launch also writes its tail (loss sums + count,
 * all-reduced in the data-parallel graph) to row s: the per-step losses the reference's logging and early-stopping
 * hooks see (scripts/runners.py:198-200,222-228), read once per launch instead of once per step. */
int gmvae_train_graph_create(const GmvaeDims* dims, int model, const uint8_t* x, int n_steps, float* params, float* m,
                             float* v, float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr,
                             float beta1, float beta2, float epsilon, float* tail_log, void** graph_out);
/* The same graph with the input pipeline inside: before each of its n_steps steps, gmvae_binarize draws that
 * step's batch from the resident uint8 `pixels` [n_rows][D] -- rows idx[s][0..B) of the int32 device buffer
 * idx [n_steps][B], which the caller refills (an epoch permutation) before every launch -- into x_scratch
 * [n_steps][B][D], with uniforms keyed by the device step counter: a new draw every step, nothing crosses PCIe. */
int gmvae_train_graph_create_pipeline(const GmvaeDims* dims, int model, const uint8_t* pixels, uint64_t n_rows,
                                      const int32_t* idx, uint8_t* x_scratch, int n_steps, float* params, float* m, float* v,
                                      float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr,
                                      float beta1, float beta2, float epsilon, float* tail_log, void** graph_out);
int gmvae_train_graph_launch(void* graph, void* stream);
int gmvae_train_graph_destroy(void* graph);

/* ---- data parallel over RCCL (no reference counterpart: the reference is single-device, scripts/runners.py:193).
 * RCCL is bound with dlopen(rccl_path or "librccl.so") at first use.  Rank 0 calls gmvae_comm_unique_id and
 * distributes the 128 bytes by any means (torch.distributed broadcast in gmvae_amd); every rank then calls
 * gmvae_comm_init.  gmvae_dp_step = gmvae_step (Philox mode) -> ONE all-reduce(SUM) of grads[P_padded + TAIL]
 * -> adam_tf_step with grad_scale = 1/tail[4], all enqueued on `stream`.  gmvae_dp_graph_create warms RCCL up
 * with one all-reduce of the (scratch) gradient buffer and changes no training state.
 * Return codes >= 1000 are 1000 + ncclResult_t. */
int gmvae_comm_unique_id(const char* rccl_path, char* out128);
/* gmvae_comm_init is BOUNDED: ncclCommInitRank runs on a helper thread and GMVAE_E_TIMEOUT comes back after
 * GMVAE_COMM_INIT_TIMEOUT seconds (environment, default 180) if some rank never joined -- the caller should then exit non-zero
 * (the helper thread stays inside RCCL).  gmvae_comm_count: the number of ranks the communicator actually spans (ncclCommCount). */
int gmvae_comm_init(const char* rccl_path, const char* id128, int rank, int world, void** comm);
int gmvae_comm_count(void* comm, int* nranks);
int gmvae_comm_destroy(void* comm);
int gmvae_dp_step(const GmvaeDims* dims, int model, const uint8_t* x, float* params, float* m, float* v,
                  float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float beta1,
                  float beta2, float epsilon, void* comm, void* stream);
int gmvae_dp_graph_create(const GmvaeDims* dims, int model, const uint8_t* x, int n_steps, float* params, float* m, float* v,
                          float* grads, void* workspace, uint64_t seed, uint64_t* step_dev, float lr, float beta1,
                          float beta2, float epsilon, void* comm, float* tail_log, void** graph_out);

/* Debugging aid: byte offset inside the workspace of a named intermediate ("hy1","hg1","hd1","y",
 * "logits","qp","pp","z","g","dqp","dpp","dlogits","dbuf0".."dbuf2","s1","s4", ...; "he<i>" / "hg<i>" / "hd<i>", i >= 1:
 * the kept input activation of layer i of the encoder (encoder_y for GMVAE) / encoder_gmm / decoder -- the parity
 * tests read the ReLU masks of a step from them). */
int gmvae_workspace_offset(const GmvaeDims* dims, int model, const char* name, uint64_t* byte_offset);

/* Debugging aid: device wall-clock stamps of the skinny schedule's launches ([10 launches][1024 blocks][8] uint64 at
 * 100 MHz).  host_out == NULL arms it (allocates the buffer; steps enqueued or captured afterwards stamp into it);
 * otherwise the buffer is copied to host_out.  tools/skstamps.py. */
int gmvae_debug_sk_stamps(unsigned long long* host_out);
/* Disarms it and frees the buffer.  Destroy every train graph captured while it was armed FIRST (their kernel arguments
 * hold the buffer's address). */
int gmvae_debug_sk_stamps_free(void);

/* Which schedule a TRAINING step of these sizes takes, as text (<= 47 chars + NUL into out48): "mega2", "mega", "skinny",
 * "fused" or "general", with "+planes" appended where the top decoder layer's GEMMs run as bf16 piece products on pre-split
 * operands (gemm.hpp plane_rounds3).  Host-side, reads the same environment switches as the step.  bench.py prices its
 * roofline line with it. */
int gmvae_step_schedule(const GmvaeDims* dims, int model, char* out48);

/* Debugging aid: resident workgroups per CU the HIP runtime reports for a kernel of the library
 * (which: 0/1/2 = grouped GEMM small/medium/large configuration, 3 = mega_fwd_bwd, 4 = finalize_adam). */
int gmvae_kernel_occupancy(int which, int* blocks_per_cu);

#ifdef __cplusplus
}
#endif
#endif /* GMVAE_HIP_H_ */
