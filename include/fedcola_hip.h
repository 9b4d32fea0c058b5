/* fedcola_hip.h -- C ABI of the MI355X-native FedCola client-step library (libfedcola_hip.so).
 *
 * Drop-in boundary for the reference's per-client local training step.  The reference has no FFI (it is pure
 * Python/PyTorch); each entry point below names the reference code it replaces so a maintainer can bind it from
 * src/models/mome.py / src/client/fedavgclient.py / src/server/fedavgserver.py with ctypes.
 *
 * Conventions: plain pointers and sizes only; every data pointer is a DEVICE pointer owned by the caller (on the
 * CURRENT device: the calls check it); all work is enqueued asynchronously on the caller's hipStream_t (passed as
 * void*); return 0 on success, <0 on error (message: fc_last_error()).  A handle is not thread-safe: use one per
 * device/stream.
 * Allocation / synchronisation: the steady state (same handle, same workspace address, same batch shape) allocates nothing
 * and never blocks the host.  The exceptions, each once: the first forward of a process creates the library's internal HIP
 * streams for the device and measures which of them run concurrently (~10 ms); the first backward over a given set of buffers and
 * batch shape uploads its device tables (weight-gradient problems, LayerNorm reductions; fc_client_step also the optimizer's chunk
 * table, a colearn_attn model the shared-gradient chunk table: < 64 KB each, one allocation + one synchronous copy).  The tables are
 * cached per process by CONTENT: another handle of the same configuration over the same addresses, or a return to an earlier
 * workspace / batch shape, finds them and uploads nothing (the cache is dropped, with one device synchronisation, after 1024 entries);
 * fc_prox_term / fc_clip_grad_norm copy a <= 50-KB table from host memory on every call (their scratch is the caller's and may
 * have been reused in between); fc_comm_create builds an RCCL communicator; the fc_k_* test entry points synchronise where their
 * comment says so.
 */
#ifndef FEDCOLA_HIP_H
#define FEDCOLA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FC_ABI_VERSION 6   /* 2: fc_comm_*, fc_aggregate*, fc_workspace_tensor, fc_k_layernorm_bwd_partial; 3: fc_model_cfg.colearn_attn, fc_k_dw;
                              4: fc_image_u8_to_f32, fc_k_gemm_epi; fc_k_layernorm_partial_floats counts fp64 rows;
                              5: fc_k_mlp_pack, fc_k_mlp_fused, fc_model_set_option;
                              6: those three left the product library: they exist in the tools build only (#ifdef FC_PROBES below); + fc_sgd_step */

enum { FC_PREC_FP32 = 0, FC_PREC_BF16 = 1 };
enum { FC_TASK_NONE = 0, FC_TASK_CLS = 1, FC_TASK_RTV = 2 };

/* Mirrors ModalityAgnosticTransformer.__init__ (src/models/mome.py:672-769) + factories (mome.py:924-1033). */
typedef struct fc_model_cfg {
  int32_t has_img, has_txt;             /* modalities = ['img'|None, 'txt'|None] */
  int32_t img_size, patch, in_chans;
  int32_t dim, depth, heads, mlp_hidden;
  int32_t vocab, max_text_len;
  int32_t task_img, task_txt;           /* FC_TASK_* per tower (tasks=[...]) */
  int32_t num_classes_img, num_classes_txt;
  int32_t with_aux, aux_trained, aux_attn_only, aux_mlp_only; /* CrossModalReparamLinear, mome.py:42-97,771-786 */
  int32_t precision;                    /* FC_PREC_*: storage type of activations / compute weights */
  int32_t colearn_attn;                 /* colearn_param == 'attn' (mome.py:836-840): with both towers present the text tower's
                                         * Attention modules ARE the image tower's -- no segments of their own, gradients of both
                                         * towers summed into the image tower's qkv / proj segments.  ABI 3. */
} fc_model_cfg;

typedef struct fc_model fc_model_t;

/* One state_dict entry of the reference model inside the flat fp32 parameter buffer. */
typedef struct fc_segment {
  char name[128];       /* exact reference state_dict key, e.g. "blockses.0.3.attn.qkv.weight" */
  int64_t offset;       /* in floats from the start of the flat buffer (64-element aligned) */
  int64_t numel;
  int32_t ndim;
  int64_t shape[4];
  int32_t trainable;    /* requires_grad (aux_weight: only when aux_trained) */
} fc_segment;

const char* fc_last_error(void);
int fc_abi_version(void);

/* ---- model handle: layout only, owns no device memory (mome.py:672-769) */
int fc_model_create(const fc_model_cfg* cfg, fc_model_t** out);
void fc_model_destroy(fc_model_t* m);
int64_t fc_model_num_params(const fc_model_t* m);        /* padded flat length in floats */
int32_t fc_model_num_segments(const fc_model_t* m);
int fc_model_segment(const fc_model_t* m, int32_t i, fc_segment* out);
int fc_model_set_trainable(fc_model_t* m, int32_t seg, int32_t trainable);  /* fedavgserver.py:422-429 freeze */

/* The library's auxiliary HIP stream (hipStream_t as void*) that carries the text tower.  It idles for most of a step, so the
 * host->device copy of the NEXT batch belongs there (fedcola_amd/loaders/prefetch.py): the GPU runs at most four hardware
 * queues well (caller, text, second image chain, weight gradients) and a fifth stream for copies costs more than it hides. */
void* fc_model_side_stream(const fc_model_t* m);

/* ---- sizes of caller-provided buffers */
size_t fc_workspace_bytes(const fc_model_t* m, int32_t B, int32_t n_txt);   /* activations saved for backward + temporaries */
size_t fc_compute_weights_bytes(const fc_model_t* m);                        /* 0 => pass the params buffer itself */

/* compute weights = cast(params) with W_eff = W + s*A folded in (mome.py:58-60). Call after params change. */
int fc_prepare_weights(const fc_model_t* m, const float* params, void* wc, void* stream);

/* ---- ModalityAgnosticTransformer.forward(x=[img|None, txt|None], feat_out) (mome.py:881-922)
 * img: float32 [B,3,H,W] (1-channel handled by the caller, mome.py:893-894); ids: int64 [B,n_txt].
 * out_img/out_txt: float32 [B, dim] (feat_out or 'rtv': unit-norm features) or [B, num_classes] ('cls' logits).
 * droppath: NULL (eval / rate 0) or float32 [2][depth][2][B] multipliers (0 or 1/keep) -- timm DropPath (mome.py:213,223).
 * Saves activations for fc_backward in `workspace`. */
int fc_forward(const fc_model_t* m, const float* params, const void* wc, const float* img, const int64_t* ids,
               int32_t B, int32_t n_txt, int32_t feat_out, const float* droppath, void* workspace, size_t workspace_bytes,
               float* out_img, float* out_txt, void* stream);

/* Backward of the last fc_forward on this workspace: loss.backward() (fedavgclient.py:97).
 * d_out_*: float32 gradients w.r.t. the forward outputs (NULL for an absent tower).
 * grads: flat float32 buffer laid out like params; must be zero-filled by the caller (gradients accumulate). */
int fc_backward(const fc_model_t* m, const float* params, const void* wc, const float* d_out_img, const float* d_out_txt,
                float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* ---- criteria.  lossbuf: float32[2] device; lossbuf[0] += loss*B (MetricManager.track, src/utils.py:337-343),
 * lossbuf[1] += loss (zero it before the call to read this step's loss). */
/* torchmultimodal ContrastiveLossWithTemperature as used at fedavgclient.py:95 (tau fixed) */
int fc_contrastive_loss_fwd_bwd(const float* a, const float* b, int32_t B, int32_t D, float tau, float* scratch,
                                size_t scratch_floats, float* lossbuf, float* da, float* db, void* stream);
size_t fc_contrastive_scratch_floats(int32_t B);
/* nn.CrossEntropyLoss() (fedavgclient.py:85,90) */
int fc_ce_loss_fwd_bwd(const float* logits, const int64_t* y, int32_t B, int32_t C, float* lossbuf, float* dlogits, void* stream);

/* ---- torch.optim.AdamW.step over the trainable ranges of the flat buffers (fedavgclient.py:63,100). step is 1-based. */
int fc_adamw_step(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* ---- torch.optim.SGD.step (fedavgclient.py:63 with --optimizer SGD, main.py:269-273: lr, momentum, nesterov, weight_decay from args; dampening 0)
 * over the trainable ranges: g' = g + wd p; buf = (step == 1) ? g' : momentum buf + g'; p -= lr (nesterov ? g' + momentum buf : buf)
 * (momentum == 0: p -= lr g', momentum_buf may be NULL).  The caller re-runs fc_prepare_weights before the next forward.  step is 1-based. */
int fc_sgd_step(const fc_model_t* m, float* params, const float* grads, float* momentum_buf, float lr, float momentum, int32_t nesterov,
                float weight_decay, int32_t step, void* stream);

/* ---- one iteration of FedavgClient.update's batch loop (fedavgclient.py:79-102), fully on device:
 * zero_grad -> forward -> criterion -> backward -> AdamW.step -> refresh compute weights.
 * labels: int64 [B] for 'cls' towers (uni-modal clients), NULL for img+txt retrieval (contrastive loss). */
int fc_client_step(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc,
                   const float* img, const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt,
                   const float* droppath, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float* lossbuf, void* workspace, size_t workspace_bytes, void* stream);

/* ---- FedproxClient.update (src/client/fedproxclient.py:29-92): the same loop with the proximal term
 *   loss += mu * 0.5 * sum over parameter tensors of ||param - global_param||_2        (fedproxclient.py:64-67; UN-squared norms)
 * global_params: flat float32 copy of the parameters taken when update() starts (copy.deepcopy(self.model), :33).
 * fc_prox_term adds 0.5*mu*(p-g)/||p-g|| per tensor to grads (0 where the norm is 0, like torch's norm backward) and the term's
 * value to lossbuf ([1] += v, [0] += v*B); fc_client_step_prox = fc_client_step with that term between backward and AdamW.step.
 * scratch: fc_prox_scratch_bytes(m) bytes of device memory (chunk tables + partial sums; fixed reduction order). */
size_t fc_prox_scratch_bytes(const fc_model_t* m);
int fc_prox_term(const fc_model_t* m, const float* params, const float* global_params, float mu, int32_t B, float* grads, float* lossbuf,
                 void* scratch, size_t scratch_bytes, void* stream);
int fc_client_step_prox(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc, const float* img,
                        const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt, const float* droppath, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int32_t step, float* lossbuf, void* workspace, size_t workspace_bytes,
                        void* stream, const float* global_params, float mu, void* prox_scratch, size_t prox_scratch_bytes);

/* ---- CreamFL (src/client/creamflclient.py, src/server/creamflserver.py).  The model work of its
 * distillation steps is fc_forward(feat_out) / fc_backward; these are the pieces around it.  Losses accumulate into lossbuf like
 * the criteria above ([1] += weight*L, [0] += weight*L*B) and write (accumulate = 0) or add (1) weight*dL/df into df. */
/* out[b] = table[idx[b]]: target features global_feature[d_idx] (creamflclient.py:160,169,203-204) */
int fc_gather_rows(const float* table, const int64_t* idx, int32_t B, int32_t D, float* out, void* stream);
/* logits = [f.target, f.old] / 0.5, label 0, cross-entropy summed over the B rows / rows_norm (creamflclient.py:175-186; the img+txt
 * client stacks both modalities' rows into one CE, :207-217: call twice with rows_norm = 2B) */
int fc_cream_moon_loss(const float* f, const float* target, const float* old_f, int32_t B, int32_t D, int32_t rows_norm, float weight,
                       float* lossbuf, float* df, int32_t accumulate, void* stream);
/* CE(f @ G^T / 0.5, labels), mean over B (creamflclient.py:165/173 + 181-182, :219-224).  G: [P, D]; scratch: fc_cream_inter_scratch_floats */
size_t fc_cream_inter_scratch_floats(int32_t B, int32_t P);
int fc_cream_inter_loss(const float* f, const float* G, const int64_t* labels, int32_t B, int32_t P, int32_t D, float weight, float* scratch,
                        size_t scratch_floats, float* lossbuf, float* df, int32_t accumulate, void* stream);
/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) over the trainable segments (creamflclient.py:232, creamflserver.py:334):
 * grads *= min(1, max_norm / (||grads||_2 + 1e-6)); total_norm_out (device float, may be NULL) receives the norm */
size_t fc_clip_scratch_bytes(const fc_model_t* m);
int fc_clip_grad_norm(const fc_model_t* m, float* grads, float max_norm, void* scratch, size_t scratch_bytes, float* total_norm_out, void* stream);
/* torch.optim.AdamW.step with torch's per-parameter bookkeeping: seg_steps (HOST int32[n_segments]) holds each segment's 1-based step
 * for this call; 0 = no gradient this step -> skipped, its step count does not advance (a classification head during feature
 * distillation).  wc (may be NULL): compute weights to refresh. */
int fc_adamw_step_segs(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                       float eps, float weight_decay, const int32_t* seg_steps, int32_t n_segments, void* wc, void* stream);
/* weight * nn.MSELoss()(out, target) over n elements (creamflserver.py:311-321) */
int fc_mse_loss_fwd_bwd(const float* out, const float* target, int64_t n, float weight, int32_t B, float* lossbuf, float* dout, void* stream);
/* server-side public-feature aggregation (creamflserver.py:373-405): w[i] = V_i.G_i - log sum_j exp(V_i.G_j) for one client's
 * features V [P, D] against the global features G [P, D] of the other modality; then out[i] = sum_c softmax_c(w[c][i]) * V_c[i]
 * (vecs_dev: device array of C device pointers; w: [C, P]) */
int fc_cream_logprob_diag(const float* V, const float* G, int32_t P, int32_t D, float* w, void* stream);
int fc_cream_combine(const float* const* vecs_dev, const float* w, int32_t C, int32_t P, int32_t D, float* out, void* stream);

/* ---- FedavgServer._aggregate blend (fedavgserver.py:656-664) in closed form, per state_dict key (segment):
 * out[seg_offset[s] + i] = w[s][0]*global[seg_offset[s] + i] + sum_j w[s][1+j] * client_bases[j][src_offset[s][j] + i].
 * Clients of other datasets hold the key at another offset of their own flat buffer (src_offset, < 0: key absent).
 * seg_weights is [n_segments, n_clients+1]; every table is device memory. */
int fc_aggregate_blend(float* out, const float* global, const float* const* client_bases, int32_t n_clients,
                       const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                       int32_t n_segments, void* stream);

/* ---- aggregation across processes (one rank per GPU) without PyTorch: an RCCL communicator behind an opaque handle.
 * Replaces the reference's single-process gather of client state_dicts into `self.global_model` (fedavgserver.py:566-589,
 * 812-819).  librccl.so is bound at run time (dlopen); single-process callers never need it.
 * fc_comm_unique_id: rank 0 fills a FC_COMM_ID_BYTES blob (ncclGetUniqueId) and hands it to every rank by its own means
 * (file, socket, MPI, torch.distributed.broadcast_object_list ...); fc_comm_create is collective over the ranks. */
#define FC_COMM_ID_BYTES 128
typedef struct fc_comm fc_comm_t;
int fc_comm_unique_id(void* id_out, size_t bytes);
int fc_comm_create(const void* id, size_t bytes, int32_t rank, int32_t world, fc_comm_t** out);
void fc_comm_destroy(fc_comm_t* comm);
int32_t fc_comm_rank(const fc_comm_t* comm);
int32_t fc_comm_world(const fc_comm_t* comm);
/* in-place sum over the ranks of a device buffer (ncclAllReduce over xGMI); no-op for comm == NULL / world 1 */
int fc_allreduce_sum(fc_comm_t* comm, float* buf, int64_t n, void* stream);
/* FedavgServer._aggregate (fedavgserver.py:591-668) for one global model, closed form:
 *   partial = [seg_weights[s][0] * global] + sum_j seg_weights[s][1+j] * client_bases[j][src_offset[s][j] ...]   (as fc_aggregate_blend)
 *   partial <- all-reduce(sum) over the ranks;   global[run] <- partial[run] for the planned ranges.
 * Each rank passes the clients it trained (weights of the others zero, src_offset < 0) and only rank 0 a non-zero global
 * weight, so the sum is the reference's blend.  client_bases is a HOST array of n_clients (<= 64) device pointers;
 * run_offset / run_numel are HOST arrays; the segment tables are device memory; partial is a numel-float device scratch.
 * comm == NULL or world 1: blends straight into `global` (no scratch traffic, no copy). */
int fc_aggregate(fc_comm_t* comm, float* global, float* partial, int64_t numel, const float* const* client_bases, int32_t n_clients,
                 const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                 int32_t n_segments, const int64_t* run_offset, const int64_t* run_numel, int32_t n_runs, void* stream);
/* the blend step of fc_aggregate alone (same arguments; `out` may be `global`): for callers that run the all-reduce themselves,
 * e.g. torch.distributed.all_reduce over RCCL */
int fc_aggregate_partial(float* out, const float* global, const float* const* client_bases, int32_t n_clients,
                         const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                         int32_t n_segments, void* stream);
/* The reference's loop itself, in its order and rounding: for j ascending, global <- global + fl32((client_j - global) * coef[s][j])
 * (fedavgserver.py:656-664; no fused multiply-add), skipping coef == 0 / src_offset < 0.  Bit-identical to the reference's fp32
 * result; used as the verification mode of the closed form.  client_bases: HOST array of device pointers (<= 64). */
int fc_aggregate_blend_seq(float* global, const float* const* client_bases, int32_t n_clients, const int64_t* seg_offset,
                           const int64_t* seg_numel, const int64_t* src_offset, const float* coef, int32_t n_segments, void* stream);
/* exact-order aggregation across ranks (one client per rank, ascending client id = rank order): ncclAllGather of every rank's
 * client buffer (slot_numel floats each, padded) into `gathered` [world * slot_numel], then fc_aggregate_blend_seq over the
 * slots.  src_offset / coef are [n_segments, world]. */
int fc_aggregate_exact(fc_comm_t* comm, float* global, const float* local_client, float* gathered, int64_t slot_numel,
                       const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* coef,
                       int32_t n_segments, void* stream);
/* outputs of the last forward on this workspace (logits of 'cls' towers / unit-norm features), for metric tracking
 * (mm.track(loss, outputs, targets), fedavgclient.py:102) */
int fc_copy_outputs(const fc_model_t* m, void* workspace, size_t workspace_bytes, float* out_img, float* out_txt, void* stream);
/* in-place per-segment scaling (pre-weighting before an RCCL all-reduce(sum)) */
int fc_scale_segments(float* buf, const int64_t* seg_offset, const int64_t* seg_numel, const float* seg_weight,
                      int32_t n_segments, void* stream);
/* FedavgClient.upload aux fold (fedavgclient.py:173-181): dst[weight] = W + A*s for every re-param linear, rest copied */
int fc_upload_fold(const fc_model_t* m, const float* params, float* dst, void* stream);

/* ---- input side: a batch of pre-decoded images kept as uint8 codes [n_images, C, HW] with the per-channel
 * table lut[C][256] = Normalize(ToTensor(u)) of the reference's transform chain (src/loaders/data.py:106-109; built by
 * fedcola_amd/loaders/cache.py, which verifies that the codes reproduce the dataset's own float tensors bit for bit) -> float32
 * [n_images, C, HW] on the device: dst = lut[c][src].  C <= 4. */
int fc_image_u8_to_f32(const uint8_t* src, const float* lut, float* dst, int64_t n_images, int32_t C, int32_t HW, void* stream);

/* ---- retrieval evaluation: COCOEvaluator.evaluate_recall (src/metrics/eval_coco.py:296-351,
 * ParallelMatMulModule :48-69) on device.  q [nq,d], g [ng,d]: float64 row-major features (the reference holds the extracted
 * features as float64, eval_coco.py:155-156, and multiplies them with Tensor.mm, :55); labels: int64 class ids
 * (image ids / class ids, eval_coco.py:166-173).  best_ranks[i] = position, in the descending-similarity order of the
 * gallery (equal similarities: lower gallery index first), of the best-placed gallery item whose label equals
 * q_labels[i] (eval_coco.py:331-334); -1 when no gallery item has that label (the reference raises there).
 * scratch: fp64 similarity rows of one query batch; fc_retrieval_scratch_bytes(nq_batch, ng) sizes it (any nq_batch >= 1). */
size_t fc_retrieval_scratch_bytes(int32_t nq_batch, int32_t ng);
int fc_retrieval_best_ranks(const double* q, const double* g, const int64_t* q_labels, const int64_t* g_labels, int32_t nq,
                            int32_t ng, int32_t d, void* scratch, size_t scratch_bytes, int64_t* best_ranks, void* stream);

/* ---- workspace inspection (tests: op-by-op parity of the bf16 path against the oracle on the library's own intermediates).
 * Byte offset / size of a saved tensor inside the workspace of (B, n_txt): tower 0 = image, 1 = text; names: per tower "x" / "gx"
 * (layer 0..depth: residual stream and its gradient), "patches", "dtok", "f", "out", "dout"; per layer "h1" "qkv" "o" "xmid" "h2"
 * "u" (bf16 mode: gelu'(u)) "gact" "gxmid" "gdm" "gda" "gdu" "gdqkv" "mean1" "rstd1" "mean2" "rstd2" "lse".
 * Activations are in the model's precision, statistics / features in fp32. */
int fc_workspace_tensor(const fc_model_t* m, int32_t B, int32_t n_txt, int32_t tower, int32_t layer, const char* name, size_t* offset,
                        size_t* bytes);

/* ---- individual kernels exposed for unit tests / reuse (dt: 0 = f32, 1 = bf16) */
int fc_k_layernorm_fwd(int32_t dt, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                       int32_t M, int32_t D, float eps, void* stream);
int fc_k_layernorm_bwd(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                       const void* res, void* dx, float* dg, float* db, int32_t M, int32_t D, void* stream);
/* the in-model form of the backward: per-block dgamma / dbeta partial rows in `partial` (fc_k_layernorm_partial_floats floats) and one
 * grouped reduction that ADDS into dg / db, instead of atomics (test entry point: synchronises the stream once) */
size_t fc_k_layernorm_partial_floats(int32_t M, int32_t D);
int fc_k_layernorm_bwd_partial(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                               const void* res, void* dx, float* dg, float* db, int32_t M, int32_t D, float* partial, void* stream);
/* kind: 0 NT (C=A.W^T, A[M,K], W[N,K]); 1 NN (C=A.W, A[M,K], W[K,N]); 2 TN (C=A^T.B, A[K,M], B[K,N]).
 * impl: 0 generic VALU, 1 MFMA bf16 (returns 1 when the shape is unsupported). dtC: type of C. bias may be NULL. */
int fc_k_gemm(int32_t impl, int32_t kind, int32_t dt_in, int32_t dt_out, const void* A, const void* B, void* C, int32_t M,
              int32_t N, int32_t K, const float* bias, int32_t gelu, void* stream);
/* the bf16 MFMA GEMM (kind 0 NT / 1 NN as above, bf16 in and out) with the fused epilogues of the model: C = A.B (+ bias) (+ res);
 * gelu_grad_out != NULL: C = gelu(A.B + bias) and gelu'(A.B + bias) is stored to gelu_grad_out (fc1 forward, mome.py:117-123);
 * mul_in != NULL: C = (A.B) * mul_in (the fc2 dX product times the saved gelu').  Unused pointers NULL. */
int fc_k_gemm_epi(int32_t kind, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, const float* bias, const void* res,
                  void* gelu_grad_out, const void* mul_in, void* stream);
/* impl: 0 generic, 1 MFMA flash (bf16, d=64) */
int fc_k_attention_fwd(int32_t impl, int32_t dt, const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H,
                       int32_t d, float scale, void* stream);
int fc_k_attention_bwd(int32_t impl, int32_t dt, const void* qkv, const void* o, const void* dout, const float* lse,
                       float* delta, void* dqkv, int32_t B, int32_t N, int32_t H, int32_t d, float scale, void* stream);
/* one weight-gradient problem of the backward through the grouped kernels: dW[out,in] (fp32) = dY[rows,out]^T . X[rows,in] (bf16
 * operands), db[out] = column sums of dY (may be NULL).  wide: 0 = 128x128 tiles; 1 / 2 = 128x384 tiles (need in % 384 == 0), 8 waves / 8 consumer + 2 LDS-DMA loader waves;
 * 3 = the fp32 mode's form: dY, X are FP32, products by split-operand MFMAs, the row reduction cut into slices added in fp64, and db is ADDED TO.
 * Test entry point: allocates its one-entry problem table and synchronises the stream. */
int fc_k_dw(int32_t wide, const void* dY, const void* X, float* dW, float* db, int32_t rows, int32_t out, int32_t in, void* stream);
int fc_k_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
               int32_t step, void* stream);
int fc_k_cast(int32_t dt_out, const float* src, void* dst, int64_t n, void* stream);
/* sims[nq,ng] = q . g^T in float64 (v_mfma_f64_16x16x4_f64) */
int fc_k_sim_f64(const double* q, const double* g, double* sims, int32_t nq, int32_t ng, int32_t d, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------------------
 * TOOLS BUILD ONLY (libfedcola_hip_probes.so: python -m fedcola_amd.build --probes, compiled with -DFC_PROBES; chosen by FC_PROBES_LIB=1 in
 * tools/ and in the tests that cover it).  The product library does not export what follows: experiments that are exact but measured no
 * faster in the client step (profiles/r05), kept buildable with their tests. */
#ifdef FC_PROBES
/* Run-time switches of the round-5 experiments.  Every option defaults to 0 = the forms that measure fastest in the ViT-S client step; the others are
 * the better kernels stand-alone and no faster in the step (profiles/r05): same results to the bit (FC_OPT_MLP_FUSED, FC_OPT_STEP_GRAPH)
 * or to the last fp32 bits of a sum whose order does not change (FC_OPT_GEMM_FORM never changes a result: same k order per element).
 *  FC_OPT_MLP_FUSED  1: fc1 -> GELU -> fc2 (and the backward mirror) as one launch per 64-row panel (bf16 mode, dim 384).  Call
 *                       fc_prepare_weights after switching it on (the fused kernels read packed weight streams behind the compute weights).
 *  FC_OPT_STEP_GRAPH 1: fc_client_step replays a captured HIP graph from the third step with the same buffers and shapes on.
 *  FC_OPT_GEMM_FORM  process-wide: tiles of the NT / NN launches that would fill less than half of the chip: 0 = 128 x 128, 64 = 64 x 128,
 *                       3 | 4 = 64 x 128 with a 3- / 4-stage staging ring for K >= 1024. */
enum { FC_OPT_MLP_FUSED = 1, FC_OPT_STEP_GRAPH = 2, FC_OPT_GEMM_FORM = 3 };
int fc_model_set_option(fc_model_t* m, int32_t option, int32_t value);
/* steps of this handle that ran as a hipGraphLaunch (FC_OPT_STEP_GRAPH) */
long fc_dbg_step_graph_hits(const fc_model_t* m);
/* the fused MLP of a Block (mome.py:117-123), bf16, D = 384, Hd % 128 == 0, one launch per call; returns 1 when the shape is not covered.
 * The weights are read from a copy packed in MFMA-fragment order: fc_k_mlp_pack writes both directions' streams (2 * D * Hd bf16 each) from
 * the row-major W1 [Hd,D] and W2 [D,Hd] (test entry point: allocates its one-entry job table and synchronises the stream).
 * bwd = 0: out[M,D] = res + rowscale[m / rows_per_sample] * (gelu(X.W1^T + b1).W2^T + b2)   (X: the LayerNorm-2 output; Wp: the forward stream);
 *          act <- gelu(u), gsave <- gelu'(u) (both [M,Hd], kept for the backward); rowscale may be NULL.
 * bwd = 1: out[M,D] = ((X.W2) * gsave).W1   (X: dY of the block's MLP branch [M,D]; Wp: the backward stream); gsave is READ,
 *          act <- (X.W2) * gsave (= du, the operand of fc1's weight gradient); b1, b2, res, rowscale unused (NULL). */
int fc_k_mlp_pack(const void* W1, const void* W2, void* stream_fwd, void* stream_bwd, int32_t D, int32_t Hd, void* stream);
int fc_k_mlp_fused(int32_t bwd, const void* X, const void* Wp, const float* b1, const float* b2, void* act, void* gsave, const void* res,
                   const float* rowscale, int32_t rows_per_sample, void* out, int32_t M, int32_t D, int32_t Hd, void* stream);
#endif /* FC_PROBES */

#ifdef __cplusplus
}
#endif
#endif
