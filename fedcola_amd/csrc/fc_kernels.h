// Host-side launchers of every HIP kernel on the hot path.  All are asynchronous on `stream`,
// return 0 on success (<0 + fc_last_error() otherwise), and never allocate.
// dt: element type of activations: FC_F32 (0) or FC_BF16 (1).
#pragma once
#include "fc_common.h"

enum { FC_F32 = 0, FC_BF16 = 1 };
static inline size_t fc_esize(int dt) { return dt == FC_BF16 ? 2 : 4; }
#define DISPATCH_DT(dt, ...)                 \
  if ((dt) == FC_F32) { typedef float T; __VA_ARGS__; } \
  else { typedef bf16_t T; __VA_ARGS__; }

// ---- layer norm (K3)
// Grouped forms (fc_ln.hip): up to two row sets -- the image and the text tower of one layer -- in ONE launch.  D % 8 == 0, D <= 1024,
// 16-byte aligned operands.  blk0 is filled in by the launcher.
struct FcLnFwdP { const void* x; void* y; const float* g; const float* b; float* mean; float* rstd; int M; int blk0; };
struct FcLnFwdArgs { FcLnFwdP p[2]; int nprob; int D; float eps; };
// dx = (res ? res : 0) + LNbwd(dy); dx_scaled (optional): a second output dx * rowscale[row / rows_per_sample] (drop-path: the next
// consumer's operand); partial: room for fc_layernorm_bwd_partial_blocks(M) * 2 * D floats, one [dgamma | dbeta] row per block, summed
// later by fc_ln_reduce_grouped over the queued FcLnReduce entries (room for ... * 2 * D fc_ln_part_t)
// The [dgamma | dbeta] partial-row buffers are sized for DOUBLES.  fp32 storage (the parity mode): products d * xhat are formed in fp32 (as
// the reference does) and summed in fp64 from the first add to the last -- no fp32 summation error, no dependence on how the rows are
// grouped.  bf16 storage: fp32 block sums in the same buffers (FcLnReduce.f64 = 0).  k_ln_reduce adds the rows of either type in fp64.
typedef double fc_ln_part_t;
struct FcLnBwdP {
  const void* dy; const void* x; const float* mean; const float* rstd; const float* g; const void* res; void* dx; void* dx_scaled;
  const float* rowscale; fc_ln_part_t* partial; int rows_per_sample; int M; int blk0; int pad;
};
struct FcLnBwdArgs { FcLnBwdP p[2]; int nprob; int D; 
#ifdef FC_PROBES
  int skip_partials;
#endif
};
int fc_layernorm_grouped_ok(int D);
int fc_layernorm_fwd_grouped(int dt, FcLnFwdArgs a, hipStream_t s);
int fc_layernorm_bwd_grouped(int dt, FcLnBwdArgs a, hipStream_t s);
// dgamma / dbeta = column sums of the partial rows (of up to two partial sets: two micro-batch chains); accumulate = 0: plain store (the
// gradient buffer need not be zeroed), 1: added to the existing value.  No atomics: two entries must not name the same dg / db.
struct FcLnReduce {
  const fc_ln_part_t* partial;
  const fc_ln_part_t* partial2;
  float* dg;
  float* db;
  int nblocks, nblocks2, D, accumulate;
  const fc_ln_part_t* partial3;   // a third micro-batch chain's partial rows (three-chain backward)
  int nblocks3, f64;       // f64: the partial rows hold doubles (fp32 storage) / floats (bf16 storage)
};
int fc_layernorm_bwd_partial_blocks(int M);
int fc_ln_reduce_grouped(const FcLnReduce* tab_dev, int n, int maxD, hipStream_t s);
// single-row-set forms.  fc_layernorm_bwd without `partial` accumulates dg / db with atomics (scalar kernel: tests and odd shapes); with
// `partial` it returns 1 and the caller must run fc_ln_reduce_grouped over its queued entries.
int fc_layernorm_fwd(int dt, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                     int M, int D, float eps, hipStream_t s);
int fc_layernorm_bwd(int dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                     const void* res, void* dx, float* dg, float* db, int M, int D, hipStream_t s, fc_ln_part_t* partial = nullptr,
                     void* dx_scaled = nullptr, const float* rowscale = nullptr, int rows_per_sample = 1);
int fc_rowscale(int dt, const void* src, void* dst, const float* rs, int rows_per_sample, int M, int D, hipStream_t s);

// ---- image embedding (K1): patches[B*np, C*P*P] (conv-weight order), cls rows, and backward pieces
int fc_patchify(int dt, const float* img, void* patches, int B, int C, int HW, int P, hipStream_t s);
int fc_cls_rows(int dt, const float* cls, const float* pos, void* x, int B, int N, int D, hipStream_t s);
int fc_img_embed_bwd(int dt, const void* dx, float* dpos, float* dcls, void* dtok, int B, int N, int D, hipStream_t s);

// ---- text embedding (K2): gather + add + LN(eps) ; backward re-gathers
int fc_txt_embed_fwd(int dt, const int64_t* ids, const float* word, const float* pos, const float* type, const float* g,
                     const float* b, void* y, float* mean, float* rstd, int B, int N, int D, int vocab, float eps, hipStream_t s);
int fc_txt_embed_bwd(int dt, const void* dy, const int64_t* ids, const float* word, const float* pos, const float* type,
                     const float* mean, const float* rstd, const float* g, float* dword, float* dpos, float* dtype,
                     float* dg, float* db, int B, int N, int D, int vocab, hipStream_t s);

// ---- final LN on cls rows + (optional) L2 normalise (K10)
int fc_head_fwd(int dt, const void* x, const float* g, const float* b, float* f, float* mean, float* rstd, float* nrm,
                float* out, int B, int N, int D, float eps, int normalize, hipStream_t s);
// din: d(out) if normalize else d(f); writes dx for ALL rows of [B,N,D] (zeros off the cls rows)
int fc_head_bwd(int dt, const float* din, const float* out, const float* nrm, int normalize, const void* x,
                const float* mean, const float* rstd, const float* g, void* dx, float* dg, float* db, int B, int N, int D,
                hipStream_t s);

// ---- losses (K11, K12).  lossbuf[0] += loss*B (running epoch sum), lossbuf[1] += loss (caller zeroes [1] per step)
int fc_contrastive_fwd_bwd(const float* a, const float* b, int B, int D, float tau, float* scratch /* 2*B*B+2*B floats */,
                           float* lossbuf, float* da, float* db, hipStream_t s);
int fc_ce_fwd_bwd(const float* logits, const int64_t* y, int B, int C, float* lossbuf, float* dlogits, hipStream_t s);

// ---- optimizer (K13): torch.optim.AdamW semantics; optionally refreshes the low-precision shadow and zeroes grads
int fc_adamw(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd,
             int step, void* shadow_bf16, int zero_grad, hipStream_t s);
// torch.optim.SGD.step over a flat range (momentum buffer `buf`; first = the parameters' first step: buf = g + wd p); shadow: bf16 copy of p or null
int fc_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, int nesterov, float wd, int first, void* shadow_bf16, hipStream_t s);
// AdamW constants of one optimizer step + the flat buffers they apply to (same element offsets in all of them).  Used by the
// stand-alone kernels and by the weight-gradient GEMM's fused epilogue (fc_gemm_tn_grouped with `opt`): one formula, one rounding
// sequence (fc_adamw_elem), so the fused and the separate optimizer produce the same bits from the same gradient.
struct FcAdamW {
  float* g0 = nullptr;         // gradients (base of the flat buffer; element index = pointer - g0)
  float* p = nullptr;          // parameters
  float* m = nullptr;          // exp_avg
  float* v = nullptr;          // exp_avg_sq
  bf16_t* shadow = nullptr;    // bf16 compute weights for the next step (may be null)
  float decay, beta1, beta2, eps, step_size, inv_bc2_sqrt;
  // non-null: {decay, step_size, inv_bc2_sqrt} are read from this device array instead (a captured step graph bakes its kernel arguments in;
  // the three values that change from step to step are written there ahead of every replay: fc_adamw_set_dyn)
  const float* dyn = nullptr;
};
int fc_adamw_set_dyn(float* dyn_dev, const FcAdamW& o, hipStream_t s);
FcAdamW fc_adamw_consts(float lr, float beta1, float beta2, float eps, float wd, int step);
#ifdef __HIPCC__
__device__ __forceinline__ FcAdamW fc_adamw_resolve(FcAdamW o) {
  if (o.dyn) { o.decay = o.dyn[0]; o.step_size = o.dyn[1]; o.inv_bc2_sqrt = o.dyn[2]; }
  return o;
}
__device__ __forceinline__ void fc_adamw_elem(float& p, float g, float& m, float& v, float decay, float beta1, float beta2, float eps, float step_size,
                                              float inv_bc2_sqrt) {
#pragma clang fp contract(off)   // every product and sum rounded on its own, wherever this is inlined: the same bits from every kernel
  const float pk = p * decay;
  m = m + (g - m) * (1.0f - beta1);            // exp_avg.lerp_(grad, 1-beta1)
  v = v * beta2 + (1.0f - beta2) * g * g;
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p = pk - step_size * (m / denom);
}
#endif
// AdamW over a table of chunks (<= FC_PROX_CHUNK consecutive elements each; FcProxChunk.seg unused): everything the fused epilogue
// does not cover, in one launch
struct FcProxChunk;
int fc_add_chunks(float* dst, const float* src, const FcProxChunk* chunks_dev, int nchunks, hipStream_t s);   // dst[c] += src[c] over the chunks
int fc_adamw_chunks(const FcProxChunk* chunks_dev, int nchunks, const FcAdamW& o, hipStream_t s);
int fc_cast(int dt_out, const float* src, void* dst, size_t n, hipStream_t s);
// W_eff = W + s*A (K9) into dst (type dt)
int fc_reparam_fold(int dt, const float* W, const float* A, const float* scale, void* dst, size_t n, hipStream_t s);
// given dW_eff in gW: ds += <gW, A>; gA = s*gW (if gA != null)
int fc_reparam_grad(const float* gW, const float* A, const float* scale, float* ds, float* gA, size_t n, hipStream_t s);
// every re-param linear of a model in ONE launch each (a client step otherwise spends ~100 small launches on them in its serial tail)
struct FcReparam { int64_t w, aux, scale, n; int32_t aux_trainable, pad; };
int fc_reparam_grad_grouped(const FcReparam* tab_dev, int nlin, const float* params, float* grads, hipStream_t s);
int fc_reparam_fold_grouped(int dt, const FcReparam* tab_dev, int nlin, const float* params, void* wc, hipStream_t s);

// ---- column sum (bias grads): db[n] (+)= sum_m dy[m,n]
int fc_colsum(int dt, const void* dy, float* db, int M, int N, int accumulate, hipStream_t s);

// ---- aggregation (K14): out[seg] = w[seg][0]*g[seg] + sum_j w[seg][1+j] * (bases[j] + src_off[seg][j])[0..len)
// (host-computed closed-form weights; src_off < 0: client j does not hold that key).  All tables are device arrays.
int fc_blend_segments(float* out, const float* g, const float* const* bases, int m, const int64_t* seg_off, const int64_t* seg_len,
                      const int64_t* src_off, const float* seg_w, int nseg, hipStream_t s);
// FedProx proximal term over parameter tensors (fedproxclient.py:64-67).  One FcProxChunk per <= FC_PROX_CHUNK consecutive elements
// of a parameter tensor; seg = dense index of the tensor among those that take part.
#define FC_PROX_CHUNK 16384
struct FcProxChunk { int64_t offset; int32_t n; int32_t seg; };
// partial[c] = sum((p-g)^2) of chunk c; norm[s] = sqrt(sum of its chunks' partials) (chunks of a tensor are consecutive: first[s] ..
// first[s+1]); loss accumulators += 0.5*mu*sum(norm) (lossbuf[1]) and that times B (lossbuf[0]); grads += 0.5*mu*(p-g)/norm (0 if norm == 0).
// Fixed reduction order throughout: bitwise reproducible.
int fc_prox_term_impl(const float* p, const float* g, const FcProxChunk* chunks, int nchunks, const int32_t* first, int nseg, float* partial,
                      float* norm, float mu, int B, float* grads, float* lossbuf, hipStream_t s);
int fc_scale_segments_impl(float* buf, const int64_t* seg_off, const int64_t* seg_len, const float* seg_w, int nseg, hipStream_t s);

// ---- generic strided GEMM (any shape; fp32 accumulate).  C[m,n] = epi( sum_k A(m,k)*B(k,n) )
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn], C row-major with leading dim ldc.
int fc_gemm_generic(int dtA, int dtB, int dtC, const void* A, long sam, long sak, const void* Bm, long sbk, long sbn,
                    void* C, long ldc, int M, int N, int K, const GemmEpi& epi, hipStream_t s);

// ---- MFMA bf16 GEMMs (gfx950): returns 1 if the shape is not supported by the fast path (caller falls to generic)
// kinds: NT: C[M,N] = A[M,K] . W[N,K]^T ; NN: C[M,N] = A[M,K] . W[K,N] ; TN: C[M,N] = A[K,M]^T . B[K,N]
enum { FC_GEMM_NT = 0, FC_GEMM_NN = 1, FC_GEMM_TN = 2 };
int fc_gemm_mfma(int kind, int dtC, const bf16_t* A, long lda, const bf16_t* Bm, long ldb, void* C, long ldc, int M, int N, int K,
                 const GemmEpi& epi, hipStream_t s);
// ---- fp32 GEMMs as three bf16 MFMA products of split operands (fc_gemm_x3.hip): the GEMM of the fp32 mode; 1 = shape not covered
int fc_gemm_x3(int kind, const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K, const GemmEpi& epi, hipStream_t s);
// ---- fp32 attention as v_mfma_f32_16x16x4_f32 chains (fc_attn_f32.hip): head_dim 64, N <= 224; 1 = shape not covered
int fc_attn_f32_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, int d, float scale, hipStream_t s);
int fc_attn_f32_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* delta /* [B, H, N] scratch */, float* dqkv, int B, int N, int H, int d,
                    float scale, hipStream_t s);
// the fp32 mode's weight gradients: dW[out,in] = dY[rows,out]^T . X[rows,in] (stored) and db += column sums of dY, reduction cut into slices
int fc_dw_x3(const float* dY, const float* X, float* dW, float* db, int rows, int out, int in, hipStream_t s);
// the same kernel over one or two problems that share N, K and the epilogue kind (image + text tower of a layer in one launch);
// returns 1 when not covered: the caller then launches the problems one by one
struct GemmProb { const bf16_t* A; const bf16_t* B; void* C; long lda, ldb, ldc; int M; GemmEpi e; };
struct GemmGroup { GemmProb p[2]; int N, K, tiles_n, tiles0, ntiles; };
int fc_gemm_mfma_grouped(int kind, int dtC, GemmGroup g, int nprob, hipStream_t s);
#ifdef FC_PROBES
void fc_gemm_set_form(int form);      // process-wide: 0 | 64 | 3 | 4 (fc_mfma.hip; tools build)
#endif

// ---- fused MLP (fc_mlp.hip): fc1 -> GELU -> fc2 per 64-row panel, or its backward mirror; D = 384 only.  The weights come from streams packed
// in MFMA-fragment order (fc_mlp_pack: one launch for a table of {W1, W2, forward stream, backward stream} jobs, fc_mlp_pack_elems bf16 each).
size_t fc_mlp_pack_elems(int D, int Hd);
int fc_mlp_fused_ok(int D, int Hd);
int fc_mlp_pack(const void* jobs_dev, int njobs, int D, int Hd, hipStream_t s);
struct FcMlpPackJob { const bf16_t* W1; const bf16_t* W2; bf16_t* fwd; bf16_t* bwd; };
int fc_mlp_fused(int bwd, const void* X, const void* Wp, const float* b1, const float* b2, void* act, void* gsave, const void* res, const float* rowscale,
                 int rps, void* out, int M, int D, int Hd, hipStream_t s);          // 1 = shape not covered

// grouped weight-gradient GEMM: C[M,N] (fp32) = A[K,M]^T . B[K,N], bias_grad[M] = column sums of A (may be null)
struct FcTnProblem {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  float* bias_grad;
  int lda, ldb, ldc;
  int M, N, K;
  int tile_start, tiles_n;
};
int fc_gemm_tn_grouped_supported(const FcTnProblem& p);
// 128 x 384 tiles for problems whose N (the linear's `in`) is a multiple of 384 (fc_gemm_dw.hip): same table format, own tile numbering
int fc_gemm_dw_wide_supported(const FcTnProblem& p);
int fc_gemm_dw_wide_tiles(const FcTnProblem& p, int* tiles_n);
// form: 1 = 8 waves that load, compute and store; 2 = 8 consumer + 2 loader (LDS-DMA) waves; 0 = FC_DW_WIDE / default
int fc_gemm_dw_wide(const FcTnProblem* probs_dev, int nprob, int total_tiles, hipStream_t s, const struct FcAdamW* opt = nullptr, int form = 0);
// opt != null: the epilogue also takes the AdamW step of every element it produced (dW tiles and the bias gradients), so the
// optimizer needs no pass of its own over the linears' weights; the gradient is still stored
int fc_gemm_tn_grouped(const FcTnProblem* probs_dev, int nprob, int total_tiles, hipStream_t s, const FcAdamW* opt = nullptr);

// ---- attention (K5).  qkv: [B,N,3,H,d] row-major (= the qkv GEMM output [B*N, 3*H*d]); o: [B,N,H*d]; lse: [B,H,N]
int fc_attn_fwd_generic(int dt, const void* qkv, void* o, float* lse, int B, int N, int H, int d, float scale, hipStream_t s);
int fc_attn_bwd_generic(int dt, const void* qkv, const void* o, const void* dout, const float* lse, float* delta,
                        void* dqkv, int B, int N, int H, int d, float scale, hipStream_t s);
int fc_attn_fwd_mfma(const bf16_t* qkv, bf16_t* o, float* lse, int B, int N, int H, int d, float scale, hipStream_t s);
int fc_attn_bwd_mfma(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, float* delta, bf16_t* dqkv,
                     int B, int N, int H, int d, float scale, hipStream_t s);
