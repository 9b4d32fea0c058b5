// CreamFL device work: the loss / clipping / optimizer pieces around fc_forward + fc_backward that
// src/client/creamflclient.py:133-233 and src/server/creamflserver.py:294-336, 372-407 run as eager PyTorch ops.
// All reductions use a fixed order (bitwise reproducible); every tensor here is small (B <= a few hundred rows, P public
// samples, D features), so the kernels are one-block-per-row fp32 code -- the heavy work stays in the model forward/backward.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/fedcola_hip.h"
#include "fc_kernels.h"

#define CREAM_TEMP 0.5f

__device__ inline float cream_block_sum(float v, float* red) {   // 256 threads, fixed tree
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}
__device__ inline float cream_block_max(float v, float* red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return r;
}

// ---- out[b] = table[idx[b]]  (target features global[d_idx], creamflclient.py:160,169,203-204)
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ table, const int64_t* __restrict__ idx, int D, float* __restrict__ out) {
  const int b = blockIdx.x;
  const float* src = table + (size_t)idx[b] * D;
  for (int i = threadIdx.x; i < D; i += 256) out[(size_t)b * D + i] = src[i];
}

// ---- logits = [f.t, f.o] / 0.5, label 0 (creamflclient.py:175-186; img+txt: :199-217 with the 2B stacked rows as denominator)
// one block: wave w walks rows w, w+4, ...; the row losses meet in a fixed order
__global__ void __launch_bounds__(256) k_cream_moon(const float* __restrict__ f, const float* __restrict__ t, const float* __restrict__ o, int B, int D,
                                                    float inv_rows, float weight, float* __restrict__ lossbuf, float* __restrict__ df, int accumulate) {
  __shared__ float red[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int b = wave; b < B; b += 4) {
    const float *fr = f + (size_t)b * D, *tr = t + (size_t)b * D, *orow = o + (size_t)b * D;
    float sp = 0.f, sn = 0.f;
    for (int i = lane; i < D; i += 64) { sp += fr[i] * tr[i]; sn += fr[i] * orow[i]; }
    const float zp = wave_sum(sp) / CREAM_TEMP, zn = wave_sum(sn) / CREAM_TEMP;
    const float mx = fmaxf(zp, zn);
    const float ep = expf(zp - mx), en = expf(zn - mx);
    const float lse = mx + logf(ep + en);
    if (lane == 0) acc += lse - zp;
    const float p0 = ep / (ep + en), p1 = en / (ep + en);
    const float dpos = weight * (p0 - 1.f) * inv_rows / CREAM_TEMP, dneg = weight * p1 * inv_rows / CREAM_TEMP;
    for (int i = lane; i < D; i += 64) {
      const float g = dpos * tr[i] + dneg * orow[i];
      if (accumulate) df[(size_t)b * D + i] += g; else df[(size_t)b * D + i] = g;
    }
  }
  const float tot = cream_block_sum(lane == 0 ? acc : 0.f, red);
  if (threadIdx.x == 0) {
    const float v = weight * tot * inv_rows;
    lossbuf[1] += v;
    lossbuf[0] += v * (float)B;
  }
}

// ---- CE(f @ G^T / 0.5, labels) (creamflclient.py:165/173 + 181-182; img+txt :219-224)
// block b: logits row -> stable log-sum-exp -> row loss and (softmax - onehot) * weight / (B * 0.5) left in z
__global__ void __launch_bounds__(256) k_cream_inter_rows(const float* __restrict__ f, const float* __restrict__ G, const int64_t* __restrict__ labels,
                                                          int B, int P, int D, float weight, float* __restrict__ z, float* __restrict__ row_loss) {
  __shared__ float red[4];
  extern __shared__ float fs[];        // the feature row
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < D; i += 256) fs[i] = f[(size_t)b * D + i];
  __syncthreads();
  float* zr = z + (size_t)b * P;
  float mx = -3.0e38f;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float* g = G + (size_t)p * D;
    float s = 0.f;
    for (int i = 0; i < D; ++i) s += fs[i] * g[i];
    s /= CREAM_TEMP;
    zr[p] = s;
    mx = fmaxf(mx, s);
  }
  mx = cream_block_max(mx, red);
  float se = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) se += expf(zr[p] - mx);
  se = cream_block_sum(se, red);
  const float lse = mx + logf(se);
  const int lab = (int)labels[b];
  if (threadIdx.x == 0) row_loss[b] = lse - zr[lab];
  __syncthreads();
  const float sc = weight / ((float)B * CREAM_TEMP);
  for (int p = threadIdx.x; p < P; p += 256) zr[p] = (expf(zr[p] - lse) - (p == lab ? 1.f : 0.f)) * sc;
}
__global__ void __launch_bounds__(256) k_cream_sum_rows(const float* __restrict__ row_loss, int B, float scale, float* __restrict__ lossbuf) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) acc += row_loss[b];
  const float tot = cream_block_sum(acc, red);
  if (threadIdx.x == 0) {
    const float v = tot * scale;
    lossbuf[1] += v;
    lossbuf[0] += v * (float)B;
  }
}
// df[b, :] (+)= dz[b, :] @ G
__global__ void __launch_bounds__(256) k_cream_inter_df(const float* __restrict__ dz, const float* __restrict__ G, int P, int D, float* __restrict__ df,
                                                        int accumulate) {
  const int b = blockIdx.x;
  const float* zr = dz + (size_t)b * P;
  for (int i = threadIdx.x; i < D; i += 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += zr[p] * G[(size_t)p * D + i];
    if (accumulate) df[(size_t)b * D + i] += s; else df[(size_t)b * D + i] = s;
  }
}

// ---- nn.MSELoss() * weight (creamflserver.py:311-321): one block, fixed order
__global__ void __launch_bounds__(256) k_mse(const float* __restrict__ out, const float* __restrict__ tgt, long n, float weight, int B,
                                             float* __restrict__ lossbuf, float* __restrict__ dout) {
  __shared__ float red[4];
  float acc = 0.f;
  const float sc = weight * 2.f / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float d = out[i] - tgt[i];
    acc += d * d;
    dout[i] = sc * d;
  }
  const float tot = cream_block_sum(acc, red);
  if (threadIdx.x == 0) {
    const float v = weight * tot / (float)n;
    lossbuf[1] += v;
    lossbuf[0] += v * (float)B;
  }
}

// ---- server feature aggregation (creamflserver.py:373-405)
// w[i] = V_i . G_i - log sum_j exp(V_i . G_j)
__global__ void __launch_bounds__(256) k_cream_logprob_diag(const float* __restrict__ V, const float* __restrict__ G, int P, int D, float* __restrict__ w) {
  __shared__ float red[4];
  extern __shared__ float vs[];
  const int i = blockIdx.x;
  for (int k = threadIdx.x; k < D; k += 256) vs[k] = V[(size_t)i * D + k];
  __syncthreads();
  float mx = -3.0e38f, diag = 0.f;
  for (int j = threadIdx.x; j < P; j += 256) {
    const float* g = G + (size_t)j * D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += vs[k] * g[k];
    if (j == i) diag = s;
    mx = fmaxf(mx, s);
  }
  mx = cream_block_max(mx, red);
  float se = 0.f;
  for (int j = threadIdx.x; j < P; j += 256) {
    const float* g = G + (size_t)j * D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += vs[k] * g[k];
    se += expf(s - mx);
  }
  se = cream_block_sum(se, red);
  diag = cream_block_sum(diag, red);          // exactly one thread holds the diagonal term
  if (threadIdx.x == 0) w[i] = diag - (mx + logf(se));
}
// out[i] = sum_c softmax_c(w[:, i]) * V_c[i]
__global__ void __launch_bounds__(256) k_cream_combine(const float* const* __restrict__ vecs, const float* __restrict__ w, int C, int P, int D,
                                                       float* __restrict__ out) {
  const int i = blockIdx.x;
  float mx = -3.0e38f;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, w[(size_t)c * P + i]);
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(w[(size_t)c * P + i] - mx);
  for (int k = threadIdx.x; k < D; k += 256) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += (expf(w[(size_t)c * P + i] - mx) / se) * vecs[c][(size_t)i * D + k];
    out[(size_t)i * D + k] = acc;
  }
}

// ---- clip_grad_norm_: chunk partial sums of g^2 -> total norm -> scale (chunk table = the FedProx one: trainable tensors)
__global__ void __launch_bounds__(256) k_sq_partial(const float* __restrict__ g, const FcProxChunk* __restrict__ chunks, float* __restrict__ partial) {
  __shared__ float red[4];
  const FcProxChunk c = chunks[blockIdx.x];
  float acc = 0.f;
  for (int i = threadIdx.x; i < c.n; i += 256) { const float v = g[c.offset + i]; acc += v * v; }
  const float r = cream_block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_clip_coef(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ coef_norm) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const float tot = sqrtf(cream_block_sum(acc, red));
  if (threadIdx.x == 0) {
    coef_norm[0] = fminf(max_norm / (tot + 1e-6f), 1.0f);    // torch: clip_coef clamped to 1
    coef_norm[1] = tot;
  }
}
__global__ void __launch_bounds__(256) k_scale_chunks(float* __restrict__ g, const FcProxChunk* __restrict__ chunks, const float* __restrict__ coef) {
  const FcProxChunk c = chunks[blockIdx.x];
  const float s = coef[0];
  if (s == 1.0f) return;
  for (int i = threadIdx.x; i < c.n; i += 256) g[c.offset + i] *= s;
}

int fc_clip_impl(float* grads, const FcProxChunk* chunks, int nchunks, float* partial, float* coef_norm, float max_norm, hipStream_t s) {
  if (nchunks <= 0) return 0;
  hipLaunchKernelGGL(k_sq_partial, dim3(nchunks), dim3(256), 0, s, grads, chunks, partial);
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(256), 0, s, partial, nchunks, max_norm, coef_norm);
  hipLaunchKernelGGL(k_scale_chunks, dim3(nchunks), dim3(256), 0, s, grads, chunks, coef_norm);
  FC_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- C ABI
extern "C" int fc_gather_rows(const float* table, const int64_t* idx, int32_t B, int32_t D, float* out, void* stream) {
  FC_REQUIRE(table && idx && out && B >= 0 && D > 0, "gather_rows: bad argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_gather_rows, dim3(B), dim3(256), 0, (hipStream_t)stream, table, idx, D, out);
  FC_LAUNCH_CHECK();
  return 0;
}
extern "C" int fc_cream_moon_loss(const float* f, const float* target, const float* old, int32_t B, int32_t D, int32_t rows_norm, float weight,
                                  float* lossbuf, float* df, int32_t accumulate, void* stream) {
  FC_REQUIRE(f && target && old && lossbuf && df && B > 0 && D > 0 && rows_norm > 0, "cream_moon_loss: bad argument");
  hipLaunchKernelGGL(k_cream_moon, dim3(1), dim3(256), 0, (hipStream_t)stream, f, target, old, B, D, 1.0f / (float)rows_norm, weight, lossbuf, df,
                     accumulate);
  FC_LAUNCH_CHECK();
  return 0;
}
extern "C" size_t fc_cream_inter_scratch_floats(int32_t B, int32_t P) { return (size_t)(B > 0 ? B : 0) * (size_t)(P > 0 ? P : 0) + (size_t)(B > 0 ? B : 0); }
extern "C" int fc_cream_inter_loss(const float* f, const float* G, const int64_t* labels, int32_t B, int32_t P, int32_t D, float weight, float* scratch,
                                   size_t scratch_floats, float* lossbuf, float* df, int32_t accumulate, void* stream) {
  FC_REQUIRE(f && G && labels && scratch && lossbuf && df && B > 0 && P > 0 && D > 0, "cream_inter_loss: bad argument");
  FC_REQUIRE(scratch_floats >= fc_cream_inter_scratch_floats(B, P), "cream_inter_loss: scratch too small");
  FC_REQUIRE(D * sizeof(float) <= 48 * 1024, "cream_inter_loss: feature width %d too large", D);
  hipStream_t s = (hipStream_t)stream;
  float* row_loss = scratch + (size_t)B * P;
  hipLaunchKernelGGL(k_cream_inter_rows, dim3(B), dim3(256), sizeof(float) * D, s, f, G, labels, B, P, D, weight, scratch, row_loss);
  hipLaunchKernelGGL(k_cream_sum_rows, dim3(1), dim3(256), 0, s, row_loss, B, weight / (float)B, lossbuf);
  hipLaunchKernelGGL(k_cream_inter_df, dim3(B), dim3(256), 0, s, scratch, G, P, D, df, accumulate);
  FC_LAUNCH_CHECK();
  return 0;
}
extern "C" int fc_mse_loss_fwd_bwd(const float* out, const float* target, int64_t n, float weight, int32_t B, float* lossbuf, float* dout, void* stream) {
  FC_REQUIRE(out && target && lossbuf && dout && n > 0, "mse_loss: bad argument");
  hipLaunchKernelGGL(k_mse, dim3(1), dim3(256), 0, (hipStream_t)stream, out, target, (long)n, weight, B, lossbuf, dout);
  FC_LAUNCH_CHECK();
  return 0;
}
extern "C" int fc_cream_logprob_diag(const float* V, const float* G, int32_t P, int32_t D, float* w, void* stream) {
  FC_REQUIRE(V && G && w && P > 0 && D > 0 && D * sizeof(float) <= 48 * 1024, "cream_logprob_diag: bad argument");
  hipLaunchKernelGGL(k_cream_logprob_diag, dim3(P), dim3(256), sizeof(float) * D, (hipStream_t)stream, V, G, P, D, w);
  FC_LAUNCH_CHECK();
  return 0;
}
extern "C" int fc_cream_combine(const float* const* vecs_dev, const float* w, int32_t C, int32_t P, int32_t D, float* out, void* stream) {
  FC_REQUIRE(vecs_dev && w && out && C > 0 && P > 0 && D > 0, "cream_combine: bad argument");
  hipLaunchKernelGGL(k_cream_combine, dim3(P), dim3(256), 0, (hipStream_t)stream, vecs_dev, w, C, P, D, out);
  FC_LAUNCH_CHECK();
  return 0;
}
