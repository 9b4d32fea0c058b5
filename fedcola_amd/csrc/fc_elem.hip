// HBM-bound kernels of the FedCola client step for gfx950: LayerNorm, embeddings, head, losses, AdamW,
// re-param fold, column sums, aggregation blend.  One wave (64 lanes) per row for the row-wise ops, wave-shuffle
// reductions, 16-byte vector accesses for the flat-buffer ops.
// Reference semantics: see fc_kernels.h / DESIGN.md (each launcher cites the reference lines it replaces).
#include <stdlib.h>
#include <string.h>

#include "fc_kernels.h"


// ======================================================================== LayerNorm (mome.py:199,203,215,751; K3)
template <typename T>
__global__ void __launch_bounds__(256) k_ln_fwd(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= M) return;
  const T* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += Io<T>::ld(xr, i);
  float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { float d = Io<T>::ld(xr, i) - mu; q += d * d; }
  float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  T* yr = y + (size_t)row * D;
  for (int i = lane; i < D; i += 64) Io<T>::st(yr, i, (Io<T>::ld(xr, i) - mu) * rs * g[i] + b[i]);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}


// ---- 16-byte vector forms (D % 8 == 0, D <= 1024): a lane owns chunks of 8 consecutive columns; one pass over HBM
template <typename T> struct V8;
template <> struct V8<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct V8<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    uint4 u = *(const uint4*)p;
    const bf16_t* h = (const bf16_t*)&u;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(h[i]);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    *(uint4*)p = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
  }
};
static bool ln_vec_ok(const void* a, const void* b, const void* c, const void* d, int D) {
  return fc_layernorm_grouped_ok(D) && !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15);
}

int fc_layernorm_fwd(int dt, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int M, int D,
                     float eps, hipStream_t s) {
  if (FC_ABLATED("ln")) return 0;
  if (M <= 0) return 0;
  if (ln_vec_ok(x, y, g, b, D)) {
    FcLnFwdArgs a{};
    a.p[0] = FcLnFwdP{x, y, g, b, mean, rstd, M, 0};
    a.nprob = 1; a.D = D; a.eps = eps;
    return fc_layernorm_fwd_grouped(dt, a, s);
  }
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_ln_fwd<T>, dim3(fc_cdiv(M, 4)), dim3(256), 0, s, (const T*)x, g, b, (T*)y, mean, rstd, M, D, eps));
  FC_LAUNCH_CHECK();
  return 0;
}

// rows handled per block in the backward (dg/db partials are reduced in-block, then one atomic per column per block)
#define LNB_ROWS 32
#define LNB_MAXV 16  // D <= 64*16
template <typename T>
__global__ void __launch_bounds__(256) k_ln_bwd(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, const float* __restrict__ g, const T* res, T* dx,
                                                float* __restrict__ dg, float* __restrict__ db, int M, int D) {
  __shared__ float red[2][4][64 * LNB_MAXV > 1024 ? 1024 : 64 * LNB_MAXV];  // [dg|db][wave][col]
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float ag[LNB_MAXV], ab[LNB_MAXV];
#pragma unroll
  for (int t = 0; t < LNB_MAXV; ++t) { ag[t] = 0.f; ab[t] = 0.f; }
  int row0 = blockIdx.x * LNB_ROWS;
  for (int r = wave; r < LNB_ROWS; r += 4) {
    int row = row0 + r;
    if (row >= M) break;
    const T* dyr = dy + (size_t)row * D;
    const T* xr = x + (size_t)row * D;
    float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < LNB_MAXV; ++t) {
      int i = lane + 64 * t;
      if (i < D) {
        float d = Io<T>::ld(dyr, i), xh = (Io<T>::ld(xr, i) - mu) * rs;
        float dxh = d * g[i];
        s1 += dxh; s2 += dxh * xh;
        ag[t] += d * xh; ab[t] += d;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    T* dxr = dx + (size_t)row * D;
#pragma unroll
    for (int t = 0; t < LNB_MAXV; ++t) {
      int i = lane + 64 * t;
      if (i < D) {
        float d = Io<T>::ld(dyr, i), xh = (Io<T>::ld(xr, i) - mu) * rs;
        float v = rs * (d * g[i] - s1 - xh * s2);
        if (res) v += Io<T>::ld(res + (size_t)row * D, i);
        Io<T>::st(dxr, i, v);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < LNB_MAXV; ++t) {
    int i = lane + 64 * t;
    if (i < D) { red[0][wave][i] = ag[t]; red[1][wave][i] = ab[t]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    float a = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    float c = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
    atomicAdd(dg + i, a);
    atomicAdd(db + i, c);
  }
}


// plain per-row scaled copy (drop-path backward where no LayerNorm backward produces the operand): 16-byte accesses when possible
template <typename T>
__global__ void __launch_bounds__(256) k_rowscale_v(const T* __restrict__ src, T* __restrict__ dst, const float* __restrict__ rs, int rows_per_sample,
                                                    size_t n8, int D8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    V8<T>::ld(src + i * 8, v);
    const float sc = rs[(i / D8) / rows_per_sample];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= sc;
    V8<T>::st(dst + i * 8, v);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k_rowscale_s(const T* __restrict__ src, T* __restrict__ dst, const float* __restrict__ rs, int rows_per_sample,
                                                    size_t n, int D) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    Io<T>::st(dst, i, Io<T>::ld(src, i) * rs[(i / D) / rows_per_sample]);
}
int fc_rowscale(int dt, const void* src, void* dst, const float* rs, int rows_per_sample, int M, int D, hipStream_t s) {
  const size_t n = (size_t)M * D;
  if (n == 0) return 0;
  if ((D & 7) == 0 && !(((uintptr_t)src | (uintptr_t)dst) & 15)) {
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    DISPATCH_DT(dt, hipLaunchKernelGGL(k_rowscale_v<T>, dim3(grid), dim3(256), 0, s, (const T*)src, (T*)dst, rs, rows_per_sample, n8, D / 8));
  } else {
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    DISPATCH_DT(dt, hipLaunchKernelGGL(k_rowscale_s<T>, dim3(grid), dim3(256), 0, s, (const T*)src, (T*)dst, rs, rows_per_sample, n, D));
  }
  FC_LAUNCH_CHECK();
  return 0;
}

int fc_layernorm_bwd(int dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g, const void* res,
                     void* dx, float* dg, float* db, int M, int D, hipStream_t s, fc_ln_part_t* partial, void* dx_scaled, const float* rowscale,
                     int rows_per_sample) {
  if (FC_ABLATED("ln")) return 0;
  if (M <= 0) return 0;
  if (partial && ln_vec_ok(dy, x, dx, res, D) && !(((uintptr_t)g | (uintptr_t)partial | (uintptr_t)dx_scaled) & 15)) {
    FcLnBwdArgs a{};
    a.p[0] = FcLnBwdP{dy, x, mean, rstd, g, res, dx, dx_scaled, rowscale, partial, rows_per_sample, M, 0, 0};
    a.nprob = 1; a.D = D;
    FC_TRY(fc_layernorm_bwd_grouped(dt, a, s));
    return 1;   // dg/db are pending in `partial` (caller queues the grouped reduction)
  }
  FC_REQUIRE(D <= 64 * LNB_MAXV, "layernorm_bwd: D=%d > %d unsupported", D, 64 * LNB_MAXV);
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_ln_bwd<T>, dim3(fc_cdiv(M, LNB_ROWS)), dim3(256), 0, s, (const T*)dy, (const T*)x, mean, rstd, g,
                                     (const T*)res, (T*)dx, dg, db, M, D));
  FC_LAUNCH_CHECK();
  if (dx_scaled) return fc_rowscale(dt, dx, dx_scaled, rowscale, rows_per_sample, M, D, s);
  return 0;
}

// ======================================================================== image embedding (mome.py:597-611, 260-266; K1)
// patches[(b*np + py*gw + px), c*P*P + ph*P + pw] = img[b, c, py*P+ph, px*P+pw]
template <typename T>
__global__ void __launch_bounds__(256) k_patchify(const float* __restrict__ img, T* __restrict__ out, int B, int C, int HW, int P) {
  int gw = HW / P, np = gw * gw, K = C * P * P;
  size_t total = (size_t)B * np * K;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    int k = (int)(idx % K);
    size_t r = idx / K;
    int p = (int)(r % np), b = (int)(r / np);
    int c = k / (P * P), ph = (k / P) % P, pw = k % P;
    int py = p / gw, px = p % gw;
    Io<T>::st(out, idx, img[(((size_t)b * C + c) * HW + (py * P + ph)) * HW + px * P + pw]);
  }
}
// 8 pixels of one patch row per thread (P % 8 == 0): two 16-B reads, one 16-B (bf16) store, output order = thread order.  The scalar form
// above moved 19 MB in 35 us at the head of every forward image chain.
template <typename T>
__global__ void __launch_bounds__(256) k_patchify_v(const float* __restrict__ img, T* __restrict__ out, int B, int C, int HW, int P) {
  const int gw = HW / P, np = gw * gw, K = C * P * P, K8 = K >> 3, PP = P * P;
  const size_t units = (size_t)B * np * K8;
  for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
    const int k = (int)(u % K8) * 8;
    const size_t r = u / K8;
    const int p = (int)(r % np), b = (int)(r / np);
    const int c = k / PP, ph = (k / P) % P, pw = k % P;
    const int py = p / gw, px = p % gw;
    float v[8];
    V8<float>::ld(img + (((size_t)b * C + c) * HW + (py * P + ph)) * HW + px * P + pw, v);
    V8<T>::st(out + u * 8, v);
  }
}
int fc_patchify(int dt, const float* img, void* patches, int B, int C, int HW, int P, hipStream_t s) {
  size_t total = (size_t)B * (HW / P) * (HW / P) * C * P * P;
  if ((P & 7) == 0 && (HW & 3) == 0 && !(((uintptr_t)img | (uintptr_t)patches) & 15)) {
    const size_t units = total / 8;
    int grid = (int)((units + 255) / 256 > 16384 ? 16384 : (units + 255) / 256);
    DISPATCH_DT(dt, hipLaunchKernelGGL(k_patchify_v<T>, dim3(grid), dim3(256), 0, s, img, (T*)patches, B, C, HW, P));
    FC_LAUNCH_CHECK();
    return 0;
  }
  int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_patchify<T>, dim3(grid), dim3(256), 0, s, img, (T*)patches, B, C, HW, P));
  FC_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ void k_cls_rows(const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ x, int B, int N, int D) {
  int b = blockIdx.x;
  for (int i = threadIdx.x; i < D; i += blockDim.x) Io<T>::st(x + (size_t)b * N * D, i, cls[i] + pos[i]);
}
int fc_cls_rows(int dt, const float* cls, const float* pos, void* x, int B, int N, int D, hipStream_t s) {
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_cls_rows<T>, dim3(B), dim3(128), 0, s, cls, pos, (T*)x, B, N, D));
  FC_LAUNCH_CHECK();
  return 0;
}

// dpos[n,i] += sum_b dx[b,n,i]; dcls[i] += sum_b dx[b,0,i]; dtok[b*(N-1)+n-1, i] = dx[b,n,i]
template <typename T>
__global__ void __launch_bounds__(256) k_img_embed_bwd(const T* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dcls,
                                                       T* __restrict__ dtok, int B, int N, int D) {
  int n = blockIdx.x;
  for (int i = threadIdx.x; i < D; i += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      T raw = dx[((size_t)b * N + n) * D + i];
      s += Io<T>::ld(&raw, 0);
      if (n > 0) dtok[((size_t)b * (N - 1) + (n - 1)) * D + i] = raw;
    }
    atomicAdd(dpos + (size_t)n * D + i, s);
    if (n == 0) atomicAdd(dcls + i, s);
  }
}
// 16-byte vector form (D % 8 == 0, D <= 1024): block = (token position n, 16 images), 16 lanes per row with 8 columns each (as in
// fc_ln.hip), so that the rows spread over N x B/16 blocks and every access is a whole 16-B chunk; the block's column sums go through
// registers and LDS to one atomic per column per block.  (The one-block-per-position form above walks the batch serially with 2-byte
// loads: 29 us at the end of each backward image chain, in front of the last weight-gradient chunk.)
template <typename T, int CH>
__global__ void __launch_bounds__(256) k_img_embed_bwd_v(const T* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dcls,
                                                         T* __restrict__ dtok, int B, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) float red_img[];   // [4 waves][D]
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, slot = lane >> 4;
  const int nc = D >> 3;
  const int bi = blockIdx.y * 16 + wave * 4 + slot;
  const bool live = bi < B;
  const T* src = dx + ((size_t)(live ? bi : 0) * N + n) * D;
  float acc[CH][8];
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
    if (c < nc && live) {
      V8<T>::ld(src + c * 8, acc[t]);
      if (n > 0) V8<T>::st(dtok + ((size_t)bi * (N - 1) + (n - 1)) * D + c * 8, acc[t]);   // same values back: a row copy
    }
  }
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[t][i] += __shfl_xor(acc[t][i], 16, 64); acc[t][i] += __shfl_xor(acc[t][i], 32, 64); }   // the wave's four rows
    if (slot == 0 && c < nc) {
      *(float4*)(red_img + wave * D + c * 8) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      *(float4*)(red_img + wave * D + c * 8 + 4) = make_float4(acc[t][4], acc[t][5], acc[t][6], acc[t][7]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    const float v = red_img[i] + red_img[D + i] + red_img[2 * D + i] + red_img[3 * D + i];
    atomicAdd(dpos + (size_t)n * D + i, v);
    if (n == 0) atomicAdd(dcls + i, v);
  }
}
int fc_img_embed_bwd(int dt, const void* dx, float* dpos, float* dcls, void* dtok, int B, int N, int D, hipStream_t s) {
  if ((D & 7) == 0 && D <= 1024 && !(((uintptr_t)dx | (uintptr_t)dtok) & 15)) {
    const int ch = fc_cdiv(D / 8, 16);
    const size_t lds = sizeof(float) * 4 * D;
#define GO(CHN) DISPATCH_DT(dt, hipLaunchKernelGGL((k_img_embed_bwd_v<T, CHN>), dim3(N, fc_cdiv(B, 16)), dim3(256), lds, s, (const T*)dx, dpos, dcls, (T*)dtok, B, N, D))
    switch (ch) {
      case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
      case 5: case 6: GO(6); break; default: GO(8); break;
    }
#undef GO
    FC_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_img_embed_bwd<T>, dim3(N), dim3(256), 0, s, (const T*)dx, dpos, dcls, (T*)dtok, B, N, D));
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== text embedding (mome.py:632-639 -> HF BertEmbeddings; K2)
__device__ inline long clamp_id(long id, int vocab) { return id < 0 ? 0 : (id >= vocab ? vocab - 1 : id); }

template <typename T>
__global__ void __launch_bounds__(256) k_txt_embed_fwd(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                       const float* __restrict__ pos, const float* __restrict__ type,
                                                       const float* __restrict__ g, const float* __restrict__ b, T* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int B, int N, int D, int vocab,
                                                       float eps) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * N) return;
  int n = row % N;
  const float* w = word + (size_t)clamp_id(ids[row], vocab) * D;
  const float* pp = pos + (size_t)n * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += w[i] + type[i] + pp[i];
  float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { float d = w[i] + type[i] + pp[i] - mu; q += d * d; }
  float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int i = lane; i < D; i += 64) Io<T>::st(y + (size_t)row * D, i, (w[i] + type[i] + pp[i] - mu) * rs * g[i] + b[i]);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
int fc_txt_embed_fwd(int dt, const int64_t* ids, const float* word, const float* pos, const float* type, const float* g, const float* b,
                     void* y, float* mean, float* rstd, int B, int N, int D, int vocab, float eps, hipStream_t s) {
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_txt_embed_fwd<T>, dim3(fc_cdiv((long)B * N, 4)), dim3(256), 0, s, ids, word, pos, type, g, b, (T*)y,
                                     mean, rstd, B, N, D, vocab, eps));
  FC_LAUNCH_CHECK();
  return 0;
}

// de = LNbwd(dy) per row; dword[id] += de (id != 0: padding_idx row gets no grad); dpos[n] += de; dtype[0] += de; dg/db.
// One block per token position n: its 4 waves walk the batch, keep dpos[n] / dtype / dgamma / dbeta partial sums in
// registers, reduce across the waves in LDS, and issue one add per column per block (only the word-row scatter stays a
// per-row atomic).
#define TEB_MAXV 16
template <typename T>
__global__ void __launch_bounds__(256) k_txt_embed_bwd(const T* __restrict__ dy, const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                       const float* __restrict__ pos, const float* __restrict__ type,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ g, float* dword, float* dpos, float* dtype, float* dg,
                                                       float* db, int B, int N, int D, int vocab) {
  __shared__ float red[3][4][64 * TEB_MAXV > 1024 ? 1024 : 64 * TEB_MAXV];
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* pp = pos + (size_t)n * D;
  float ade[TEB_MAXV], agm[TEB_MAXV], abt[TEB_MAXV];
#pragma unroll
  for (int t = 0; t < TEB_MAXV; ++t) { ade[t] = 0.f; agm[t] = 0.f; abt[t] = 0.f; }
  for (int bi = wave; bi < B; bi += 4) {
    const int row = bi * N + n;
    const long id = clamp_id(ids[row], vocab);
    const float* w = word + (size_t)id * D;
    const float mu = mean[row], rs = rstd[row];
    const T* dyr = dy + (size_t)row * D;
    float d[TEB_MAXV], xh[TEB_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < TEB_MAXV; ++t) {
      int i = lane + 64 * t;
      d[t] = 0.f; xh[t] = 0.f;
      if (i < D) {
        d[t] = Io<T>::ld(dyr, i);
        xh[t] = (w[i] + type[i] + pp[i] - mu) * rs;
        s1 += d[t] * g[i]; s2 += d[t] * g[i] * xh[t];
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int t = 0; t < TEB_MAXV; ++t) {
      int i = lane + 64 * t;
      if (i < D) {
        float de = rs * (d[t] * g[i] - s1 - xh[t] * s2);
        if (id != 0) atomicAdd(dword + (size_t)id * D + i, de);
        ade[t] += de; agm[t] += d[t] * xh[t]; abt[t] += d[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TEB_MAXV; ++t) {
    int i = lane + 64 * t;
    if (i < D) { red[0][wave][i] = ade[t]; red[1][wave][i] = agm[t]; red[2][wave][i] = abt[t]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    float de = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    atomicAdd(dpos + (size_t)n * D + i, de);
    atomicAdd(dtype + i, de);
    atomicAdd(dg + i, red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i]);
    atomicAdd(db + i, red[2][0][i] + red[2][1][i] + red[2][2][i] + red[2][3][i]);
  }
}
// 16-byte vector form (D % 8 == 0, D <= 1024): block = (token position n, 16 samples), 16 lanes per row as in fc_ln.hip, so that B x N
// rows spread over N x B/16 blocks (the one-block-per-position form above runs on 32 of 256 CUs at N = 32: 139 us at B = 64).
// dpos[n] / dtype / dgamma / dbeta: summed over the block's rows in registers and LDS, one atomic per column per block.
template <typename T, int CH>
__global__ void __launch_bounds__(256) k_txt_embed_bwd_v(const T* __restrict__ dy, const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ g, float* dword, float* dpos, float* dtype, float* dg,
                                                         float* db, int B, int N, int D, int vocab) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];   // [4 waves][3][D] | [4 waves][4 rows][144]: transposition scratch of the word-gradient atomics
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, slot = lane >> 4;
  const int nc = D >> 3;
  const int bi = blockIdx.y * 16 + wave * 4 + slot;
  const bool live = bi < B;
  const int row = (live ? bi : B - 1) * N + n;
  const long id = clamp_id(ids[row], vocab);
  const float mu = mean[row], rs = rstd[row];
  float d[CH][8], xh[CH][8], gg[CH][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[t][i] = 0.f; xh[t][i] = 0.f; gg[t][i] = 0.f; }
    if (c < nc) {
      float w[8], ty[8], pp[8];
      V8<T>::ld(dy + (size_t)row * D + c * 8, d[t]);
      V8<float>::ld(word + (size_t)id * D + c * 8, w);
      V8<float>::ld(type + c * 8, ty);
      V8<float>::ld(pos + (size_t)n * D + c * 8, pp);
      V8<float>::ld(g + c * 8, gg[t]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xh[t][i] = (w[i] + ty[i] + pp[i] - mu) * rs;
        s1 += d[t][i] * gg[t][i]; s2 += d[t][i] * gg[t][i] * xh[t][i];
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  s1 /= (float)D; s2 /= (float)D;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
    float de[8], gm[8], bt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      de[i] = live ? rs * (d[t][i] * gg[t][i] - s1 - xh[t][i] * s2) : 0.f;
      gm[i] = live ? d[t][i] * xh[t][i] : 0.f;
      bt[i] = live ? d[t][i] : 0.f;
    }
    // word-table gradient: a lane holds 8 CONSECUTIVE columns of its row, which as atomics would touch four 128-B lines per instruction and
    // row (16 lanes x 32-B stride).  The 16 lanes of a row swap through LDS so that instruction j adds columns {lane + 16 j}: one 64-B run
    // per row and instruction, a quarter of the line requests at the L2 atomic units (12 of the kernel's 26 us were these atomics).
    {
      float* tr = red_dyn + 12 * D + (wave * 4 + slot) * 144;          // 128 columns of this row's chunk, rows 144 floats apart (banks)
      *(float4*)(tr + 8 * sub) = make_float4(de[0], de[1], de[2], de[3]);
      *(float4*)(tr + 8 * sub + 4) = make_float4(de[4], de[5], de[6], de[7]);
      // same wave, same row group: the LDS write above is visible to the reads below once lgkmcnt drains (no barrier needed)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (live && id != 0) {
        float* wrow = dword + (size_t)id * D + 128 * t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = sub + 16 * j;
          if (128 * t + col < D) atomicAdd(wrow + col, tr[col]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {    // the wave's four rows
      de[i] += __shfl_xor(de[i], 16, 64); de[i] += __shfl_xor(de[i], 32, 64);
      gm[i] += __shfl_xor(gm[i], 16, 64); gm[i] += __shfl_xor(gm[i], 32, 64);
      bt[i] += __shfl_xor(bt[i], 16, 64); bt[i] += __shfl_xor(bt[i], 32, 64);
    }
    if (slot == 0 && c < nc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red_dyn[(wave * 3 + 0) * D + c * 8 + i] = de[i];
        red_dyn[(wave * 3 + 1) * D + c * 8 + i] = gm[i];
        red_dyn[(wave * 3 + 2) * D + c * 8 + i] = bt[i];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    const float de = red_dyn[(0 * 3 + 0) * D + i] + red_dyn[(1 * 3 + 0) * D + i] + red_dyn[(2 * 3 + 0) * D + i] + red_dyn[(3 * 3 + 0) * D + i];
    atomicAdd(dpos + (size_t)n * D + i, de);
    atomicAdd(dtype + i, de);
    atomicAdd(dg + i, red_dyn[(0 * 3 + 1) * D + i] + red_dyn[(1 * 3 + 1) * D + i] + red_dyn[(2 * 3 + 1) * D + i] + red_dyn[(3 * 3 + 1) * D + i]);
    atomicAdd(db + i, red_dyn[(0 * 3 + 2) * D + i] + red_dyn[(1 * 3 + 2) * D + i] + red_dyn[(2 * 3 + 2) * D + i] + red_dyn[(3 * 3 + 2) * D + i]);
  }
}
int fc_txt_embed_bwd(int dt, const void* dy, const int64_t* ids, const float* word, const float* pos, const float* type, const float* mean,
                     const float* rstd, const float* g, float* dword, float* dpos, float* dtype, float* dg, float* db, int B, int N, int D,
                     int vocab, hipStream_t s) {
  if ((D & 7) == 0 && D <= 1024 && !(((uintptr_t)dy | (uintptr_t)word | (uintptr_t)pos | (uintptr_t)type | (uintptr_t)g) & 15)) {
    const int ch = fc_cdiv(D / 8, 16);
    const size_t lds = sizeof(float) * (12 * D + 16 * 144);
#define GO(CHN) DISPATCH_DT(dt, hipLaunchKernelGGL((k_txt_embed_bwd_v<T, CHN>), dim3(N, fc_cdiv(B, 16)), dim3(256), lds, s, (const T*)dy, ids, word, pos, type, \
                                                   mean, rstd, g, dword, dpos, dtype, dg, db, B, N, D, vocab))
    switch (ch) {
      case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
      case 5: case 6: GO(6); break; default: GO(8); break;
    }
#undef GO
    FC_LAUNCH_CHECK();
    return 0;
  }
  FC_REQUIRE(D <= 64 * TEB_MAXV, "txt_embed_bwd: D=%d > %d unsupported", D, 64 * TEB_MAXV);
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_txt_embed_bwd<T>, dim3(N), dim3(256), 0, s, (const T*)dy, ids, word, pos, type, mean, rstd, g, dword,
                                     dpos, dtype, dg, db, B, N, D, vocab));
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== head: final LN on cls rows (+ L2 normalise) (mome.py:906,915,657-659; K10)
template <typename T>
__global__ void __launch_bounds__(64) k_head_fwd(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, float* f,
                                                 float* mean, float* rstd, float* nrm, float* out, int B, int N, int D, float eps,
                                                 int normalize) {
  int bi = blockIdx.x, lane = threadIdx.x;
  const T* xr = x + (size_t)bi * N * D;
  if (D <= 1024) {                                      // the row lives in registers: one memory round trip instead of three dependent ones
    float xv[16], gv[16], bv[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      xv[k] = i < D ? Io<T>::ld(xr, i) : 0.f;
      gv[k] = i < D ? g[i] : 0.f;
      bv[k] = i < D ? b[i] : 0.f;
      s += xv[k];
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const float d = lane + 64 * k < D ? xv[k] - mu : 0.f; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    float n2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      xv[k] = (xv[k] - mu) * rs * gv[k] + bv[k];
      if (lane + 64 * k < D) n2 += xv[k] * xv[k];
    }
    const float nr = sqrtf(wave_sum(n2));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      if (i < D) {
        f[(size_t)bi * D + i] = xv[k];
        if (normalize) out[(size_t)bi * D + i] = xv[k] / nr;
      }
    }
    if (lane == 0) { mean[bi] = mu; rstd[bi] = rs; nrm[bi] = nr; }
    return;
  }
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += Io<T>::ld(xr, i);
  float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { float d = Io<T>::ld(xr, i) - mu; q += d * d; }
  float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float n2 = 0.f;
  for (int i = lane; i < D; i += 64) {
    float v = (Io<T>::ld(xr, i) - mu) * rs * g[i] + b[i];
    f[(size_t)bi * D + i] = v;
    n2 += v * v;
  }
  float nr = sqrtf(wave_sum(n2));
  if (normalize)
    for (int i = lane; i < D; i += 64) out[(size_t)bi * D + i] = f[(size_t)bi * D + i] / nr;
  if (lane == 0) { mean[bi] = mu; rstd[bi] = rs; nrm[bi] = nr; }
}
int fc_head_fwd(int dt, const void* x, const float* g, const float* b, float* f, float* mean, float* rstd, float* nrm, float* out, int B,
                int N, int D, float eps, int normalize, hipStream_t s) {
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_head_fwd<T>, dim3(B), dim3(64), 0, s, (const T*)x, g, b, f, mean, rstd, nrm, out, B, N, D, eps, normalize));
  FC_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_head_bwd(const float* __restrict__ din, const float* __restrict__ out, const float* __restrict__ nrm,
                                                  int normalize, const T* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ g, T* __restrict__ dx, float* dg,
                                                  float* db, int B, int N, int D) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * N) return;
  T* dxr = dx + (size_t)row * D;
  if (row % N != 0) {                                   // no gradient reaches the non-cls rows through the head: zero rows, 16 bytes per lane
    if (((D * sizeof(T)) & 15) == 0 && (((uintptr_t)dx) & 15) == 0) {
      const int n16 = (int)(D * sizeof(T) / 16);
      for (int c = lane; c < n16; c += 64) ((uint4*)dxr)[c] = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (int i = lane; i < D; i += 64) Io<T>::st(dxr, i, 0.f);
    }
    return;
  }
  int bi = row / N;
  const float* dr = din + (size_t)bi * D;
  float dot = 0.f, inv = 1.f;
  if (normalize) {
    const float* o = out + (size_t)bi * D;
    for (int i = lane; i < D; i += 64) dot += o[i] * dr[i];
    dot = wave_sum(dot);
    inv = 1.0f / nrm[bi];
  }
  const T* xr = x + (size_t)row * D;
  float mu = mean[bi], rs = rstd[bi];
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < D; i += 64) {
    float df = normalize ? (dr[i] - out[(size_t)bi * D + i] * dot) * inv : dr[i];
    float xh = (Io<T>::ld(xr, i) - mu) * rs;
    s1 += df * g[i]; s2 += df * g[i] * xh;
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
  for (int i = lane; i < D; i += 64) {
    float df = normalize ? (dr[i] - out[(size_t)bi * D + i] * dot) * inv : dr[i];
    float xh = (Io<T>::ld(xr, i) - mu) * rs;
    Io<T>::st(dxr, i, rs * (df * g[i] - s1 - xh * s2));
    atomicAdd(dg + i, df * xh);
    atomicAdd(db + i, df);
  }
}
int fc_head_bwd(int dt, const float* din, const float* out, const float* nrm, int normalize, const void* x, const float* mean,
                const float* rstd, const float* g, void* dx, float* dg, float* db, int B, int N, int D, hipStream_t s) {
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_head_bwd<T>, dim3(fc_cdiv((long)B * N, 4)), dim3(256), 0, s, din, out, nrm, normalize, (const T*)x,
                                     mean, rstd, g, (T*)dx, dg, db, B, N, D));
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== contrastive loss (torchmultimodal ContrastiveLossWithTemperature; K11)
// scratch: L[B*B] | dL[B*B] | lse_r[B] | lse_c[B]
__global__ void __launch_bounds__(64) k_con_lse(const float* __restrict__ L, float* lse_r, float* lse_c, float* lossbuf, int B) {
  int idx = blockIdx.x, lane = threadIdx.x;
  bool isrow = idx < B;
  int i = isrow ? idx : idx - B;
  float mx = -INFINITY;
  for (int j = lane; j < B; j += 64) mx = fmaxf(mx, isrow ? L[(size_t)i * B + j] : L[(size_t)j * B + i]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < B; j += 64) s += expf((isrow ? L[(size_t)i * B + j] : L[(size_t)j * B + i]) - mx);
  s = wave_sum(s);
  float lse = mx + logf(s);
  if (lane == 0) {
    (isrow ? lse_r : lse_c)[i] = lse;
    float contrib = 0.5f * (lse - L[(size_t)i * B + i]) / (float)B;
    atomicAdd(lossbuf + 1, contrib);
    atomicAdd(lossbuf + 0, contrib * (float)B);
  }
}
// L[i][j] = tau * <a_i, b_j>: one block per row i, thread t -> column t>>2, quarter t&3 of the feature range
__global__ void __launch_bounds__(256) k_con_logits(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ L, int B, int D, float tau) {
  const int i = blockIdx.x, q = threadIdx.x & 3;
  const float* ar = a + (size_t)i * D;
  for (int j = threadIdx.x >> 2; j < B; j += 64) {
    const float* br = b + (size_t)j * D;
    float s = 0.f;
    for (int d = q; d < D; d += 4) s = fmaf(ar[d], br[d], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q == 0) L[(size_t)i * B + j] = tau * s;
  }
}
// blocks [0,B): da_i = tau * sum_j dL_ij b_j ; blocks [B,2B): db_j = tau * sum_i dL_ij a_i ; dL formed on the fly
__global__ void __launch_bounds__(256) k_con_grads(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ L,
                                                   const float* __restrict__ lse_r, const float* __restrict__ lse_c, float* __restrict__ da,
                                                   float* __restrict__ db, int B, int D, float tau) {
  extern __shared__ float w[];   // dL row / column
  const bool isrow = blockIdx.x < (unsigned)B;
  const int r = isrow ? blockIdx.x : blockIdx.x - B;
  for (int o = threadIdx.x; o < B; o += 256) {
    int i = isrow ? r : o, j = isrow ? o : r;
    float l = L[(size_t)i * B + j];
    w[o] = (0.5f / (float)B) * (expf(l - lse_r[i]) + expf(l - lse_c[j]) - (i == j ? 2.0f : 0.0f)) * tau;
  }
  __syncthreads();
  const float* src = isrow ? b : a;
  float* dst = (isrow ? da : db) + (size_t)r * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int o = 0; o < B; ++o) s = fmaf(w[o], src[(size_t)o * D + d], s);
    dst[d] = s;
  }
}
// ---- two-launch form (round 6; D % 4 == 0): the loss phase sits between the forward's join and the backward's fork with the chip empty, so
// every launch and every boundary in it is step time.  Launch 1, block i: row i of L AND column i of L (row i of L^T: tau <a_j, b_i>) from the
// inputs alone, so that both log-sum-exps of index i and its loss terms are local to the block -- no pass over L in between.  Launch 2: the
// gradients, with the column case reading L^T rows (contiguous) and the D range cut into float4 columns x o-groups so that all loads of a thread
// are independent.  Same sums as the three-kernel form up to the order of the fp32 adds.
__global__ void __launch_bounds__(256) k_con_rowcol(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ L, float* __restrict__ LT,
                                                    float* __restrict__ lse_r, float* __restrict__ lse_c, float* lossbuf, int B, int D, float tau) {
  extern __shared__ float sh[];      // row[B] | col[B]
  float* rowv = sh; float* colv = rowv + B;
  const int i = blockIdx.x, t = threadIdx.x, q = t & 3, nc = D >> 2;
  const float4* ai = (const float4*)(a + (size_t)i * D);      // the same address for every lane of a quarter: one broadcast request
  const float4* bi = (const float4*)(b + (size_t)i * D);
  for (int j0 = 0; j0 < B; j0 += 64) {
    const int j = j0 + (t >> 2);
    float s1 = 0.f, s2 = 0.f;
    if (j < B) {
      const float4* bj = (const float4*)(b + (size_t)j * D);
      const float4* aj = (const float4*)(a + (size_t)j * D);
      int c = q;
      for (; c + 28 < nc; c += 32) {      // eight chunks of each of the four rows per batch: 32 independent 16-byte loads, one round trip, then their FMAs
        float4 x[8], y[8], u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = bj[c + 4 * k]; y[k] = aj[c + 4 * k]; u[k] = ai[c + 4 * k]; v[k] = bi[c + 4 * k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s1 = fmaf(u[k].x, x[k].x, s1); s1 = fmaf(u[k].y, x[k].y, s1); s1 = fmaf(u[k].z, x[k].z, s1); s1 = fmaf(u[k].w, x[k].w, s1);
          s2 = fmaf(y[k].x, v[k].x, s2); s2 = fmaf(y[k].y, v[k].y, s2); s2 = fmaf(y[k].z, v[k].z, s2); s2 = fmaf(y[k].w, v[k].w, s2);
        }
      }
      for (; c < nc; c += 4) {
        const float4 x = bj[c], y = aj[c], u = ai[c], v = bi[c];
        s1 = fmaf(u.x, x.x, s1); s1 = fmaf(u.y, x.y, s1); s1 = fmaf(u.z, x.z, s1); s1 = fmaf(u.w, x.w, s1);
        s2 = fmaf(y.x, v.x, s2); s2 = fmaf(y.y, v.y, s2); s2 = fmaf(y.z, v.z, s2); s2 = fmaf(y.w, v.w, s2);
      }
    }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    if (q == 0 && j < B) {
      const float l1 = tau * s1, l2 = tau * s2;
      rowv[j] = l1; colv[j] = l2;
      L[(size_t)i * B + j] = l1; LT[(size_t)i * B + j] = l2;
    }
  }
  __syncthreads();
  const int wave = t >> 6, lane = t & 63;
  if (wave < 2) {
    const float* v = wave == 0 ? rowv : colv;
    float mx = -INFINITY;
    for (int j = lane; j < B; j += 64) mx = fmaxf(mx, v[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < B; j += 64) sum += expf(v[j] - mx);
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    if (lane == 0) {
      (wave == 0 ? lse_r : lse_c)[i] = lse;
      const float contrib = 0.5f * (lse - rowv[i]) / (float)B;      // L_ii is the diagonal of both
      atomicAdd(lossbuf + 1, contrib);
      atomicAdd(lossbuf + 0, contrib * (float)B);
    }
  }
}
// blocks [0, B): da_r = sum_j dL_rj b_j ; blocks [B, 2B): db_r = sum_i dL_ir a_i ; dL formed on the fly (times tau).  G o-groups x nc float4 columns.
__global__ void __launch_bounds__(256) k_con_grads2(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ L,
                                                    const float* __restrict__ LT, const float* __restrict__ lse_r, const float* __restrict__ lse_c,
                                                    float* __restrict__ da, float* __restrict__ db, int B, int D, float tau, int G) {
  extern __shared__ float sh[];      // w[B] | partial sums [G][D]
  float* w = sh; float4* part = (float4*)(sh + ((B + 3) & ~3));
  const bool isrow = blockIdx.x < (unsigned)B;
  const int r = isrow ? blockIdx.x : blockIdx.x - B, t = threadIdx.x, nc = D >> 2;
  const float c0 = (0.5f / (float)B) * tau;
  for (int o = t; o < B; o += 256) {
    const float l = (isrow ? L : LT)[(size_t)r * B + o];
    const float e1 = isrow ? expf(l - lse_r[r]) : expf(l - lse_r[o]);
    const float e2 = isrow ? expf(l - lse_c[o]) : expf(l - lse_c[r]);
    w[o] = c0 * (e1 + e2 - (o == r ? 2.0f : 0.0f));
  }
  __syncthreads();
  const float4* src = (const float4*)(isrow ? b : a);
  for (int c0i = 0; c0i < nc; c0i += 256 / G) {      // (one pass when G * nc <= 256: D = 384 -> nc = 96, G = 2)
    const int g = t / (256 / G), c = c0i + t % (256 / G);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nc && c - c0i < 256 / G) {
      int o = g;
      for (; o + 7 * G < B; o += 8 * G) {      // eight independent loads in flight
        float4 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = src[(size_t)(o + k * G) * nc + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float wo = w[o + k * G];
          acc.x = fmaf(wo, x[k].x, acc.x); acc.y = fmaf(wo, x[k].y, acc.y); acc.z = fmaf(wo, x[k].z, acc.z); acc.w = fmaf(wo, x[k].w, acc.w);
        }
      }
      for (; o < B; o += G) {
        const float4 x = src[(size_t)o * nc + c];
        const float wo = w[o];
        acc.x = fmaf(wo, x.x, acc.x); acc.y = fmaf(wo, x.y, acc.y); acc.z = fmaf(wo, x.z, acc.z); acc.w = fmaf(wo, x.w, acc.w);
      }
      part[g * nc + c] = acc;
    }
  }
  __syncthreads();
  float4* dst = (float4*)((isrow ? da : db) + (size_t)r * D);
  for (int c = t; c < nc; c += 256) {
    float4 s4 = part[c];
    for (int g = 1; g < G; ++g) { const float4 p = part[g * nc + c]; s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w; }
    dst[c] = s4;
  }
}
int fc_contrastive_fwd_bwd(const float* a, const float* b, int B, int D, float tau, float* scratch, float* lossbuf, float* da, float* db,
                           hipStream_t s) {
  float* L = scratch;
  float* LT = scratch + (size_t)B * B;
  float* lse_r = scratch + 2 * (size_t)B * B;
  float* lse_c = lse_r + B;
  const bool al = !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)da | (uintptr_t)db) & 15);
  if ((D & 3) == 0 && al && (size_t)(2 * D + 2 * B) * sizeof(float) <= 60000) {
    const int nc = D >> 2;
    int G = 256 / nc;                                   // o-groups of the gradient launch: as many as fit the block, at most 8
    G = G < 1 ? 1 : (G > 8 ? 8 : G);
    while (256 % G) --G;
    const size_t lds2 = sizeof(float) * (((size_t)B + 3) & ~(size_t)3) + sizeof(float) * (size_t)G * D;
    if (lds2 <= 60000) {
      hipLaunchKernelGGL(k_con_rowcol, dim3(B), dim3(256), sizeof(float) * (2 * B), s, a, b, L, LT, lse_r, lse_c, lossbuf, B, D, tau);
      hipLaunchKernelGGL(k_con_grads2, dim3(2 * B), dim3(256), lds2, s, a, b, L, LT, lse_r, lse_c, da, db, B, D, tau, G);
      FC_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL(k_con_logits, dim3(B), dim3(256), 0, s, a, b, L, B, D, tau);
  hipLaunchKernelGGL(k_con_lse, dim3(2 * B), dim3(64), 0, s, L, lse_r, lse_c, lossbuf, B);
  hipLaunchKernelGGL(k_con_grads, dim3(2 * B), dim3(256), sizeof(float) * B, s, a, b, L, lse_r, lse_c, da, db, B, D, tau);
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== cross entropy (fedavgclient.py:85,90; K12)
__global__ void __launch_bounds__(64) k_ce(const float* __restrict__ logits, const int64_t* __restrict__ y, int B, int C, float* lossbuf,
                                           float* __restrict__ dl) {
  int i = blockIdx.x, lane = threadIdx.x;
  const float* lr = logits + (size_t)i * C;
  float mx = -INFINITY;
  for (int j = lane; j < C; j += 64) mx = fmaxf(mx, lr[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += expf(lr[j] - mx);
  s = wave_sum(s);
  float lse = mx + logf(s);
  int yi = (int)y[i];
  for (int j = lane; j < C; j += 64) dl[(size_t)i * C + j] = (expf(lr[j] - lse) - (j == yi ? 1.f : 0.f)) / (float)B;
  if (lane == 0) {
    float c = (lse - lr[yi]) / (float)B;
    atomicAdd(lossbuf + 1, c);
    atomicAdd(lossbuf + 0, c * (float)B);
  }
}
int fc_ce_fwd_bwd(const float* logits, const int64_t* y, int B, int C, float* lossbuf, float* dlogits, hipStream_t s) {
  hipLaunchKernelGGL(k_ce, dim3(B), dim3(64), 0, s, logits, y, B, C, lossbuf, dlogits);
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== AdamW (fedavgclient.py:63,100 -> torch.optim.AdamW; K13)
__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               size_t n, float decay, float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt,
                                               bf16_t* __restrict__ shadow, int zero_grad) {
  size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], gg = ((float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    float* P = (float*)&pp; float* G = (float*)&gg; float* Mm = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) fc_adamw_elem(P[k], G[k], Mm[k], V[k], decay, beta1, beta2, eps, step_size, inv_bc2_sqrt);
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (shadow) {
      ushort4 sh; sh.x = f2bf(P[0]); sh.y = f2bf(P[1]); sh.z = f2bf(P[2]); sh.w = f2bf(P[3]);
      ((ushort4*)shadow)[i] = sh;
    }
  }
}
int fc_adamw(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd, int step,
             void* shadow_bf16, int zero_grad, hipStream_t s) {
  if (FC_ABLATED("adamw")) return 0;
  FC_REQUIRE(n % 4 == 0 && ((uintptr_t)p % 16 == 0), "adamw: buffer must be 16B aligned and a multiple of 4 elements");
  const FcAdamW o = fc_adamw_consts(lr, beta1, beta2, eps, wd, step);
  const float step_size = o.step_size, inv_bc2_sqrt = o.inv_bc2_sqrt, decay = o.decay;
  size_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_adamw, dim3(grid), dim3(256), 0, s, p, g, m, v, n, decay, beta1, beta2, eps, step_size, inv_bc2_sqrt,
                     (bf16_t*)shadow_bf16, zero_grad);
  FC_LAUNCH_CHECK();
  return 0;
}

// ---- torch.optim.SGD.step (fedavgclient.py:63 with --optimizer SGD: lr, momentum, weight_decay, nesterov from args; dampening 0)
//   g' = g + wd p ; buf = first ? g' : momentum buf + g' ; d = nesterov ? g' + momentum buf : buf (momentum == 0: d = g') ; p -= lr d
__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n, float lr, float momentum,
                                             int nesterov, float wd, int first, bf16_t* __restrict__ shadow) {
#pragma clang fp contract(off)      // torch rounds every product and sum on its own (add_(alpha) aside: one multiply, one add)
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], bb = momentum != 0.f && !first ? ((float4*)buf)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* P = (float*)&pp; float* G = (float*)&gg; float* Bf = (float*)&bb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = G[k];
      if (wd != 0.f) d = d + wd * P[k];
      if (momentum != 0.f) {
        Bf[k] = first ? d : momentum * Bf[k] + d;
        d = nesterov ? d + momentum * Bf[k] : Bf[k];
      }
      P[k] = P[k] - lr * d;
    }
    ((float4*)p)[i] = pp;
    if (momentum != 0.f) ((float4*)buf)[i] = bb;
    if (shadow) {
      ushort4 sh; sh.x = f2bf(P[0]); sh.y = f2bf(P[1]); sh.z = f2bf(P[2]); sh.w = f2bf(P[3]);
      ((ushort4*)shadow)[i] = sh;
    }
  }
}
int fc_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, int nesterov, float wd, int first, void* shadow_bf16, hipStream_t s) {
  FC_REQUIRE(n % 4 == 0 && ((uintptr_t)p % 16 == 0), "sgd: buffer must be 16B aligned and a multiple of 4 elements");
  const size_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_sgd, dim3(grid), dim3(256), 0, s, p, g, buf, n, lr, momentum, nesterov, wd, first, (bf16_t*)shadow_bf16);
  FC_LAUNCH_CHECK();
  return 0;
}

FcAdamW fc_adamw_consts(float lr, float beta1, float beta2, float eps, float wd, int step) {
  FcAdamW o;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  o.step_size = (float)((double)lr / bc1);
  o.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  o.decay = (float)(1.0 - (double)lr * (double)wd);
  o.beta1 = beta1; o.beta2 = beta2; o.eps = eps;
  return o;
}
// one workgroup per chunk; float4 accesses where the chunk is 16-B aligned (every tensor of the models here is), scalar otherwise
__global__ void __launch_bounds__(256) k_adamw_chunks(const FcProxChunk* __restrict__ chunks, FcAdamW o_) {
  const FcAdamW o = fc_adamw_resolve(o_);
  const FcProxChunk c = chunks[blockIdx.x];
  float* p = o.p + c.offset; float* g = o.g0 + c.offset; float* m = o.m + c.offset; float* v = o.v + c.offset;
  if (((c.offset | c.n) & 3) == 0) {
    for (int i = threadIdx.x; i < c.n / 4; i += 256) {
      float4 pp = ((float4*)p)[i], gg = ((float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
      float* P = (float*)&pp; float* G = (float*)&gg; float* Mm = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
      for (int k = 0; k < 4; ++k) fc_adamw_elem(P[k], G[k], Mm[k], V[k], o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
      ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
      if (o.shadow) {
        ushort4 sh; sh.x = f2bf(P[0]); sh.y = f2bf(P[1]); sh.z = f2bf(P[2]); sh.w = f2bf(P[3]);
        ((ushort4*)(o.shadow + c.offset))[i] = sh;
      }
    }
  } else {
    for (int i = threadIdx.x; i < c.n; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      fc_adamw_elem(pp, g[i], mm, vv, o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (o.shadow) o.shadow[c.offset + i] = f2bf(pp);
    }
  }
}
__global__ void __launch_bounds__(256) k_add_chunks(float* __restrict__ dst, const float* __restrict__ src, const FcProxChunk* __restrict__ chunks) {
  const FcProxChunk c = chunks[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) dst[c.offset + i] += src[c.offset + i];
}
int fc_add_chunks(float* dst, const float* src, const FcProxChunk* chunks_dev, int nchunks, hipStream_t s) {
  if (nchunks <= 0) return 0;
  hipLaunchKernelGGL(k_add_chunks, dim3(nchunks), dim3(256), 0, s, dst, src, chunks_dev);
  FC_LAUNCH_CHECK();
  return 0;
}
__global__ void k_adamw_set_dyn(float* dyn, float decay, float step_size, float inv_bc2_sqrt) { dyn[0] = decay; dyn[1] = step_size; dyn[2] = inv_bc2_sqrt; }
int fc_adamw_set_dyn(float* dyn_dev, const FcAdamW& o, hipStream_t s) {
  hipLaunchKernelGGL(k_adamw_set_dyn, dim3(1), dim3(1), 0, s, dyn_dev, o.decay, o.step_size, o.inv_bc2_sqrt);
  FC_LAUNCH_CHECK();
  return 0;
}
int fc_adamw_chunks(const FcProxChunk* chunks_dev, int nchunks, const FcAdamW& o, hipStream_t s) {
  if (nchunks <= 0 || FC_ABLATED("adamw")) return 0;
  hipLaunchKernelGGL(k_adamw_chunks, dim3(nchunks), dim3(256), 0, s, chunks_dev, o);
  FC_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_cast(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) Io<T>::st(dst, i, src[i]);
}
int fc_cast(int dt_out, const float* src, void* dst, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  DISPATCH_DT(dt_out, hipLaunchKernelGGL(k_cast<T>, dim3(grid), dim3(256), 0, s, src, (T*)dst, n));
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== CrossModalReparamLinear (mome.py:58-60; K9)
template <typename T>
__global__ void __launch_bounds__(256) k_reparam_fold(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ sc,
                                                      T* __restrict__ dst, size_t n) {
  float s = sc[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
#pragma clang fp contract(off)      // the reference rounds the product, then the sum (fedavgclient.py:176): no fma, so that the upload is bit-identical
    const float t = s * A[i];
    Io<T>::st(dst, i, W[i] + t);
  }
}
int fc_reparam_fold(int dt, const float* W, const float* A, const float* scale, void* dst, size_t n, hipStream_t s) {
  int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_reparam_fold<T>, dim3(grid), dim3(256), 0, s, W, A, scale, (T*)dst, n));
  FC_LAUNCH_CHECK();
  return 0;
}
__global__ void __launch_bounds__(256) k_reparam_grad(const float* __restrict__ gW, const float* __restrict__ A, const float* __restrict__ sc,
                                                      float* ds, float* __restrict__ gA, size_t n) {
  __shared__ float red[4];
  float s = sc[0], acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gw = gW[i];
    acc += gw * A[i];
    if (gA) gA[i] += s * gw;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(ds, red[0] + red[1] + red[2] + red[3]);
}
int fc_reparam_grad(const float* gW, const float* A, const float* scale, float* ds, float* gA, size_t n, hipStream_t s) {
  int grid = (int)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256);
  hipLaunchKernelGGL(k_reparam_grad, dim3(grid), dim3(256), 0, s, gW, A, scale, ds, gA, n);
  FC_LAUNCH_CHECK();
  return 0;
}

// grouped forms: blockIdx.y = linear, blockIdx.x = slice of it (same arithmetic as the single-linear kernels above)
#define RP_SLICES 64
__global__ void __launch_bounds__(256) k_reparam_grad_grouped(const FcReparam* __restrict__ tab, const float* __restrict__ params, float* __restrict__ grads) {
  __shared__ float red[4];
  const FcReparam e = tab[blockIdx.y];
  const float* gW = grads + e.w;
  const float* A = params + e.aux;
  float* gA = e.aux_trainable ? grads + e.aux : nullptr;
  const float s = params[e.scale];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)e.n; i += (size_t)RP_SLICES * 256) {
    const float gw = gW[i];
    acc += gw * A[i];
    if (gA) gA[i] += s * gw;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(grads + e.scale, red[0] + red[1] + red[2] + red[3]);
}
int fc_reparam_grad_grouped(const FcReparam* tab_dev, int nlin, const float* params, float* grads, hipStream_t s) {
  if (nlin <= 0) return 0;
  hipLaunchKernelGGL(k_reparam_grad_grouped, dim3(RP_SLICES, nlin), dim3(256), 0, s, tab_dev, params, grads);
  FC_LAUNCH_CHECK();
  return 0;
}
template <typename T>
__global__ void __launch_bounds__(256) k_reparam_fold_grouped(const FcReparam* __restrict__ tab, const float* __restrict__ params, T* __restrict__ wc) {
  const FcReparam e = tab[blockIdx.y];
  const float* W = params + e.w;
  const float* A = params + e.aux;
  const float s = params[e.scale];
  T* dst = wc + e.w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)e.n; i += (size_t)RP_SLICES * 256) {
#pragma clang fp contract(off)      // as k_reparam_fold: product rounded, then the sum (fedavgclient.py:176)
    const float t = s * A[i];
    Io<T>::st(dst, i, W[i] + t);
  }
}
int fc_reparam_fold_grouped(int dt, const FcReparam* tab_dev, int nlin, const float* params, void* wc, hipStream_t s) {
  if (nlin <= 0) return 0;
  DISPATCH_DT(dt, hipLaunchKernelGGL(k_reparam_fold_grouped<T>, dim3(RP_SLICES, nlin), dim3(256), 0, s, tab_dev, params, (T*)wc));
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== pre-decoded image batches: uint8 -> the transform's float values
// A client's decoded-image cache (fedcola_amd/loaders/cache.py) keeps images as uint8 codes with a per-channel table
// lut[c][u] = Normalize(ToTensor(u)) (src/loaders/data.py:106-109 as torch ops): the batch crosses PCIe as uint8 (9.6 MB instead of
// 38.5 MB at B = 64) and is expanded here, on the copy stream.  16 pixels per thread: one 16-byte load, four 16-byte stores.
__global__ void __launch_bounds__(256) k_u8_lut(const uint8_t* __restrict__ src, const float* __restrict__ lut, float* __restrict__ dst, size_t n16,
                                                int C, int HW16) {
  __shared__ float tab[4 * 256];
  for (int i = threadIdx.x; i < C * 256; i += 256) tab[i] = lut[i];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i / HW16) % C);
    const uint4 u = ((const uint4*)src)[i];
    const float* t = tab + c * 256;
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ((float4*)dst)[i * 4 + q] = make_float4(t[w[q] & 255], t[(w[q] >> 8) & 255], t[(w[q] >> 16) & 255], t[w[q] >> 24]);
  }
}
__global__ void __launch_bounds__(256) k_u8_lut_s(const uint8_t* __restrict__ src, const float* __restrict__ lut, float* __restrict__ dst, size_t n, int C,
                                                  int HW) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = lut[(int)((i / HW) % C) * 256 + src[i]];
}
extern "C" int fc_image_u8_to_f32(const uint8_t* src, const float* lut, float* dst, int64_t n_images, int32_t C, int32_t HW, void* stream) {
  FC_REQUIRE(src && lut && dst, "fc_image_u8_to_f32: null buffer");
  FC_REQUIRE(C >= 1 && C <= 4 && HW > 0 && n_images >= 0, "fc_image_u8_to_f32: bad geometry (channels %d, pixels %d)", C, HW);
  const size_t n = (size_t)n_images * C * HW;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if ((HW & 15) == 0 && !(((uintptr_t)src | (uintptr_t)dst) & 15)) {
    const size_t n16 = n / 16;
    const int grid = (int)((n16 + 255) / 256 > 2048 ? 2048 : (n16 + 255) / 256);
    hipLaunchKernelGGL(k_u8_lut, dim3(grid), dim3(256), 0, s, src, lut, dst, n16, C, HW / 16);
  } else {
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(k_u8_lut_s, dim3(grid), dim3(256), 0, s, src, lut, dst, n, C, HW);
  }
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== column sums (bias gradients)
// fp32 input (the parity mode, and the fp32 classification heads of every mode): ONE block per 64 columns walks all rows, sums in
// fp64 in a fixed order and stores once -- reproducible, and exact to the final fp32 rounding (these sums cancel to ~1e-3 of their
// summands; an fp32 accumulation in any order costs ~1e-4 of the result there, which is the whole parity budget).
// bf16 input (fallback paths only; the weight-gradient GEMMs produce the bias gradients of the timed mode): row chunks + atomics.
template <typename T>
__global__ void __launch_bounds__(256) k_colsum(const T* __restrict__ dy, float* __restrict__ db, int M, int N, int rows_per_block) {
  __shared__ float red[4][64];
  int col = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc = 0.f;
  if (col < N)
    for (int r = r0 + wave; r < r1; r += 4) acc += Io<T>::ld(dy, (size_t)r * N + col);
  red[wave][threadIdx.x & 63] = acc;
  __syncthreads();
  if (wave == 0 && col < N) atomicAdd(db + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void __launch_bounds__(1024) k_colsum_f64(const float* __restrict__ dy, float* __restrict__ db, int M, int N, int accumulate) {
  __shared__ double red[16][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  double acc = 0.0;
  if (col < N)
    for (int r = wave; r < M; r += 16) acc += (double)dy[(size_t)r * N + col];
  red[wave][threadIdx.x & 63] = acc;
  __syncthreads();
  if (wave == 0 && col < N) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][threadIdx.x];
    db[col] = accumulate ? (float)((double)db[col] + v) : (float)v;
  }
}
int fc_colsum(int dt, const void* dy, float* db, int M, int N, int accumulate, hipStream_t s) {
  if (dt == FC_F32) {
    if (M <= 0) { if (!accumulate) FC_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * N, s)); return 0; }
    hipLaunchKernelGGL(k_colsum_f64, dim3(fc_cdiv(N, 64)), dim3(1024), 0, s, (const float*)dy, db, M, N, accumulate);
    FC_LAUNCH_CHECK();
    return 0;
  }
  if (!accumulate) FC_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * N, s));
  if (M <= 0) return 0;
  int rpb = 256;
  hipLaunchKernelGGL(k_colsum<bf16_t>, dim3(fc_cdiv(N, 64), fc_cdiv(M, rpb)), dim3(256), 0, s, (const bf16_t*)dy, db, M, N, rpb);
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== aggregation blend (fedavgserver.py:656-664 closed form; K14)
__global__ void __launch_bounds__(256) k_blend(float* __restrict__ out, const float* __restrict__ g, const float* const* __restrict__ bases, int m,
                                               const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_len,
                                               const int64_t* __restrict__ src_off, const float* __restrict__ seg_w) {
  int sgi = blockIdx.y;
  int64_t off = seg_off[sgi], len = seg_len[sgi];
  const float* w = seg_w + (size_t)sgi * (m + 1);
  const int64_t* so = src_off + (size_t)sgi * m;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
    float acc = w[0] != 0.f ? w[0] * g[off + i] : 0.f;
    for (int j = 0; j < m; ++j) {
      float wj = w[1 + j];
      if (wj != 0.f && so[j] >= 0) acc += wj * bases[j][so[j] + i];
    }
    out[off + i] = acc;
  }
}
int fc_blend_segments(float* out, const float* g, const float* const* bases, int m, const int64_t* seg_off, const int64_t* seg_len,
                      const int64_t* src_off, const float* seg_w, int nseg, hipStream_t s) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(k_blend, dim3(64, nseg), dim3(256), 0, s, out, g, bases, m, seg_off, seg_len, src_off, seg_w);
  FC_LAUNCH_CHECK();
  return 0;
}
__global__ void __launch_bounds__(256) k_scale_seg(float* __restrict__ buf, const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_len,
                                                   const float* __restrict__ seg_w) {
  int sgi = blockIdx.y;
  int64_t off = seg_off[sgi], len = seg_len[sgi];
  float w = seg_w[sgi];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) buf[off + i] *= w;
}
int fc_scale_segments_impl(float* buf, const int64_t* seg_off, const int64_t* seg_len, const float* seg_w, int nseg, hipStream_t s) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(k_scale_seg, dim3(64, nseg), dim3(256), 0, s, buf, seg_off, seg_len, seg_w);
  FC_LAUNCH_CHECK();
  return 0;
}

// ======================================================================== FedProx proximal term (fedproxclient.py:64-67)
__device__ inline float block_sum_256(float v, float* red) {   // fixed tree: the same order on every run
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}
__global__ void __launch_bounds__(256) k_prox_partial(const float* __restrict__ p, const float* __restrict__ g, const FcProxChunk* __restrict__ chunks,
                                                      float* __restrict__ partial) {
  __shared__ float red[4];
  const FcProxChunk c = chunks[blockIdx.x];
  float acc = 0.f;
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float d = p[c.offset + i] - g[c.offset + i];
    acc += d * d;
  }
  const float r = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// block s: norm[s]; the last block to finish (a counter would do) is avoided: a second single-block launch sums the norms
__global__ void __launch_bounds__(256) k_prox_norm(const float* __restrict__ partial, const int32_t* __restrict__ first, float* __restrict__ norm) {
  __shared__ float red[4];
  const int s = blockIdx.x;
  float acc = 0.f;
  for (int c = first[s] + threadIdx.x; c < first[s + 1]; c += 256) acc += partial[c];
  const float r = block_sum_256(acc, red);
  if (threadIdx.x == 0) norm[s] = sqrtf(r);
}
__global__ void __launch_bounds__(256) k_prox_loss(const float* __restrict__ norm, int nseg, float mu, int B, float* __restrict__ lossbuf) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int s = threadIdx.x; s < nseg; s += 256) acc += norm[s];
  const float r = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    const float v = mu * (0.5f * r);
    lossbuf[1] += v;
    lossbuf[0] += v * (float)B;
  }
}
__global__ void __launch_bounds__(256) k_prox_grad(const float* __restrict__ p, const float* __restrict__ g, const FcProxChunk* __restrict__ chunks,
                                                   const float* __restrict__ norm, float mu, float* __restrict__ grads) {
  const FcProxChunk c = chunks[blockIdx.x];
  const float nr = norm[c.seg];
  if (!(nr > 0.f)) return;                      // torch's norm backward: zero gradient at a zero norm
  const float coef = (mu * 0.5f) / nr;
  for (int i = threadIdx.x; i < c.n; i += 256) grads[c.offset + i] += coef * (p[c.offset + i] - g[c.offset + i]);
}
int fc_prox_term_impl(const float* p, const float* g, const FcProxChunk* chunks, int nchunks, const int32_t* first, int nseg, float* partial,
                      float* norm, float mu, int B, float* grads, float* lossbuf, hipStream_t s) {
  if (nchunks <= 0 || nseg <= 0) return 0;
  hipLaunchKernelGGL(k_prox_partial, dim3(nchunks), dim3(256), 0, s, p, g, chunks, partial);
  hipLaunchKernelGGL(k_prox_norm, dim3(nseg), dim3(256), 0, s, partial, first, norm);
  hipLaunchKernelGGL(k_prox_loss, dim3(1), dim3(256), 0, s, norm, nseg, mu, B, lossbuf);
  hipLaunchKernelGGL(k_prox_grad, dim3(nchunks), dim3(256), 0, s, p, g, chunks, norm, mu, grads);
  FC_LAUNCH_CHECK();
  return 0;
}
