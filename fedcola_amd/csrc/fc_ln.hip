// LayerNorm forward / backward for gfx950, grouped over up to two row sets (the image and the text tower of one layer in ONE launch).
// Reference semantics: nn.LayerNorm inside Block / BertEmbeddings, /root/reference/src/models/mome.py:199,203,215,751.
//
// Layout: 16 lanes own one row, a lane owns the 16-byte chunks {sub, sub+16, ...} of it, so at D = 384 (48 chunks) all 64 lanes of a
// wave carry data (the one-wave-per-row form of rounds 1-2 left 16 of 64 lanes idle) and a wave works on 4 rows at a time; row
// reductions are 4 xor-shuffles inside the 16-lane group.  Every load of a wave's rows is issued before the first use.  HBM-bound:
// forward 2 x M x D elements, backward 4 x M x D (dy, x, residual in; dx out).
// dgamma / dbeta: per-lane column sums over the block's rows -> one partial row [dgamma | dbeta] per block (fixed order, no atomics),
// summed by k_ln_reduce once per backward for every LayerNorm of the model.
#include <stdlib.h>

#include "fc_kernels.h"

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 u;
  __device__ __forceinline__ void ld(const bf16_t* p) { u = *(const uint4*)p; }
  __device__ __forceinline__ void zero() { u = make_uint4(0u, 0u, 0u, 0u); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    *(uint4*)p = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float sum16(float v) {   // over the 16 lanes that share a row
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

template <typename T> struct LnAcc { typedef float type; };          // in-lane column sums: see k_ln_bwd_g
template <> struct LnAcc<float> { typedef double type; };
#define LN_FWD_RG 2                         // row groups (of 4 rows) per wave
#define LN_FWD_ROWS (16 * LN_FWD_RG)        // rows per block (4 waves)

template <typename T, int CH>
__global__ void __launch_bounds__(256) k_ln_fwd_g(FcLnFwdArgs a) {
  const bool second = a.nprob > 1 && (int)blockIdx.x >= a.p[1].blk0;
  const FcLnFwdP& P = second ? a.p[1] : a.p[0];
  const int D = a.D, nc = D >> 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, slot = lane >> 4;
  const int rb = ((int)blockIdx.x - P.blk0) * LN_FWD_ROWS + wave * (4 * LN_FWD_RG);
  const T* x = (const T*)P.x;
  T* y = (T*)P.y;
  Raw8<T> raw[LN_FWD_RG][CH];
  int row[LN_FWD_RG];
#pragma unroll
  for (int q = 0; q < LN_FWD_RG; ++q) {
    row[q] = rb + 4 * q + slot;
    const int rc = row[q] < P.M ? row[q] : P.M - 1;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int c = sub + 16 * t;
      if (c < nc) raw[q][t].ld(x + (size_t)rc * D + c * 8); else raw[q][t].zero();
    }
  }
  float gg[CH][8], bb[CH][8];
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
    if (c < nc) { ld8f(P.g + c * 8, gg[t]); ld8f(P.b + c * 8, bb[t]); }
  }
  const float invD = 1.0f / (float)D;
#pragma unroll
  for (int q = 0; q < LN_FWD_RG; ++q) {
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      raw[q][t].get(v[t]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[t][i];
    }
    const float mu = sum16(s) * invD;
    float qq = 0.f;
#pragma unroll
    for (int t = 0; t < CH; ++t)
      if (sub + 16 * t < nc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[t][i] - mu; qq += d * d; }
      }
    const float rs = 1.0f / sqrtf(sum16(qq) * invD + a.eps);
    if (row[q] < P.M) {
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int c = sub + 16 * t;
        if (c < nc) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (v[t][i] - mu) * rs * gg[t][i] + bb[t][i];
          Raw8<T>::st(y + (size_t)row[q] * D + c * 8, o);
        }
      }
      if (sub == 0) { P.mean[row[q]] = mu; P.rstd[row[q]] = rs; }
    }
  }
}

// ---- backward.  Block = 4 waves x RG row groups x 4 rows; dx = res + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)).
template <typename T, int CH, int LN_BWD_RG>
__global__ void __launch_bounds__(256) k_ln_bwd_g(FcLnBwdArgs a) {
  constexpr int LN_BWD_ROWS = 16 * LN_BWD_RG;
  typedef typename LnAcc<T>::type acc_t;   // fp32 storage (parity mode): fp64 column sums from the first add; bf16 storage: fp32
  extern __shared__ char red_raw[];
  acc_t* red_dyn = (acc_t*)red_raw;        // [4 waves][2][D]
  const bool second = a.nprob > 1 && (int)blockIdx.x >= a.p[1].blk0;
  const FcLnBwdP& P = second ? a.p[1] : a.p[0];
  const int D = a.D, nc = D >> 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, slot = lane >> 4;
  const int lblk = (int)blockIdx.x - P.blk0;
  const int rb = lblk * LN_BWD_ROWS + wave * (4 * LN_BWD_RG);
  const T* dy = (const T*)P.dy;
  const T* x = (const T*)P.x;
  const T* res = (const T*)P.res;
  T* dx = (T*)P.dx;
  T* dxs = (T*)P.dx_scaled;
  Raw8<T> rd[LN_BWD_RG][CH], rx[LN_BWD_RG][CH], rr[LN_BWD_RG][CH];
  int row[LN_BWD_RG];
  float mu[LN_BWD_RG], rs[LN_BWD_RG];
#pragma unroll
  for (int q = 0; q < LN_BWD_RG; ++q) {          // every load of the wave's rows first
    row[q] = rb + 4 * q + slot;
    const int rc = row[q] < P.M ? row[q] : P.M - 1;
    mu[q] = P.mean[rc]; rs[q] = P.rstd[rc];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int c = sub + 16 * t;
      if (c < nc) {
        rd[q][t].ld(dy + (size_t)rc * D + c * 8);
        rx[q][t].ld(x + (size_t)rc * D + c * 8);
        if (res) rr[q][t].ld(res + (size_t)rc * D + c * 8); else rr[q][t].zero();
      } else {
        rd[q][t].zero(); rx[q][t].zero(); rr[q][t].zero();
      }
    }
  }
  // Column sums: fp32 storage (the parity mode) adds in fp64 from the first add to the partial row (the sums cancel to ~1e-3 of their
  // summands: fp32 accumulation alone costs ~1e-4 of the result there, the whole parity budget).  bf16 storage keeps fp32 up to the
  // block's partial row -- its inputs carry 2^-9 each -- because fp64 adds, shuffles and partial rows made this HBM-bound kernel slower
  // (10.6 -> 15.1 us at 12 608 rows all-fp64, 4.57 -> 4.63 ms per step with fp64 behind the lane sums); the partial rows of either type
  // are summed in fp64 by k_ln_reduce.
  float gg[CH][8];
  acc_t ag[CH][8], ab[CH][8];
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ag[t][i] = 0; ab[t][i] = 0; gg[t][i] = 0.f; }
    if (c < nc) ld8f(P.g + c * 8, gg[t]);
  }
  const float invD = 1.0f / (float)D;
#pragma unroll
  for (int q = 0; q < LN_BWD_RG; ++q) {
    const bool live = row[q] < P.M;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      float d[8], xv[8];
      rd[q][t].get(d); rx[q][t].get(xv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[i] - mu[q]) * rs[q];
        const float dxh = d[i] * gg[t][i];
        s1 += dxh; s2 += dxh * xh;
        if (live) { ag[t][i] += (acc_t)(d[i] * xh); ab[t][i] += (acc_t)d[i]; }
      }
    }
    const float m1 = sum16(s1) * invD, m2 = sum16(s2) * invD;
    if (live) {
      const float sc = dxs ? P.rowscale[row[q] / P.rows_per_sample] : 0.f;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int c = sub + 16 * t;
        if (c < nc) {
          float d[8], xv[8], r[8], o[8];
          rd[q][t].get(d); rx[q][t].get(xv); rr[q][t].get(r);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float xh = (xv[i] - mu[q]) * rs[q];
            o[i] = rs[q] * (d[i] * gg[t][i] - m1 - xh * m2) + r[i];
          }
          Raw8<T>::st(dx + (size_t)row[q] * D + c * 8, o);
          if (dxs) {      // drop-path: the consumer of this gradient wants it times the per-sample multiplier (of the STORED value)
            float os[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) os[i] = Io<T>::rt(o[i]) * sc;
            Raw8<T>::st(dxs + (size_t)row[q] * D + c * 8, os);
          }
        }
      }
    }
  }
#ifdef FC_PROBES
  if (a.skip_partials) return;      // measurement only (FC_LN_NOPART=1): what the column-sum tail costs; dgamma / dbeta are wrong
#endif
  // column sums: the four row slots of the wave (lanes l, l+16, l+32, l+48 hold the same columns), then the four waves through LDS
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = sub + 16 * t;
    acc_t* r0 = red_dyn + (wave * 2 + 0) * D + c * 8;
    acc_t* r1 = red_dyn + (wave * 2 + 1) * D + c * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc_t a0 = ag[t][i], b0 = ab[t][i];
      a0 += __shfl_xor(a0, 16, 64); a0 += __shfl_xor(a0, 32, 64);
      b0 += __shfl_xor(b0, 16, 64); b0 += __shfl_xor(b0, 32, 64);
      if (slot == 0 && c < nc) { r0[i] = a0; r1[i] = b0; }
    }
  }
  __syncthreads();
  acc_t* pp = (acc_t*)P.partial + (size_t)lblk * 2 * D;      // (the buffer is sized for fp64 rows; bf16 storage writes fp32 ones: FcLnReduce.f64)
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int h = i >= D, col = i - h * D;
    pp[i] = red_dyn[(0 * 2 + h) * D + col] + red_dyn[(1 * 2 + h) * D + col] + red_dyn[(2 * 2 + h) * D + col] + red_dyn[(3 * 2 + h) * D + col];
  }
}

// One partial row per block of LN_BWD_ROWS rows -> dgamma / dbeta.  Block (x = 64-column slab of [dgamma | dbeta], y = LayerNorm
// instance): 16 waves stride the partial rows, combine in LDS, ONE plain store per column (accumulate = 1: an add into the existing
// value, for callers that keep a running gradient) -- no atomics, fixed order.
__global__ void __launch_bounds__(1024) k_ln_reduce(const FcLnReduce* __restrict__ tab) {
  __shared__ double red[16][64];
  const FcLnReduce e = tab[blockIdx.y];
  const int W = 2 * e.D;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= W) return;
  double acc = 0.0;
  const void* sets[3] = {e.partial, e.partial2, e.partial3};
  const int nb[3] = {e.nblocks, e.nblocks2, e.nblocks3};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (!sets[q] || col >= W) continue;
    if (e.f64) { for (int bk = wave; bk < nb[q]; bk += 16) acc += ((const double*)sets[q])[(size_t)bk * W + col]; }
    else { for (int bk = wave; bk < nb[q]; bk += 16) acc += (double)((const float*)sets[q])[(size_t)bk * W + col]; }
  }
  red[wave][threadIdx.x & 63] = acc;
  __syncthreads();
  if (wave == 0 && col < W) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][threadIdx.x];
    float* dst = col < e.D ? e.dg + col : e.db + col - e.D;
    *dst = e.accumulate ? (float)((double)*dst + v) : (float)v;
  }
}
int fc_ln_reduce_grouped(const FcLnReduce* tab_dev, int n, int maxD, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ln_reduce, dim3(fc_cdiv(2 * maxD, 64), n), dim3(1024), 0, s, tab_dev);
  FC_LAUNCH_CHECK();
  return 0;
}

// row groups per wave in the backward: 2 = 32 rows per block (half the partial rows; 178 VGPRs), 1 = 16 rows per block
static int ln_bwd_rg() {
#ifdef FC_PROBES
  static const int v = getenv("FC_LN_BWD_RG") ? atoi(getenv("FC_LN_BWD_RG")) : 2;
  return v == 1 ? 1 : 2;
#else
  return 2;
#endif
}
int fc_layernorm_bwd_partial_blocks(int M) { return fc_cdiv(M, 16 * ln_bwd_rg()); }

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
int fc_layernorm_grouped_ok(int D) { return (D % 8 == 0) && D <= 1024; }

int fc_layernorm_fwd_grouped(int dt, FcLnFwdArgs a, hipStream_t s) {
  FC_REQUIRE(a.nprob >= 1 && a.nprob <= 2 && fc_layernorm_grouped_ok(a.D), "layernorm_fwd_grouped: unsupported shape (D = %d)", a.D);
  int blocks = 0;
  for (int i = 0; i < a.nprob; ++i) {
    FC_REQUIRE(al16(a.p[i].x) && al16(a.p[i].y) && al16(a.p[i].g) && al16(a.p[i].b), "layernorm_fwd_grouped: operands must be 16-byte aligned");
    a.p[i].blk0 = blocks;
    blocks += fc_cdiv(a.p[i].M, LN_FWD_ROWS);
  }
  if (blocks == 0) return 0;
  const int ch = fc_cdiv(a.D / 8, 16);
#define GO(CHN) DISPATCH_DT(dt, hipLaunchKernelGGL((k_ln_fwd_g<T, CHN>), dim3(blocks), dim3(256), 0, s, a))
  switch (ch) {
    case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
    case 5: case 6: GO(6); break; default: GO(8); break;
  }
#undef GO
  FC_LAUNCH_CHECK();
  return 0;
}
int fc_layernorm_bwd_grouped(int dt, FcLnBwdArgs a, hipStream_t s) {
  FC_REQUIRE(a.nprob >= 1 && a.nprob <= 2 && fc_layernorm_grouped_ok(a.D), "layernorm_bwd_grouped: unsupported shape (D = %d)", a.D);
  int blocks = 0;
  for (int i = 0; i < a.nprob; ++i) {
    const FcLnBwdP& p = a.p[i];
    FC_REQUIRE(al16(p.dy) && al16(p.x) && al16(p.res) && al16(p.dx) && al16(p.dx_scaled) && al16(p.g) && al16(p.partial),
               "layernorm_bwd_grouped: operands must be 16-byte aligned");
    a.p[i].blk0 = blocks;
    blocks += fc_layernorm_bwd_partial_blocks(p.M);
  }
  if (blocks == 0) return 0;
  const int ch = fc_cdiv(a.D / 8, 16);
#ifdef FC_PROBES
  static const int nopart = fc_knob("FC_LN_NOPART", 0);
  a.skip_partials = nopart;
#endif
  const size_t lds = (dt == FC_F32 ? sizeof(double) : sizeof(float)) * 8 * a.D;
#define GO(CHN)                                                                                                     \
  do {                                                                                                              \
    if (ln_bwd_rg() == 1) { DISPATCH_DT(dt, hipLaunchKernelGGL((k_ln_bwd_g<T, CHN, 1>), dim3(blocks), dim3(256), lds, s, a)); } \
    else { DISPATCH_DT(dt, hipLaunchKernelGGL((k_ln_bwd_g<T, CHN, 2>), dim3(blocks), dim3(256), lds, s, a)); }      \
  } while (0)
  switch (ch) {
    case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
    case 5: case 6: GO(6); break; default: GO(8); break;
  }
#undef GO
  FC_LAUNCH_CHECK();
  return 0;
}
