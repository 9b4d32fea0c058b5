// bf16 MFMA GEMMs for gfx950 (CDNA4), written for the FedCola client step's shapes (K = 384..1536, M = B*197).
//
//   NT: C[M,N] = A[M,K] . W[N,K]^T     forward linears            (both operands k-contiguous: "KC")
//   NN: C[M,N] = A[M,K] . W[K,N]       dX = dY . W                (B operand has k as its row index: "KR")
//   TN: C[M,N] = A[K,M]^T . B[K,N]     dW = dY^T . X (fp32 out)   (both operands KR; split-K over the long reduction)
//
// Structure: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16
// accumulators.  Global -> registers -> LDS staging with the next tile's loads issued before the current tile's MFMAs
// and written to the other LDS buffer afterwards (one barrier per k-tile).  LDS images are XOR-swizzled so that both
// fragment read forms are bank-conflict free:
//   KC tile [128 rows][64 k]  (128 B rows): 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7); fragments by ds_read_b128.
//   KR tile [64 k][128 cols]  (256 B rows): 32-B unit u of row k lives at unit u ^ (((k>>3)&1)<<2 | (k&3)); fragments by
//   ds_read_b64_tr_b16 (the CDNA4 transposing LDS read), two per 8-k fragment.
// Epilogue: accumulators -> LDS (fp32, padded rows) -> 8 consecutive columns per thread -> fused bias / GELU(+pre-act
// store) / GELU' multiply / drop-path row scale / residual / patch-embed row remap, 16-byte global stores.
// Block ids are remapped so that the tiles sharing an A row-panel run on the same XCD (private L2).
#include "fc_kernels.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BM 128
#define BN 128
#define BK 64
#define CS_LD 132  // padded fp32 row of the epilogue image
enum { KC = 0, KR = 1 };

__device__ __forceinline__ int kc_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int kr_off(int k, int c) {
  int s = (((k >> 3) & 1) << 2) | (k & 3);
  return k * 256 + (((c >> 1) ^ s) << 5) + ((c & 1) << 4);
}

// ---- staging: each thread moves 4 x 16 B per operand per k-tile
template <int MODE>
__device__ __forceinline__ void stage_load(uint4 (&r)[4], const bf16_t* __restrict__ P, long ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (MODE == KC) {  // P[row][k]
      int c = tid & 7, row = (tid >> 3) + 32 * p;
      int gr = row0 + row, gk = k0 + c * 8;
      if (gr < nrows && gk < K) v = *(const uint4*)(P + (size_t)gr * ld + gk);
    } else {  // P[k][col]
      int c = tid & 15, k = (tid >> 4) + 16 * p;
      int gk = k0 + k, gc = row0 + c * 8;
      if (gk < K && gc < nrows) v = *(const uint4*)(P + (size_t)gk * ld + gc);
    }
    r[p] = v;
  }
}
template <int MODE>
__device__ __forceinline__ void stage_store(const uint4 (&r)[4], char* lds, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (MODE == KC) {
      int c = tid & 7, row = (tid >> 3) + 32 * p;
      *(uint4*)(lds + kc_off(row, c)) = r[p];
    } else {
      int c = tid & 15, k = (tid >> 4) + 16 * p;
      *(uint4*)(lds + kr_off(k, c)) = r[p];
    }
  }
}

// ---- fragment reads.  rb = first row (KC) / first column (KR) of the 16-wide block inside the tile; ks = k-step (0/1)
template <int MODE>
__device__ __forceinline__ bf16x8 frag_read(const char* lds, int rb, int ks, int lane) {
  if (MODE == KC) {
    int row = rb + (lane & 15), c = ks * 4 + (lane >> 4);
    return *(const bf16x8*)(lds + kc_off(row, c));
  } else {
    int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    int col = rb + 4 * p;
    int cbyte = (col & 7) * 2, c = col >> 3;
    int k0 = ks * 32 + 8 * g + q;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + kr_off(k0, c) + cbyte));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + kr_off(k0 + 4, c) + cbyte));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// ---- epilogue on 8 consecutive columns of one row
template <typename TC> struct Vec8;
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    uint4 u = *(const uint4*)p;
    const bf16_t* h = (const bf16_t*)&u;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(h[i]);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    uint4 u;
    bf16_t* h = (bf16_t*)&u;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = f2bf(v[i]);
    *(uint4*)p = u;
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

template <typename TC, bool ATOMIC>
__device__ __forceinline__ void epi_store8(TC* C, long ldc, int m, int n, float (&v)[8], const GemmEpi& e, int N) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (e.bias) {
    float b[8];
    Vec8<float>::ld(e.bias + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  long orow = m;
  if (e.patch_rows > 0) {
    orow = (long)m + m / e.patch_rows + 1;
    float b[8];
    Vec8<float>::ld(e.pos + (size_t)(1 + m % e.patch_rows) * N + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  size_t o = (size_t)orow * ldc + n;
  if (e.preact) {
    Vec8<TC>::st((TC*)e.preact + o, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
  }
  if (e.gelu_in) {
    float u[8];
    Vec8<TC>::ld((const TC*)e.gelu_in + o, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= gelu_erf_grad(u[i]);
  }
  if (e.rowscale) {
    float s = e.rowscale[m / e.rows_per_sample];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= s;
  }
  if (e.res) {
    float r[8];
    Vec8<TC>::ld((const TC*)e.res + o, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (ATOMIC) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd((float*)C + o + i, v[i]);
  } else {
    if (e.accumulate) {
      float r[8];
      Vec8<TC>::ld(C + o, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    Vec8<TC>::st(C + o, v);
  }
}

template <int AMODE, int BMODE, typename TC, bool ATOMIC>
__global__ void __launch_bounds__(256, 2)
k_gemm_mfma(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bm, long ldb, TC* C, long ldc, int M, int N, int K,
            int tiles_n, int ktiles_per_split, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A 16 KB | B 16 KB); reused as Cs[128][CS_LD] fp32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware bijective remap of the tile id (blocks b, b+8, ... share an XCD)
  int nwg = gridDim.x, b = blockIdx.x;
  int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int tile_m = idx / tiles_n, tile_n = idx % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int T = (K + BK - 1) / BK;
  const int t_beg = blockIdx.y * ktiles_per_split;
  const int t_end = min(T, t_beg + ktiles_per_split);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (t_beg < t_end) {
    stage_load<AMODE>(ra, A, lda, m0, M, t_beg * BK, K, tid);
    stage_load<BMODE>(rb, Bm, ldb, n0, N, t_beg * BK, K, tid);
    stage_store<AMODE>(ra, smem, tid);
    stage_store<BMODE>(rb, smem + 16384, tid);
  }
  __syncthreads();
  for (int t = t_beg; t < t_end; ++t) {
    const int cur = (t - t_beg) & 1;
    const char* la = smem + cur * 32768;
    const char* lb = la + 16384;
    const bool more = (t + 1 < t_end);
    if (more) {
      stage_load<AMODE>(ra, A, lda, m0, M, (t + 1) * BK, K, tid);
      stage_load<BMODE>(rb, Bm, ldb, n0, N, (t + 1) * BK, K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_read<AMODE>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BMODE>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * 32768;
      stage_store<AMODE>(ra, na, tid);
      stage_store<BMODE>(rb, na + 16384, tid);
    }
    __syncthreads();
  }
  // ---- epilogue through LDS (all waves are past the last barrier: the staging buffers are free)
  float* Cs = (float*)smem;
  {
    const int g = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int x = 0; x < 4; ++x) Cs[(wm * 64 + i * 16 + g * 4 + x) * CS_LD + wn * 64 + j * 16 + cl] = acc[i][j][x];
  }
  __syncthreads();
  if (ATOMIC && t_beg >= t_end) return;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int row = (tid >> 4) + 16 * p, c8 = (tid & 15) * 8;
    int m = m0 + row, n = n0 + c8;
    if (m < M && n < N) {
      float v[8];
      float4 x0 = *(const float4*)(Cs + row * CS_LD + c8), x1 = *(const float4*)(Cs + row * CS_LD + c8 + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      epi_store8<TC, ATOMIC>(C, ldc, m, n, v, e, N);
    }
  }
}

// ======================================================================== grouped weight-gradient GEMM
// All dW = dY^T . X products of a backward pass (every linear of every layer of both towers) in ONE launch, each output
// tile owned by exactly one workgroup that walks the whole reduction: no split-K, no atomics, bitwise reproducible.
// The bias gradient (column sums of dY) falls out of the A-operand staging registers of the tile_n == 0 workgroups.
__global__ void __launch_bounds__(256, 2) k_gemm_tn_grouped(const FcTnProblem* __restrict__ probs, int nprob) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int nwg = gridDim.x, b = blockIdx.x;
  int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  int pi = 0;
  while (pi + 1 < nprob && probs[pi + 1].tile_start <= idx) ++pi;
  const FcTnProblem P = probs[pi];
  const int local = idx - P.tile_start;
  const int tile_m = local / P.tiles_n, tile_n = local % P.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = P.M, N = P.N, K = P.K;
  const int T = (K + BK - 1) / BK;
  const bool do_colsum = (tile_n == 0) && (P.bias_grad != nullptr);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) cs[i] = 0.f;

  uint4 ra[4], rb[4];
  stage_load<KR>(ra, P.A, P.lda, m0, M, 0, K, tid);
  stage_load<KR>(rb, P.B, P.ldb, n0, N, 0, K, tid);
  stage_store<KR>(ra, smem, tid);
  stage_store<KR>(rb, smem + 16384, tid);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    const char* la = smem + cur * 32768;
    const char* lb = la + 16384;
    if (do_colsum) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bf16_t* hh = (const bf16_t*)&ra[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[i] += bf2f(hh[i]);
      }
    }
    const bool more = (t + 1 < T);
    if (more) {
      stage_load<KR>(ra, P.A, P.lda, m0, M, (t + 1) * BK, K, tid);
      stage_load<KR>(rb, P.B, P.ldb, n0, N, (t + 1) * BK, K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_read<KR>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = frag_read<KR>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * 32768;
      stage_store<KR>(ra, na, tid);
      stage_store<KR>(rb, na + 16384, tid);
    }
    __syncthreads();
  }
  float* Cs = (float*)smem;
  if (do_colsum) {  // thread (c = tid&15, kgroup = tid>>4) holds sums of columns 8c..8c+7 over its k rows
    float* R = Cs;  // [16 kgroups][128 cols]
#pragma unroll
    for (int i = 0; i < 8; ++i) R[(tid >> 4) * 128 + (tid & 15) * 8 + i] = cs[i];
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) s += R[kg * 128 + tid];
      if (m0 + tid < M) P.bias_grad[m0 + tid] = s;
    }
    __syncthreads();
  }
  {
    const int g = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int x = 0; x < 4; ++x) Cs[(wm * 64 + i * 16 + g * 4 + x) * CS_LD + wn * 64 + j * 16 + cl] = acc[i][j][x];
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int row = (tid >> 4) + 16 * p, c8 = (tid & 15) * 8;
    int m = m0 + row, n = n0 + c8;
    if (m < M && n < N) {
      float4 x0 = *(const float4*)(Cs + row * CS_LD + c8), x1 = *(const float4*)(Cs + row * CS_LD + c8 + 4);
      float* dst = P.C + (size_t)m * P.ldc + n;
      *(float4*)dst = x0;
      *(float4*)(dst + 4) = x1;
    }
  }
}

int fc_gemm_tn_grouped_supported(const FcTnProblem& p) {
  return !((p.M & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) ||
           ((uintptr_t)p.C & 15));
}
int fc_gemm_tn_grouped(const FcTnProblem* probs_dev, int nprob, int total_tiles, hipStream_t s) {
  if (nprob <= 0 || total_tiles <= 0) return 0;
  const int lds = BM * CS_LD * 4;
  static bool done = false;
  if (!done) { FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_tn_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  hipLaunchKernelGGL(k_gemm_tn_grouped, dim3(total_tiles), dim3(256), lds, s, probs_dev, nprob);
  FC_LAUNCH_CHECK();
  return 0;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int AM, int BMo, typename TC, bool ATOM>
static int launch_gemm(dim3 grid, const bf16_t* A, long lda, const bf16_t* Bm, long ldb, void* C, long ldc, int M, int N, int K, int tiles_n,
                       int kps, const GemmEpi& epi, hipStream_t s) {
  const int lds = BM * CS_LD * 4;  // 67,584 B (> the 65,536 B staging image)
  auto kfn = k_gemm_mfma<AM, BMo, TC, ATOM>;
  static bool attr_done = false;  // one flag per instantiation
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, A, lda, Bm, ldb, (TC*)C, ldc, M, N, K, tiles_n, kps, epi);
  FC_LAUNCH_CHECK();
  return 0;
}

int fc_gemm_mfma(int kind, int dtC, const bf16_t* A, long lda, const bf16_t* Bm, long ldb, void* C, long ldc, int M, int N, int K,
                 const GemmEpi& epi, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // vector-width constraints of this kernel; anything else goes to the generic path
  if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 7) || !aligned16(A) || !aligned16(Bm) || !aligned16(C)) return 1;
  if (kind != FC_GEMM_TN && (K & 7)) return 1;
  if (kind == FC_GEMM_TN && (M & 7)) return 1;
  if (epi.bias && !aligned16(epi.bias)) return 1;
  if (epi.res && !aligned16(epi.res)) return 1;
  if (epi.preact && !aligned16(epi.preact)) return 1;
  if (epi.gelu_in && !aligned16(epi.gelu_in)) return 1;
  if (epi.pos && !aligned16(epi.pos)) return 1;
  int tiles_m = fc_cdiv(M, BM), tiles_n = fc_cdiv(N, BN);
  int tiles = tiles_m * tiles_n;
  int T = fc_cdiv(K, BK);
  if (kind == FC_GEMM_NT) {
    if (dtC == FC_BF16) return launch_gemm<KC, KC, bf16_t, false>(dim3(tiles, 1), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, T, epi, s);
    return launch_gemm<KC, KC, float, false>(dim3(tiles, 1), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, T, epi, s);
  }
  if (kind == FC_GEMM_NN) {
    if (dtC == FC_BF16) return launch_gemm<KC, KR, bf16_t, false>(dim3(tiles, 1), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, T, epi, s);
    return launch_gemm<KC, KR, float, false>(dim3(tiles, 1), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, T, epi, s);
  }
  if (dtC != FC_F32) return 1;
  // TN (weight gradients): split the long reduction so that ~2 workgroups per CU are in flight; the partial tiles are
  // combined with fp32 atomics into a zeroed (or accumulating) output
  bool plain = !epi.bias && !epi.res && !epi.preact && !epi.gelu_in && !epi.rowscale && epi.patch_rows == 0;
  int splits = plain ? (512 + tiles - 1) / tiles : 1;
  if (splits > T) splits = T;
  if (splits < 1) splits = 1;
  int kps = fc_cdiv(T, splits);
  splits = fc_cdiv(T, kps);
  if (splits == 1) return launch_gemm<KR, KR, float, false>(dim3(tiles, 1), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, T, epi, s);
  if (!epi.accumulate && !epi.out_zeroed) FC_CHECK_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, s));
  GemmEpi e2 = epi;
  e2.accumulate = 0;
  return launch_gemm<KR, KR, float, true>(dim3(tiles, splits), A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, kps, e2, s);
}
