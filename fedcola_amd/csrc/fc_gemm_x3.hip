// fp32 GEMMs on the matrix cores: every fp32 operand element a is split into THREE bf16 values, hi = bf16(a), mid = bf16(a - hi),
// lo = bf16(a - hi - mid) (both differences are exact in fp32, so hi + mid + lo carries 24 significant bits: a itself up to the last
// rounding), and a product tile is accumulated in fp32 from the SIX bf16 MFMAs whose weight is >= 2^-16 of the leading one,
//   A.B ~= Ah.Bh + (Ah.Bm + Am.Bh) + (Ah.Bl + Al.Bh + Am.Bm)          (dropped: Am.Bl, Al.Bm, Al.Bl <= 3 x 2^-24 |a||b| per term),
// i.e. the rounding class of an fp32 FMA chain.  (Two pieces and three products leave 1.5 x 2^-17 per term: measured 1.1e-4 on a LayerNorm
// weight gradient of the small golden model -- over the reference's 1e-4 bar.)  This is the GEMM of the library's precision = 'fp32' mode
// (fp32 storage everywhere; the mode that holds the <= 1e-4 bar) wherever the shapes allow 16-byte accesses; the VALU kernel of
// fc_generic.hip takes the rest.  Reference: the linears of Attention / Mlp, /root/reference/src/models/mome.py:117-123,150-168.
//
//   NT: C[M,N] = A[M,K] . W[N,K]^T   (forward)      NN: C[M,N] = A[M,K] . W[K,N]   (dX)      TN: C[M,N] = A[K,M]^T . B[K,N]   (dW)
//
// 128 x 128 output tile, BK = 64, 256 threads = 2 x 2 waves of 64 x 64 (4 x 4 v_mfma_f32_16x16x32_bf16 accumulators, operands swapped so
// that a lane owns 4 consecutive output columns), one workgroup per CU (96 KB of LDS): 192 MFMAs per wave between two barriers -- the
// kernel is bound by the matrix pipe, at a sixth of the bf16 rate.  Per k-tile a thread loads 8 + 8 float4 of the two operands (the NEXT
// tile's loads are in flight under this tile's MFMAs), splits them and writes the three images of each operand in the bf16 kernels' LDS
// layouts (fc_mfma_dev.h: k-contiguous rows read by ds_read_b128, k-row tiles read by ds_read_b64_tr_b16), 6 x 16 KB.
#include <map>
#include <mutex>

#include "fc_kernels.h"
#include "fc_mfma_dev.h"

struct X3Regs { float4 v[8]; };

// One operand tile: KC = P[row][k] (k contiguous; thread owns the 8-k chunk c = tid & 7 of rows (tid >> 3) + 32 p), KR = P[k][col] (thread owns the
// 8-column chunk c = tid & 15 of k rows (tid >> 4) + 16 p).  Out-of-range rows / columns / k read as zero.
template <int MODE>
__device__ __forceinline__ void x3_load(X3Regs& R, const float* __restrict__ P, long ld, int r0, int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    bool ok;
    const float* src;
    if (MODE == KC) {
      const int row = r0 + (tid >> 3) + 32 * p, k = k0 + (tid & 7) * 8;
      ok = row < nrows && k < K;
      src = P + (size_t)(ok ? row : 0) * ld + (ok ? k : 0);
    } else {
      const int k = k0 + (tid >> 4) + 16 * p, col = r0 + (tid & 15) * 8;
      ok = k < K && col < nrows;
      src = P + (size_t)(ok ? k : 0) * ld + (ok ? col : 0);
    }
    const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
    R.v[2 * p] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
    R.v[2 * p + 1] = ok ? b : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void x3_split(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const bf16_t ha = f2bf(a), hb = f2bf(b);
  const float ra = a - bf2f(ha), rb = b - bf2f(hb);          // exact
  const bf16_t ma = f2bf(ra), mb = f2bf(rb);
  hi = (unsigned)ha | ((unsigned)hb << 16);
  mid = (unsigned)ma | ((unsigned)mb << 16);
  lo = f2bf2(ra - bf2f(ma), rb - bf2f(mb));                  // exact differences again
}
template <int MODE>
__device__ __forceinline__ void x3_store(const X3Regs& R, char* img, int tid) {   // img: hi | mid | lo images, 16 KB apart
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint4 h, m, l;
    x3_split(R.v[2 * p].x, R.v[2 * p].y, h.x, m.x, l.x);
    x3_split(R.v[2 * p].z, R.v[2 * p].w, h.y, m.y, l.y);
    x3_split(R.v[2 * p + 1].x, R.v[2 * p + 1].y, h.z, m.z, l.z);
    x3_split(R.v[2 * p + 1].z, R.v[2 * p + 1].w, h.w, m.w, l.w);
    const int off = MODE == KC ? kc_off((tid >> 3) + 32 * p, tid & 7) : kr_off((tid >> 4) + 16 * p, tid & 15);
    *(uint4*)(img + off) = h;
    *(uint4*)(img + 16384 + off) = m;
    *(uint4*)(img + 32768 + off) = l;
  }
}

// the epilogue of one lane's 4 consecutive columns of row m (fp32 in, fp32 out; erf-GELU as the reference's nn.GELU)
__device__ __forceinline__ void x3_epi4(float* C, long ldc, int m, int n, f32x4 acc, const GemmEpi& e, int N) {
  float v[4] = {acc[0] * e.alpha, acc[1] * e.alpha, acc[2] * e.alpha, acc[3] * e.alpha};
  if (e.bias) {
    const float4 b = *(const float4*)(e.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  long orow = m;
  if (e.patch_rows > 0) {
    orow = (long)m + m / e.patch_rows + 1;
    const float4 b = *(const float4*)(e.pos + (size_t)(1 + m % e.patch_rows) * N + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  const size_t o = (size_t)orow * ldc + n;
  if (e.preact) {
    float4 u;
    if (e.gelu_saved_grad) u = make_float4(gelu_erf_grad(v[0]), gelu_erf_grad(v[1]), gelu_erf_grad(v[2]), gelu_erf_grad(v[3]));
    else u = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)e.preact + o) = u;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
  }
  if (e.gelu_in) {
    const float4 u = *(const float4*)((const float*)e.gelu_in + o);
    const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= e.gelu_saved_grad ? uu[i] : gelu_erf_grad(uu[i]);
  }
  if (e.rowscale) {
    const float s = e.rowscale[m / e.rows_per_sample];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= s;
  }
  if (e.res) {
    const float4 r = *(const float4*)((const float*)e.res + o);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  if (e.accumulate) {
    const float4 r = *(const float4*)(C + o);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  *(float4*)(C + o) = make_float4(v[0], v[1], v[2], v[3]);
}

// one 64-k tile of the six-product sum from the LDS images into the running fp32 accumulators
template <int AMODE, int BMODE>
__device__ __forceinline__ void x3_compute(const char* ai, const char* bi, f32x4 (&acc)[4][4], int wm, int wn, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 fa[3][4], fb[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[c][i] = frag_read<AMODE>(ai + c * 16384, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[c][j] = frag_read<BMODE>(bi + c * 16384, wn * 64 + j * 16, ks, lane);
      if (AMODE == KR) frag_fence(fa[c]);
      if (BMODE == KR) frag_fence(fb[c]);
    }
    // (and in every second half-step with the A fragments negated and the partial subtracted: a rounding toward -inf becomes one toward
    // +inf there, the two directions cancel on average)
    if (ks & 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint4 u = *(uint4*)&fa[c][i];
          u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
          fa[c][i] = *(bf16x8*)&u;
        }
    }
    // The matrix pipe's fp32 accumulate truncates (measured: a systematic error of -6e-11 |result| per k element when every product of a
    // 12 608-long reduction lands in one accumulator -- harmless per element, but the column sums downstream add it 12 608 times).  So the
    // six products of a 32-k half-step go into a fresh accumulator and join the running sum by a round-to-nearest VALU add.
    f32x4 part[4][4];
#define X3_MFMA0(ca, cb)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
  part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cb][j], fa[ca][i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
#define X3_MFMA(ca, cb)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
  part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cb][j], fa[ca][i], part[i][j], 0, 0, 0)
    X3_MFMA0(2, 0); X3_MFMA(0, 2); X3_MFMA(1, 1);         // 2^-16 terms first, the leading product last
    X3_MFMA(1, 0); X3_MFMA(0, 1);
    X3_MFMA(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (ks & 1) ? acc[i][j] - part[i][j] : acc[i][j] + part[i][j];
#undef X3_MFMA0
#undef X3_MFMA
  }
}

template <int AMODE, int BMODE>
__global__ void __launch_bounds__(256, 1) k_gemm_x3(const float* __restrict__ A, long lda, const float* __restrict__ Bm, long ldb, float* __restrict__ C, long ldc,
                                                    int M, int N, int K, int tiles_n, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A: hi | mid | lo, then B: hi | mid | lo, 16 KB each
  char* ai = smem;
  char* bi = smem + 49152;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int T = (K + BK - 1) / BK;
  X3Regs ra, rb;
  x3_load<AMODE>(ra, A, lda, m0, M, 0, K, tid);
  x3_load<BMODE>(rb, Bm, ldb, n0, N, 0, K, tid);
  for (int kt = 0; kt < T; ++kt) {
    x3_store<AMODE>(ra, ai, tid);
    x3_store<BMODE>(rb, bi, tid);
    lds_barrier();
    if (kt + 1 < T) {                                      // the next tile's loads fly under this tile's MFMAs
      x3_load<AMODE>(ra, A, lda, m0, M, (kt + 1) * BK, K, tid);
      x3_load<BMODE>(rb, Bm, ldb, n0, N, (kt + 1) * BK, K, tid);
    }
    x3_compute<AMODE, BMODE>(ai, bi, acc, wm, wn, lane);
    lds_barrier();                                         // fragment reads done before the images are overwritten
  }
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + cl;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (n < N) x3_epi4(C, ldc, m, n, acc[i][j], e, N);
    }
  }
}

// ======================================================================== weight gradients of the fp32 mode
// dW[out, in] = dY[rows, out]^T . X[rows, in] and db[out] = column sums of dY: the output has 9 - 72 tiles and the reduction runs over all
// the rows of the batch (12 608), so the reduction is cut into S slices (tiles x S ~ two workgroups per CU); slice s writes its raw partial
// tile to part[s] and -- the workgroups of column tile 0 -- the fp64 column sums of its rows of dY (taken from the fp32 staging registers,
// before the split) to colp[s]; k_dw_x3_reduce then adds the slices in a fixed order in fp64.  No atomics: the same bits every run.
__global__ void __launch_bounds__(256, 1) k_dw_x3(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part, double* __restrict__ colp,
                                                  int rows, int out, int in, int tiles_n, int tiles, int kt_per) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ai = smem;
  char* bi = smem + 49152;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int sl = blockIdx.x / tiles, t = blockIdx.x % tiles;
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const bool do_col = colp != nullptr && (t % tiles_n) == 0;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  double cs[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  const int T = (rows + BK - 1) / BK;
  const int k_beg = sl * kt_per, k_end = (k_beg + kt_per < T) ? k_beg + kt_per : T;
  X3Regs ra, rb;
  if (k_beg < k_end) {
    x3_load<KR>(ra, dY, out, m0, out, k_beg * BK, rows, tid);
    x3_load<KR>(rb, X, in, n0, in, k_beg * BK, rows, tid);
  }
  for (int kt = k_beg; kt < k_end; ++kt) {
    if (do_col) {                                          // this thread's 8 columns x 4 k rows of the tile
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        cs[0] += ra.v[2 * p].x; cs[1] += ra.v[2 * p].y; cs[2] += ra.v[2 * p].z; cs[3] += ra.v[2 * p].w;
        cs[4] += ra.v[2 * p + 1].x; cs[5] += ra.v[2 * p + 1].y; cs[6] += ra.v[2 * p + 1].z; cs[7] += ra.v[2 * p + 1].w;
      }
    }
    x3_store<KR>(ra, ai, tid);
    x3_store<KR>(rb, bi, tid);
    lds_barrier();
    if (kt + 1 < k_end) {
      x3_load<KR>(ra, dY, out, m0, out, (kt + 1) * BK, rows, tid);
      x3_load<KR>(rb, X, in, n0, in, (kt + 1) * BK, rows, tid);
    }
    x3_compute<KR, KR>(ai, bi, acc, wm, wn, lane);
    lds_barrier();
  }
  const int g = lane >> 4, cl = lane & 15;
  float* P = part + (size_t)sl * out * in;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + cl;
    if (m >= out) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (n < in) *(float4*)(P + (size_t)m * in + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
  if (do_col) {                                            // 16 threads (tid >> 4) share a column chunk (tid & 15): add them through LDS
    double* red = (double*)smem;                           // [16 k-groups][128 columns]
#pragma unroll
    for (int x = 0; x < 8; ++x) red[(tid >> 4) * 128 + (tid & 15) * 8 + x] = cs[x];
    __syncthreads();
    if (tid < 128) {
      double v = 0.;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[r * 128 + tid];
      if (m0 + tid < out) colp[(size_t)sl * out + m0 + tid] = v;
    }
  }
}
__global__ void __launch_bounds__(256) k_dw_x3_reduce(const float* __restrict__ part, const double* __restrict__ colp, float* __restrict__ dW, float* __restrict__ db,
                                                      long n, int out, int S) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i < n) {
    double a = 0., b = 0., c = 0., d = 0.;
    for (int s = 0; s < S; ++s) {
      const float4 v = *(const float4*)(part + (size_t)s * n + i);
      a += v.x; b += v.y; c += v.z; d += v.w;
    }
    *(float4*)(dW + i) = make_float4((float)a, (float)b, (float)c, (float)d);
  }
  if (db && blockIdx.x == 0)
    for (int m = threadIdx.x; m < out; m += 256) {
      double v = 0.;
      for (int s = 0; s < S; ++s) v += colp[(size_t)s * out + m];
      db[m] = (float)((double)db[m] + v);                  // the bias gradient accumulates, as fc_colsum(accumulate = 1) did
    }
}

// per-stream scratch for the slices' partial tiles (a stream runs one weight-gradient product at a time)
static std::mutex g_x3_mu;
static std::map<hipStream_t, std::pair<char*, size_t>> g_x3_ws;
static int x3_scratch(hipStream_t s, size_t bytes, char** out) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  auto& w = g_x3_ws[s];
  if (w.second < bytes) {
    if (w.first) {
      FC_CHECK_HIP(hipStreamSynchronize(s));
      FC_CHECK_HIP(hipFree(w.first));
      w = {nullptr, 0};
    }
    const size_t want = bytes + bytes / 4;
    FC_CHECK_HIP(hipMalloc((void**)&w.first, want));
    w.second = want;
  }
  *out = w.first;
  return 0;
}
// dW = dY^T . X (stored), db += column sums of dY (db may be null); 1 = shape not covered
int fc_dw_x3(const float* dY, const float* X, float* dW, float* db, int rows, int out, int in, hipStream_t s) {
  static const int on = fc_knob("FC_X3", 1);
  if (!on) return 1;
  if (rows <= 0 || out <= 0 || in <= 0) return 1;
  if ((out & 7) || (in & 7) || ((uintptr_t)dY & 15) || ((uintptr_t)X & 15) || ((uintptr_t)dW & 15)) return 1;
  const int tiles_n = fc_cdiv(in, BN), tiles = fc_cdiv(out, BM) * tiles_n, T = fc_cdiv(rows, BK);
  int S = 512 / tiles;
  if (S > 32) S = 32;
  if (S > T) S = T;
  if (S < 1) S = 1;
  const int kt_per = fc_cdiv(T, S);
  S = fc_cdiv(T, kt_per);
  const size_t n = (size_t)out * in, pbytes = (size_t)S * n * sizeof(float), cbytes = (size_t)S * out * sizeof(double);
  char* ws = nullptr;
  FC_TRY(x3_scratch(s, pbytes + cbytes, &ws));
  float* part = (float*)ws;
  double* colp = db ? (double*)(ws + pbytes) : nullptr;
  static bool attr_done = false;
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_dw_x3, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_dw_x3, dim3(tiles * S), dim3(256), 98304, s, dY, X, part, colp, rows, out, in, tiles_n, tiles, kt_per);
  FC_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dw_x3_reduce, dim3(fc_cdiv((long)n / 4, 256)), dim3(256), 0, s, part, colp, dW, db, (long)n, out, S);
  FC_LAUNCH_CHECK();
  return 0;
}

static bool x3_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// returns 0 = launched, 1 = shape not covered (the caller takes the VALU kernel), < 0 = error
int fc_gemm_x3(int kind, const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  static const int on = fc_knob("FC_X3", 1);
  if (!on) return 1;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((N & 7) || (K & 7) || (lda & 3) || (ldb & 3) || (ldc & 3) || !x3_al16(A) || !x3_al16(Bm) || !x3_al16(C)) return 1;
  if (kind == FC_GEMM_TN && (M & 7)) return 1;
  if (e.dbg) return 1;
  if ((e.bias && !x3_al16(e.bias)) || (e.res && !x3_al16(e.res)) || (e.preact && !x3_al16(e.preact)) || (e.gelu_in && !x3_al16(e.gelu_in)) || (e.pos && !x3_al16(e.pos)))
    return 1;
  const int tiles_n = fc_cdiv(N, BN), tiles = fc_cdiv(M, BM) * tiles_n;
  const int lds = 98304;
#define X3_GO(AM, BMo)                                                                                                  \
  do {                                                                                                                  \
    auto kfn = k_gemm_x3<AM, BMo>;                                                                                      \
    static bool attr_done = false;                                                                                      \
    if (!attr_done) {                                                                                                   \
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));             \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), lds, s, A, lda, Bm, ldb, C, ldc, M, N, K, tiles_n, e);              \
  } while (0)
  if (kind == FC_GEMM_NT) X3_GO(KC, KC);
  else if (kind == FC_GEMM_NN) X3_GO(KC, KR);
  else if (kind == FC_GEMM_TN) X3_GO(KR, KR);
  else return 1;
#undef X3_GO
  FC_LAUNCH_CHECK();
  return 0;
}
