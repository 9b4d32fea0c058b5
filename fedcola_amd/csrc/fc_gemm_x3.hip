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
// 128 x 128 output tile, 32-k tiles, 256 threads = 2 x 2 waves of 64 x 64 (4 x 4 v_mfma_f32_16x16x32_bf16 accumulators, operands swapped so
// that a lane owns 4 consecutive output columns), 48 KB of LDS: two or three workgroups per CU, so that one's split / store phase runs under
// another's 96 MFMAs per wave, and the 297-tile launches of the N = 384 linears fit the chip in one round.  Per k-tile a thread loads 4 + 4
// float4 of the two operands (the NEXT tile's loads are in flight under this tile's MFMAs), splits them and writes the three images of each
// operand, 6 x 8 KB: k-contiguous operands as [128 rows][32 k] (64-byte rows, 16-byte chunk g of row r at g ^ ((r >> 3) & 1) << 1: conflict-
// free ds_read_b128 fragments), k-row operands as the first 32 rows of the bf16 kernels' k-row layout (fc_mfma_dev.h, ds_read_b64_tr_b16).
#include <map>
#include <mutex>

#include "fc_kernels.h"
#include "fc_mfma_dev.h"

#define XK 32            // k per tile
#define XIMG 8192        // bytes of one bf16 image of an operand tile
struct X3Regs { float4 v[4]; };
__device__ __forceinline__ int kc32_off(int row, int c) { return row * 64 + ((c ^ (((row >> 3) & 1) << 1)) << 4); }

// One operand tile: KC = P[row][k] (k contiguous; thread owns the 8-k chunk c = tid & 3 of rows (tid >> 2) + 64 p), KR = P[k][col] (thread owns the
// 8-column chunk c = tid & 15 of k rows (tid >> 4) + 16 p).  Out-of-range rows / columns / k read as zero.
template <int MODE>
__device__ __forceinline__ void x3_load(X3Regs& R, const float* __restrict__ P, long ld, int r0, int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    bool ok;
    const float* src;
    if (MODE == KC) {
      const int row = r0 + (tid >> 2) + 64 * p, k = k0 + (tid & 3) * 8;
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
__device__ __forceinline__ void x3_store(const X3Regs& R, char* img, int tid) {   // img: hi | mid | lo images, XIMG apart
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    uint4 h, m, l;
    x3_split(R.v[2 * p].x, R.v[2 * p].y, h.x, m.x, l.x);
    x3_split(R.v[2 * p].z, R.v[2 * p].w, h.y, m.y, l.y);
    x3_split(R.v[2 * p + 1].x, R.v[2 * p + 1].y, h.z, m.z, l.z);
    x3_split(R.v[2 * p + 1].z, R.v[2 * p + 1].w, h.w, m.w, l.w);
    const int off = MODE == KC ? kc32_off((tid >> 2) + 64 * p, tid & 3) : kr_off((tid >> 4) + 16 * p, tid & 15);
    *(uint4*)(img + off) = h;
    *(uint4*)(img + XIMG + off) = m;
    *(uint4*)(img + 2 * XIMG + off) = l;
  }
}
// fragment of the 16-wide block at rb: lane (r = lane & 15, g = lane >> 4) gets k = 8 g .. 8 g + 7 of row (column) rb + r
template <int MODE>
__device__ __forceinline__ bf16x8 x3_frag(const char* img, int rb, int lane) {
  if (MODE == KC) return *(const bf16x8*)(img + kc32_off(rb + (lane & 15), lane >> 4));
  return frag_read<KR>(img, rb, 0, lane);
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

// one 32-k tile of the six-product sum from the LDS images into the running fp32 accumulators.
// The matrix pipe's fp32 accumulate rounds toward -inf (measured: a systematic error of -6e-11 |result| per k element when every product of a
// 12 608-long reduction lands in one accumulator -- harmless per element, but the column sums downstream add it 12 608 times).  So the six
// products of a tile go into a fresh accumulator that joins the running sum by a round-to-nearest VALU add, and every second tile runs with
// the A fragments negated and is subtracted: a rounding toward -inf becomes one toward +inf there, the two directions cancel on average.
template <int AMODE, int BMODE>
__device__ __forceinline__ void x3_compute(const char* ai, const char* bi, f32x4 (&acc)[4][4], int wm, int wn, int lane, bool odd) {
  const unsigned flip = odd ? 0x80008000u : 0u;
  f32x4 part[4][4];
#define X3_LOAD_A(f, c)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = x3_frag<AMODE>(ai + (c) * XIMG, wm * 64 + i * 16, lane);          \
  if (AMODE == KR) frag_fence(f);                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                        \
    uint4 u = *(uint4*)&f[i];                                                                                            \
    u.x ^= flip; u.y ^= flip; u.z ^= flip; u.w ^= flip;                                                                  \
    f[i] = *(bf16x8*)&u;                                                                                                 \
  }
#define X3_LOAD_B(f, c)                                                                                                  \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) f[j] = x3_frag<BMODE>(bi + (c) * XIMG, wn * 64 + j * 16, lane);          \
  if (BMODE == KR) frag_fence(f);
#define X3_MFMA0(fa_, fb_)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
  part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_[j], fa_[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
#define X3_MFMA(fa_, fb_)                                                                                                \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
  part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_[j], fa_[i], part[i][j], 0, 0, 0)
  // the 2^-16 terms first, the leading product last; at most four fragment sets (A hi, B hi, A mid, B mid) are live at a time
  bf16x8 ah[4], bh[4];
  X3_LOAD_A(ah, 0);
  {
    bf16x8 bl[4];
    X3_LOAD_B(bl, 2);
    X3_MFMA0(ah, bl);
  }
  __builtin_amdgcn_sched_barrier(0);
  X3_LOAD_B(bh, 0);
  {
    bf16x8 al[4];
    X3_LOAD_A(al, 2);
    X3_MFMA(al, bh);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    bf16x8 am[4], bm[4];
    X3_LOAD_A(am, 1);
    X3_LOAD_B(bm, 1);
    X3_MFMA(am, bm);
    X3_MFMA(am, bh);
    X3_MFMA(ah, bm);
  }
  __builtin_amdgcn_sched_barrier(0);
  X3_MFMA(ah, bh);
#undef X3_LOAD_A
#undef X3_LOAD_B
#undef X3_MFMA0
#undef X3_MFMA
  const float sgn = odd ? -1.0f : 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] += sgn * part[i][j];
}

template <int AMODE, int BMODE>
__global__ void __launch_bounds__(256, 2) k_gemm_x3(const float* __restrict__ A, long lda, const float* __restrict__ Bm, long ldb, float* __restrict__ C, long ldc,
                                                    int M, int N, int K, int tiles_n, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A: hi | mid | lo, then B: hi | mid | lo, 8 KB each
  char* ai = smem;
  char* bi = smem + 3 * XIMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int T = (K + XK - 1) / XK;
  X3Regs ra, rb;
  x3_load<AMODE>(ra, A, lda, m0, M, 0, K, tid);
  x3_load<BMODE>(rb, Bm, ldb, n0, N, 0, K, tid);
  for (int kt = 0; kt < T; ++kt) {
    x3_store<AMODE>(ra, ai, tid);
    x3_store<BMODE>(rb, bi, tid);
    lds_barrier();
    if (kt + 1 < T) {                                      // the next tile's loads fly under this tile's MFMAs
      x3_load<AMODE>(ra, A, lda, m0, M, (kt + 1) * XK, K, tid);
      x3_load<BMODE>(rb, Bm, ldb, n0, N, (kt + 1) * XK, K, tid);
    }
    x3_compute<AMODE, BMODE>(ai, bi, acc, wm, wn, lane, kt & 1);
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

// out[c][r] = in[r][c] for an fp32 matrix of `rows` x `cols` (row stride ld): 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256) k_x3_transpose(const float* __restrict__ in, long ld, float* __restrict__ out, int rows, int cols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    t[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < cols && r < rows) out[(size_t)c * rows + r] = t[tx][ty + 8 * i];
  }
}

// ======================================================================== weight gradients of the fp32 mode
// dW[out, in] = dY[rows, out]^T . X[rows, in] and db[out] = column sums of dY: the output has 9 - 72 tiles and the reduction runs over all
// the rows of the batch (12 608), so the reduction is cut into S slices (tiles x S ~ two workgroups per CU); slice s writes its raw partial
// tile to part[s], k_colsum_slices the fp64 column sums of a slice of dY's rows to colp[s] (in the product kernel they cost 48 spilled
// registers), and k_dw_x3_reduce adds the slices in a fixed order in fp64.  No atomics: the same bits every run.
__global__ void __launch_bounds__(256) k_colsum_slices(const float* __restrict__ dY, double* __restrict__ colp, int rows, int out, int rows_per) {
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, col = blockIdx.x * 64 + tx * 4, sl = blockIdx.y;
  const int r_end = (sl + 1) * rows_per < rows ? (sl + 1) * rows_per : rows;
  double a = 0., b = 0., c = 0., d = 0.;
  if (col < out)
    for (int r = sl * rows_per + ty; r < r_end; r += 16) {
      const float4 v = *(const float4*)(dY + (size_t)r * out + col);
      a += v.x; b += v.y; c += v.z; d += v.w;
    }
  red[ty][tx * 4] = a; red[ty][tx * 4 + 1] = b; red[ty][tx * 4 + 2] = c; red[ty][tx * 4 + 3] = d;
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < out) {
    double v = 0.;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r][threadIdx.x];
    colp[(size_t)sl * out + blockIdx.x * 64 + threadIdx.x] = v;
  }
}
__global__ void __launch_bounds__(256, 2) k_dw_x3(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part, int rows, int out, int in,
                                                  int tiles_n, int tiles, int kt_per) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ai = smem;
  char* bi = smem + 3 * XIMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int sl = blockIdx.x / tiles, t = blockIdx.x % tiles;
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int T = (rows + XK - 1) / XK;
  const int k_beg = sl * kt_per, k_end = (k_beg + kt_per < T) ? k_beg + kt_per : T;
  X3Regs ra, rb;
  if (k_beg < k_end) {
    x3_load<KR>(ra, dY, out, m0, out, k_beg * XK, rows, tid);
    x3_load<KR>(rb, X, in, n0, in, k_beg * XK, rows, tid);
  }
  for (int kt = k_beg; kt < k_end; ++kt) {
    x3_store<KR>(ra, ai, tid);
    x3_store<KR>(rb, bi, tid);
    lds_barrier();
    if (kt + 1 < k_end) {
      x3_load<KR>(ra, dY, out, m0, out, (kt + 1) * XK, rows, tid);
      x3_load<KR>(rb, X, in, n0, in, (kt + 1) * XK, rows, tid);
    }
    x3_compute<KR, KR>(ai, bi, acc, wm, wn, lane, kt & 1);
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
static std::map<std::pair<int, hipStream_t>, std::pair<char*, size_t>> g_x3_ws;      // keyed by (device, stream): the null stream / equal handles on two devices must not share an allocation
static int x3_scratch(hipStream_t s, size_t bytes, char** out) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  int dev = 0;
  FC_CHECK_HIP(hipGetDevice(&dev));
  auto& w = g_x3_ws[std::make_pair(dev, s)];
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
  const int tiles_n = fc_cdiv(in, BN), tiles = fc_cdiv(out, BM) * tiles_n, T = fc_cdiv(rows, XK);
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
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_dw_x3, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * XIMG));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_dw_x3, dim3(tiles * S), dim3(256), 6 * XIMG, s, dY, X, part, rows, out, in, tiles_n, tiles, kt_per);
  FC_LAUNCH_CHECK();
  if (colp) {
    hipLaunchKernelGGL(k_colsum_slices, dim3(fc_cdiv(out, 64), S), dim3(256), 0, s, dY, colp, rows, out, fc_cdiv(rows, S));
    FC_LAUNCH_CHECK();
  }
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
  const int lds = 6 * XIMG;
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
  else if (kind == FC_GEMM_NN) {
    // dX = dY . W with W[K][N] row-major: its tile would be a k-row image (transposing LDS reads, two per fragment, and 20 more live registers:
    // 364 against 536 TFLOP/s of MFMA rate on 12 608 x 384 x 1536).  W is small (<= 2.4 MB): transpose it into the stream's scratch and run the
    // k-contiguous form.
    static const int tr = fc_knob("FC_X3_NN_TRANSPOSE", 1);
    if (tr) {
      char* ws = nullptr;
      FC_TRY(x3_scratch(s, (size_t)N * K * sizeof(float), &ws));
      float* WT = (float*)ws;
      hipLaunchKernelGGL(k_x3_transpose, dim3(fc_cdiv(N, 32), fc_cdiv(K, 32)), dim3(256), 0, s, Bm, ldb, WT, K, N);
      FC_LAUNCH_CHECK();
      Bm = WT;
      ldb = K;
      X3_GO(KC, KC);
    } else {
      X3_GO(KC, KR);
    }
  } else if (kind == FC_GEMM_TN) X3_GO(KR, KR);
  else return 1;
#undef X3_GO
  FC_LAUNCH_CHECK();
  return 0;
}
