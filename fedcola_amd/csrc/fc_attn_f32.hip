// fp32 attention on the matrix cores for the library's precision = 'fp32' mode: head_dim 64, whole key range resident (N <= 224).
// Reference semantics: Attention.forward, /root/reference/src/models/mome.py:150-168 (scores and softmax in fp32).
//
// Every product is a chain of v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate; A: lane (c = lane & 15, g = lane >> 4) holds A[m = c][k = g],
// B: B[k = g][n = c], D: register v of lane (c, g) holds D[m = 4g + v][n = c]).  The reduction index of a product may be visited in any order,
// as long as both operands agree, which is what lets every operand be read 16 bytes at a time and every accumulator tile feed the next
// product with no lane movement:
//   * a product over the head dimension (S = Q.K^T, dP = dO.V^T) takes, in step (j, x) of 16, d = 16 j + 4 g + x from both sides: a lane reads
//     four float4 of its row (one 16-row block of the LDS tile, or its own query / key row from global) and feeds them element by element;
//   * a product over keys or queries (O = P.V, dQ = dS.K, dV = P^T.dO, dK = dS^T.Q) takes, in step v of 4 per 16-row block, row 4 g + v: exactly
//     the row whose probability sits in register v of the accumulator the lane already holds; the other operand is ONE float4 of that row of
//     the LDS tile (columns 4 c .. 4 c + 3: the four output blocks db use element db), and the output column m = c stands for head dimension
//     4 c + db -- so a lane ends up with 16 consecutive head dimensions of its row: 64-byte stores.
// The matrix pipe's fp32 accumulate rounds toward -inf (fc_gemm_x3.hip measured it): a systematic -3e-8 of every element that the column sums
// downstream (bias / LayerNorm gradients over 12 608 rows) add up linearly.  So every chain here is TWO chains: half of its steps run with one
// operand negated into a second accumulator and the result is their difference -- the rounding directions cancel, and the two independent
// chains also pipeline better.
// One workgroup (8 waves) per (batch, head); forward: K and V tiles in LDS; backward: TWO launches: the dQ workgroups (K, V in LDS) first --
// they also produce delta --, then the dK / dV workgroups (Q, dO in LDS, lse and delta from global); no atomics.  P is recomputed from the
// forward's log-sum-exp; nothing of size N x N goes to HBM.
// delta (round 6): the softmax backward is dS = P o (dP - delta) with delta_i = sum_j P_ij dP_ij.  The flash form delta_i = rowsum(dO_i o O_i) is
// the same number in exact arithmetic, but in fp32 the two differ by the rounding of two differently-ordered, cancelling sums, and then
// rowsum_j(dS_ij) is not ~0: the residual r_i enters dK as sum_i r_i Q_i -- a term that does not average out over the rows, so the column sums
// downstream (norm1.bias = colsum(dqkv) . W) carry it: 1.0e-4 of the worst tensor of the 768-wide parity case against the fp32 oracle's 2.7e-5
// (tools/fp32_colsum_probe.py: everything upstream of this kernel was 4x closer to exact than the oracle, its output 2x further).  So delta is
// formed the way torch's softmax backward forms it: from the SAME P and dP the dS uses, delta_i = (sum_j P_ij dP_ij) / (sum_j P_ij) (a first pass of
// the dQ workgroup over its keys; the dK / dV side recomputes bit-identical P and dP: the products commute and both sides visit the head
// dimension in the same order).
// LDS tiles are [rows][64] fp32 with 256-byte rows and the 16-byte chunk q of row r stored at chunk q ^ sig(r), sig(r) = (r & 3) | tau((r >> 2) & 3) << 2,
// tau = (0, 3, 1, 2): conflict-free for BOTH read forms (row reads: 16 rows x one chunk; row-group reads: 4 rows x 16 chunks) under
// ds_read_b128's lane grouping {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
#include "fc_kernels.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define AW 8          // waves per workgroup

__device__ __forceinline__ int sig(int row) { return (row & 3) | (((0x9C >> (((row >> 2) & 3) * 2)) & 3) << 2); }
__device__ __forceinline__ const float4* at4(const float* T, int row, int chunk) { return (const float4*)(T + row * 64 + ((chunk ^ sig(row)) << 2)); }

// stage a [N][64] fp32 slice (row stride ld floats) into a swizzled tile of NP rows (zero rows past N)
__device__ __forceinline__ void stage_f32(float* T, const float* __restrict__ src, long ld, int N, int NP, int tid) {
  for (int idx = tid; idx < NP * 16; idx += 64 * AW) {
    const int row = idx >> 4, q = idx & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N) v = *(const float4*)(src + (size_t)row * ld + q * 4);
    *(float4*)(T + row * 64 + ((q ^ sig(row)) << 2)) = v;
  }
}
// the four float4 of a lane's own row that a head-dimension product consumes: elements d = 16 j + 4 g + x
__device__ __forceinline__ void row_frag_g(float4 (&f)[4], const float* __restrict__ base, long ld, int row, int N, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = row < N ? *(const float4*)(base + (size_t)row * ld + 16 * j + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void row_frag_l(float4 (&f)[4], const float* T, int row, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = *at4(T, row, 4 * j + g);
}
__device__ __forceinline__ void neg_frag(float4 (&n)[4], const float4 (&a)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) n[j] = make_float4(-a[j].x, -a[j].y, -a[j].z, -a[j].w);
}
// D[m = a's row][n = b's row] = sum over the head dimension; nb = -b (steps j = 1, 3 go to the second accumulator)
__device__ __forceinline__ f32x4 dot_hd(const float4 (&a)[4], const float4 (&b)[4], const float4 (&nb)[4]) {
  f32x4 p = {0.f, 0.f, 0.f, 0.f}, n = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    p = MF(a[j].x, b[j].x, p);
    n = MF(a[j + 1].x, nb[j + 1].x, n);
    p = MF(a[j].y, b[j].y, p);
    n = MF(a[j + 1].y, nb[j + 1].y, n);
    p = MF(a[j].z, b[j].z, p);
    n = MF(a[j + 1].z, nb[j + 1].z, n);
    p = MF(a[j].w, b[j].w, p);
    n = MF(a[j + 1].w, nb[j + 1].w, n);
  }
  return p - n;
}
// out[db][.] (column m = c <-> head dimension 4 c + db) += sum over the 16 rows rb .. rb + 15 of T of T[row][4 c + db] * w[row's register]
__device__ __forceinline__ void acc_rows(f32x4 (&out)[4], const float* T, int rb, const f32x4& w, int c, int g) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float4 a = *at4(T, rb + 4 * g + v, c);
    out[0] = MF(a.x, w[v], out[0]);
    out[1] = MF(a.y, w[v], out[1]);
    out[2] = MF(a.z, w[v], out[2]);
    out[3] = MF(a.w, w[v], out[3]);
  }
}
// a lane's 16 consecutive head dimensions 16 g .. 16 g + 15 of its row: element (v, db) = out[db][v]
__device__ __forceinline__ void store_row16(float* __restrict__ p, const f32x4 (&op)[4], const f32x4 (&on)[4], float mul) {
#pragma unroll
  for (int v = 0; v < 4; ++v)
    *(float4*)(p + 4 * v) = make_float4((op[0][v] - on[0][v]) * mul, (op[1][v] - on[1][v]) * mul, (op[2][v] - on[2][v]) * mul, (op[3][v] - on[3][v]) * mul);
}

// ======================================================================== forward
template <int NF>
__global__ void __launch_bounds__(64 * AW) k_attn_f32_fwd(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse, int B, int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  constexpr int NP = 16 * NF;
  float* Ks = smf;
  float* Vs = smf + NP * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const float* base = qkv + (size_t)b * N * D3 + h * 64;
  stage_f32(Ks, base + Dm, D3, N, NP, tid);
  stage_f32(Vs, base + 2 * Dm, D3, N, NP, tid);
  __syncthreads();
  const int nqb = (N + 15) >> 4;
  for (int qb = wave; qb < nqb; qb += AW) {
    const int qrow = qb * 16 + c;
    float4 qf[4], nqf[4];
    row_frag_g(qf, base, D3, qrow, N, g);
    neg_frag(nqf, qf);
    f32x4 s[NF];
    float m = -INFINITY;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float4 kf[4];
      row_frag_l(kf, Ks, f * 16 + c, g);
      f32x4 a = dot_hd(kf, qf, nqf);                                  // S^T[key = 16 f + 4 g + v][q = c], raw
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        a[v] = (f * 16 + 4 * g + v < N) ? a[v] * scale : -INFINITY;
        m = fmaxf(m, a[v]);
      }
      s[f] = a;
      __builtin_amdgcn_sched_barrier(0);     // keep one key block's fragments live at a time (the unrolled loop otherwise hoists every block's reads: spills)
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int v = 0; v < 4; ++v) { const float p = expf(s[f][v] - m); s[f][v] = p; sum += p; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 oacc[4], onac[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; onac[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int f = 0; f < NF; ++f) {                                          // O[q = c][d] over the keys (padded keys: p = 0); odd blocks negated
      if (f & 1) acc_rows(onac, Vs, f * 16, -s[f], c, g);
      else acc_rows(oacc, Vs, f * 16, s[f], c, g);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qrow < N) {
      store_row16(o + ((size_t)b * N + qrow) * Dm + h * 64 + 16 * g, oacc, onac, 1.0f / sum);
      if (g == 0) lse[((size_t)b * H + h) * N + qrow] = m + logf(sum);
    }
  }
}

// ======================================================================== backward
__device__ __forceinline__ void attn_f32_dq_body(float* smf, int bh, const float* __restrict__ qkv, const float* __restrict__ dout,
                                                 const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, int N, int H, int nf, float scale) {
  const int NP = 16 * nf;
  float* Ks = smf;
  float* Vs = smf + NP * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = bh / H, h = bh % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const float* base = qkv + (size_t)b * N * D3 + h * 64;
  const float* dobase = dout + (size_t)b * N * Dm + h * 64;
  stage_f32(Ks, base + Dm, D3, N, NP, tid);
  stage_f32(Vs, base + 2 * Dm, D3, N, NP, tid);
  __syncthreads();
  const int nqb = (N + 15) >> 4;
  for (int qb = wave; qb < nqb; qb += AW) {
    const int qrow = qb * 16 + c;
    float4 qf[4], dof[4];
    row_frag_g(qf, base, D3, qrow, N, g);
    row_frag_g(dof, dobase, Dm, qrow, N, g);
    const float lq = qrow < N ? lse[((size_t)b * H + h) * N + qrow] : 1e30f;
    float4 nqf[4], ndof[4];
    neg_frag(nqf, qf);
    neg_frag(ndof, dof);
    // pass 1: delta[q = c] = (sum_j P dP) / (sum_j P) over the keys (fixed order: v, key block, then the four lane groups).  The division matters:
    // P = exp(S - lse) sums to 1 + eps (the fp32 log-sum-exp), and dP_ij = dO_i . V_j carries a component common to all j; with delta = sum P dP
    // alone that component survives in dS as eps x common mode (measured: 4x WORSE than the flash form), divided by sum P it cancels exactly
    // and rowsum_j(dS_ij) = sum P dP - delta sum P = 0 to the last bit of the two sums -- what torch's softmax backward gets from its normalised P.
    float sd = 0.f, sp = 0.f;
    for (int f = 0; f < nf; ++f) {
      float4 kf[4], vf[4];
      row_frag_l(kf, Ks, f * 16 + c, g);
      row_frag_l(vf, Vs, f * 16 + c, g);
      const f32x4 st = dot_hd(kf, qf, nqf);
      const f32x4 dpt = dot_hd(vf, dof, ndof);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (16 * f + 4 * g + v < N) { const float pv = expf(st[v] * scale - lq); sp += pv; sd += pv * dpt[v]; }
    }
    sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
    sp += __shfl_xor(sp, 16, 64); sp += __shfl_xor(sp, 32, 64);
    const float dl = qrow < N ? sd / sp : 0.f;                          // delta[q = c]
    if (g == 0 && qrow < N) delta[((size_t)b * H + h) * N + qrow] = dl;
    f32x4 dq[4], dqn[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dq[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dqn[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int f = 0; f < nf; ++f) {
      float4 kf[4], vf[4];
      row_frag_l(kf, Ks, f * 16 + c, g);
      row_frag_l(vf, Vs, f * 16 + c, g);
      const f32x4 st = dot_hd(kf, qf, nqf);                               // S^T[key][q], raw
      const f32x4 dpt = dot_hd(vf, dof, ndof);                            // dP^T[key][q]
      f32x4 ds;
#pragma unroll
      for (int v = 0; v < 4; ++v)      // padded keys are masked explicitly, as in the forward: with lse below about -88 their exp overflows and inf x 0 (their zero K rows) is NaN
        ds[v] = (16 * f + 4 * g + v < N) ? expf(st[v] * scale - lq) * (dpt[v] - dl) : 0.f;
      if (f & 1) acc_rows(dqn, Ks, f * 16, -ds, c, g);                      // dQ[q = c][d] over the keys; odd blocks negated
      else acc_rows(dq, Ks, f * 16, ds, c, g);
    }
    if (qrow < N) store_row16(dqkv + ((size_t)b * N + qrow) * D3 + h * 64 + 16 * g, dq, dqn, scale);
  }
}

__device__ __forceinline__ void attn_f32_dkv_body(float* smf, int bh, const float* __restrict__ qkv, const float* __restrict__ dout,
                                                  const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv, int N, int H, int nf, float scale) {
  const int NP = 16 * nf;
  float* Qs = smf;
  float* Ds = smf + NP * 64;
  float* lse_s = Ds + NP * 64;
  float* del_s = lse_s + NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int b = bh / H, h = bh % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const float* base = qkv + (size_t)b * N * D3 + h * 64;
  const float* dobase = dout + (size_t)b * N * Dm + h * 64;
  stage_f32(Qs, base, D3, N, NP, tid);
  stage_f32(Ds, dobase, Dm, N, NP, tid);
  for (int row = tid; row < NP; row += 64 * AW) {            // lse and delta (the dQ launch wrote it) of every query; padded queries: P = exp(. - 1e30) = 0
    lse_s[row] = row < N ? lse[((size_t)b * H + h) * N + row] : 1e30f;
    del_s[row] = row < N ? delta[((size_t)b * H + h) * N + row] : 0.f;
  }
  __syncthreads();
  const int nkb = (N + 15) >> 4;
  for (int kb = wave; kb < nkb; kb += AW) {
    const int krow = kb * 16 + c;
    float4 kf[4], vf[4];
    row_frag_g(kf, base + Dm, D3, krow, N, g);
    row_frag_g(vf, base + 2 * Dm, D3, krow, N, g);
    float4 nkf[4], nvf[4];
    neg_frag(nkf, kf);
    neg_frag(nvf, vf);
    f32x4 dv[4], dk[4], dvn[4], dkn[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      dv[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dvn[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkn[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int f = 0; f < nf; ++f) {
      float4 qf[4], df[4];
      row_frag_l(qf, Qs, f * 16 + c, g);
      row_frag_l(df, Ds, f * 16 + c, g);
      const f32x4 sa = dot_hd(qf, kf, nkf);                               // S[q = 16 f + 4 g + v][key = c], raw
      const f32x4 dpa = dot_hd(df, vf, nvf);                              // dP[q][key]
      const float4 l4 = *(const float4*)(lse_s + f * 16 + 4 * g), d4 = *(const float4*)(del_s + f * 16 + 4 * g);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
      f32x4 p, ds;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        p[v] = expf(sa[v] * scale - lv[v]);
        ds[v] = p[v] * (dpa[v] - dl[v]);
      }
      if (f & 1) {                                                         // odd query blocks negated
        acc_rows(dvn, Ds, f * 16, -p, c, g);
        acc_rows(dkn, Qs, f * 16, -ds, c, g);
      } else {
        acc_rows(dv, Ds, f * 16, p, c, g);                                 // dV[key = c][d] over the queries
        acc_rows(dk, Qs, f * 16, ds, c, g);                                // dK[key = c][d]
      }
    }
    if (krow < N) {
      float* row = dqkv + ((size_t)b * N + krow) * D3 + h * 64 + 16 * g;
      store_row16(row + Dm, dk, dkn, scale);
      store_row16(row + 2 * Dm, dv, dvn, 1.0f);
    }
  }
}

template <int PART>      // 0: the dQ workgroups (write delta), 1: the dK / dV workgroups (read it): two launches, in this order
__global__ void __launch_bounds__(64 * AW) k_attn_f32_bwd(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, float* __restrict__ dqkv, int B, int N, int H, int nf, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  if (PART == 0) attn_f32_dq_body(smf, blockIdx.x, qkv, dout, lse, delta, dqkv, N, H, nf, scale);
  else attn_f32_dkv_body(smf, blockIdx.x, qkv, dout, lse, delta, dqkv, N, H, nf, scale);
}

// ======================================================================== launchers (1 = shape not covered: the caller takes the VALU kernels)
static int f32_nf(int N) { return N <= 32 ? 2 : N <= 48 ? 3 : N <= 64 ? 4 : N <= 208 ? 13 : N <= 224 ? 14 : 0; }
static bool f32_ok(int N, int d, const void* a, const void* b, const void* c) {
  return d == 64 && f32_nf(N) != 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
template <int NF>
static int launch_f32_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, float scale, hipStream_t s) {
  const int lds = 2 * 16 * NF * 256;
  auto k = k_attn_f32_fwd<NF>;
  static bool done = false;
  if (!done) { FC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  hipLaunchKernelGGL(k, dim3(B * H), dim3(64 * AW), lds, s, qkv, o, lse, B, N, H, scale);
  FC_LAUNCH_CHECK();
  return 0;
}
int fc_attn_f32_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, int d, float scale, hipStream_t s) {
  static const int on = fc_knob("FC_ATTN_F32_MFMA", 1);
  if (!on || !f32_ok(N, d, qkv, o, qkv)) return 1;
  switch (f32_nf(N)) {
    case 2: return launch_f32_fwd<2>(qkv, o, lse, B, N, H, scale, s);
    case 3: return launch_f32_fwd<3>(qkv, o, lse, B, N, H, scale, s);
    case 4: return launch_f32_fwd<4>(qkv, o, lse, B, N, H, scale, s);
    case 13: return launch_f32_fwd<13>(qkv, o, lse, B, N, H, scale, s);
    case 14: return launch_f32_fwd<14>(qkv, o, lse, B, N, H, scale, s);
  }
  return 1;
}
int fc_attn_f32_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* delta, float* dqkv, int B, int N, int H, int d, float scale,
                    hipStream_t s) {
  static const int on = fc_knob("FC_ATTN_F32_MFMA", 1);
  (void)o;                                                  // (delta no longer comes from rowsum(dO o O): see the header)
  if (!on || !delta || !f32_ok(N, d, qkv, dout, dqkv)) return 1;
  const int nf = f32_nf(N), lds = 2 * 16 * nf * 256 + 2 * 16 * nf * 4;
  static int lds_set = 0;
  if (lds > lds_set) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_attn_f32_bwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_attn_f32_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(k_attn_f32_bwd<0>, dim3(B * H), dim3(64 * AW), lds, s, qkv, dout, lse, delta, dqkv, B, N, H, nf, scale);
  hipLaunchKernelGGL(k_attn_f32_bwd<1>, dim3(B * H), dim3(64 * AW), lds, s, qkv, dout, lse, delta, dqkv, B, N, H, nf, scale);
  FC_LAUNCH_CHECK();
  return 0;
}
