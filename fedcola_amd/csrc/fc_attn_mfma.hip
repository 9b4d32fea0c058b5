// bf16 MFMA attention for gfx950, head_dim 64, whole key range resident (N <= 256: ViT tokens 197, captions <= 64).
// Reference semantics: Attention.forward, /root/reference/src/models/mome.py:150-168 (scores and softmax in fp32).
//
// One workgroup (4 waves) per (batch, head).  K and V (and, for the backward, Q and dO) tiles [Npad][64] bf16 sit in
// LDS as 128-B rows with ONE swizzle that serves both read forms without bank conflicts (chunk c of row r at
// c ^ hd(r), hd below): row reads (ds_read_b128, MFMA operands that are k-contiguous) and transposed 4x16 block reads
// (ds_read_b64_tr_b16, operands whose k index is the tile's row).
//
// Every product is arranged so that an accumulator tile feeds the next MFMA as an operand with no lane movement:
//   fwd : S^T = K.Q^T (rows = keys, cols = queries) -> softmax down the rows (in-lane + 2 shuffles) -> O^T = V^T.P^T
//   bwd1: S^T, dP^T = V.dO^T -> dS^T -> dQ = (dS^T)^T.K            (waves split the query blocks)
//   bwd2: S = Q.K^T, dP = dO.V^T -> P, dS -> dV = P^T.dO, dK = dS^T.Q   (waves split the key blocks; no atomics)
// P is recomputed from the forward's log-sum-exp; nothing of size N x N ever goes to HBM.
#include <stdlib.h>
#include <string.h>

#include "fc_kernels.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS3(p) ((__attribute__((address_space(3))) s16x4*)(p))

__device__ __forceinline__ int hd(int row) {
  int t = (row >> 1) & 7;
  return ((((t >> 1) ^ (t >> 2)) & 1) << 2) | ((t & 1) << 1) | (t >> 2);
}
__device__ __forceinline__ int at_off(int row, int c) { return row * 128 + ((c ^ hd(row)) << 4); }

// row read: lane (r = lane&15, g = lane>>4) gets T[rb + r][32*ks + 8g .. +7]
__device__ __forceinline__ bf16x8 row_frag(const char* T, int rb, int ks, int lane) {
  return *(const bf16x8*)(T + at_off(rb + (lane & 15), ks * 4 + (lane >> 4)));
}
// transposed read for a 32-row k-step starting at row r0: lane (c = lane&15, g) gets
//   { T[r0 + 4g + j][cb + c] (j = 0..3), T[r0 + 16 + 4g + j][cb + c] (j = 0..3) }
__device__ __forceinline__ bf16x8 tr_frag(const char* T, int r0, int cb, int lane) {
  int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  int col = cb + 4 * p, c = col >> 3, cbyte = (col & 7) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(T + at_off(r0 + 4 * g + q, c) + cbyte));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(T + at_off(r0 + 16 + 4 * g + q, c) + cbyte));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  uint4 u = make_uint4(f2bf2(a[0], a[1]), f2bf2(a[2], a[3]), f2bf2(b[0], b[1]), f2bf2(b[2], b[3]));
  return *(bf16x8*)&u;
}
// stage a [N][64] slice (row stride ld elements) into a swizzled LDS tile of NP rows (zero rows past N).  All global loads
// of the tile are issued before the first LDS write (clamped row + select instead of a branch), so the staging pays ONE
// memory round trip instead of one per 256-thread sweep.
template <int NP, int NT = 256>
__device__ __forceinline__ void stage_tile(char* T, const bf16_t* __restrict__ src, long ld, int N, int tid) {
  constexpr int IT = (NP * 8 + NT - 1) / NT;
  uint4 v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int idx = tid + NT * i, row = idx >> 3, c = idx & 7;
    int rc = row < N ? row : N - 1;
    v[i] = *(const uint4*)(src + (size_t)rc * ld + c * 8);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int idx = tid + NT * i, row = idx >> 3, c = idx & 7;
    if (NP * 8 % NT == 0 || row < NP) *(uint4*)(T + at_off(row, c)) = row < N ? v[i] : make_uint4(0u, 0u, 0u, 0u);
  }
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ======================================================================== forward
// 8 waves per (batch, head): a wave owns the query blocks {wave, wave + 8, ...} (two at N = 197) and requests their Q fragments BEFORE the
// K / V staging, so that one memory round trip covers all of a workgroup's loads (the 4-wave form of rounds 1-2 walked four query blocks per
// wave and paid a dependent Q load in each: 18.8 us at B = 64 for 3.8 GFLOP).
#define AF_WAVES 8
#define AF_QB 2        // query blocks per wave held in registers (N <= 16 * AF_WAVES * AF_QB = 256)
// NV = key blocks that hold real keys (compile time, so that the block loop stays straight-line code: a run-time test per block
// serialises it into read -> wait -> MFMA -> wait chains); MASKALL = false promises N > 16 (NV - 1): only block NV - 1 straddles N.
template <int NF, int NV = NF, bool MASKALL = true>  // key fragments of 16 (Npad = 16*NF, NF even)
__global__ void __launch_bounds__(64 * AF_WAVES) k_attn_fwd_mfma(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse, int B, int N,
                                                                 int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = 16 * NF;
  char* Ks = smem;
  char* Vs = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cl = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const bf16_t* base = qkv + (size_t)b * N * D3 + h * 64;
  const int nqb = (N + 15) >> 4;
  bf16x8 qf[AF_QB][2];
#pragma unroll
  for (int r = 0; r < AF_QB; ++r) {
    const int qrow = (wave + AF_WAVES * r) * 16 + cl;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (qrow < N) v = *(const uint4*)(base + (size_t)qrow * D3 + ks * 32 + g * 8);
      qf[r][ks] = *(bf16x8*)&v;
    }
  }
  stage_tile<NP, 64 * AF_WAVES>(Ks, base + Dm, D3, N, tid);
  stage_tile<NP, 64 * AF_WAVES>(Vs, base + 2 * Dm, D3, N, tid);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < AF_QB; ++r) {
    const int qb = wave + AF_WAVES * r;
    if (qb >= nqb) break;
    const int qrow = qb * 16 + cl;
    f32x4 s[NF];
    float m = -INFINITY;
    // key blocks past N (block 13 of 14 at N = 197) are skipped, and only the block that straddles N pays the mask
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 a = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (f < NV) {
        a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a = MFMA(row_frag(Ks, f * 16, ks, lane), qf[r][ks], a);   // S^T[key = 16f+4g+x][q = cl]
        if (MASKALL || f == NV - 1) {
#pragma unroll
          for (int x = 0; x < 4; ++x) a[x] = (f * 16 + 4 * g + x < N) ? a[x] : -INFINITY;   // raw scores: the scale rides in the exponent's fma below
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) m = fmaxf(m, a[x]);
      }
      s[f] = a;
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
    const float sc2 = scale * 1.4426950408889634f, m2 = m * sc2;                    // exp(scale (s - m)) = 2^(s sc2 - m sc2): one fma + v_exp_f32 per score
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f < NV) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][x], sc2, -m2)); s[f][x] = p; sum += p; }
      } else {
        s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < (NV + 1) / 2; ++ss) {
      bf16x8 pb = pack8(s[2 * ss], s[2 * ss + 1]);                      // P^T[key(8g+j)][q = cl]
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[db] = MFMA(tr_frag(Vs, 32 * ss, db * 16, lane), pb, oacc[db]);   // O^T[d = 16db+4g+x][q]
    }
    if (qrow < N) {
      float inv = 1.0f / sum;
      bf16_t* orow = o + ((size_t)b * N + qrow) * Dm + h * 64 + 4 * g;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        *(uint2*)(orow + db * 16) = make_uint2(f2bf2(oacc[db][0] * inv, oacc[db][1] * inv), f2bf2(oacc[db][2] * inv, oacc[db][3] * inv));
      }
      if (g == 0) lse[((size_t)b * H + h) * N + qrow] = m * scale + __logf(sum);
    }
  }
}

// ======================================================================== backward
#ifdef FC_PROBES
__device__ long long* g_ab_stamps = nullptr;      // tools build: s_memtime stamps of workgroup 0 of each body ([body][wave][64])
#define AB_STAMP(body, k) do { long long* sp_ = g_ab_stamps; if (sp_ && bh == 0 && lane == 0) sp_[(body) * 512 + wave * 64 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_STAMP(body, k) do {} while (0)
#endif
// Two bodies, each with two tiles (56 KB for N = 197) in LDS so that two workgroups share a CU, one workgroup per (batch, head)
// and body:
//   k_attn_bwd_dq : K, V tiles in LDS; Q / dO / O fragments straight from global; waves split the query blocks
//   k_attn_bwd_dkv: Q, dO tiles in LDS; K / V fragments straight from global; waves split the key blocks (no atomics)
// delta = rowsum(dO * O) is recomputed inside both kernels (no separate pass, no delta round trip through HBM).
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* __restrict__ base, long ld, int row, int N, int ks, int g) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (row < N) v = *(const uint4*)(base + (size_t)row * ld + ks * 32 + g * 8);
  return *(bf16x8*)&v;
}
// branch-free form: the row is clamped and the fragment of a row past N zeroed afterwards (no exec-mask branch between the loads)
__device__ __forceinline__ bf16x8 gfrag_nb(const bf16_t* __restrict__ base, long ld, int row, int N, int ks, int g) {
  const int rc = row < N ? row : N - 1;
  return *(const bf16x8*)(base + (size_t)rc * ld + ks * 32 + g * 8);
}
// the same loads as inline asm: hipcc sinks plain loads towards their first use (below an LDS-DMA issue, where waiting for them would drain the
// DMA); these stay where they are written.  The compiler does not count them: the consumer waits with frag_wait<K>() (K = younger VMEM operations).
__device__ __forceinline__ void gload16_asm(bf16x8& dst, const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload4_asm(float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
template <int K>
__device__ __forceinline__ void frag_wait(bf16x8 (&a)[2], bf16x8 (&b)[2], bf16x8 (&c)[2], float& l) {
  asm volatile("s_waitcnt vmcnt(%7)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]), "+v"(l) : "n"(K) : "memory");
}
__device__ __forceinline__ bf16x8 zero_past(bf16x8 v, bool ok) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? v : z;
}
__device__ __forceinline__ float dot8(const bf16x8& a, const bf16x8& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += bf2f((bf16_t)a[i]) * bf2f((bf16_t)b[i]);
  return s;
}

// Register blocking: a wave works on a PAIR of 16-row blocks (32 queries in the dQ body, 32 keys in the dK/dV body) at a time, so
// every K / V (Q / dO) fragment read from LDS feeds two MFMAs instead of one: half the LDS reads per MFMA of the one-block form
// (the LDS port, not the matrix pipe, bounded that form), and twice the independent MFMA chains per step to cover their latency.
// PIPE (U == 1): software pipeline over the key pairs: the S^T / dP^T products of pair ss + 1 are issued BEFORE the exponent,
// the packing and the dQ products of pair ss, so that the matrix pipe works on the next pair's scores while the VALU turns this pair's
// into dS -- two independent chains per wave instead of one (read -> MFMA -> exp -> pack -> MFMA).
template <int NF, int U, int NW, int PIPE = 0, bool EARLY = false>
__device__ __forceinline__ void attn_bwd_dq_body(char* smem, int bh, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                 const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int B, int N,
                                                 int H, float scale) {
  constexpr int NP = 16 * NF;
  char* Ks = smem;
  char* Vs = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cl = lane & 15;
  const float sc2 = scale * 1.4426950408889634f;   // exp(scale s - lse) = 2^(s sc2 - lse log2 e)
  const int b = bh / H, h = bh % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const bf16_t* base = qkv + (size_t)b * N * D3 + h * 64;
  const bf16_t* obase = o + (size_t)b * N * Dm + h * 64;
  const bf16_t* dobase = dout + (size_t)b * N * Dm + h * 64;
  AB_STAMP(0, 0);
  bf16x8 qf[U][2], dof[U][2];
  float dl[U], lq[U];
  int qrow[U];
  bf16x8 of_[U][2];
  float lraw[U];
  auto frag_issue = [&](int qp) {    // the query group's Q / dO / O fragments and lse: loads only, branch-free, nothing consumed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      qrow[u] = (U * qp + u) * 16 + cl;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[u][ks] = gfrag_nb(base, D3, qrow[u], N, ks, g);
        dof[u][ks] = gfrag_nb(dobase, Dm, qrow[u], N, ks, g);
        of_[u][ks] = gfrag_nb(obase, Dm, qrow[u], N, ks, g);
      }
      lraw[u] = lse[((size_t)b * H + h) * N + (qrow[u] < N ? qrow[u] : N - 1)];
    }
  };
  auto frag_finish = [&]() {         // delta = rowsum(dO * O), lse in log2 units, rows past N zeroed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = qrow[u] < N;
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[u][ks] = zero_past(qf[u][ks], ok);
        dof[u][ks] = zero_past(dof[u][ks], ok);
        acc += dot8(dof[u][ks], of_[u][ks]);
      }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);                                    // delta[q = cl]
      dl[u] = acc;
      lq[u] = ok ? lraw[u] * 1.4426950408889634f : 1e30f;                // log2 units, like sc2
    }
  };
  auto fragments = [&](int qp) { frag_issue(qp); frag_finish(); };
  // EARLY: the first group's fragments are requested BEFORE the K / V staging (one memory round trip for both, as in the forward kernel)
  // and consumed after it; a later group's right after the previous group's pair loop, ahead of its stores
  if (EARLY && wave * 16 * U < N) frag_issue(wave);
  if (EARLY) __builtin_amdgcn_sched_barrier(0);
  stage_tile<NP, 64 * NW>(Ks, base + Dm, D3, N, tid);
  stage_tile<NP, 64 * NW>(Vs, base + 2 * Dm, D3, N, tid);
  __syncthreads();
  AB_STAMP(0, 1);
  if (EARLY && wave * 16 * U < N) frag_finish();
  bf16_t* dbase = dqkv + (size_t)b * N * D3 + h * 64;
  for (int qp = wave; qp < NF / U; qp += NW) {           // query group: rows 16 U qp .. 16 U (qp + 1) - 1
    if (qp * 16 * U >= N) break;
    AB_STAMP(0, 2 + 4 * (qp / NW));
    if (!EARLY) fragments(qp);
    f32x4 dq[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[u][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    AB_STAMP(0, 3 + 4 * (qp / NW) + (dl[0] == 123.456f));      // (the comparison makes the stamp wait for the fragments)
    if (PIPE && U == 1) {
      f32x4 st[2], dpt[2];
      auto scores = [&](int ss, f32x4 (&st_)[2], f32x4 (&dpt_)[2]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          st_[hh] = (f32x4){0.f, 0.f, 0.f, 0.f}; dpt_[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            st_[hh] = MFMA(row_frag(Ks, (2 * ss + hh) * 16, ks, lane), qf[0][ks], st_[hh]);
            dpt_[hh] = MFMA(row_frag(Vs, (2 * ss + hh) * 16, ks, lane), dof[0][ks], dpt_[hh]);
          }
        }
      };
      scores(0, st, dpt);
#pragma unroll 1
      for (int ss = 0; ss < NF / 2; ++ss) {
        f32x4 stn[2], dptn[2];
        const int sn = ss + 1 < NF / 2 ? ss + 1 : ss;      // (the last iteration recomputes its own pair: branch-free, result unused)
        scores(sn, stn, dptn);
        f32x4 ds[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int x = 0; x < 4; ++x) ds[hh][x] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[hh][x], sc2, -lq[0])) * (dpt[hh][x] - dl[0]);
        const bf16x8 bfg = pack8(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[0][db] = MFMA(tr_frag(Ks, 32 * ss, db * 16, lane), bfg, dq[0][db]);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) { st[hh] = stn[hh]; dpt[hh] = dptn[hh]; }
      }
    } else
#pragma unroll 1
    for (int ss = 0; ss < NF / 2; ++ss) {                  // key pair: keys 32 ss .. 32 ss + 31
      bf16x8 kf[2][2], vf[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[hh][ks] = row_frag(Ks, (2 * ss + hh) * 16, ks, lane); vf[hh][ks] = row_frag(Vs, (2 * ss + hh) * 16, ks, lane); }
      bf16x8 bfg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f32x4 ds[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            st = MFMA(kf[hh][ks], qf[u][ks], st);          // S^T[key][q]
            dpt = MFMA(vf[hh][ks], dof[u][ks], dpt);       // dP^T[key][q]
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) ds[hh][x] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[x], sc2, -lq[u])) * (dpt[x] - dl[u]);
        }
        bfg[u] = pack8(ds[0], ds[1]);                      // B[k = key(8g+j)][col = q = cl]
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 kt = tr_frag(Ks, 32 * ss, db * 16, lane);
#pragma unroll
        for (int u = 0; u < U; ++u) dq[u][db] = MFMA(kt, bfg[u], dq[u][db]);   // dQ^T[d = 16db+4g+x][q = cl]
      }
    }
    AB_STAMP(0, 4 + 4 * (qp / NW) + (dq[0][0][0] == 123.456f));
    uint2 pk[U][4];
    bf16_t* prow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      prow[u] = qrow[u] < N ? dbase + (size_t)qrow[u] * D3 + 4 * g : nullptr;
#pragma unroll
      for (int db = 0; db < 4; ++db) pk[u][db] = make_uint2(f2bf2(dq[u][db][0] * scale, dq[u][db][1] * scale), f2bf2(dq[u][db][2] * scale, dq[u][db][3] * scale));
    }
    const bool more = EARLY && qp + NW < NF / U && (qp + NW) * 16 * U < N;
    if (more) frag_issue(qp + NW);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (prow[u]) {       // a lane owns 4 consecutive head dims of its query row: 8-byte stores
#pragma unroll
        for (int db = 0; db < 4; ++db) *(uint2*)(prow[u] + db * 16) = pk[u][db];
      }
    if (more) frag_finish();
  }
  AB_STAMP(0, 20);
}

// ---- chunk-pipelined form of the dQ body (round 5): K and V go to LDS by LDS-DMA (buffer_load ... lds: no registers, the swizzle is applied on
// the SOURCE side -- lane l of 1-KB piece p fetches the logical chunk that physical slot (row 8p + l / 8, l % 8) holds), one 32-key chunk per wave
// and key pair in the order the pair loop consumes them, behind counted s_waitcnt vmcnt(6 - ss): pair ss starts when ITS chunk has landed while
// the chunks of the later pairs are still in flight.  Every LDS read of the loop is inline asm (hipcc puts s_waitcnt vmcnt(0) in front of any LDS
// access it sees while an LDS-DMA is in flight, fc_mfma_dev.h) with explicit lgkmcnt fences that tie the fragments.
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
__device__ __forceinline__ bf16x8 row_frag_asm(const char* T, int rb, int ks, int lane) {
  bf16x8 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(T + at_off(rb + (lane & 15), ks * 4 + (lane >> 4)))) : "memory");
  return f;
}
__device__ __forceinline__ void tr_frag_asm(const char* T, int r0, int cb, int lane, s16x4& lo, s16x4& hi) {
  int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  int col = cb + 4 * p, c = col >> 3, cbyte = (col & 7) * 2;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(T + at_off(r0 + 4 * g + q, c) + cbyte)) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(T + at_off(r0 + 16 + 4 * g + q, c) + cbyte)) : "memory");
}
__device__ __forceinline__ void fence8(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
}
__device__ __forceinline__ void fence_tr(s16x4 (&lo)[4], s16x4 (&hi)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])::"memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t attn_rsrc(const bf16_t* p, long bytes) {
  unsigned long long base = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
// one 1-KB piece (8 tile rows) of a [N][64] slice -> its place in the swizzled tile; rows past N read as zero (out-of-range offset)
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t r, char* T, int piece, long ld, int N, int lane) {
  const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ hd(row);
  const unsigned vo = row < N ? (unsigned)((row * ld + c * 8) * 2) : 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (attn_lds_ptr_t)(T + piece * 1024), 16, vo, 0, 0, 0);
}
template <int K> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }
__device__ __forceinline__ void wait_vm_dyn(int k) {      // k in 0 .. 6
  switch (k) {
    case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; default: wait_vm<6>(); break;
  }
}

template <int NF, int NW>
__device__ __forceinline__ void attn_bwd_dq_dma(char* smem, int bh, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int B, int N,
                                                int H, float scale) {
  static_assert(NW == 8 && NF == 14, "8 waves: waves 0-3 fetch K, waves 4-7 V, one 1-KB piece per wave and 32-key chunk, 7 chunks");
  constexpr int NP = 16 * NF;
  char* Ks = smem;
  char* Vs = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, cl = lane & 15;
  const float sc2 = scale * 1.4426950408889634f;
  const int b = bh / H, h = bh % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const bf16_t* base = qkv + (size_t)b * N * D3 + h * 64;
  const bf16_t* obase = o + (size_t)b * N * Dm + h * 64;
  const bf16_t* dobase = dout + (size_t)b * N * Dm + h * 64;
  AB_STAMP(0, 0);
  bf16x8 qf[2], dof[2], of_[2];
  float dl, lq, lraw;
  int qrow;
  auto frag_issue = [&](int qp) {
    qrow = qp * 16 + cl;
    const int rc = qrow < N ? qrow : N - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      gload16_asm(qf[ks], base + (size_t)rc * D3 + ks * 32 + g * 8);
      gload16_asm(dof[ks], dobase + (size_t)rc * Dm + ks * 32 + g * 8);
      gload16_asm(of_[ks], obase + (size_t)rc * Dm + ks * 32 + g * 8);
    }
    gload4_asm(lraw, lse + ((size_t)b * H + h) * N + rc);
  };
  auto frag_finish = [&]() {
    const bool ok = qrow < N;
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = zero_past(qf[ks], ok);
      dof[ks] = zero_past(dof[ks], ok);
      acc += dot8(dof[ks], of_[ks]);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    dl = acc;
    lq = ok ? lraw * 1.4426950408889634f : 1e30f;
  };
  frag_issue(wave);                                      // OLDER than the DMA below: consumed behind vmcnt(7) while all seven chunks fly
  __builtin_amdgcn_sched_barrier(0);                     // (hipcc otherwise sinks some of these loads below the DMA: waiting for them would drain it)
  {
    const bool isv = wave >= 4;
    const __amdgpu_buffer_rsrc_t r = attn_rsrc(base + (isv ? 2 : 1) * Dm, ((long)(N - 1) * D3 + 64) * 2);
    char* T = isv ? Vs : Ks;
#pragma unroll
    for (int c = 0; c < NF / 2; ++c) dma_piece(r, T, 4 * c + (wave & 3), D3, N, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  frag_wait<NF / 2>(qf, dof, of_, lraw);                 // the seven younger operations are this wave's DMA pieces
  frag_finish();
  bf16_t* dbase = dqkv + (size_t)b * N * D3 + h * 64;
  for (int qp = wave; qp < NF; qp += NW) {
    if (qp * 16 >= N) break;
    const bool first = qp == wave;
    if (!first) { AB_STAMP(0, 2 + 4 * (qp / NW)); }
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    AB_STAMP(0, 3 + 4 * (qp / NW) + (dl == 123.456f));
#pragma unroll 1
    for (int ss = 0; ss < NF / 2; ++ss) {
      if (first) {                                       // this pair's chunk has landed (this wave's piece: vmcnt; everyone's: the barrier)
        wait_vm_dyn(NF / 2 - 1 - ss);
        __builtin_amdgcn_s_barrier();
        if (ss == 0) { AB_STAMP(0, 1); }
      }
      bf16x8 kf[4], vf[4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[2 * hh + ks] = row_frag_asm(Ks, (2 * ss + hh) * 16, ks, lane); vf[2 * hh + ks] = row_frag_asm(Vs, (2 * ss + hh) * 16, ks, lane); }
      fence8(kf, vf);
      f32x4 st[2], dpt[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        st[hh] = (f32x4){0.f, 0.f, 0.f, 0.f}; dpt[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { st[hh] = MFMA(kf[2 * hh + ks], qf[ks], st[hh]); dpt[hh] = MFMA(vf[2 * hh + ks], dof[ks], dpt[hh]); }
      }
      s16x4 tlo[4], thi[4];                              // the transposed K fragments are requested before the exponent work that hides them
#pragma unroll
      for (int db = 0; db < 4; ++db) tr_frag_asm(Ks, 32 * ss, db * 16, lane, tlo[db], thi[db]);
      f32x4 ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int x = 0; x < 4; ++x) ds[hh][x] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[hh][x], sc2, -lq)) * (dpt[hh][x] - dl);
      const bf16x8 bfg = pack8(ds[0], ds[1]);
      fence_tr(tlo, thi);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x8 kt;
        kt[0] = tlo[db][0]; kt[1] = tlo[db][1]; kt[2] = tlo[db][2]; kt[3] = tlo[db][3];
        kt[4] = thi[db][0]; kt[5] = thi[db][1]; kt[6] = thi[db][2]; kt[7] = thi[db][3];
        dq[db] = MFMA(kt, bfg, dq[db]);
      }
    }
    AB_STAMP(0, 4 + 4 * (qp / NW) + (dq[0][0] == 123.456f));
    uint2 pk[4];
    bf16_t* prow = qrow < N ? dbase + (size_t)qrow * D3 + 4 * g : nullptr;
#pragma unroll
    for (int db = 0; db < 4; ++db) pk[db] = make_uint2(f2bf2(dq[db][0] * scale, dq[db][1] * scale), f2bf2(dq[db][2] * scale, dq[db][3] * scale));
    const bool more = qp + NW < NF && (qp + NW) * 16 < N;
    if (more) frag_issue(qp + NW);
    if (prow) {
#pragma unroll
      for (int db = 0; db < 4; ++db) *(uint2*)(prow + db * 16) = pk[db];
    }
    if (more) { frag_wait<0>(qf, dof, of_, lraw); frag_finish(); }
  }
  AB_STAMP(0, 20);
}

template <int NF, int U, int NW, int PIPE = 0, bool EARLY = false>
__device__ __forceinline__ void attn_bwd_dkv_body(char* smem, int bh, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                  const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int B, int N,
                                                  int H, float scale) {
  constexpr int NP = 16 * NF;
  char* Qs = smem;
  char* Ds = smem + NP * 128;
  float* lse_s = (float*)(Ds + NP * 128);
  float* del_s = lse_s + NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cl = lane & 15;
  const float sc2 = scale * 1.4426950408889634f;   // exp(scale s - lse) = 2^(s sc2 - lse log2 e)
  const int b = bh / H, h = bh % H;
  const long D3 = 3L * H * 64, Dm = (long)H * 64;
  const bf16_t* base = qkv + (size_t)b * N * D3 + h * 64;
  const bf16_t* obase = o + (size_t)b * N * Dm + h * 64;
  const bf16_t* dobase = dout + (size_t)b * N * Dm + h * 64;
  AB_STAMP(1, 0);
  bf16x8 kfb[U][2], vfb[U][2];
  int krow[U];
  auto fragments = [&](int kp) {     // the key group's K / V fragments straight from global
#pragma unroll
    for (int u = 0; u < U; ++u) {
      krow[u] = (U * kp + u) * 16 + cl;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kfb[u][ks] = gfrag(base + Dm, D3, krow[u], N, ks, g); vfb[u][ks] = gfrag(base + 2 * Dm, D3, krow[u], N, ks, g); }
    }
  };
  if (EARLY && wave * 16 * U < N) fragments(wave);      // ahead of the staging: one memory round trip for both
  stage_tile<NP, 64 * NW>(Qs, base, D3, N, tid);
  // stage dO and form delta = rowsum(dO * O): a row's 8 chunks sit in 8 consecutive lanes
  {
    constexpr int NT = 64 * NW, IT = (NP * 8 + NT - 1) / NT;
    uint4 dv_[IT], ov_[IT];
    float ls_[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int idx = tid + NT * i, row = idx >> 3, c = idx & 7;
      int rc = row < N ? row : N - 1;
      dv_[i] = *(const uint4*)(dobase + (size_t)rc * Dm + c * 8);
      ov_[i] = *(const uint4*)(obase + (size_t)rc * Dm + c * 8);
      ls_[i] = lse[((size_t)b * H + h) * N + rc];
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int idx = tid + NT * i, row = idx >> 3, c = idx & 7;
      uint4 v = row < N ? dv_[i] : make_uint4(0u, 0u, 0u, 0u);
      if (NP * 8 % NT == 0 || row < NP) *(uint4*)(Ds + at_off(row, c)) = v;
      float d = dot8(*(bf16x8*)&v, *(bf16x8*)&ov_[i]);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if (c == 0 && (NP * 8 % NT == 0 || row < NP)) {
        del_s[row] = d;
        lse_s[row] = row < N ? ls_[i] * 1.4426950408889634f : 1e30f;   // padded queries: P = exp(. - 1e30) = 0
      }
    }
  }
  __syncthreads();
  AB_STAMP(1, 1);
  bf16_t* dbase = dqkv + (size_t)b * N * D3 + h * 64;
  for (int kp = wave; kp < NF / U; kp += NW) {           // key group: keys 16 U kp .. 16 U (kp + 1) - 1
    if (kp * 16 * U >= N) break;
    AB_STAMP(1, 2 + 4 * (kp / NW));
    if (!EARLY) fragments(kp);
    f32x4 dv[U][4], dk[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) { dv[u][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[u][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    AB_STAMP(1, 3 + 4 * (kp / NW) + (kfb[0][0][0] == 12345 && vfb[0][1][0] == 12345));
    if (PIPE && U == 1) {
      f32x4 sa[2], dpa[2];
      auto scores = [&](int qp, f32x4 (&sa_)[2], f32x4 (&dpa_)[2]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          sa_[hh] = (f32x4){0.f, 0.f, 0.f, 0.f}; dpa_[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            sa_[hh] = MFMA(row_frag(Qs, (2 * qp + hh) * 16, ks, lane), kfb[0][ks], sa_[hh]);
            dpa_[hh] = MFMA(row_frag(Ds, (2 * qp + hh) * 16, ks, lane), vfb[0][ks], dpa_[hh]);
          }
        }
      };
      scores(0, sa, dpa);
#pragma unroll 1
      for (int qp = 0; qp < NF / 2; ++qp) {
        f32x4 san[2], dpan[2];
        const int qn = qp + 1 < NF / 2 ? qp + 1 : qp;
        scores(qn, san, dpan);
        f32x4 P[2], dS[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float4 l4 = *(const float4*)(lse_s + (2 * qp + hh) * 16 + 4 * g), d4 = *(const float4*)(del_s + (2 * qp + hh) * 16 + 4 * g);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[hh][x], sc2, -lv[x]));
            P[hh][x] = pr;
            dS[hh][x] = pr * (dpa[hh][x] - dlv[x]);
          }
        }
        const bf16x8 pa = pack8(P[0], P[1]), dsa = pack8(dS[0], dS[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dv[0][db] = MFMA(tr_frag(Ds, 32 * qp, db * 16, lane), pa, dv[0][db]);
          dk[0][db] = MFMA(tr_frag(Qs, 32 * qp, db * 16, lane), dsa, dk[0][db]);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) { sa[hh] = san[hh]; dpa[hh] = dpan[hh]; }
      }
    } else
#pragma unroll 1
    for (int qp = 0; qp < NF / 2; ++qp) {                  // query pair: rows 32 qp .. 32 qp + 31
      bf16x8 qf[2][2], df[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { qf[hh][ks] = row_frag(Qs, (2 * qp + hh) * 16, ks, lane); df[hh][ks] = row_frag(Ds, (2 * qp + hh) * 16, ks, lane); }
      float lv[2][4], dl[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float4 l4 = *(const float4*)(lse_s + (2 * qp + hh) * 16 + 4 * g), d4 = *(const float4*)(del_s + (2 * qp + hh) * 16 + 4 * g);
        lv[hh][0] = l4.x; lv[hh][1] = l4.y; lv[hh][2] = l4.z; lv[hh][3] = l4.w;
        dl[hh][0] = d4.x; dl[hh][1] = d4.y; dl[hh][2] = d4.z; dl[hh][3] = d4.w;
      }
      bf16x8 pa[U], dsa[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f32x4 P[2], dS[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dpa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            sa = MFMA(qf[hh][ks], kfb[u][ks], sa);         // S[q = 4g+x][key = cl]
            dpa = MFMA(df[hh][ks], vfb[u][ks], dpa);       // dP[q][key]
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[x], sc2, -lv[hh][x]));
            P[hh][x] = p;
            dS[hh][x] = p * (dpa[x] - dl[hh][x]);
          }
        }
        pa[u] = pack8(P[0], P[1]);                         // B[k = q(8g+j)][col = key = cl]
        dsa[u] = pack8(dS[0], dS[1]);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 dt = tr_frag(Ds, 32 * qp, db * 16, lane), qt = tr_frag(Qs, 32 * qp, db * 16, lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          dv[u][db] = MFMA(dt, pa[u], dv[u][db]);          // dV^T[d = 16db+4g+x][key = cl]
          dk[u][db] = MFMA(qt, dsa[u], dk[u][db]);         // dK^T[d][key]
        }
      }
    }
    AB_STAMP(1, 4 + 4 * (kp / NW) + (dv[0][0][0] == 123.456f && dk[0][0][0] == 123.456f));
    uint2 pkk[U][4], pkv[U][4];
    bf16_t* prow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      prow[u] = krow[u] < N ? dbase + (size_t)krow[u] * D3 + Dm + 4 * g : nullptr;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        pkk[u][db] = make_uint2(f2bf2(dk[u][db][0] * scale, dk[u][db][1] * scale), f2bf2(dk[u][db][2] * scale, dk[u][db][3] * scale));
        pkv[u][db] = make_uint2(f2bf2(dv[u][db][0], dv[u][db][1]), f2bf2(dv[u][db][2], dv[u][db][3]));
      }
    }
    if (EARLY && kp + NW < NF / U && (kp + NW) * 16 * U < N) fragments(kp + NW);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (prow[u]) {       // a lane owns 4 consecutive head dims of its key row: 8-byte stores
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          *(uint2*)(prow[u] + db * 16) = pkk[u][db];
          *(uint2*)(prow[u] + Dm + db * 16) = pkv[u][db];
        }
      }
  }
  AB_STAMP(1, 20);
}

// ======================================================================== fused backward (N > 64): EXPERIMENT, tools build only
// Measured on MI355X (B = 64, N = 197, H = 6; profiles/r03/attn_bwd_fused.txt): 41.7 us against 36.1 us for the two-body form above, and
// the same step time (4.62 ms either way).  Staging every operand once costs 6 us (77 MB instead of 126 MB), but the dQ and dK/dV phases
// take 14 + 20 us: each is bound by the SUM of its LDS reads (0.5 KB per MFMA: every wave re-reads the tiles it does not own), its MFMAs and
// its exponent VALU work, with 2 waves per SIMD and one workgroup per CU (114 KB of LDS) there is nothing to overlap them with, and 384
// workgroups on 256 CUs leave the second round half empty.  The exponent is folded to one fma + v_exp_f32 here (-2.7 us); the same fold made
// the 32-row two-body kernel 3 us SLOWER stand-alone (39.1 vs 36.1 us, fewer instructions, same registers) and the 16-row one 1 us faster.
// (A single-pass form -- one 8-wave workgroup per (batch, head) with Q, K, V and dO all resident in LDS, every input read once -- was built
// and measured slower, 41.7 vs 36.1 us: profiles/r04/removed_experiments.patch, profiles/r03/attn_bwd_fused.txt.)

// ======================================================================== launchers
static int pick_nf(int N) { return N <= 32 ? 2 : N <= 64 ? 4 : N <= 224 ? 14 : N <= 256 ? 16 : 0; }

template <int NF, int NV = NF, bool MASKALL = true>
static int launch_fwd(const bf16_t* qkv, bf16_t* o, float* lse, int B, int N, int H, float scale, hipStream_t s) {
  const int lds = 2 * 16 * NF * 128;
  auto k = k_attn_fwd_mfma<NF, NV, MASKALL>;
  static bool done = false;
  if (!done) { FC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  hipLaunchKernelGGL(k, dim3(B * H), dim3(64 * AF_WAVES), lds, s, qkv, o, lse, B, N, H, scale);
  FC_LAUNCH_CHECK();
  return 0;
}
// dQ and dK/dV of one attention backward in ONE launch: blocks [0, BH) run the dQ bodies, blocks [BH, 2BH) the dK/dV ones.
// The two are independent; as separate launches on one stream the second waited for the first to drain.
// U = 16-row blocks a wave owns, NW = waves per workgroup: <2, 4> = 32-row register blocking at 2 waves per SIMD (207 VGPRs);
// <1, 8> = 16-row blocking under 128 VGPRs, 4 waves per SIMD (twice the LDS fragment reads per MFMA, twice the waves to hide the
// MFMA -> exponent -> MFMA dependency chain behind)
template <int NF, int U, int NW, int PIPE = 0>
__global__ void __launch_bounds__(64 * NW, (NW >= 16 ? 4 : NW / 2)) k_attn_bwd(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int B, int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int half = B * H;
  // the dK/dV workgroups are the longer ones (two accumulators per pair): they are dispatched FIRST, so that when the grid does not fit the chip
  // in one round (B = 64: 768 workgroups, 512 resident) the short dQ workgroups form the partial second round (PIPE & 16: the former order)
  const bool dq_first = (PIPE & 16) != 0;
  const int blk = blockIdx.x;
  bool is_dq;
  int bh;
  if constexpr ((PIPE & 64) != 0) {
    // paired mapping: the two workgroups of a (batch, head) run on the SAME XCD (workgroup b runs on XCD b % 8), eight dispatch slots apart, so
    // the operands one of them has pulled through the fabric are L2 hits for the other: groups of 16 = {dK/dV of 8 heads, dQ of the same 8}
    const int p = blk >> 4, r = blk & 15;
    is_dq = r >= 8;
    bh = p * 8 + (r & 7);
    if (bh >= half) return;
  } else {
    is_dq = dq_first ? blk < half : blk >= half;
    bh = (is_dq == dq_first) ? blk : blk - half;
  }
  if (is_dq) {
    if constexpr ((PIPE & 8) != 0) attn_bwd_dq_dma<NF, NW>(smem, bh, qkv, o, dout, lse, dqkv, B, N, H, scale);
    else attn_bwd_dq_body<NF, U, NW, (PIPE & 1), (PIPE & 4) != 0>(smem, bh, qkv, o, dout, lse, dqkv, B, N, H, scale);
  }
  else attn_bwd_dkv_body<NF, U, NW, (PIPE & 2), (PIPE & 4) != 0>(smem, bh, qkv, o, dout, lse, dqkv, B, N, H, scale);
}
#ifdef FC_PROBES
static long long* g_ab_stamps_host = nullptr;
extern "C" int fc_dbg_attn_stamps(long long* out1024) {      // tools build: the stamps of the last backward launch (call after a synchronise)
  if (!g_ab_stamps_host) return -1;
  return hipMemcpy(out1024, g_ab_stamps_host, 1024 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
template <int NF, int U = 2, int NW = 4, int PIPE = 0>
static int launch_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, bf16_t* dqkv, int B, int N, int H, float scale,
                      hipStream_t s) {
  const int lds_q = 2 * 16 * NF * 128, lds_kv = 2 * 16 * NF * 128 + 2 * 16 * NF * 4;
#ifdef FC_PROBES
  static long long* stamps = nullptr;
  if (fc_knob("FC_ATTN_STAMPS", 0)) {
    if (!stamps) { FC_CHECK_HIP(hipMalloc(&stamps, 1024 * sizeof(long long))); FC_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ab_stamps), &stamps, sizeof(stamps))); g_ab_stamps_host = stamps; }
    FC_CHECK_HIP(hipMemsetAsync(stamps, 0, 1024 * sizeof(long long), s));
  }
#endif
  auto kb = k_attn_bwd<NF, U, NW, PIPE>;
  const int lds = lds_kv > lds_q ? lds_kv : lds_q;
  static bool done = false;
  if (!done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
  }
  const int grid = (PIPE & 64) ? ((B * H + 7) / 8) * 16 : B * H * 2;
  hipLaunchKernelGGL(kb, dim3(grid), dim3(64 * NW), lds, s, qkv, o, dout, lse, dqkv, B, N, H, scale);
  FC_LAUNCH_CHECK();
  return 0;
}

int fc_attn_fwd_mfma(const bf16_t* qkv, bf16_t* o, float* lse, int B, int N, int H, int d, float scale, hipStream_t s) {
  if (FC_ABLATED("attn")) return 0;
  if (d != 64 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 7)) return 1;
  switch (pick_nf(N)) {
    case 2: return launch_fwd<2>(qkv, o, lse, B, N, H, scale, s);
    case 4: return launch_fwd<4>(qkv, o, lse, B, N, H, scale, s);
    case 14:
      if (N > 192 && N <= 208) return launch_fwd<14, 13, false>(qkv, o, lse, B, N, H, scale, s);   // ViT /16 at 224: 197 tokens = 12 full blocks + 5 keys
      return launch_fwd<14>(qkv, o, lse, B, N, H, scale, s);
    case 16: return launch_fwd<16>(qkv, o, lse, B, N, H, scale, s);
  }
  return 1;
}


int fc_attn_bwd_mfma(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, float* delta, bf16_t* dqkv, int B, int N, int H,
                     int d, float scale, hipStream_t s) {
  if (FC_ABLATED("attn")) return 0;
  (void)delta;   // recomputed in-kernel
  if (d != 64 || ((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)o & 15)) return 1;
  // 16-row blocks per wave at twice the waves (8 per workgroup for the image sequences: 110 VGPRs, 4 waves per SIMD) instead of 32-row
  // blocks at 2 waves per SIMD (206 VGPRs): twice the LDS fragment reads per MFMA -- the LDS is 13 % busy -- for twice the waves to
  // cover the MFMA -> exponent -> MFMA dependency chain with: 36.0 -> 33.0 us at B = 64, N = 197, 32.0 with the exponent folded into one fma +
  // v_exp_f32 (that fold made the 32-row form 3 us slower), step -0.5 .. -1 % (profiles/r03/attn_bwd_u1.txt).  Requesting both jobs' Q / dO / O
  // fragments ahead of the staging (128 VGPRs) measured 33.8 us, skipping the all-padding half of the last 32-row pair in the inner loops
  // (a wave-uniform branch, 1 / 14 of the work) 33.0 us: neither kept.
  // FC_ATTN_BWD_U1=0 (tools build) restores the 32-row form.
  static const int u1 = fc_knob("FC_ATTN_BWD_U1", 1);
#ifdef FC_PROBES      // round 5's measured variants of the N = 197 kernel (profiles/r05/attn_bwd_pipeline.txt): none beat the default, tools build only
  static const int pipe = fc_knob("FC_ATTN_BWD_PIPE", 0);
  if (pipe && u1 && pick_nf(N) == 14) {                       // bits: 1 the dQ half's pair loop pipelined, 2 the dK/dV half's, 4 early fragment requests, 8 the dQ half chunk-pipelined behind LDS-DMA
    switch (pipe) {
      case 1: return launch_bwd<14, 1, 8, 1>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
      case 2: return launch_bwd<14, 1, 8, 2>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
      case 3: return launch_bwd<14, 1, 8, 3>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
      case 4: return launch_bwd<14, 1, 8, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
      case 5: return launch_bwd<14, 1, 8, 5>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
      case 12: if (N > 112) return launch_bwd<14, 1, 8, 12>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);      // 8: the dQ half chunk-pipelined behind LDS-DMA
               break;
      case 16: return launch_bwd<14, 1, 8, 16>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);                     // 16: dQ workgroups first (rounds 2-4)
      case 64: return launch_bwd<14, 1, 8, 64>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);                     // 64: both workgroups of a (batch, head) on one XCD, dispatched together
      case 32: return launch_bwd<14, 1, 16, 0>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);                     // 32: 16 waves per workgroup, one 16-row block per wave
      case 36: return launch_bwd<14, 1, 16, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);                     //     ... with the fragments requested ahead of the staging
    }
  }
#endif
  switch (pick_nf(N)) {
    case 2: return u1 ? launch_bwd<2, 1, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s) : launch_bwd<2>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
    case 4: return u1 ? launch_bwd<4, 1, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s) : launch_bwd<4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
    case 14: return u1 ? launch_bwd<14, 1, 8>(qkv, o, dout, lse, dqkv, B, N, H, scale, s) : launch_bwd<14>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
    case 16: return u1 ? launch_bwd<16, 1, 8>(qkv, o, dout, lse, dqkv, B, N, H, scale, s) : launch_bwd<16>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  }
  return 1;
}
