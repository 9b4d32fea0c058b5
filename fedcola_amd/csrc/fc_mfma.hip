// bf16 MFMA GEMMs for gfx950 (CDNA4), written for the FedCola client step's shapes (K = 384..1536, M = B*197).
//
//   NT: C[M,N] = A[M,K] . W[N,K]^T     forward linears            (both operands k-contiguous: "KC")
//   NN: C[M,N] = A[M,K] . W[K,N]       dX = dY . W                (B operand has k as its row index: "KR")
//   TN: C[M,N] = A[K,M]^T . B[K,N]     dW = dY^T . X (fp32 out)   (both operands KR; split-K over the long reduction)
//
// Structure: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16
// accumulators with swapped operands (a lane owns 4 consecutive output columns).
//   NT / NN (k_gemm_mfma): persistent workgroups (<= 2 per CU) walk the output tiles; operand tiles stream global -> LDS by
//   LDS-DMA (buffer_load ... lds, no staging registers), double-buffered, one barrier per k-tile; the epilogue kind is a
//   template parameter and goes through the idle staging buffer in two 64-row halves.
//   TN (k_gemm_tn_grouped): one workgroup per output tile over the whole (long) reduction, operands staged through
//   registers with two k-tiles of loads in flight; every dW / db of a group of layers in one launch.
// LDS images are XOR-swizzled (fc_mfma_dev.h) so that both fragment read forms are bank-conflict free:
//   KC tile [128 rows][64 k]  (128 B rows): 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7); fragments by ds_read_b128.
//   KR tile [64 k][128 cols]  (256 B rows): 32-B unit u of row k lives at unit u ^ (((k>>3)&1)<<2 | (k&3)); fragments by
//   ds_read_b64_tr_b16 (the CDNA4 transposing LDS read), two per 8-k fragment.
// Block ids are remapped so that the tiles sharing an A row-panel run on the same XCD (private L2).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>

#include "fc_kernels.h"
#include "fc_mfma_dev.h"

template <int MODE>
__device__ __forceinline__ void retarget_operand(Operand& o, long ld, int row0, int nrows, int tid, bool valid) {
  // move the per-lane offsets to another tile of the same matrix (row0 = first row (KC) / first column (KR))
  if (MODE == KC) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = row0 + (tid >> 3) + 32 * p;
      o.voff[p] = (valid && r < nrows) ? (unsigned)((r * ld + o.c8) * 2) : FC_OOB;
    }
  } else {
    int col = row0 + (tid & 15) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) o.voff[p] = (valid && col < nrows) ? (unsigned)((((tid >> 4) + 16 * p) * ld + col) * 2) : FC_OOB;
  }
}
template <int MODE>
__device__ __forceinline__ Operand make_operand(const bf16_t* P, long ld, int row0, int nrows, int K, int tid) {
  Operand o;
  // total extent in bytes (rows x ld for KC, K x ld for KR); the pointer and sizes are wave-uniform by construction
  unsigned long long base = (unsigned long long)P;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const void* up = (const void*)(((unsigned long long)hi << 32) | lo);
  long rows = (MODE == KC) ? nrows : K;
  unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(((rows - 1) * ld + ((MODE == KC) ? K : nrows)) * 2));
  o.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)up, 0, (int)bytes, 0x00020000);
  if (MODE == KC) {  // P[row][k]: lane chunk c = tid&7, rows (tid>>3) + 32p
    o.c8 = (tid & 7) * 8;
    o.kstride = 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = row0 + (tid >> 3) + 32 * p;
      o.voff[p] = r < nrows ? (unsigned)((r * ld + o.c8) * 2) : FC_OOB;
    }
  } else {  // P[k][col]: lane chunk c = tid&15, k rows (tid>>4) + 16p
    o.c8 = 0;
    o.kstride = (unsigned)(ld * 2);
    int col = row0 + (tid & 15) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) o.voff[p] = col < nrows ? (unsigned)((((tid >> 4) + 16 * p) * ld + col) * 2) : FC_OOB;
  }
  return o;
}
template <int MODE>
__device__ __forceinline__ void stage_load(uint4 (&r)[4], const Operand& o, int k0, int K) {
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)k0 * o.kstride));   // scalar offset (SGPR)
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned vo = o.voff[p];
    if (MODE == KC) vo = (k0 + o.c8 < K) ? vo : FC_OOB;   // k tail of the last tile (K % 64 != 0)
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(o.rsrc, vo, soff, 0);
    r[p] = *(uint4*)&v;
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


template <typename TC>
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
    if (e.gelu_saved_grad) {
      float gp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) gp[i] = gelu_erf_grad(v[i]);
      Vec8<TC>::st((TC*)e.preact + o, gp);
    } else {
      Vec8<TC>::st((TC*)e.preact + o, v);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
  }
  if (e.gelu_in) {
    float u[8];
    Vec8<TC>::ld((const TC*)e.gelu_in + o, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.gelu_saved_grad ? u[i] : gelu_erf_grad(u[i]);
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
  if (e.accumulate) {
    float r[8];
    Vec8<TC>::ld(C + o, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  Vec8<TC>::st(C + o, v);
}

// accumulators -> LDS image (fp32, rows padded to CS_LD): lane (g, cl) owns C[16i + cl][16j + 4g .. +3] (swapped operands)
__device__ __forceinline__ void acc_to_lds(float* Cs, const f32x4 (&acc)[4][4], int wm, int wn, int lane) {
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(Cs + (wm * 64 + i * 16 + cl) * CS_LD + wn * 64 + j * 16 + 4 * g) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
}



// LDS image -> global with the fused epilogue.  The epilogue KIND is a compile-time parameter: every load it needs is
// unconditional (rows past M are clamped for the loads and masked for the stores), so hipcc can issue the whole batch and
// use counted waits.  (With run-time option flags each optional load sat in its own basic block behind an
// s_waitcnt vmcnt(0), which also drained the register prefetch pipeline: 17k cycles per tile instead of ~2k.)
enum { EPI_PLAIN = 0, EPI_BIAS, EPI_RES, EPI_RES_SCALE, EPI_GELU, EPI_GELU_GRAD, EPI_PATCH, EPI_GENERIC, EPI_GELU_SG, EPI_MUL };


// 64-row epilogue image in ONE 32-KB staging buffer: rows of 128 floats, float4 slot s of row r stored at slot s ^ (r & 7)
// (conflict-free for the accumulator writes -- 8 lanes, 8 rows, one column slot -- and for the row reads).
__device__ __forceinline__ int cs_slot(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 2); }
// half hp of the tile = rows {64 wm + 32 hp + r, r < 32}: image row = 32 wm + r, every wave contributes two row-fragments
template <int HP, int NI>
__device__ __forceinline__ void acc_to_lds_half(float* Cs, const f32x4 (&acc)[NI][4], int wm, int wn, int lane) {
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(Cs + cs_slot(wm * 32 + ii * 16 + cl, wn * 16 + j * 4 + g)) =
          make_float4(acc[2 * HP + ii][j][0], acc[2 * HP + ii][j][1], acc[2 * HP + ii][j][2], acc[2 * HP + ii][j][3]);
}
// store instructions one thread issues per output tile with the compile-time epilogues (0: unknown -> full drain)
template <int EPI, typename TC, int MT = 128> struct EpiStores { static constexpr int n = EPI == EPI_GENERIC ? 0 : ((EPI == EPI_GELU || EPI == EPI_GELU_SG) ? 16 : 8) * (sizeof(TC) == 2 ? 1 : 2) * MT / 128; };
// Global inputs of a tile's epilogue (bias, residual / GELU' operand, drop-path scale), requested for BOTH halves before the
// tile's last k-step: vmcnt retires in order, so a load issued after a store cannot be waited for without also waiting
// out that store's write latency -- loading inside each half serialised every half behind the previous half's stores.
struct EpiRegs { float bias[8]; uint4 rin0[4], rin1[4]; float sc0[4], sc1[4]; };   // always a local of the kernel: lives in VGPRs
template <int EPI, typename TC> struct EpiPre {
  static constexpr bool on = sizeof(TC) == 2 && EPI != EPI_GENERIC && EPI != EPI_PATCH;
};
#define TILE_ROW2(row, hp) (MT == 64 ? (row) : ((((row) >> 5) << 6) + (hp) * 32 + ((row) & 31)))
template <int EPI, typename TC, int MT = 128>
__device__ __forceinline__ void epi_prefetch(EpiRegs& R, long ldc, int m0, int n0, int M, int N, const GemmEpi& e, int tid) {
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_RES || EPI == EPI_RES_SCALE || EPI == EPI_GELU || EPI == EPI_GELU_SG;
  constexpr bool HAS_RES = EPI == EPI_RES || EPI == EPI_RES_SCALE;
  constexpr bool HAS_IN = EPI == EPI_GELU_GRAD || EPI == EPI_MUL;
  const int c8 = (tid & 15) * 8, n = n0 + c8, r0 = tid >> 4;
  const int nc = n < N ? n : N - 8;
  if (HAS_BIAS) Vec8<float>::ld(e.bias + nc, R.bias);
  if (HAS_RES || HAS_IN || EPI == EPI_RES_SCALE) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ma = m0 + TILE_ROW2(r0 + 16 * q, 0), mb = m0 + TILE_ROW2(r0 + 16 * q, 1);
      const int mca = ma < M ? ma : M - 1, mcb = mb < M ? mb : M - 1;
      const size_t offa = (size_t)mca * ldc + nc, offb = (size_t)mcb * ldc + nc;
      if (HAS_RES) { R.rin0[q] = *(const uint4*)((const bf16_t*)e.res + offa); if (MT == 128) R.rin1[q] = *(const uint4*)((const bf16_t*)e.res + offb); }
      if (HAS_IN) { R.rin0[q] = *(const uint4*)((const bf16_t*)e.gelu_in + offa); if (MT == 128) R.rin1[q] = *(const uint4*)((const bf16_t*)e.gelu_in + offb); }
      if (EPI == EPI_RES_SCALE) { R.sc0[q] = e.rowscale[mca / e.rows_per_sample]; if (MT == 128) R.sc1[q] = e.rowscale[mcb / e.rows_per_sample]; }
    }
  }
}
// one half (64 rows starting at tile row `rbase`) of the fused epilogue; thread owns columns 8*(tid&15).. and rows (tid>>4) + 16q
template <int EPI, typename TC, bool PRE = false, int MT = 128>
__device__ __forceinline__ void half_epilogue(const float* Cs, TC* C, long ldc, int m0, int n0, int M, int N, const GemmEpi& e, int tid, int hp,
                                              __amdgpu_buffer_rsrc_t crs, __amdgpu_buffer_rsrc_t prs, const float (&pbias)[8],
                                              const uint4 (&prin)[4], const float (&psc)[4]) {
  const int c8 = (tid & 15) * 8, n = n0 + c8, r0 = tid >> 4;
#define TILE_ROW(row) (MT == 64 ? (row) : ((((row) >> 5) << 6) + hp * 32 + ((row) & 31)))   /* image row -> row inside the tile */
  if (EPI == EPI_GENERIC) {
    if (n >= N) return;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      int row = r0 + 16 * q, m = m0 + TILE_ROW(row);
      if (m < M) {
        float v[8];
        float4 x0 = *(const float4*)(Cs + cs_slot(row, (tid & 15) * 2)), x1 = *(const float4*)(Cs + cs_slot(row, (tid & 15) * 2 + 1));
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        epi_store8<TC>(C, ldc, m, n, v, e, N);
      }
    }
    return;
  }
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_RES || EPI == EPI_RES_SCALE || EPI == EPI_GELU || EPI == EPI_GELU_SG || EPI == EPI_PATCH;
  constexpr bool HAS_RES = EPI == EPI_RES || EPI == EPI_RES_SCALE;
  const int nc = n < N ? n : N - 8;
  constexpr bool use_pre = EpiPre<EPI, TC>::on && PRE;
  float bias[8];
  if (HAS_BIAS) {
    if (use_pre) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = pbias[i];
    } else {
      Vec8<float>::ld(e.bias + nc, bias);
    }
  }
  float rin[4][8], rpos[4][8], sc[4];
  size_t off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + TILE_ROW(r0 + 16 * q);
    const int mc = m < M ? m : M - 1;
    long orow = mc;
    if (EPI == EPI_PATCH) orow = (long)mc + mc / e.patch_rows + 1;
    off[q] = (size_t)orow * ldc + nc;
    if (use_pre) {
      if (HAS_RES || EPI == EPI_GELU_GRAD || EPI == EPI_MUL) {
        const uint4 u = prin[q];
        rin[q][0] = __uint_as_float(u.x << 16); rin[q][1] = __uint_as_float(u.x & 0xffff0000u);
        rin[q][2] = __uint_as_float(u.y << 16); rin[q][3] = __uint_as_float(u.y & 0xffff0000u);
        rin[q][4] = __uint_as_float(u.z << 16); rin[q][5] = __uint_as_float(u.z & 0xffff0000u);
        rin[q][6] = __uint_as_float(u.w << 16); rin[q][7] = __uint_as_float(u.w & 0xffff0000u);
      }
      if (EPI == EPI_RES_SCALE) sc[q] = psc[q];
      continue;
    }
    if (HAS_RES) Vec8<TC>::ld((const TC*)e.res + off[q], rin[q]);
    if (EPI == EPI_GELU_GRAD || EPI == EPI_MUL) Vec8<TC>::ld((const TC*)e.gelu_in + off[q], rin[q]);
    if (EPI == EPI_PATCH) Vec8<float>::ld(e.pos + (size_t)(1 + mc % e.patch_rows) * N + nc, rpos[q]);
    if (EPI == EPI_RES_SCALE) sc[q] = e.rowscale[mc / e.rows_per_sample];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = r0 + 16 * q;
    const int trow = TILE_ROW(row);
    float v[8];
    float4 x0 = *(const float4*)(Cs + cs_slot(row, (tid & 15) * 2)), x1 = *(const float4*)(Cs + cs_slot(row, (tid & 15) * 2 + 1));
    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
    if (HAS_BIAS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += bias[i];
    }
    if (EPI == EPI_PATCH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += rpos[q][i];
    }
    const bool ok = (m0 + trow < M) && (n < N);
    if (EPI == EPI_GELU) {
      buf_store8<TC>(prs, off[q], ok, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = gelu_fast(v[i]);
    }
    if (EPI == EPI_GELU_SG) {   // one exp / rcp per element serves both gelu(u) (output) and gelu'(u) (saved for the backward)
      float gp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float cdf, pdf;
        gelu_fast_parts(v[i], cdf, pdf);
        gp[i] = cdf + v[i] * pdf;
        v[i] *= cdf;
      }
      buf_store8<TC>(prs, off[q], ok, gp);
    }
    if (EPI == EPI_GELU_GRAD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= gelu_fast_grad(rin[q][i]);
    }
    if (EPI == EPI_MUL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= rin[q][i];
    }
    if (EPI == EPI_RES_SCALE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= sc[q];
    }
    if (HAS_RES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += rin[q][i];
    }
    buf_store8<TC>(crs, off[q], ok, v);
  }
#undef TILE_ROW
}
template <int EPI, typename TC>
__device__ __forceinline__ void half_epilogue(const float* Cs, TC* C, long ldc, int m0, int n0, int M, int N, const GemmEpi& e, int tid, int hp,
                                              __amdgpu_buffer_rsrc_t crs, __amdgpu_buffer_rsrc_t prs) {
  const float b[8] = {};
  const uint4 r[4] = {};
  const float c[4] = {};
  half_epilogue<EPI, TC, false, 128>(Cs, C, ldc, m0, n0, M, N, e, tid, hp, crs, prs, b, r, c);
}

// ---- main loop shared by the single-problem and the grouped kernels.
// Operands are swapped in the MFMA (D' = B.A^T) so that acc[i][j][x] = C[m = 16i + lane&15][n = 16j + 4(lane>>4) + x]:
// a lane then owns 4 consecutive output columns and the epilogue needs no LDS round trip.
// Two register sets keep TWO k-tiles of global loads in flight behind the MFMAs (load -> use distance = 2 tiles).
struct StageRegs { uint4 a[4], b[4]; };

template <int AMODE, int BMODE>
__device__ __forceinline__ void tile_load(StageRegs& R, const Operand& oa, const Operand& ob, int k0, int K) {
  stage_load<AMODE>(R.a, oa, k0, K);
  stage_load<BMODE>(R.b, ob, k0, K);
}
template <int AMODE, int BMODE, bool COLSUM>
__device__ __forceinline__ void tile_store(const StageRegs& R, char* buf, int tid, float (&cs)[8], bool do_colsum) {
  stage_store<AMODE>(R.a, buf, tid);
  stage_store<BMODE>(R.b, buf + 16384, tid);
  if (COLSUM && do_colsum) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bf16_t* hh = (const bf16_t*)&R.a[p];
#pragma unroll
      for (int i = 0; i < 8; ++i) cs[i] += bf2f(hh[i]);
    }
  }
}
// NI = row blocks of 16 per wave: 4 (128-row tile) or 2 (64-row tile: a wave's rows start at 32 wm)
template <int AMODE, int BMODE, int NI = 4>
__device__ __forceinline__ void tile_compute(const char* buf, f32x4 (&acc)[NI][4], int wm, int wn, int lane) {
  const char* la = buf;
  const char* lb = buf + 16384;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[NI], bfr[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) af[i] = frag_read<AMODE>(la, wm * (16 * NI) + i * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BMODE>(lb, wn * 64 + j * 16, ks, lane);
    if (AMODE == KR) { static_assert(AMODE != KR || NI == 4, "KR A operand: 128-row tiles only"); frag_fence(*(bf16x8(*)[4])&af); }
    if (BMODE == KR) frag_fence(bfr);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
}

template <int AMODE, int BMODE, bool COLSUM>
__device__ __forceinline__ void gemm_mainloop(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bm, long ldb, int m0, int n0, int M,
                                              int N, int K, char* smem, f32x4 (&acc)[4][4], float (&cs)[8], bool do_colsum, int tid, int wm, int wn,
                                              int lane, int dbg = 0) {
  const int T = (K + BK - 1) / BK;
  char* buf0 = smem;
  char* buf1 = smem + 32768;
  StageRegs R0, R1;
  const Operand oa = make_operand<AMODE>(A, lda, m0, M, K, tid);
  const Operand ob = make_operand<BMODE>(Bm, ldb, n0, N, K, tid);
  // Loads past the last tile fall outside the descriptor (zeros) and their stores go to a buffer nobody reads again, so
  // the loop body is branch-free: hipcc can then count the loads in flight (s_waitcnt vmcnt(8), never 0, inside the loop).
  tile_load<AMODE, BMODE>(R0, oa, ob, 0, K);
  tile_load<AMODE, BMODE>(R1, oa, ob, BK, K);
  tile_store<AMODE, BMODE, COLSUM>(R0, buf0, tid, cs, do_colsum);
  lds_barrier();
  for (int t = 0; t < T; t += 2) {
    if (!(dbg & 1)) tile_load<AMODE, BMODE>(R0, oa, ob, (t + 2) * BK, K);
    if (!(dbg & 4)) tile_compute<AMODE, BMODE>(buf0, acc, wm, wn, lane);
    tile_store<AMODE, BMODE, COLSUM>(R1, buf1, tid, cs, do_colsum);
    lds_barrier();
    if (!(dbg & 1)) tile_load<AMODE, BMODE>(R1, oa, ob, (t + 3) * BK, K);
    if (!(dbg & 4)) tile_compute<AMODE, BMODE>(buf1, acc, wm, wn, lane);   // an odd T ends on an all-zero tile: adds nothing
    tile_store<AMODE, BMODE, COLSUM>(R0, buf0, tid, cs, do_colsum);
    lds_barrier();
  }
}


// Persistent kernel over the tiles of ONE or TWO problems that share N, K and the epilogue kind (the image and the text tower of a
// layer: one launch instead of three, one read of each weight matrix, 345 instead of 150 tiles for the N = 384 shapes): the grid is
// at most 2 workgroups per CU and every workgroup walks the tiles id, id + grid, id + 2*grid, ... of the concatenated tile list.
// Operand tiles stream global -> LDS directly (LDS-DMA), double-buffered: the loads of k-tile t+1 (or of the NEXT output tile's
// first k-tile) are in flight while k-tile t feeds the MFMAs and while the epilogue runs.  One barrier per k-tile: wait for own DMA
// (vmcnt) -> barrier -> issue next DMA -> compute.
// MT: rows of an output tile.  128 (default) or 64: launches whose 128-row tiles would leave most of the chip idle (a 4 334-row chain with N = 384:
// 102 tiles on 256 CUs, every workgroup walking K = 1 536 alone) are cut into twice as many 64 x 128 tiles -- what the vendor library's
// heuristic does on these shapes (MT128x64, profiles/r05/vendor_yardstick_*.txt: 18.2 against 23.5 us on 4 334 x 384 x 1 536).
template <int AMODE, int BMODE, typename TC, int EPI, int MT = 128>
__global__ void __launch_bounds__(256, 2) k_gemm_mfma(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A 16 KB | B 16 KB) + a separate epilogue image
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);
  const int N = g.N, K = g.K, tiles_n = g.tiles_n, ntiles = g.ntiles;
  const int T = (K + BK - 1) / BK;
  constexpr int NI = MT / 32, NPA = MT / 32;      // row blocks per wave; 1-KB pieces of the A tile per wave
  static_assert(MT == 128 || (MT == 64 && AMODE == KC), "64-row tiles: k-contiguous A operand only");
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // load side: tile `lt` (of problem `lprob`), k-tile `lk`, destination buffer parity `lb`
  int lt = first, lk = 0, lb = 0, lprob = -1;
  Operand oa, ob;
  auto retarget = [&]() {
    const bool valid = lt < ntiles;
    const int t = valid ? lt : first;
    const int pr = t >= g.tiles0;
    const GemmProb& P = g.p[pr];
    if (pr != lprob) {
      oa = make_operand_glds<AMODE>(P.A, P.lda, 0, P.M, K, wave, lane);
      ob = make_operand_glds<BMODE>(P.B, P.ldb, 0, N, K, wave, lane);
      lprob = pr;
    }
    const int l = t - (pr ? g.tiles0 : 0);
    retarget_glds<AMODE, NPA>(oa, P.lda, (l / tiles_n) * MT, P.M, wave, lane, valid);
    retarget_glds<BMODE>(ob, P.ldb, (l % tiles_n) * BN, N, wave, lane, valid);
  };
  retarget();
#define ISSUE_NEXT()                                                                   \
  do {                                                                                 \
    char* dst = smem + lb * 32768;                                                     \
    stage_glds<AMODE, NPA>(oa, dst, lk * BK, K, wave, lane);                           \
    stage_glds<BMODE>(ob, dst + 16384, lk * BK, K, wave, lane);                        \
    lb ^= 1;                                                                           \
    if (++lk == T) {                                                                   \
      lk = 0;                                                                          \
      lt += G;                                                                         \
      retarget();                                                                      \
    }                                                                                  \
  } while (0)
  ISSUE_NEXT();                      // k-tile 0 of the first tile -> buffer 0
  int cb = 0;                        // buffer holding the k-tile to compute next
  constexpr int NST = EpiStores<EPI, TC, MT>::n;
  EpiRegs pre;
  for (int ct = first; ct < ntiles; ct += G) {
    const int pr = ct >= g.tiles0;
    const GemmProb& P = g.p[pr];
    const GemmEpi& e = P.e;
    const int M = P.M;
    const long ldc = P.ldc;
    TC* C = (TC*)P.C;
    const int lct = ct - (pr ? g.tiles0 : 0);
    const int m0 = (lct / tiles_n) * MT, n0 = (lct % tiles_n) * BN;
    for (int k = 0; k < T; ++k) {
      // this wave's pieces of the current k-tile have landed.  Right after an epilogue the youngest NST operations are that
      // epilogue's stores (a fixed count per thread): skip them instead of draining the HBM write latency.
      if (NST > 0 && k == 0 && ct != first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // ... everyone's have, and the other buffer is no longer being read
      asm volatile("" ::: "memory");
      ISSUE_NEXT();                                        // next k-tile (possibly of the next output tile) -> other buffer
      if (EpiPre<EPI, TC>::on && k == T - 1) epi_prefetch<EPI, TC, MT>(pre, ldc, m0, n0, M, N, e, tid);   // lands under the last MFMAs
      tile_compute<AMODE, BMODE, NI>(smem + cb * 32768, acc, wm, wn, lane);
      cb ^= 1;
    }
    // ---- epilogue, two 64-row halves through the staging buffer that was computed last (the other one is receiving the
    // next tile's first k-tile by DMA meanwhile)
    const long crows = (long)M + (e.patch_rows > 0 ? M / e.patch_rows + 2 : 0);
    const __amdgpu_buffer_rsrc_t crs = make_store_rsrc((void*)C, crows * ldc * (long)sizeof(TC));
    const __amdgpu_buffer_rsrc_t prs = make_store_rsrc(e.preact ? e.preact : (void*)C, crows * ldc * (long)sizeof(TC));
    float* Cs = (float*)(smem + (cb ^ 1) * 32768);
    lds_barrier();                                         // last MFMA fragment reads of this buffer are done
    acc_to_lds_half<0, NI>(Cs, acc, wm, wn, lane);
    lds_barrier();
    half_epilogue<EPI, TC, true, MT>(Cs, C, ldc, m0, n0, M, N, e, tid, 0, crs, prs, pre.bias, pre.rin0, pre.sc0);
    if (MT == 128) {
      lds_barrier();
      acc_to_lds_half<(MT == 128 ? 1 : 0), NI>(Cs, acc, wm, wn, lane);
      lds_barrier();
      half_epilogue<EPI, TC, true, MT>(Cs, C, ldc, m0, n0, M, N, e, tid, 1, crs, prs, pre.bias, pre.rin1, pre.sc1);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();                                         // image reads done before the next DMA may target this buffer
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_NEXT
}


#ifdef FC_PROBES      // round-5 experiments (64-row ring, split reduction): faster stand-alone, slower in the step (profiles/r05) -- tools build only
// ======================================================================== 64 x 128 tiles, one per workgroup, deep staging ring
// The launches this is for: a chain's N = 384 products with a long reduction (fc2 forward, fc1 dX: K = 1 536; qkv dX: K = 1 152) at
// 4 334 rows -- 102 tiles of 128 x 128, one workgroup alone on each of 102 CUs walking 18-24 k-steps, each k-step waiting out a whole
// L2 / HBM round trip (one k-tile in flight): 24.4 us, the longest kernels of a layer, where the vendor library's 128 x 64 tiles take
// 18.2 / 15.8 us (profiles/r05/vendor_yardstick_*.txt).  Here: twice as many tiles (64 rows), NS stages of 24 KB (A 8 KB | B 16 KB) with
// NS - 1 k-tiles in flight behind a counted vmcnt, one tile per workgroup (no cross-tile bookkeeping), 72-96 KB of LDS so that a 64-KB
// workgroup of another chain still fits beside it (the 128-KB four-stage form of round 4 shut the other chains out and lost in the step).
template <int BMODE, typename TC, int EPI, int NS>
__global__ void __launch_bounds__(256, 1) k_gemm_d64(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // NS x (A [64][64] 8 KB | B 16 KB); the 32-KB epilogue image aliases it
  constexpr int MT = 64, NI = 2, SLOT = 24576;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int N = g.N, K = g.K, tiles_n = g.tiles_n;
  const int T = (K + BK - 1) / BK;
  const int ct = xcd_remap(blockIdx.x, gridDim.x);
  const int pr = ct >= g.tiles0;
  const GemmProb& P = g.p[pr];
  const GemmEpi& e = P.e;
  const int M = P.M;
  const long ldc = P.ldc;
  TC* C = (TC*)P.C;
  const int lct = ct - (pr ? g.tiles0 : 0);
  const int m0 = (lct / tiles_n) * MT, n0 = (lct % tiles_n) * BN;
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Operand oa = make_operand_glds<KC>(P.A, P.lda, 0, M, K, wave, lane);
  Operand ob = make_operand_glds<BMODE>(P.B, P.ldb, 0, N, K, wave, lane);
  retarget_glds<KC, 2>(oa, P.lda, m0, M, wave, lane, true);
  retarget_glds<BMODE>(ob, P.ldb, n0, N, wave, lane, true);
  // k-tiles past the end fall outside the descriptors (zeros into a stage nobody reads again): every wave issues 6 DMAs per step, so the
  // counted wait is the same immediate from the first step to the last
#define D64_ISSUE(kt)                                                                  \
  do {                                                                                 \
    char* dst = smem + ((kt) % NS) * SLOT;                                             \
    stage_glds<KC, 2>(oa, dst, (kt) * BK, K, wave, lane);                              \
    stage_glds<BMODE>(ob, dst + 8192, (kt) * BK, K, wave, lane);                       \
  } while (0)
#pragma unroll 1
  for (int d = 0; d < NS - 1; ++d) D64_ISSUE(d);
  EpiRegs pre;
  const int kpre = T >= 2 ? T - 2 : 0;
#pragma unroll 1
  for (int k = 0; k < T; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (NS - 2)) : "memory");      // this wave's pieces of k-tile k have landed; NS - 2 younger k-tiles stay in flight
    __builtin_amdgcn_s_barrier();                                            // ... everyone's have, and the stage computed last is free again
    asm volatile("" ::: "memory");
    D64_ISSUE(k + NS - 1);
    if (EpiPre<EPI, TC>::on && k == kpre) epi_prefetch<EPI, TC, MT>(pre, ldc, m0, n0, M, N, e, tid);
    {
      const char* buf = smem + (k % NS) * SLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[NI], bfr[4];
#pragma unroll
        for (int i = 0; i < NI; ++i) af[i] = frag_read<KC>(buf, wm * 32 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BMODE>(buf + 8192, wn * 64 + j * 16, ks, lane);
        if (BMODE == KR) frag_fence(bfr);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }
#undef D64_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the over-issued (all-zero) k-tiles land before the image reuses the ring
  const long crows = (long)M + (e.patch_rows > 0 ? M / e.patch_rows + 2 : 0);
  const __amdgpu_buffer_rsrc_t crs = make_store_rsrc((void*)C, crows * ldc * (long)sizeof(TC));
  const __amdgpu_buffer_rsrc_t prs = make_store_rsrc(e.preact ? e.preact : (void*)C, crows * ldc * (long)sizeof(TC));
  float* Cs = (float*)smem;
  lds_barrier();
  acc_to_lds_half<0, NI>(Cs, acc, wm, wn, lane);
  lds_barrier();
  half_epilogue<EPI, TC, true, MT>(Cs, C, ldc, m0, n0, M, N, e, tid, 0, crs, prs, pre.bias, pre.rin0, pre.sc0);
}

// ======================================================================== split reduction ACROSS workgroups (round 5; EXPERIMENT, tools build only)
// The same under-filled long-reduction launches (102 tiles of 18-24 k-steps at chain size), a third way: each output tile is computed by TWO
// workgroups of the ordinary shape (128 x 128 tile, two 32-KB stages, 64 KB of LDS: two per CU beside anything else), each walking HALF of the
// k-tiles.  The 64-row forms above either kept the k-chain (64-row tiles: twice the workgroups for the same time = twice the CU-slot time) or
// bought the speed with LDS (ring: 72-96 KB).  This one halves the serial chain at the SAME CU-slot time: 204 workgroups x ~half the duration.
// Exchange: a workgroup writes its partial accumulators (fp32, register order: 64 KB, coalesced) to a per-stream scratch, releases, and bumps
// the tile's flag; the workgroup that finds the flag already bumped acquires, adds the other partial and runs the epilogue.  a + b == b + a
// bit for bit, so the result does not depend on which half arrives last; it differs from the one-workgroup kernel's by the order of summation.
// Measured (profiles/r05/gemm_splitk_cold.txt, gemm_splitk_in_step_ab.txt): 24.5 -> 19.2 us (fc2 forward) and 20-24 -> 15.3 us (fc1 / qkv dX) at chain
// size, 24.3 -> 18.6 / 14.4 at the text tower's 2 048 rows, 30.6 -> 42.4 at the full batch -- and the B = 64 step 4.55 -> 4.70 ms (+3 %): 204
// workgroups x 15-19 us is MORE workgroup-time than 102 x 24.5, and the step pays for workgroup-time, not for one kernel's latency.
template <int BMODE, typename TC, int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_sk(GemmGroup g, float* __restrict__ part, unsigned* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x (A 16 KB | B 16 KB); the epilogue image aliases buffer 0
  constexpr int MT = 128, NI = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int N = g.N, K = g.K, tiles_n = g.tiles_n;
  const int T = K / BK, Th = T >> 1;                               // the launcher checks K % (2 BK) == 0
  const int id = xcd_remap(blockIdx.x, gridDim.x);                 // the two halves of a tile are neighbours: same XCD, same L2
  const int ct = id >> 1, kh = id & 1;
  const GemmProb& P = g.p[0];
  const GemmEpi& e = P.e;
  const int M = P.M;
  const long ldc = P.ldc;
  TC* C = (TC*)P.C;
  const int m0 = (ct / tiles_n) * MT, n0 = (ct % tiles_n) * BN;
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Operand oa = make_operand_glds<KC>(P.A, P.lda, 0, M, K, wave, lane);
  Operand ob = make_operand_glds<BMODE>(P.B, P.ldb, 0, N, K, wave, lane);
  retarget_glds<KC>(oa, P.lda, m0, M, wave, lane, true);
  retarget_glds<BMODE>(ob, P.ldb, n0, N, wave, lane, true);
  const int k0 = kh * Th;
#define SK_ISSUE(kt, b)                                                                \
  do {                                                                                 \
    char* dst = smem + (b) * 32768;                                                    \
    stage_glds<KC>(oa, dst, (kt) * BK, K, wave, lane);                                 \
    stage_glds<BMODE>(ob, dst + 16384, (kt) * BK, K, wave, lane);                      \
  } while (0)
  SK_ISSUE(k0, 0);
#pragma unroll 1
  for (int k = 0; k < Th; ++k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (k + 1 < Th) SK_ISSUE(k0 + k + 1, (k + 1) & 1);
    tile_compute<KC, BMODE, NI>(smem + (k & 1) * 32768, acc, wm, wn, lane);
  }
#undef SK_ISSUE
  // ---- hand the partial over, or take the other one.  Coherence by ACCESS, not by fence: the partial tiles are stored and loaded with the
  // system-scope cache policy (sc0 sc1: written through to memory, read past the caches), the flag is an agent-scope atomic.  A release /
  // acquire fence pair (__threadfence) costs a write-back of the XCD's whole L2 per wave here: 24 -> 53 us per launch, measured.
  float* mine = part + ((size_t)ct * 2 + kh) * (MT * BN);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(mine + ((i * 4 + j) * 256 + tid) * 4), "v"(acc[i][j]) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this thread's partial has reached memory ...
  __shared__ unsigned arrived;
  __builtin_amdgcn_s_barrier();                                    // ... and so has every other thread's, before the flag moves
  if (tid == 0) arrived = __hip_atomic_fetch_add(flags + ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (arrived == 0) return;                                        // first of the two: the other half finishes the tile
  EpiRegs pre;
  if (EpiPre<EPI, TC>::on) epi_prefetch<EPI, TC, MT>(pre, ldc, m0, n0, M, N, e, tid);
  const float* other = part + ((size_t)ct * 2 + (kh ^ 1)) * (MT * BN);
  f32x4 o[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(o[i][j]) : "v"(other + ((i * 4 + j) * 256 + tid) * 4) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(o[i][j]));                            // (ties the loaded registers behind the wait)
      acc[i][j] += o[i][j];
    }
  if (tid == 0) __hip_atomic_store(flags + ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
  const long crows = (long)M + (e.patch_rows > 0 ? M / e.patch_rows + 2 : 0);
  const __amdgpu_buffer_rsrc_t crs = make_store_rsrc((void*)C, crows * ldc * (long)sizeof(TC));
  const __amdgpu_buffer_rsrc_t prs = make_store_rsrc(e.preact ? e.preact : (void*)C, crows * ldc * (long)sizeof(TC));
  float* Cs = (float*)smem;
  lds_barrier();
  acc_to_lds_half<0, NI>(Cs, acc, wm, wn, lane);
  lds_barrier();
  half_epilogue<EPI, TC, true, MT>(Cs, C, ldc, m0, n0, M, N, e, tid, 0, crs, prs, pre.bias, pre.rin0, pre.sc0);
  lds_barrier();
  acc_to_lds_half<1, NI>(Cs, acc, wm, wn, lane);
  lds_barrier();
  half_epilogue<EPI, TC, true, MT>(Cs, C, ldc, m0, n0, M, N, e, tid, 1, crs, prs, pre.bias, pre.rin1, pre.sc1);
}
#endif      // FC_PROBES (split reduction across workgroups)

// ======================================================================== grouped weight-gradient GEMM
// All dW = dY^T . X products of a backward pass (every linear of every layer of both towers) in ONE launch, each output
// tile owned by exactly one workgroup that walks the whole reduction: no split-K, no atomics, bitwise reproducible.
// The bias gradient (column sums of dY) falls out of the A-operand staging registers of the tile_n == 0 workgroups.
// OPT: the epilogue also takes the AdamW step of the elements it produced (parameters, moments, bf16 shadow at the same element
// offsets as the gradient): the optimizer's pass over the linears' weights -- 86 % of the parameters, 7 HBM streams -- disappears into
// the weight-gradient launches that run under the backward.
template <bool OPT>
__global__ void __launch_bounds__(256, 2) k_gemm_tn_grouped(const FcTnProblem* __restrict__ probs, int nprob, FcAdamW o_) {
  const FcAdamW o = fc_adamw_resolve(o_);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int idx = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  while (pi + 1 < nprob && probs[pi + 1].tile_start <= idx) ++pi;
  const FcTnProblem P = probs[pi];
  const int local = idx - P.tile_start;
  const int tile_m = local / P.tiles_n, tile_n = local % P.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = P.M, N = P.N;
  const bool do_colsum = (tile_n == 0) && (P.bias_grad != nullptr);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) cs[i] = 0.f;
  gemm_mainloop<KR, KR, true>(P.A, P.lda, P.B, P.ldb, m0, n0, M, N, P.K, smem, acc, cs, do_colsum, tid, wm, wn, lane);
  if (do_colsum) {  // thread (c = tid&15, kgroup = tid>>4) holds sums of columns 8c..8c+7 over its k rows
    float* R = (float*)smem;  // [16 kgroups][128 cols]; the staging buffers are free after the main loop's last barrier
#pragma unroll
    for (int i = 0; i < 8; ++i) R[(tid >> 4) * 128 + (tid & 15) * 8 + i] = cs[i];
    lds_barrier();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) sum += R[kg * 128 + tid];
      if (m0 + tid < M) {
        P.bias_grad[m0 + tid] = sum;
        if (OPT) {
          const size_t idx = (size_t)(P.bias_grad + m0 + tid - o.g0);
          float pp = o.p[idx], mm = o.m[idx], vv = o.v[idx];
          fc_adamw_elem(pp, sum, mm, vv, o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
          o.p[idx] = pp; o.m[idx] = mm; o.v[idx] = vv;
          if (o.shadow) o.shadow[idx] = f2bf(pp);
        }
      }
    }
  }
  lds_barrier();
  float* Cs = (float*)smem;
  acc_to_lds(Cs, acc, wm, wn, lane);
  lds_barrier();
  if (!OPT) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int row = (tid >> 4) + 16 * p, c8 = (tid & 15) * 8;
      int m = m0 + row, n = n0 + c8;
      if (m < M && n < N) {
        float* dst = P.C + (size_t)m * P.ldc + n;
        *(float4*)dst = *(const float4*)(Cs + row * CS_LD + c8);
        *(float4*)(dst + 4) = *(const float4*)(Cs + row * CS_LD + c8 + 4);
      }
    }
    return;
  }
  // 16 consecutive threads own one 512-B row segment of each stream; two rows (12 x 16-B loads) in flight per thread
#pragma unroll 2
  for (int p = 0; p < 8; ++p) {
    const int row = (tid >> 4) + 16 * p, c8 = (tid & 15) * 8;
    const int m = m0 + row, n = n0 + c8;
    if (m < M && n < N) {
      float* dst = P.C + (size_t)m * P.ldc + n;
      const size_t idx = (size_t)(dst - o.g0);
      float4 g[2], pp[2], mm[2], vv[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pp[h] = *(const float4*)(o.p + idx + 4 * h);
        mm[h] = *(const float4*)(o.m + idx + 4 * h);
        vv[h] = *(const float4*)(o.v + idx + 4 * h);
        g[h] = *(const float4*)(Cs + row * CS_LD + c8 + 4 * h);
      }
      bf16_t sh[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float* Pp = (float*)&pp[h]; float* G = (float*)&g[h]; float* Mm = (float*)&mm[h]; float* V = (float*)&vv[h];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fc_adamw_elem(Pp[k], G[k], Mm[k], V[k], o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
          sh[4 * h + k] = f2bf(Pp[k]);
        }
        *(float4*)(dst + 4 * h) = g[h];
        *(float4*)(o.p + idx + 4 * h) = pp[h];
        *(float4*)(o.m + idx + 4 * h) = mm[h];
        *(float4*)(o.v + idx + 4 * h) = vv[h];
      }
      if (o.shadow) *(uint4*)(o.shadow + idx) = *(const uint4*)sh;
    }
  }
}

int fc_gemm_tn_grouped_supported(const FcTnProblem& p) {
  return !((p.M & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) ||
           ((uintptr_t)p.C & 15));
}
int fc_gemm_tn_grouped(const FcTnProblem* probs_dev, int nprob, int total_tiles, hipStream_t s, const FcAdamW* opt) {
  if (nprob <= 0 || total_tiles <= 0) return 0;
  const int lds = BM * CS_LD * 4;
  static bool done = false;
  if (!done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_tn_grouped<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_tn_grouped<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
  }
  if (opt) hipLaunchKernelGGL(k_gemm_tn_grouped<true>, dim3(total_tiles), dim3(256), lds, s, probs_dev, nprob, *opt);
  else hipLaunchKernelGGL(k_gemm_tn_grouped<false>, dim3(total_tiles), dim3(256), lds, s, probs_dev, nprob, FcAdamW());
  FC_LAUNCH_CHECK();
  return 0;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int AM, int BMo, typename TC, int EPI, int MT = 128>
static int launch_gemm_epi(const GemmGroup& g, hipStream_t s) {
  const int lds = 65536;  // two 32-KB staging buffers; the epilogue image aliases the idle one
  auto kfn = k_gemm_mfma<AM, BMo, TC, EPI, MT>;
  static bool attr_done = false;  // one flag per instantiation
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  static int max_wg = 0;
  if (!max_wg) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    max_wg = 2 * cus;   // 64 KB of LDS per workgroup: two per CU
  }
  int grid = g.ntiles < max_wg ? g.ntiles : max_wg;
  // FC_GEMM_TPW (tools build): at least this many tiles per workgroup, so that a workgroup's epilogue runs under its next tile's loads
  // even when the launch has fewer tiles than the chip has slots; FC_GEMM_MAXWG caps the grid
  static const int tpw = fc_knob("FC_GEMM_TPW", 1), cap = fc_knob("FC_GEMM_MAXWG", 0);
  if (tpw > 1 && grid > (g.ntiles + tpw - 1) / tpw) grid = (g.ntiles + tpw - 1) / tpw;
  if (cap > 0 && grid > cap) grid = cap;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, g);
  FC_LAUNCH_CHECK();
  return 0;
}
static int epi_kind(const GemmEpi& e) {
  if (e.accumulate || e.dbg) return EPI_GENERIC;
  int extras = (e.res != nullptr) + (e.preact != nullptr) + (e.gelu_in != nullptr) + (e.patch_rows > 0);
  if (extras > 1) return EPI_GENERIC;
  if (e.patch_rows > 0) return e.bias ? EPI_PATCH : EPI_GENERIC;
  if (e.preact) return (e.bias && !e.rowscale) ? (e.gelu_saved_grad ? EPI_GELU_SG : EPI_GELU) : EPI_GENERIC;
  if (e.gelu_in) return (!e.bias && !e.rowscale) ? (e.gelu_saved_grad ? EPI_MUL : EPI_GELU_GRAD) : EPI_GENERIC;
  if (e.res) return !e.bias ? EPI_GENERIC : (e.rowscale ? EPI_RES_SCALE : EPI_RES);
  if (e.rowscale) return EPI_GENERIC;
  return e.bias ? EPI_BIAS : EPI_PLAIN;
}
// the epilogue kinds each GEMM form is instantiated for (anything else takes the run-time-flag EPI_GENERIC body)
template <int AM, int BMo, typename TC>
static int launch_gemm(const GemmGroup& g, int k, hipStream_t s, bool mt64 = false) {
#define GO(E) return launch_gemm_epi<AM, BMo, TC, E>(g, s)
#define GO64(E) return launch_gemm_epi<AM, BMo, TC, E, 64>(g, s)
  if (AM == KC && BMo == KC && sizeof(TC) == 2) {   // forward linears
#ifdef FC_PROBES
    if constexpr (AM == KC && BMo == KC && sizeof(TC) == 2) {
      if (mt64) switch (k) { case EPI_BIAS: GO64(EPI_BIAS); case EPI_RES: GO64(EPI_RES); case EPI_RES_SCALE: GO64(EPI_RES_SCALE); }
    }
#endif
    switch (k) { case EPI_BIAS: GO(EPI_BIAS); case EPI_RES: GO(EPI_RES); case EPI_RES_SCALE: GO(EPI_RES_SCALE); case EPI_GELU: GO(EPI_GELU);
                 case EPI_PATCH: GO(EPI_PATCH); case EPI_PLAIN: GO(EPI_PLAIN); case EPI_GELU_SG: GO(EPI_GELU_SG); }
  } else if (AM == KC && BMo == KR && sizeof(TC) == 2) {   // dX
#ifdef FC_PROBES
    if constexpr (AM == KC && BMo == KR && sizeof(TC) == 2) {
      if (mt64 && k == EPI_PLAIN) GO64(EPI_PLAIN);
    }
#endif
    switch (k) { case EPI_PLAIN: GO(EPI_PLAIN); case EPI_GELU_GRAD: GO(EPI_GELU_GRAD); case EPI_BIAS: GO(EPI_BIAS); case EPI_MUL: GO(EPI_MUL); }
  } else {
    switch (k) { case EPI_PLAIN: GO(EPI_PLAIN); case EPI_BIAS: GO(EPI_BIAS); }
  }
  GO(EPI_GENERIC);
#undef GO
#undef GO64
}
#ifdef FC_PROBES
template <int BMo, typename TC, int EPI, int NS>
static int launch_gemm_d64(const GemmGroup& g, hipStream_t s) {
  const int lds = NS * 24576;
  auto kfn = k_gemm_d64<BMo, TC, EPI, NS>;
  static bool attr_done = false;
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(g.ntiles), dim3(256), lds, s, g);
  FC_LAUNCH_CHECK();
  return 0;
}
// the deep 64-row form: bf16, long reductions, the epilogues of the model's N = 384 products.  Returns 1 when it has no such instantiation.
static int launch_gemm_deep(int kind, int ek, int stages, const GemmGroup& g, hipStream_t s) {
#define GO_D(BMo, E) return stages == 3 ? launch_gemm_d64<BMo, bf16_t, E, 3>(g, s) : launch_gemm_d64<BMo, bf16_t, E, 4>(g, s)
  if (kind == FC_GEMM_NT) {
    switch (ek) { case EPI_RES: GO_D(KC, EPI_RES); case EPI_RES_SCALE: GO_D(KC, EPI_RES_SCALE); case EPI_BIAS: GO_D(KC, EPI_BIAS); }
  } else if (kind == FC_GEMM_NN) {
    if (ek == EPI_PLAIN) GO_D(KR, EPI_PLAIN);
  }
#undef GO_D
  return 1;
}
// per-(device, stream) scratch of the split-reduction form: partial tiles + one flag per tile (a stream runs one GEMM at a time)
#define SK_MAX_TILES 512
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, char*> g_sk_ws;
static int sk_scratch(hipStream_t s, float** part, unsigned** flags) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  int dev = 0;
  FC_CHECK_HIP(hipGetDevice(&dev));
  char*& w = g_sk_ws[std::make_pair(dev, s)];
  const size_t pbytes = (size_t)SK_MAX_TILES * 2 * BM * BN * sizeof(float);
  if (!w) {                                                        // once per stream and process (allocates and zero-fills the flags)
    FC_CHECK_HIP(hipMalloc((void**)&w, pbytes + SK_MAX_TILES * sizeof(unsigned)));
    FC_CHECK_HIP(hipMemset(w + pbytes, 0, SK_MAX_TILES * sizeof(unsigned)));
  }
  *part = (float*)w;
  *flags = (unsigned*)(w + pbytes);
  return 0;
}
template <int BMo, typename TC, int EPI>
static int launch_gemm_sk(const GemmGroup& g, hipStream_t s) {
  const int lds = 65536;
  auto kfn = k_gemm_sk<BMo, TC, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  float* part; unsigned* flags;
  FC_TRY(sk_scratch(s, &part, &flags));
  hipLaunchKernelGGL(kfn, dim3(2 * g.ntiles), dim3(256), lds, s, g, part, flags);
  FC_LAUNCH_CHECK();
  return 0;
}
static int launch_gemm_splitk(int kind, int ek, const GemmGroup& g, hipStream_t s) {
  if (kind == FC_GEMM_NT) {
    switch (ek) { case EPI_RES: return launch_gemm_sk<KC, bf16_t, EPI_RES>(g, s); case EPI_RES_SCALE: return launch_gemm_sk<KC, bf16_t, EPI_RES_SCALE>(g, s);
                  case EPI_BIAS: return launch_gemm_sk<KC, bf16_t, EPI_BIAS>(g, s); }
  } else if (kind == FC_GEMM_NN) {
    if (ek == EPI_PLAIN) return launch_gemm_sk<KR, bf16_t, EPI_PLAIN>(g, s);
  }
  return 1;
}
#endif
// process-wide form of the under-filled launches (fc_model_set_option FC_OPT_GEMM_FORM; tools build: FC_GEMM_MT64 / FC_GEMM_DEEP64):
// 0 = 128-row tiles, 64 = 64-row tiles, 3 / 4 = 64-row tiles with a 3- / 4-stage ring
#ifdef FC_PROBES
static int g_gemm_form = 0;
void fc_gemm_set_form(int form) { g_gemm_form = form; }
#endif
#ifdef FC_PROBES
// 64-row tiles exist for these (kind, output type, epilogue) combinations
static bool gemm_has_mt64(int kind, int dtC, int ek) {
  if (dtC != FC_BF16) return false;
  if (kind == FC_GEMM_NT) return ek == EPI_BIAS || ek == EPI_RES || ek == EPI_RES_SCALE;
  if (kind == FC_GEMM_NN) return ek == EPI_PLAIN;
  return false;
}

#endif
static int gemm_prob_ok(int kind, const GemmProb& p, int N, int K) {
  // vector-width constraints of this kernel; anything else goes to the generic path
  if (p.M <= 0) return 0;
  if ((N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 7) || !aligned16(p.A) || !aligned16(p.B) || !aligned16(p.C)) return 0;
  if (kind != FC_GEMM_TN && (K & 7)) return 0;
  if (kind == FC_GEMM_TN && (p.M & 7)) return 0;
  const GemmEpi& e = p.e;
  if (e.bias && !aligned16(e.bias)) return 0;
  if (e.res && !aligned16(e.res)) return 0;
  if (e.preact && !aligned16(e.preact)) return 0;
  if (e.gelu_in && !aligned16(e.gelu_in)) return 0;
  if (e.pos && !aligned16(e.pos)) return 0;
  return 1;
}
// one or two problems (same N, K, epilogue kind) in one launch; returns 1 when not covered (the caller launches them one by one)
int fc_gemm_mfma_grouped(int kind, int dtC, GemmGroup g, int nprob, hipStream_t s) {
  if (FC_ABLATED("gemm")) return 0;
  if (nprob < 1 || nprob > 2 || g.N <= 0 || g.K <= 0) return 1;
  const int ek = epi_kind(g.p[0].e);
  for (int i = 0; i < nprob; ++i) {
    if (!gemm_prob_ok(kind, g.p[i], g.N, g.K)) return 1;
    if (epi_kind(g.p[i].e) != ek) return 1;
#ifdef FC_PROBES
    static const int dbg = getenv("FC_GEMM_DBG") ? atoi(getenv("FC_GEMM_DBG")) : 0;
    g.p[i].e.dbg = dbg;
#endif
  }
  g.tiles_n = fc_cdiv(g.N, BN);
  int mt = BM;
#ifdef FC_PROBES
  {
    // a launch whose 128-row tiles would occupy less than about half of the chip's workgroup slots is cut into 64-row tiles instead
    static const int thr_env = fc_knob("FC_GEMM_MT64", 0);      // (measured: 23.4 / 21.3 against 24.4 / 25.0 us stand-alone, +1 % in the step: off)
    const int thr = thr_env > 0 ? thr_env : (g_gemm_form == 64 ? 128 : 0);
    int t128 = 0;
    for (int i = 0; i < nprob; ++i) t128 += fc_cdiv(g.p[i].M, BM) * g.tiles_n;
    if (thr > 0 && t128 <= thr && gemm_has_mt64(kind, dtC, ek)) mt = 64;
  }
#endif
#ifdef FC_PROBES
  {
    // ... or, when its reduction is long (K >= 1 024), computed by two workgroups per tile, half of the k-tiles each (tools build: FC_GEMM_SPLITK=1)
    static const int sk_env = fc_knob("FC_GEMM_SPLITK", 0);
    const bool sk = sk_env > 0;
    const int t128 = fc_cdiv(g.p[0].M, BM) * g.tiles_n;
    if (sk && nprob == 1 && dtC == FC_BF16 && t128 <= SK_MAX_TILES && g.K >= 1024 && (g.K % (2 * BK)) == 0 && gemm_has_mt64(kind, dtC, ek) &&
        g.p[0].e.patch_rows == 0) {
      GemmGroup d = g;
      d.tiles0 = t128; d.ntiles = t128; d.p[1] = d.p[0];
      const int r = launch_gemm_splitk(kind, ek, d, s);
      if (r <= 0) return r;
    }
  }
#endif
#ifdef FC_PROBES
  {
    // ... and when its reduction is long (K >= 1 024: fc2 forward, fc1 / qkv dX of a chain) into 64-row tiles with a deep staging ring, one per workgroup
    static const int deep_env = fc_knob("FC_GEMM_DEEP64", 0);      // stages (3 | 4), 0 = off (stand-alone 24.4 -> 15.2 us, in the step +2 ... +6 %: off)
    const int deep = deep_env >= 3 ? deep_env : ((g_gemm_form == 3 || g_gemm_form == 4) ? g_gemm_form : 0);
    int t128 = 0;
    for (int i = 0; i < nprob; ++i) t128 += fc_cdiv(g.p[i].M, BM) * g.tiles_n;
    if (deep >= 3 && dtC == FC_BF16 && t128 <= 128 && g.K >= 1024 && (g.K % BK) == 0 && gemm_has_mt64(kind, dtC, ek)) {
      GemmGroup d = g;
      d.tiles0 = fc_cdiv(d.p[0].M, 64) * d.tiles_n;
      d.ntiles = d.tiles0 + (nprob > 1 ? fc_cdiv(d.p[1].M, 64) * d.tiles_n : 0);
      if (nprob == 1) d.p[1] = d.p[0];
      const int r = launch_gemm_deep(kind, ek, deep, d, s);
      if (r <= 0) return r;
    }
  }
#endif
  g.tiles0 = fc_cdiv(g.p[0].M, mt) * g.tiles_n;
  g.ntiles = g.tiles0 + (nprob > 1 ? fc_cdiv(g.p[1].M, mt) * g.tiles_n : 0);
  if (nprob == 1) g.p[1] = g.p[0];
  if (kind == FC_GEMM_NT) {
    if (dtC == FC_BF16) return launch_gemm<KC, KC, bf16_t>(g, ek, s, mt == 64);
    return launch_gemm<KC, KC, float>(g, ek, s);
  }
  if (kind == FC_GEMM_NN) {
    if (dtC == FC_BF16) return launch_gemm<KC, KR, bf16_t>(g, ek, s, mt == 64);
    return launch_gemm<KC, KR, float>(g, ek, s);
  }
  // TN (weight gradients normally go through fc_gemm_tn_grouped instead)
  if (dtC != FC_F32) return 1;
  return launch_gemm<KR, KR, float>(g, ek, s);
}

int fc_gemm_mfma(int kind, int dtC, const bf16_t* A, long lda, const bf16_t* Bm, long ldb, void* C, long ldc, int M, int N, int K,
                 const GemmEpi& epi_in, hipStream_t s) {
  if (FC_ABLATED("gemm")) return 0;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  GemmGroup g{};
  g.p[0] = GemmProb{A, Bm, C, lda, ldb, ldc, M, epi_in};
  g.N = N; g.K = K;
  return fc_gemm_mfma_grouped(kind, dtC, g, 1, s);
}
