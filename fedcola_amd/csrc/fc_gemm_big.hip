// Large-tile bf16 MFMA GEMM for gfx950, grouped over up to two row sets (the image and the text tower of one layer in ONE launch).
//
//   C_p[M_p, N] = A_p[M_p, K] . W_p[N, K]^T  (+ fused epilogue),  p = 0, 1        (forward linears, mome.py:117-123, 150-168)
//
// The dX products of the backward (dX = dY . W) run through the same kernel against a TRANSPOSED bf16 copy of the weights that the
// library refreshes once per optimizer step (fc_transpose_linears): one operand form (both k-contiguous), one LDS image, one swizzle.
//
// Why another GEMM: a CU takes in ~70 GB/s through the vector-memory -> LDS path, and an LDS-DMA instruction costs its wave 40-180
// issue cycles per 1-KB piece (MI355X_MICROARCH.md, "ldsdma-fill", "LDS-DMA piece issue cost").  The 128x128 tile of fc_mfma.hip does
// 64 FLOP per staged byte and 0.25 pieces per MFMA: a k-step costs ~1400 cycles for 512 cycles of MFMA.  Here one
// 8-wave workgroup per CU owns a 256x192 tile (110 FLOP per staged byte, 0.15 pieces per MFMA; two waves per SIMD, so one wave's DMA
// issue runs under its partner's MFMAs) or, where N = 384 leaves too few 256-row tiles, a 128x192 tile with a three-stage ring.
// 192 divides D, 3D and 4D for every width the reference builds (192, 384, 768).
//
// Structure: persistent workgroups (<= 1 per CU) walk the tile list of both problems; operand tiles [rows][64 k] stream global -> LDS by
// LDS-DMA into an NS-stage ring (swizzled on the source address: fc_mfma_dev.h kc_off), the load side runs NS-1 k-steps ahead of the
// MFMAs ACROSS tile boundaries (the next tile's first k-steps land under the epilogue); one barrier per k-step.  Wave (wm, wn) owns a
// 64 x (192 / WN) block: 4 x NJ accumulators of v_mfma_f32_16x16x32_bf16 with swapped operands (a lane owns 4 consecutive columns).
// Epilogue: per 16-row fragment the wave writes its fp32 block to a PRIVATE region of the idle ring stage, reads it back as whole
// 16-byte output chunks (rows of 192 B / 96 B per wave), applies bias / GELU / residual / drop-path scale in fp32 and stores bf16.  No
// workgroup barrier inside the epilogue.  Every global input of the epilogue is requested before the tile's last k-step; stores use
// out-of-range offsets for masked lanes so that every wave issues the same number and the main loop's waits are counted.
#ifdef FC_PROBES   // FC_PROBES: whole file -- an experiment kept in the tools build only (large-tile grouped NT GEMM); the product library does not contain it
#include <stdlib.h>
#include <string.h>

#include "fc_kernels.h"
#include "fc_mfma_dev.h"

#define GB_BN 192
#define GB_BK 64

enum { GB_PLAIN = 0, GB_BIAS, GB_RES, GB_RES_SCALE, GB_GELU_SG, GB_MUL };

template <int EPI> struct GbEpi {
  static constexpr bool has_bias = EPI == GB_BIAS || EPI == GB_RES || EPI == GB_RES_SCALE || EPI == GB_GELU_SG;
  static constexpr bool has_in = EPI == GB_RES || EPI == GB_RES_SCALE || EPI == GB_MUL;
  static constexpr int n_st = EPI == GB_GELU_SG ? 2 : 1;
};

template <int TBM, int NS, int NW> struct GbCfg {     // tile rows, ring stages, waves per workgroup
  static constexpr int WM = TBM / 64;                 // waves along M
  static constexpr int WN = NW / WM;                  // waves along N
  static constexpr int WTN = GB_BN / WN;              // columns per wave: 96 / 48
  static constexpr int NJ = WTN / 16;                 // column fragments per wave: 6 / 3
  static constexpr int A_BYTES = TBM * 128;
  static constexpr int W_BYTES = GB_BN * 128;
  static constexpr int STAGE = A_BYTES + W_BYTES;     // 57344 / 40960
  static constexpr int PA = TBM / 8 / NW;             // A pieces (8 rows x 128 B) per wave per k-step
  static constexpr int PW = GB_BN / 8 / NW;           // W pieces per wave per k-step
  static constexpr int PIECES = PA + PW;
  static constexpr int CHUNKS = WTN / 8;              // 16-byte output chunks per row of the wave's block: 12 / 6
  static constexpr int ITEMS = (16 * CHUNKS + 63) / 64;   // (row, chunk) items per lane per 16-row fragment: 3 / 2
  static constexpr int XMASK = WTN == 96 ? 7 : 3;     // slot swizzle (stays inside aligned groups of 8 / 4 float4 slots)
  static constexpr int EPI_BYTES = 16 * WTN * 4;      // private staging per wave: 6144 / 3072
  static constexpr int BIAS_OFF = NW * EPI_BYTES;     // [192] floats behind the staging regions, inside the idle ring stage
  static constexpr int LDS = NS * STAGE;              // 114688 (256 rows, 2 stages) / 122880 (128, 3) / 81920 (128, 2, 4 waves: two per CU)
  static_assert(BIAS_OFF + GB_BN * 4 <= STAGE, "epilogue staging + bias must fit the idle ring stage");
  static_assert(PA >= 1 && PA <= 4 && PW >= 1 && PW <= 6, "piece counts");
};

struct GbLoad {           // load side of one operand: descriptor + per-lane offsets of this wave's pieces
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[6];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gb_rsrc(const void* p, long bytes) {
  unsigned long long base = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  void* up = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(up, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
// piece = 8 rows x 128 B of a [rows][64 k] tile; lane -> (row = 8 piece + lane/8, physical chunk lane%8); the swizzle is applied to the
// SOURCE address (the LDS-DMA destination is lane-linear)
template <int NP>
__device__ __forceinline__ void gb_retarget(GbLoad& o, long ld, int row0, int nrows, int piece0, int lane, bool valid) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int row = (piece0 + p) * 8 + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((row >> 1) & 7);
    const int r = row0 + row;
    o.voff[p] = (valid && r < nrows) ? (unsigned)((r * ld + c * 8) * 2) : FC_OOB;
  }
}
template <int NP>
__device__ __forceinline__ void gb_issue(const GbLoad& o, char* tile, int piece0, int k0) {
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(k0 * 2);
#pragma unroll
  for (int p = 0; p < NP; ++p)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(o.rsrc, (lds_ptr_t)(tile + (piece0 + p) * 1024), 16, o.voff[p], soff, 0, 0);
}

struct GbTile { int prob, m0, n0; };
__device__ __forceinline__ GbTile gb_tile(int t, int tiles0, int tiles_n, int TBM) {
  GbTile r;
  r.prob = t >= tiles0;
  const int l = t - (r.prob ? tiles0 : 0);
  r.m0 = (l / tiles_n) * TBM;
  r.n0 = (l % tiles_n) * GB_BN;
  return r;
}

template <int TBM, int NS, int NW, int EPI>
__global__ void __launch_bounds__(NW * 64, 2) k_gemm_big(FcGemmGrouped g, int tiles0, int ntiles, int tiles_n) {
  using Cf = GbCfg<TBM, NS, NW>;
  using Ep = GbEpi<EPI>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cf::WN, wn = wave % Cf::WN;
  const int G = gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);
  const int N = g.N, K = g.K, T = K / GB_BK;

  // ---- load side: tile lt, k-step lk, ring stage ls; runs NS-1 k-steps ahead of the compute side
  GbLoad la, lw;
  int lt = first, lk = 0, ls = 0;
  int lprob = -1;
  auto retarget = [&]() {
    const bool valid = lt < ntiles;
    const GbTile t = gb_tile(valid ? lt : first, tiles0, tiles_n, TBM);
    if (t.prob != lprob) {
      const FcGemmProb& P = g.p[t.prob];
      la.rsrc = gb_rsrc(P.A, ((long)(P.M - 1) * P.lda + K) * 2);
      lw.rsrc = gb_rsrc(P.W, ((long)(N - 1) * P.ldw + K) * 2);
      lprob = t.prob;
    }
    const FcGemmProb& P = g.p[t.prob];
    gb_retarget<Cf::PA>(la, P.lda, t.m0, P.M, wave * Cf::PA, lane, valid);
    gb_retarget<Cf::PW>(lw, P.ldw, t.n0, N, wave * Cf::PW, lane, valid);
  };
  auto issue_next = [&]() {
    char* dst = smem + ls * Cf::STAGE;
    gb_issue<Cf::PA>(la, dst, wave * Cf::PA, lk * GB_BK);
    gb_issue<Cf::PW>(lw, dst + Cf::A_BYTES, wave * Cf::PW, lk * GB_BK);
    ls = ls + 1 == NS ? 0 : ls + 1;
    if (++lk == T) {
      lk = 0;
      lt += G;
      retarget();
    }
  };
  retarget();
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue_next();

  // bias of a tile: loaded one tile ahead (before the previous tile's stores: vmcnt retires in order, a load behind stores waits them
  // out), parked in registers of threads 0..47, written to LDS in front of the epilogue's barrier
  float4 nbias = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_bias = [&](int t) {
    if (Ep::has_bias && tid < GB_BN / 4 && t < ntiles) {
      const GbTile tb = gb_tile(t, tiles0, tiles_n, TBM);
      const int n = tb.n0 + tid * 4;
      nbias = n < N ? *(const float4*)(g.p[tb.prob].bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_bias(first);
  f32x4 acc[4][Cf::NJ];
  int cs = 0;                                    // ring stage of the k-step to compute next
  constexpr int NST = 4 * Cf::ITEMS * Ep::n_st;  // store instructions per wave per tile
  for (int ct = first; ct < ntiles; ct += G) {
    const GbTile tl = gb_tile(ct, tiles0, tiles_n, TBM);
    const FcGemmProb& P = g.p[tl.prob];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < Cf::NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // epilogue items of this lane: (row, chunk) inside the wave's 16 x WTN fragment
    int irow[Cf::ITEMS], ichk[Cf::ITEMS];
#pragma unroll
    for (int p = 0; p < Cf::ITEMS; ++p) {
      const int idx = lane + 64 * p;
      irow[p] = idx / Cf::CHUNKS;                // >= 16: no item (masked)
      ichk[p] = idx % Cf::CHUNKS;
    }
    // epilogue inputs (residual / GELU' operand, drop-path scale) of 16-row fragments 2h, 2h+1: every one of them is requested BEFORE the
    // tile's first store (a load behind a store cannot be waited for without waiting out the store), half of them before the last k-step
    uint4 pin[4][Cf::ITEMS];
    float psc[4][Cf::ITEMS];
    auto load_inputs = [&](int h) {
      const bf16_t* in = (const bf16_t*)(EPI == GB_MUL ? P.mul_in : P.res);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int p = 0; p < Cf::ITEMS; ++p) {
          const int i = 2 * h + ii;
          const int m = tl.m0 + wm * 64 + i * 16 + irow[p], n = tl.n0 + wn * Cf::WTN + ichk[p] * 8;
          const int mc = (irow[p] < 16 && m < P.M) ? m : P.M - 1, nc = n < N ? n : N - 8;
          pin[i][p] = *(const uint4*)(in + (size_t)mc * P.ldc + nc);
          if (EPI == GB_RES_SCALE) psc[i][p] = P.rowscale[mc / P.rows_per_sample];
        }
    };
    for (int k = 0; k < T; ++k) {
      // this wave's pieces of k-step k have landed: all but the younger NS-2 k-steps' DMA (and, on the first k-steps of a tile, the
      // previous epilogue's stores, which are younger than that DMA) must be done
      if (ct != first && k < NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * Cf::PIECES + NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * Cf::PIECES) : "memory");
      __builtin_amdgcn_s_barrier();              // ... everyone's have, and the stage computed last is free
      asm volatile("" ::: "memory");
      issue_next();
      if (Ep::has_in && k == T - 1) load_inputs(0);   // epilogue inputs of fragments 0, 1: they land under the last MFMAs
      const char* st = smem + cs * Cf::STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], wf[Cf::NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag_read<KC>(st, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < Cf::NJ; ++j) wf[j] = frag_read<KC>(st + Cf::A_BYTES, wn * Cf::WTN + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < Cf::NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
      cs = cs + 1 == NS ? 0 : cs + 1;
    }
    // ---- epilogue through the stage computed last (no DMA targets it before the next k-step's barrier)
    char* idle = smem + (cs == 0 ? NS - 1 : cs - 1) * Cf::STAGE;
    float* Es = (float*)(idle + wave * Cf::EPI_BYTES);
    float* bias_s = (float*)(idle + Cf::BIAS_OFF);
    if (Ep::has_in) load_inputs(1);
    lds_barrier();                               // every wave has finished its fragment reads of that stage
    if (Ep::has_bias) {
      if (tid < GB_BN / 4) *(float4*)(bias_s + tid * 4) = nbias;
      load_bias(ct + G);
      lds_barrier();                             // the bias is visible
    }
    const __amdgpu_buffer_rsrc_t crs = make_store_rsrc((void*)P.C, (long)P.M * P.ldc * 2);
    const __amdgpu_buffer_rsrc_t prs = make_store_rsrc(EPI == GB_GELU_SG ? P.preact : (void*)P.C, (long)P.M * P.ldc * 2);
    const int g4 = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < Cf::NJ; ++j)
        *(float4*)(Es + cl * Cf::WTN + (((4 * j + g4) ^ (cl & Cf::XMASK)) << 2)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
#pragma unroll
      for (int p = 0; p < Cf::ITEMS; ++p) {
        const int r = irow[p] < 16 ? irow[p] : 0, c = ichk[p];
        const float4 x0 = *(const float4*)(Es + r * Cf::WTN + (((2 * c) ^ (r & Cf::XMASK)) << 2));
        const float4 x1 = *(const float4*)(Es + r * Cf::WTN + (((2 * c + 1) ^ (r & Cf::XMASK)) << 2));
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const int m = tl.m0 + wm * 64 + i * 16 + irow[p], n = tl.n0 + wn * Cf::WTN + c * 8;
        const bool ok = irow[p] < 16 && m < P.M && n < N;
        const size_t off = (size_t)m * P.ldc + n;
        if (Ep::has_bias) {
          const float4 b0 = *(const float4*)(bias_s + wn * Cf::WTN + c * 8), b1 = *(const float4*)(bias_s + wn * Cf::WTN + c * 8 + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        float rin[8];
        if (Ep::has_in) {
          const uint4 u = pin[i][p];
          rin[0] = __uint_as_float(u.x << 16); rin[1] = __uint_as_float(u.x & 0xffff0000u);
          rin[2] = __uint_as_float(u.y << 16); rin[3] = __uint_as_float(u.y & 0xffff0000u);
          rin[4] = __uint_as_float(u.z << 16); rin[5] = __uint_as_float(u.z & 0xffff0000u);
          rin[6] = __uint_as_float(u.w << 16); rin[7] = __uint_as_float(u.w & 0xffff0000u);
        }
        if (EPI == GB_GELU_SG) {   // one exp / rcp per element serves both gelu(u) (output) and gelu'(u) (saved for the backward)
          float gp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float cdf, pdf;
            gelu_fast_parts(v[e], cdf, pdf);
            gp[e] = cdf + v[e] * pdf;
            v[e] *= cdf;
          }
          buf_store8<bf16_t>(prs, off, ok, gp);
        }
        if (EPI == GB_MUL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= rin[e];
        }
        if (EPI == GB_RES_SCALE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= psc[i][p];
        }
        if (EPI == GB_RES || EPI == GB_RES_SCALE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rin[e];
        }
        buf_store8<bf16_t>(crs, off, ok, v);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------- host side
template <int TBM, int NS, int NW, int EPI>
static int gb_launch(const FcGemmGrouped& g, hipStream_t s) {
  using Cf = GbCfg<TBM, NS, NW>;
  auto kfn = k_gemm_big<TBM, NS, NW, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS));
    attr_done = true;
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  const int tiles_n = fc_cdiv(g.N, GB_BN);
  const int tiles0 = fc_cdiv(g.p[0].M, TBM) * tiles_n;
  const int tiles1 = g.nprob > 1 ? fc_cdiv(g.p[1].M, TBM) * tiles_n : 0;
  const int ntiles = tiles0 + tiles1;
  if (ntiles == 0) return 0;
  const int max_wg = cus * (163840 / Cf::LDS);      // persistent workgroups: as many as are resident at once
  const int grid = ntiles < max_wg ? ntiles : max_wg;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), Cf::LDS, s, g, tiles0, ntiles, tiles_n);
  FC_LAUNCH_CHECK();
  return 0;
}
template <int TBM, int NS, int NW>
static int gb_launch_epi(const FcGemmGrouped& g, hipStream_t s) {
  switch (g.epi) {
    case GB_PLAIN: return gb_launch<TBM, NS, NW, GB_PLAIN>(g, s);
    case GB_BIAS: return gb_launch<TBM, NS, NW, GB_BIAS>(g, s);
    case GB_RES: return gb_launch<TBM, NS, NW, GB_RES>(g, s);
    case GB_RES_SCALE: return gb_launch<TBM, NS, NW, GB_RES_SCALE>(g, s);
    case GB_GELU_SG: return gb_launch<TBM, NS, NW, GB_GELU_SG>(g, s);
    case GB_MUL: return gb_launch<TBM, NS, NW, GB_MUL>(g, s);
  }
  return 1;
}
static bool gb_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int fc_gemm_grouped_epi(const GemmEpi& e) {     // which fused epilogue a GemmEpi asks for (-1: none of this kernel's)
  if (e.accumulate || e.dbg || e.patch_rows > 0 || e.alpha != 1.0f) return -1;
  const int extras = (e.res != nullptr) + (e.preact != nullptr) + (e.gelu_in != nullptr);
  if (extras > 1) return -1;
  if (e.preact) return (e.bias && !e.rowscale && e.gelu_saved_grad) ? GB_GELU_SG : -1;
  if (e.gelu_in) return (!e.bias && !e.rowscale && e.gelu_saved_grad) ? GB_MUL : -1;
  if (e.res) return !e.bias ? -1 : (e.rowscale ? GB_RES_SCALE : GB_RES);
  if (e.rowscale) return -1;
  return e.bias ? GB_BIAS : GB_PLAIN;
}

int fc_gemm_nt_grouped(const FcGemmGrouped& g, hipStream_t s, int force_bm) {
  if (FC_ABLATED("gemm")) return 0;
  if (g.nprob < 1 || g.nprob > 2 || g.N <= 0 || g.K <= 0) return 1;
  if ((g.N % 8) || (g.K % GB_BK) || g.K < 3 * GB_BK || g.epi < 0 || g.epi > GB_MUL) return 1;
  for (int i = 0; i < g.nprob; ++i) {
    const FcGemmProb& p = g.p[i];
    if (p.M <= 0) return 1;
    if ((p.lda & 7) || (p.ldw & 7) || (p.ldc & 7) || !gb_al16(p.A) || !gb_al16(p.W) || !gb_al16(p.C)) return 1;
    if (p.bias && !gb_al16(p.bias)) return 1;
    if (p.res && !gb_al16(p.res)) return 1;
    if (p.preact && !gb_al16(p.preact)) return 1;
    if (p.mul_in && !gb_al16(p.mul_in)) return 1;
    if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.M * p.ldc * 2 >= (1L << 31)) return 1;   // 32-bit buffer offsets
  }
  // force_bm: 256 = 256 x 192 tiles, 8 waves, one workgroup per CU; 128 = 128 x 192, 8 waves, three-stage ring, one per CU;
  // 2 (default) = 128 x 192, 4 waves, two workgroups per CU (one's DMA issue and epilogue run under the other's MFMAs)
  if (force_bm == 256) return gb_launch_epi<256, 2, 8>(g, s);
  if (force_bm == 128) return gb_launch_epi<128, 3, 8>(g, s);
  return gb_launch_epi<128, 2, 4>(g, s);
}

// ---- transposed bf16 copies of the linears' compute weights: dst[in][out] = src[out][in], one launch for every linear of the model
// (table of FcTranspose).  32 x 32 tiles through LDS, 16-byte global accesses on both sides.
__global__ void __launch_bounds__(256) k_transpose_linears(const FcTranspose* __restrict__ tab, const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
  __shared__ bf16_t tile[64][72];
  const FcTranspose e = tab[blockIdx.y];
  const int tiles_c = (e.in + 63) / 64, tiles_r = (e.out + 63) / 64;
  for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {      // 64 rows x 8 chunks of 8
      const int idx = threadIdx.x + 256 * p, r = idx >> 3, c = (idx & 7) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r0 + r < e.out && c0 + c < e.in) v = *(const uint4*)(src + e.src + (size_t)(r0 + r) * e.in + c0 + c);
      *(uint4*)&tile[r][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = threadIdx.x + 256 * p, c = idx >> 3, r = (idx & 7) * 8;   // output row = input column
      if (c0 + c < e.in && r0 + r < e.out) {
        bf16_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = tile[r + i][c];
        *(uint4*)(dst + e.dst + (size_t)(c0 + c) * e.out + r0 + r) = *(const uint4*)o;
      }
    }
  }
}
int fc_transpose_linears(const FcTranspose* tab_dev, int n, const bf16_t* src, bf16_t* dst, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_transpose_linears, dim3(48, n), dim3(256), 0, s, tab_dev, src, dst);
  FC_LAUNCH_CHECK();
  return 0;
}

// ---- kernel-level test entry point (include/fedcola_hip.h): one or two NT problems through the large-tile kernel.
// in*: residual (epi 2, 3) or multiplier (epi 5); out2_*: gelu' output of epi 4.  Returns 1 when the shape is not covered.
extern "C" int fc_k_gemm_big(const void* A0, const void* W0, void* C0, int32_t M0, const void* A1, const void* W1, void* C1, int32_t M1, int32_t N,
                             int32_t K, int32_t epi, const float* bias0, const float* bias1, const void* in0, const void* in1, void* out2_0,
                             void* out2_1, const float* rowscale0, const float* rowscale1, int32_t rows_per_sample, int32_t force_bm, void* stream) {
  FcGemmGrouped g{};
  g.nprob = (A1 && M1 > 0) ? 2 : 1;
  g.N = N; g.K = K; g.epi = epi;
  const void* As[2] = {A0, A1}; const void* Ws[2] = {W0, W1}; void* Cs[2] = {C0, C1};
  const float* bs[2] = {bias0, bias1}; const void* ins[2] = {in0, in1}; void* o2[2] = {out2_0, out2_1}; const float* rsc[2] = {rowscale0, rowscale1};
  const int Ms[2] = {M0, M1};
  for (int i = 0; i < g.nprob; ++i) {
    FcGemmProb& p = g.p[i];
    p.A = (const bf16_t*)As[i]; p.W = (const bf16_t*)Ws[i]; p.C = (bf16_t*)Cs[i];
    p.bias = bs[i];
    p.res = (epi == GB_RES || epi == GB_RES_SCALE) ? ins[i] : nullptr;
    p.mul_in = epi == GB_MUL ? ins[i] : nullptr;
    p.preact = o2[i];
    p.rowscale = rsc[i];
    p.lda = K; p.ldw = K; p.ldc = N;
    p.M = Ms[i]; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  }
  return fc_gemm_nt_grouped(g, (hipStream_t)stream, force_bm);
}
// dst[in][out] = src[out][in] for one matrix (test entry of the transposed-weight pass)
extern "C" int fc_k_transpose(const void* src, void* dst, int32_t out, int32_t in, void* stream) {
  FcTranspose e{0, 0, out, in};
  FcTranspose* dev = nullptr;
  FC_CHECK_HIP(hipMalloc(&dev, sizeof(e)));
  FC_CHECK_HIP(hipMemcpy(dev, &e, sizeof(e), hipMemcpyHostToDevice));
  int r = fc_transpose_linears(dev, 1, (const bf16_t*)src, (bf16_t*)dst, (hipStream_t)stream);
  FC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  FC_CHECK_HIP(hipFree(dev));
  return r;
}

#endif  // FC_PROBES
