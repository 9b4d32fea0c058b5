// Weight-stationary bf16 MFMA GEMM for gfx950, for the linears whose reduction length fits one resident weight chunk
// (K = 384: the ViT-S qkv / proj / fc1 forward products and the fc2 / proj dX products; mome.py:117-123,150-168).
//
//   NT: C[M,N] = A[M,K] . W[N,K]^T      NN: C[M,N] = A[M,K] . W[K,N]        (A, C bf16 row-major, fp32 accumulate)
//
// Why: a CU ingests ~30 B/clk through the vector-memory -> LDS path (MI355X_MICROARCH.md, "Indexed rows: gather into LDS";
// measured on the 128x128-tile kernel of fc_mfma.hip).  A 128x128 tile that stages BOTH operands does
// 64 FLOP per staged byte, i.e. at most ~45 % of the CU's MFMA rate.  Here a workgroup keeps a 128-column chunk of W
// ([128][384] or [384][128], 96 KB) RESIDENT in LDS and streams only A row tiles (128 rows x 64 k = 16-KB pieces, a
// three-slot ring, two pieces in flight behind the MFMAs): 128 FLOP per staged byte.
//
// Work split: one persistent 8-wave workgroup per CU (all 160 KB of LDS; two waves per SIMD, each wave a 32x64 block of the
// tile).  The M tiles are cut into 8 contiguous ranges, one per XCD label (blockIdx % 8: the blocks that share an L2), so that
// an XCD's slice of A (<= 1.3 MB for ViT-S) and all of W stay in its 4-MB L2 while the N chunks re-read them.  Inside an XCD
// the (chunk, m-tile) items are ordered chunk-major and cut into equal contiguous shares: a workgroup touches at most two
// chunks (one reload of the resident W).  Placement is a speed matter only; any block -> XCD mapping gives the same results.
//
// Epilogue: software-pipelined.  The accumulators of tile t are kept (ping-pong register sets) while tile t+1 is
// accumulated, and the two 16-row halves of a wave's block are finished under k-steps 1 and 3 of tile t+1: their VALU work
// (bias, GELU, residual) fills issue slots between the MFMAs instead of idling the matrix pipe.  A half goes through a private
// 2-KB LDS region: the residual / GELU' operand rows arrive there by LDS-DMA (so every vector-memory operation of the kernel
// is a DMA or a store, and all waits are counted by hand), the results are written back packed and leave as whole 128-B rows.
#ifdef FC_PROBES   // FC_PROBES: whole file -- an experiment kept in the tools build only (weight-stationary K = 384 GEMM); the product library does not contain it
#include <stdlib.h>
#include <string.h>

#include "fc_kernels.h"
#include "fc_mfma_dev.h"

enum { WS_PLAIN = 0, WS_BIAS, WS_RES, WS_GELU_SG, WS_MUL };

#define WS_KT 6                          // k-tiles of 64: K = 384
#define WS_WBYTES (WS_KT * 16384)        // resident W chunk: 6 k-tiles of [128][64] / [64][128]
#define WS_RING 3
#define WS_ABYTES (WS_RING * 16384)
#define WS_SCRATCH 16384                 // 8 waves x 2 KB epilogue regions
#define WS_LDS (WS_WBYTES + WS_ABYTES + WS_SCRATCH)   // 163840 = all of a CU's LDS
#define WS_PPW 2                         // 1-KB DMA pieces per wave per 16-KB tile (8 waves)

template <int EPI> struct WsEpi {
  static constexpr bool has_bias = EPI == WS_BIAS || EPI == WS_RES || EPI == WS_GELU_SG;
  static constexpr bool has_in = EPI == WS_RES || EPI == WS_MUL;      // one bf16 [M,N] input
  static constexpr int n_in = has_in ? 2 : 0;                         // DMA operations per wave per half
  static constexpr int n_st = EPI == WS_GELU_SG ? 2 : 1;              // 16-B stores per thread per half-part
};

struct WsArgs {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  bf16_t* C; long ldc;
  int M, N, K;
  const float* bias;         // [N]
  const bf16_t* in;          // residual (RES) or multiplier (MUL), [M][ldc]
  bf16_t* out2;              // GELU_SG: gelu'(u) goes here, gelu(u) to C
#ifdef FC_PROBES
  int dbg;                   // tools build only (FC_WS_DBG): 1 = no A DMA, 2 = no MFMA, 4 = no epilogue halves, 8 = no W load
  long long* stamps;         // tools build only (FC_WS_STAMPS): [workgroup][256] s_memtime stamps of wave 0 ([0],[1] = s_memrealtime at start / end)
#endif
};
#ifdef FC_PROBES
#define WS_DBG(c, bit) ((c).dbg & (bit))
#define WS_STAMP(c) do { if ((c).stamps && threadIdx.x == 0 && (c).sidx < 256) (c).stamps[(size_t)blockIdx.x * 256 + (c).sidx++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_DBG(c, bit) false
#define WS_STAMP(c) do {} while (0)
#endif

// ---- LDS-DMA staging for 8 waves: wave w owns the 1-KB pieces 2w, 2w+1 of a 16-KB operand tile (same images as fc_mfma_dev.h)
template <int MODE>
__device__ __forceinline__ void ws_retarget(unsigned (&voff)[WS_PPW], long ld, int row0, int nrows, int wave, int lane, bool valid) {
#pragma unroll
  for (int p = 0; p < WS_PPW; ++p) {
    const int piece = wave * WS_PPW + p;
    if (MODE == KC) {
      int row = piece * 8 + (lane >> 3), pc = lane & 7;
      int c = pc ^ ((row >> 1) & 7);
      int r = row0 + row;
      voff[p] = (valid && r < nrows) ? (unsigned)((r * ld + c * 8) * 2) : FC_OOB;
    } else {
      int k = piece * 4 + (lane >> 4), pc = lane & 15;
      int sw = (((k >> 3) & 1) << 2) | (k & 3);
      int c = (((pc >> 1) ^ sw) << 1) | (pc & 1);
      int col = row0 + c * 8;
      voff[p] = (valid && col < nrows) ? (unsigned)((k * ld + col) * 2) : FC_OOB;
    }
  }
}
__device__ __forceinline__ void ws_stage(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[WS_PPW], char* buf, unsigned soff_bytes, int wave) {
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff_bytes);
#pragma unroll
  for (int p = 0; p < WS_PPW; ++p)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + (wave * WS_PPW + p) * 1024), 16, voff[p], soff, 0, 0);
}

template <int BMODE>
__device__ __forceinline__ void ws_compute(const char* la, const char* lb, f32x4 (&acc)[2][4], int wm, int wn, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[2], bfr[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = frag_read<KC>(la, wm * 32 + i * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BMODE>(lb, wn * 64 + j * 16, ks, lane);
    if (BMODE == KR) frag_fence(bfr);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
}

#define WS_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// LDS accesses of the epilogue regions go through inline asm: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of any DS
// access it cannot prove disjoint from an in-flight LDS-DMA destination, which drained the A-piece pipeline at every epilogue
// part (measured: ~2000 cycles each).  The compiler does not track these, so each read is followed by ws_lds_wait on its result.
__device__ __forceinline__ unsigned ws_lds_addr(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ws_lds_write_b64(const char* p, uint2 v) {
  u32x2 t = {v.x, v.y};
  asm volatile("ds_write_b64 %0, %1" ::"v"(ws_lds_addr(p)), "v"(t) : "memory");
}
__device__ __forceinline__ u32x2 ws_lds_read_b64(const char* p) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ws_lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ u32x4 ws_lds_read_b128(const char* p) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ws_lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ void ws_lds_wait(u32x2& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void ws_lds_wait(u32x4& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); }

struct WsCtx {
  __amdgpu_buffer_rsrc_t ars, irs, crs, ors;
  char* Wres; char* Aring; char* R;
  int wave, lane, wm, wn;
  int M, N, n0;
  long lda, ldc;
#ifdef FC_PROBES
  int dbg;
  long long* stamps;
  mutable int sidx;
#endif
  float bias[16];            // C layout: bias[4*jj + x] for column n0 + 64 wn + 16 jj + 4 g + x
};

// Epilogue region image: [16 rows][128 B], 16-B chunk c of row r stored at chunk c ^ (r & 7).  Without the swizzle the
// accumulator-layout accesses (16 lanes = 16 rows at one column offset, rows 128 B apart) are 16-way bank conflicts: measured
// ~2000 LDS cycles per half.  The DMA that fills the region writes linearly, so its SOURCE column is permuted instead.
__device__ __forceinline__ int ws_r_off(int row, int unit8) { return row * 128 + ((((unit8 >> 1) ^ row) & 7) << 4) + ((unit8 & 1) << 3); }

// request the input rows of half h (16 rows x 64 columns of this wave's block of the tile at m0) into the wave's LDS region
template <int EPI>
__device__ __forceinline__ void ws_half_request(const WsCtx& c, int m0, int h, char* dst) {
  if (!WsEpi<EPI>::has_in) return;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int r = 8 * o + (c.lane >> 3);
    const int col = c.n0 + c.wn * 64 + 8 * ((c.lane ^ r) & 7);
    const int row = m0 + c.wm * 32 + 16 * h + r;
    const unsigned vo = (row < c.M && col < c.N) ? (unsigned)(((long)row * c.ldc + col) * 2) : FC_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.irs, (lds_ptr_t)(dst + o * 1024), 16, vo, 0, 0, 0);
  }
}
// Finish half h (16 rows x 64 columns of this wave's block) of the tile at m0, in two parts that sit under two different
// k-steps so that a CU's stores trickle out at about the rate its store path sustains (~10 B/clk) instead of in one burst:
//   part A: bias / GELU / residual / multiply in the accumulator layout, results packed to bf16 into the wave's LDS region,
//           rows 0..7 read back as whole 128-B segments and stored (GELU_SG: all 16 rows of gelu(u); gelu'(u) stays in `keep`)
//   part B: rows 8..15 stored (GELU_SG: gelu'(u) packed, read back and stored to the second output)
// The caller has waited for the half's input DMA before part A.
template <int EPI>
__device__ __forceinline__ void ws_store_rows(const WsCtx& c, int m0, int h, int o, __amdgpu_buffer_rsrc_t dst) {
  const int lane = c.lane;
  u32x4 rowv = ws_lds_read_b128(c.R + o * 1024 + lane * 16);
  const int r = 8 * o + (lane >> 3);
  const int row = m0 + c.wm * 32 + 16 * h + r, col = c.n0 + c.wn * 64 + 8 * ((lane ^ r) & 7);
  const unsigned vo = (row < c.M && col < c.N) ? (unsigned)(((long)row * c.ldc + col) * 2) : FC_OOB;
  ws_lds_wait(rowv);
  __builtin_amdgcn_raw_buffer_store_b128(*(const v4u*)&rowv, dst, vo, 0, 0);
}
__device__ __forceinline__ void ws_pack_to_region(const WsCtx& c, const float (&s)[4][4]) {
  const int g = c.lane >> 4, cl = c.lane & 15;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) ws_lds_write_b64(c.R + ws_r_off(cl, 4 * jj + g), make_uint2(f2bf2(s[jj][0], s[jj][1]), f2bf2(s[jj][2], s[jj][3])));
}
template <int EPI>
__device__ __forceinline__ void ws_half_part_a(const WsCtx& c, const f32x4 (&acc)[4], int m0, int h, const char* in_region, float (&keep)[4][4]) {
  const int lane = c.lane, g = lane >> 4, cl = lane & 15;
  float v[4][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int x = 0; x < 4; ++x) v[jj][x] = acc[jj][x] + (WsEpi<EPI>::has_bias ? c.bias[4 * jj + x] : 0.f);
  if (WsEpi<EPI>::has_in) {
    u32x2 uin[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) uin[jj] = ws_lds_read_b64(in_region + ws_r_off(cl, 4 * jj + g));
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) ws_lds_wait(uin[jj]);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const u32x2 u = uin[jj];
      const float i0 = __uint_as_float(u.x << 16), i1 = __uint_as_float(u.x & 0xffff0000u);
      const float i2 = __uint_as_float(u.y << 16), i3 = __uint_as_float(u.y & 0xffff0000u);
      if (EPI == WS_MUL) { v[jj][0] *= i0; v[jj][1] *= i1; v[jj][2] *= i2; v[jj][3] *= i3; }
      else { v[jj][0] += i0; v[jj][1] += i1; v[jj][2] += i2; v[jj][3] += i3; }
    }
  }
  if (EPI == WS_GELU_SG) {     // one exp / rcp per element serves gelu(u) (output) and gelu'(u) (saved for the backward)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        float cdf, pdf;
        gelu_fast_parts(v[jj][x], cdf, pdf);
        keep[jj][x] = cdf + v[jj][x] * pdf;
        v[jj][x] *= cdf;
      }
  }
  ws_pack_to_region(c, v);
  ws_store_rows<EPI>(c, m0, h, 0, c.crs);
  if (EPI == WS_GELU_SG) ws_store_rows<EPI>(c, m0, h, 1, c.crs);
}
template <int EPI>
__device__ __forceinline__ void ws_half_part_b(const WsCtx& c, int m0, int h, const float (&keep)[4][4]) {
  if (EPI == WS_GELU_SG) {
    ws_pack_to_region(c, keep);
    ws_store_rows<EPI>(c, m0, h, 0, c.ors);
    ws_store_rows<EPI>(c, m0, h, 1, c.ors);
  } else {
    ws_store_rows<EPI>(c, m0, h, 1, c.crs);
  }
}

// One tile: six k-steps into `cur`, with the previous tile's two halves finished underneath: input rows requested at steps
// 0 / 2, part A at steps 1 / 4, part B at steps 2 / 5.  Per wave and step the vector-memory queue receives, in this order:
// the A piece two steps ahead (2 DMAs), then that step's stores (n_st) and / or input request (n_in).  The counted waits
// below follow from that order.  PREV: there is a previous tile to finish; PREV2: the tile before it had one too.
template <int BMODE, int EPI, bool PREV, bool PREV2>
__device__ __forceinline__ void ws_tile(const WsCtx& c, f32x4 (&cur)[2][4], const f32x4 (&prev)[2][4], int m0_prev,
                                        const unsigned (&voff_cur)[WS_PPW], const unsigned (&voff_next)[WS_PPW]) {
  constexpr int I = PREV ? WsEpi<EPI>::n_in : 0, S = PREV ? WsEpi<EPI>::n_st : 0, S2 = PREV2 ? WsEpi<EPI>::n_st : 0;
  float keep[4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define WS_EPI_WORK(k)                                                                                              \
  do {                                                                                                              \
    if ((k) == 1 || (k) == 4) {                                                                                     \
      if (WsEpi<EPI>::has_in) { if ((k) == 1) WS_WAIT(2); else WS_WAIT(4); }                                         \
      ws_half_part_a<EPI>(c, prev[(k) == 4], m0_prev, (k) == 4, c.R, keep);                                         \
    }                                                                                                               \
    if ((k) == 2 || (k) == 5) ws_half_part_b<EPI>(c, m0_prev, (k) == 5, keep);                                      \
    if ((k) == 2 && WsEpi<EPI>::has_in) {                                                                           \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
      ws_half_request<EPI>(c, m0_prev, 1, c.R);                                                                     \
    }                                                                                                               \
  } while (0)
#define WS_STEP(k, WAITN)                                                                                           \
  do {                                                                                                              \
    WS_STAMP(c);                                                                                                    \
    WS_WAIT(WAITN);                                                                                                 \
    WS_STAMP(c);                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_s_barrier();                                                                                   \
    asm volatile("" ::: "memory");                                                                                  \
    WS_STAMP(c);                                                                                                    \
    if (WS_DBG(c, 1)) {}                                                                                            \
    else if ((k) + 2 < WS_KT) ws_stage(c.ars, voff_cur, c.Aring + (((k) + 2) % WS_RING) * 16384, ((k) + 2) * 128, c.wave); \
    else ws_stage(c.ars, voff_next, c.Aring + (((k) + 2) % WS_RING) * 16384, ((k) + 2 - WS_KT) * 128, c.wave);     \
    if (PREV && (k) == 0 && !WS_DBG(c, 4)) ws_half_request<EPI>(c, m0_prev, 0, c.R);                                \
    if (PREV && (k) != 0 && (k) != 3 && !WS_DBG(c, 4) && c.wave >= 4) WS_EPI_WORK(k);   /* SIMD partners take opposite orders */ \
    if (!WS_DBG(c, 2)) ws_compute<BMODE>(c.Aring + ((k) % WS_RING) * 16384, c.Wres + (k) * 16384, cur, c.wm, c.wn, c.lane); \
    if (PREV && (k) != 0 && (k) != 3 && !WS_DBG(c, 4) && c.wave < 4) WS_EPI_WORK(k);                                \
    WS_STAMP(c);                                                                                                    \
  } while (0)
  WS_STEP(0, 2 + 2 * S2);
  WS_STEP(1, 2 + I + S2);
  WS_STEP(2, 2 + I + S);
  WS_STEP(3, 2 + I + 2 * S);
  WS_STEP(4, 2 + I + S);
  WS_STEP(5, 2 + S);
#undef WS_STEP
#undef WS_EPI_WORK
}

template <int BMODE, int EPI>
__global__ void __launch_bounds__(512, 1) k_gemm_ws(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K;
  // ---- this workgroup's share: XCD label x owns m-tiles [mlo, mhi); its U workgroups cut the chunk-major item list evenly
  const int G = gridDim.x, x = blockIdx.x & 7, u = blockIdx.x >> 3;
  const int U = (G - x + 7) >> 3;
  const int MT = (M + 127) >> 7, NC = (N + 127) >> 7;
  const int mlo = (int)((long)MT * x / 8), mhi = (int)((long)MT * (x + 1) / 8);
  const int mt = mhi - mlo;
  const long I = (long)NC * mt;
  int it = (int)(I * u / U);
  const int it_end = (int)(I * (u + 1) / U);
  if (it >= it_end) return;

  WsCtx c;
  c.Wres = smem; c.Aring = smem + WS_WBYTES; c.R = smem + WS_WBYTES + WS_ABYTES + wave * 2048;
  c.wave = wave; c.lane = lane; c.wm = wave >> 1; c.wn = wave & 1;
  c.M = M; c.N = N; c.lda = a.lda; c.ldc = a.ldc;
#ifdef FC_PROBES
  c.dbg = a.dbg; c.stamps = a.stamps; c.sidx = 2;
  if (c.stamps && threadIdx.x == 0) c.stamps[(size_t)blockIdx.x * 256] = __builtin_amdgcn_s_memrealtime();
  WS_STAMP(c);
#endif
  {
    Operand t = make_operand_glds<KC>(a.A, a.lda, 0, M, K, wave, lane);
    c.ars = t.rsrc;
  }
  Operand ow = make_operand_glds<BMODE>(a.W, a.ldw, 0, N, K, wave, lane);
  c.crs = make_store_rsrc((void*)a.C, (long)M * a.ldc * 2);
  c.ors = make_store_rsrc(EPI == WS_GELU_SG ? (void*)a.out2 : (void*)a.C, (long)M * a.ldc * 2);
  c.irs = make_store_rsrc(WsEpi<EPI>::has_in ? (void*)a.in : (void*)a.C, (long)M * a.ldc * 2);
  const __amdgpu_buffer_rsrc_t brs = make_store_rsrc(WsEpi<EPI>::has_bias ? (void*)a.bias : (void*)a.C, (long)N * 4);

  while (it < it_end) {
    // ---- segment: items [it, seg_end) share chunk j
    const int j = it / mt;
    const int seg_end = min(it_end, (j + 1) * mt);
    c.n0 = j * 128;
    const int m = mlo + (it - j * mt);              // first m-tile of the segment
    const int nitems = seg_end - it;
    // every wave is done with the previous chunk's W, with the ring and with its epilogue region; its DMAs have landed (only
    // the previous segment's last stores may still be in flight)
    WS_WAIT(4 * WsEpi<EPI>::n_st);
    lds_barrier();
    if (WsEpi<EPI>::has_bias) {   // this wave's 64 bias values -> its LDS region (lanes 0..15 carry 16 B each), read back after the W wait
      const unsigned vo = lane < 16 ? (unsigned)((c.n0 + c.wn * 64 + 4 * lane) * 4) : FC_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr_t)c.R, 16, vo, 0, 0, 0);
    }
    {
      unsigned wv[WS_PPW];
      ws_retarget<BMODE>(wv, a.ldw, c.n0, N, wave, lane, true);
      for (int kt = 0; kt < WS_KT && !WS_DBG(c, 8); ++kt) ws_stage(ow.rsrc, wv, c.Wres + kt * 16384, (unsigned)kt * 64u * ow.kstride, wave);
    }
    unsigned va[WS_PPW], vb[WS_PPW];
    ws_retarget<KC>(va, a.lda, m * 128, M, wave, lane, true);
    ws_stage(c.ars, va, c.Aring, 0, wave);
    ws_stage(c.ars, va, c.Aring + 16384, 128, wave);
    if (WsEpi<EPI>::has_bias) {
      WS_WAIT(2 * WS_KT + 4);                      // the bias DMA is the oldest operation: W and the first two pieces may fly on
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        u32x4 b4 = ws_lds_read_b128(c.R + (16 * jj + 4 * (lane >> 4)) * 4);
        ws_lds_wait(b4);
        c.bias[4 * jj + 0] = __uint_as_float(b4.x); c.bias[4 * jj + 1] = __uint_as_float(b4.y);
        c.bias[4 * jj + 2] = __uint_as_float(b4.z); c.bias[4 * jj + 3] = __uint_as_float(b4.w);
      }
    }
    f32x4 accA[2][4], accB[2][4];
    // first tile of the segment: nothing to finish underneath it
    ws_retarget<KC>(vb, a.lda, (m + 1) * 128, M, wave, lane, 1 < nitems);
    ws_tile<BMODE, EPI, false, false>(c, accA, accB, 0, va, vb);
    int ci = 1;
    for (; ci + 1 < nitems; ci += 2) {
      ws_retarget<KC>(va, a.lda, (m + ci + 1) * 128, M, wave, lane, true);
      if (ci == 1) ws_tile<BMODE, EPI, true, false>(c, accB, accA, (m + ci - 1) * 128, vb, va);
      else ws_tile<BMODE, EPI, true, true>(c, accB, accA, (m + ci - 1) * 128, vb, va);
      ws_retarget<KC>(vb, a.lda, (m + ci + 2) * 128, M, wave, lane, ci + 2 < nitems);
      ws_tile<BMODE, EPI, true, true>(c, accA, accB, (m + ci) * 128, va, vb);
    }
    const bool odd_tail = ci < nitems;              // one more tile, accumulated into accB
    if (odd_tail) {
      ws_retarget<KC>(va, a.lda, 0, M, wave, lane, false);
      if (ci == 1) ws_tile<BMODE, EPI, true, false>(c, accB, accA, (m + ci - 1) * 128, vb, va);
      else ws_tile<BMODE, EPI, true, true>(c, accB, accA, (m + ci - 1) * 128, vb, va);
    }
    // ---- the last tile's halves, with nothing left to hide them under.  Their input rows are requested together: half 0 into
    // the wave's region, half 1 into the wave's 2 KB of ring slot 2 (idle once every wave has left the last k-step; the
    // prefetcher's trailing zero-fill pieces go to slots 0 and 1).
    const int m0_last = (m + nitems - 1) * 128;
    char* const R1 = c.Aring + 2 * 16384 + wave * 2048;
    if (WsEpi<EPI>::has_in) {
      lds_barrier();
      ws_half_request<EPI>(c, m0_last, 0, c.R);
      ws_half_request<EPI>(c, m0_last, 1, R1);
      WS_WAIT(0);
    }
    float keep[4][4];
    if (odd_tail) {
      ws_half_part_a<EPI>(c, accB[0], m0_last, 0, c.R, keep); ws_half_part_b<EPI>(c, m0_last, 0, keep);
      ws_half_part_a<EPI>(c, accB[1], m0_last, 1, R1, keep); ws_half_part_b<EPI>(c, m0_last, 1, keep);
    } else {
      ws_half_part_a<EPI>(c, accA[0], m0_last, 0, c.R, keep); ws_half_part_b<EPI>(c, m0_last, 0, keep);
      ws_half_part_a<EPI>(c, accA[1], m0_last, 1, R1, keep); ws_half_part_b<EPI>(c, m0_last, 1, keep);
    }
    it = seg_end;
    WS_STAMP(c);
  }
#ifdef FC_PROBES
  WS_STAMP(c);
  if (c.stamps && threadIdx.x == 0) { c.stamps[(size_t)blockIdx.x * 256 + 1] = __builtin_amdgcn_s_memrealtime(); c.stamps[(size_t)blockIdx.x * 256 + 255] = c.sidx; }
#endif
  WS_WAIT(4 * WsEpi<EPI>::n_st);    // every LDS-DMA of this workgroup has landed before its LDS is released (stores may drain after the end)
}

template <int BMODE, int EPI>
static int ws_launch(const WsArgs& a, hipStream_t s) {
  auto kfn = k_gemm_ws<BMODE, EPI>;
  static bool attr_done = false;   // one flag per instantiation
  if (!attr_done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    attr_done = true;
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  hipLaunchKernelGGL(kfn, dim3(cus), dim3(512), WS_LDS, s, a);
  FC_LAUNCH_CHECK();
  return 0;
}

#ifdef FC_PROBES
static long long* g_ws_stamps = nullptr;
extern "C" int fc_dbg_ws_read_stamps(long long* host, int n) {   // tools build only
  if (!g_ws_stamps) return -1;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_ws_stamps, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost);
}
#endif
static bool ws_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// returns 1 when the shape / epilogue is not covered (the caller then takes the tiled kernel of fc_mfma.hip)
int fc_gemm_ws(int kind, const bf16_t* A, long lda, const bf16_t* Wt, long ldw, bf16_t* C, long ldc, int M, int N, int K, const GemmEpi& e,
               hipStream_t s) {
  if (kind != FC_GEMM_NT && kind != FC_GEMM_NN) return 1;
  if (K != 64 * WS_KT || (N & 7) || (lda & 7) || (ldw & 7) || (ldc & 7) || M < 128 || N < 8) return 1;
  if ((long)M * ldc * 2 >= (1L << 31) || (long)M * lda * 2 >= (1L << 31)) return 1;     // 32-bit buffer offsets
  if (!ws_al16(A) || !ws_al16(Wt) || !ws_al16(C)) return 1;
  if (e.accumulate || e.patch_rows > 0 || e.alpha != 1.0f || e.dbg || e.rowscale) return 1;
  if ((e.bias && !ws_al16(e.bias)) || (e.res && !ws_al16(e.res)) || (e.preact && !ws_al16(e.preact)) || (e.gelu_in && !ws_al16(e.gelu_in))) return 1;
  WsArgs a{A, lda, Wt, ldw, C, ldc, M, N, K, e.bias, nullptr, nullptr};
#ifdef FC_PROBES
  static const int ws_dbg = getenv("FC_WS_DBG") ? atoi(getenv("FC_WS_DBG")) : 0;
  a.dbg = ws_dbg;
  a.stamps = nullptr;
  static const bool want_stamps = getenv("FC_WS_STAMPS") != nullptr;
  if (want_stamps) {
    if (!g_ws_stamps) (void)hipMalloc(&g_ws_stamps, 1024 * 256 * sizeof(long long));
    (void)hipMemsetAsync(g_ws_stamps, 0, 1024 * 256 * sizeof(long long), s);
    a.stamps = g_ws_stamps;
  }
#endif
  int kindE = -1;
  const int extras = (e.res != nullptr) + (e.preact != nullptr) + (e.gelu_in != nullptr);
  if (extras > 1) return 1;
  if (e.preact) {
    if (!e.bias || !e.gelu_saved_grad) return 1;
    kindE = WS_GELU_SG; a.out2 = (bf16_t*)e.preact;
  } else if (e.gelu_in) {
    if (e.bias || !e.gelu_saved_grad) return 1;
    kindE = WS_MUL; a.in = (const bf16_t*)e.gelu_in;
  } else if (e.res) {
    if (!e.bias) return 1;
    kindE = WS_RES; a.in = (const bf16_t*)e.res;
  } else {
    kindE = e.bias ? WS_BIAS : WS_PLAIN;
  }
#define GO(B, E) return ws_launch<B, E>(a, s)
  if (kind == FC_GEMM_NT) {
    switch (kindE) { case WS_BIAS: GO(KC, WS_BIAS); case WS_RES: GO(KC, WS_RES); case WS_GELU_SG: GO(KC, WS_GELU_SG); case WS_PLAIN: GO(KC, WS_PLAIN); }
  } else {
    switch (kindE) { case WS_PLAIN: GO(KR, WS_PLAIN); case WS_MUL: GO(KR, WS_MUL); }
  }
#undef GO
  return 1;
}

#endif  // FC_PROBES
