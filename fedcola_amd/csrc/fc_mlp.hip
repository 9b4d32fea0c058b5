#ifdef FC_PROBES      // tools build only (python -m fedcola_amd.build --probes): exact, measured no faster in the step (profiles/r05/fused_mlp.md)
// Fused MLP for gfx950 (round 5): fc1 -> GELU -> fc2 of a Block (mome.py:117-123) in ONE launch per 64-row panel, and its mirror
// image in the backward (dh = dm . W2, du = dh * gelu'(u), dx = du . W1).  The hidden activation of a panel never leaves the CU between
// the two products: per 128-wide hidden chunk, u = X . W1c^T accumulates in registers, goes through the activation into a 16-KB LDS
// image, and feeds y += h . W2c^T; y (64 x D fp32) stays in registers over all chunks.  What the two separate GEMMs paid for and this
// kernel does not: one dependent launch per layer and chain, the cold ingest of the [rows, 4D] operand by the second product
// (38.7 MB per layer at full batch) and the tile quantisation of a 102-tile launch on 256 CUs.
//
// ACTIVATIONS THROUGH LDS, WEIGHTS THROUGH REGISTERS.  The X panel (64 x D) is resident in LDS, the activation chunk goes through two
// alternating LDS images; the weights never touch LDS: they are read from a copy PACKED IN MFMA-FRAGMENT ORDER (fc_mlp_pack, once per
// optimizer step), 16 bytes per lane, 1 KB per wave-instruction, every byte exactly once per workgroup (a wave owns all 64 rows x 32 of
// the 128 columns of a piece, so no two waves need the same weight fragment), ML_NB pieces ahead into a register ring.  The first version
// of this kernel (profiles/r05/mlp_fused_v1_ring.hip.txt) staged the weight pieces by LDS-DMA through a five-slot ring with two loader
// waves and one barrier per piece: bit-identical results, 75-92 us per launch at every row count -- each 1-KB LDS-DMA instruction cost its
// loader wave ~140 cycles at issue beside the consumers' 48 KB of fragment reads per piece (MI355X_MICROARCH.md, 'LDS-DMA piece issue
// cost'), 16 of them per piece.  Here a piece costs 16 coalesced global loads (4 per wave) and 32 ds_read_b128, no barrier.
//
// 512 threads = 8 waves, two per SIMD, IN TWO ROLES: waves 0-3 run the first product of chunk c and its activation (-> H image c & 1),
// waves 4-7 the second product of chunk c - 1 (<- H image (c - 1) & 1) at the same time; ONE workgroup barrier per chunk hands an H image
// over.  The two waves of a SIMD are therefore always in different phases: the activation's VALU work and the waits of one run under the
// MFMAs of the other.  Every wave owns 32 of the 128 columns of its pieces over all 64 rows and sums its k in ascending order, exactly as
// the separate kernels do: the results are bit-identical to fc1 (+ GELU epilogue) followed by fc2 (tests/test_gpu_kernels.py).
// Earlier forms, same results, kept as records: 4 waves in one role (profiles/r05/mlp_fused_v2_4waves.hip.txt: 54 us warm / 69 cold per
// launch, the phases adding up -- ablation: weight loads 25, MFMAs 27, activation 27 us) and 8 waves with the pair splitting the k-steps
// (mlp_fused_v3_ksplit.hip.txt: the same 55 / 69 us -- two waves in lockstep through the same phases overlap nothing).
// Piece order per chunk: 2 ND pieces of the first product (k-tiles of X), then ND x 2 of the second (128 output columns x 64 hidden each).
//
// Forward  (BWD = 0): first product W1 [Hd, D], second W2 [D, Hd]; act = gelu(u) and gsave = gelu'(u) are stored for the backward.
// Backward (BWD = 1): first product W2^T (k = D, columns = hidden), second W1^T (k = hidden, columns = D): fc_mlp_pack transposes while
//                     it packs; gsave = gelu'(u) is read, act = du is stored (the operand of dW1 / db1).
#include "fc_kernels.h"
#include "fc_mfma_dev.h"

#define ML_ROWS 64
#define ML_HC 128
#define ML_NB 6            // weight pieces in flight per wave of the second role (register ring); divides the 2 ND pieces a role has per chunk
#define ML_NBA 6           // ... of the first role, which also issues the activation stores: its loads queue behind them in vmcnt order
#define ML_PIECE 16384     // packed bytes per piece: 4 waves x (2 k-steps x 2 column blocks) x 1 KB
#define ML_TILE 8192       // one [64 rows][64 k] KC image

// ---------------------------------------------------------------- packing
// stream[chunk c][piece p][wave w][k-step ks][column block j][lane][8]: lane (r = lane & 15, g = lane >> 4) holds
//   p <  2 ND: Wa_eff[n = 128 c + 32 w + 16 j + r][k = 64 p + 32 ks + 8 g .. + 7]                         (first product, k-tile p of D)
//   p >= 2 ND: Wb_eff[n = 128 nb + 32 w + 16 j + r][k = 128 c + 64 kt2 + 32 ks + 8 g .. + 7], (nb, kt2) = ((p - 2 ND) >> 1, (p - 2 ND) & 1)
// forward:  Wa_eff = W1 [Hd][D], Wb_eff = W2 [D][Hd] (rows k-contiguous: 16-byte copies)
// backward: Wa_eff[n][k] = W2[k][n], Wb_eff[n][k] = W1[k][n] (gathers)
typedef FcMlpPackJob MlPackJob;
template <int ND>
__global__ void __launch_bounds__(256) k_mlp_pack(const MlPackJob* __restrict__ jobs, int Hd) {
  constexpr int NT1 = 2 * ND, PPC = 4 * ND, D = 128 * ND;
  const MlPackJob J = jobs[blockIdx.y];
  const int bwd = blockIdx.z;
  const int u = blockIdx.x * 256 + threadIdx.x;           // 16-byte unit of the stream
  const int NCH = Hd / ML_HC;
  if (u >= NCH * PPC * 1024) return;
  const int lane = u & 63, j = (u >> 6) & 1, ks = (u >> 7) & 1, w = (u >> 8) & 3, q = u >> 10;
  const int c = q / PPC, p = q % PPC, r = lane & 15, g = lane >> 4;
  int n, k0;
  const bool first = p < NT1;
  if (first) { n = 128 * c + 32 * w + 16 * j + r; k0 = 64 * p + 32 * ks + 8 * g; }
  else { const int nb = (p - NT1) >> 1, kt2 = (p - NT1) & 1; n = 128 * nb + 32 * w + 16 * j + r; k0 = 128 * c + 64 * kt2 + 32 * ks + 8 * g; }
  uint4 v;
  if (!bwd) {
    v = first ? *(const uint4*)(J.W1 + (size_t)n * D + k0) : *(const uint4*)(J.W2 + (size_t)n * Hd + k0);
  } else {
    bf16_t e[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) e[x] = first ? J.W2[(size_t)(k0 + x) * Hd + n] : J.W1[(size_t)(k0 + x) * D + n];
    v = *(const uint4*)e;
  }
  *(uint4*)((bwd ? J.bwd : J.fwd) + (size_t)u * 8) = v;
}
size_t fc_mlp_pack_elems(int D, int Hd) { return 2 * (size_t)D * Hd; }      // elements of ONE direction's stream (both weight matrices)
int fc_mlp_fused_ok(int D, int Hd) { return D == 384 && Hd > 0 && (Hd % ML_HC) == 0; }
// jobs_dev: device array of njobs {W1, W2, fwd stream, bwd stream}; packs both directions of every job in one launch
int fc_mlp_pack(const void* jobs_dev, int njobs, int D, int Hd, hipStream_t s) {
  if (njobs <= 0) return 0;
  FC_REQUIRE(fc_mlp_fused_ok(D, Hd), "fc_mlp_pack: D = %d, Hd = %d not covered", D, Hd);
  const int units = (Hd / ML_HC) * 12 * 1024;
  hipLaunchKernelGGL(k_mlp_pack<3>, dim3(fc_cdiv(units, 256), njobs, 2), dim3(256), 0, s, (const MlPackJob*)jobs_dev, Hd);
  FC_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- the kernel
__device__ __forceinline__ unsigned ml_lane_off(int lane, int ks) {   // kc_off(16 i + r, 4 ks + g) - 2048 i, r = lane & 15, g = lane >> 4
  const int r = lane & 15, g = lane >> 4;
  return (unsigned)(r * 128 + ((((ks << 2) | g) ^ (r >> 1)) << 4));
}
struct MlA { bf16x8 f[4]; };         // the LDS operand's fragments of ONE 32-k step of a piece: [16-row block]
struct MlB { bf16x8 f[2][2]; };      // the weight fragments of one piece: [k-step][16-column block]
__device__ __forceinline__ void ml_read_a(MlA& A, const char* t) {   // t: image base + this lane's offset for the k-step
#pragma unroll
  for (int i = 0; i < 4; ++i) A.f[i] = *(const bf16x8*)(t + i * 2048);
}
// acc[i][j][x] = C[16 i + (lane & 15)][16 j + 4 (lane >> 4) + x] of the wave's 64 x 32 block (swapped operands, as in fc_mfma.hip).
// A fragment register is re-read for the NEXT piece (`next`: that piece's image + this lane's offset for the k-step; null: no re-read) as
// soon as its two MFMAs have issued, so every LDS read has 14 MFMAs (224 cycles) to land: with whole k-step sets re-read behind their eight
// MFMAs a wave alone on the matrix pipe took 520 cycles per 16-MFMA piece (profiles/r05/mlp_fused_v4d_stamps.txt).
template <int KS, bool REREAD>
__device__ __forceinline__ void ml_mfma(MlA& A, const MlB& Bf, f32x4 (&acc)[4][2], const char* next) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf.f[KS][j], A.f[i], acc[i][j], 0, 0, 0);
    if (REREAD) A.f[i] = *(const bf16x8*)(next + i * 2048);
  }
}

// gelu(v) and gelu'(v) of N elements, stage by stage over all of them (the same arithmetic as gelu_fast_parts, element by element: the
// Horner chain of one element is nine dependent instructions; written per element hipcc scheduled the chains one after the other with an
// s_nop between the packed operations, 137 cycles per element -- profiles/r05/mlp_fused_v4c_stamps.txt)
template <int N>
__device__ __forceinline__ void ml_gelu(const float (&v)[N], float (&h)[N], float (&gp)[N]) {
  float t[N], ex[N], poly[N];
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = __builtin_amdgcn_rcpf(fmaf(fabsf(v[k]), FC_GELU_P, 1.0f));
#pragma unroll
  for (int k = 0; k < N; ++k) { const float w = v[k] * FC_GELU_K; ex[k] = __builtin_amdgcn_exp2f(-w * w); }
#pragma unroll
  for (int k = 0; k < N; ++k) poly[k] = -1.453152027f + t[k] * 1.061405429f;
#pragma unroll
  for (int k = 0; k < N; ++k) poly[k] = 1.421413741f + t[k] * poly[k];
#pragma unroll
  for (int k = 0; k < N; ++k) poly[k] = -0.284496736f + t[k] * poly[k];
#pragma unroll
  for (int k = 0; k < N; ++k) poly[k] = 0.254829592f + t[k] * poly[k];
#pragma unroll
  for (int k = 0; k < N; ++k) poly[k] = t[k] * poly[k];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float erfz = 1.0f - poly[k] * ex[k];
    const float cdf = 0.5f + copysignf(0.5f * erfz, v[k]);
    const float pdf = ex[k] * 0.39894228040143267794f;
    gp[k] = cdf + v[k] * pdf;
    h[k] = v[k] * cdf;
  }
}

#ifndef ML_PHASES
#define ML_PHASES 1
#endif
__device__ __forceinline__ void ml_barrier_x() {      // phase boundary: no LDS hand-off rides on it
#if ML_PHASES
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}

struct MlpArgs {
  const bf16_t* X;          // [M, D]: forward LN2 output; backward dm (= dx_{l+1} [* drop-path scale])
  const bf16_t* Wp;         // the packed weight stream of this direction (fc_mlp_pack)
  const float* b1;          // forward: [Hd]
  const float* b2;          // forward: [D]
  bf16_t* act;              // [M, Hd]: forward gelu(u) (out); backward du (out)
  bf16_t* gsave;            // [M, Hd]: forward gelu'(u) (out); backward gelu'(u) (in)
  const bf16_t* res;        // forward: residual [M, D]
  const float* rowscale;    // forward: drop-path scale per sample (may be null)
  bf16_t* out;              // [M, D]
  int M, Hd, rps;
  int dbg;                  // tools build only (FC_MLP_DBG): 1 no activation arithmetic, 4 no weight loads
  long long* stamps;        // tools build only
};
#ifdef FC_PROBES
#define ML_DBG(bit) (a.dbg & (bit))
// in-kernel stamps (tools/mlp_stamps.py): shader-clock time of workgroup 0's waves at the phase boundaries, [wave][64]
#define ML_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[wave * 64 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ML_DBG(bit) 0
#define ML_STAMP(k) do {} while (0)
#endif

template <bool BWD, int ND, int WAUX>
__global__ void __launch_bounds__(512) k_mlp_fused(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT1 = 2 * ND, PPC = 4 * ND, D = 128 * ND;
  constexpr int XB = NT1 * ML_TILE;                       // the two H buffers (2 images each) follow the X panel
  constexpr int GB = XB + 4 * ML_TILE;                    // forward: then two buffers of the same shape for gelu'(u) on its way to global memory
  static_assert(NT1 % ML_NB == 0 && NT1 % ML_NBA == 0, "the register ring is indexed by the piece number inside a role's half chunk");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave & 3, role = wave >> 2;               // column block of 32 | 0: first product + activation, 1: second product + output
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * ML_ROWS;
  const int M = a.M, Hd = a.Hd;
  const int NCH = Hd / ML_HC, Q = NCH * PPC;
  // ---- the X panel: NT1 images of [64 rows][64 k] by LDS-DMA, 8 one-KB pieces per image, 6 pieces per wave; rows past M read as zeros
  {
    const __amdgpu_buffer_rsrc_t rsX = make_store_rsrc((void*)a.X, (long)M * D * 2);
    const int rl = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int s6 = wave * NT1 + t, kt = s6 >> 3, sp = s6 & 7, row = 8 * sp + rl, c = pc ^ (((sp & 1) << 2) | (rl >> 1));
      const unsigned vo = (m0 + row < M) ? (unsigned)(((m0 + row) * D + kt * 64 + c * 8) * 2) : FC_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(smem + kt * ML_TILE + sp * 1024), 16, vo, 0, 0, 0);
    }
  }
  // ---- the weight stream: piece q, column block w = 4 KB at q * 16 KB + w * 4 KB, fragment (ks, j) at + (2 ks + j) KB, lane at + 16 lane.
  // Role 0 reads the pieces 0 .. NT1-1 of every chunk, role 1 the pieces NT1 .. PPC-1.
  const __amdgpu_buffer_rsrc_t rsW = make_store_rsrc((void*)a.Wp, (long)Q * ML_PIECE);       // pieces past the end read as zeros
  const unsigned wv = (unsigned)(w * 4096 + lane * 16);
  MlB Bq[ML_NBA];
  auto load_b = [&](MlB& dst, int q) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(q * ML_PIECE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv + (unsigned)((2 * ks + j) * 1024), so, WAUX);
        dst.f[ks][j] = *(bf16x8*)&v;
      }
  };
  // (one if / else over the role from here to the output phase: with the ring filled in a separate if the allocator kept both roles' registers
  // alive in both branches and spilled 114 of them)
  const char* xa0 = smem + ml_lane_off(lane, 0);
  const char* xa1 = smem + ml_lane_off(lane, 1);
  typedef __attribute__((vector_size(8))) unsigned int v2u;
  MlA A0, A1;      // the fragments of k-step 0 / k-step 1 of the current piece: each is re-read for the next piece as soon as its MFMAs have issued
  // a role's piece P (0 .. NT1-1 inside its half of the chunk): ring slot P % ML_NB, refilled with the role's piece P + ML_NB
#define ML_REFILL(P, NBX, ROLE) load_b(Bq[(P) % (NBX)], ((P) + (NBX) < NT1 ? c * PPC : (c + 1) * PPC - NT1) + (ROLE) * NT1 + (P) + (NBX))
  if (role == 0) {
    // ================================================== first product + activation
#pragma unroll
    for (int k = 0; k < ML_NBA; ++k) load_b(Bq[k], k);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * ML_NBA) : "memory");        // the X panel has landed (the weight loads are younger)
    ML_STAMP(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ML_STAMP(1);
    f32x4 uacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) uacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsAct = make_store_rsrc((void*)a.act, (long)M * Hd * 2);
    const __amdgpu_buffer_rsrc_t rsG = make_store_rsrc((void*)a.gsave, (long)M * Hd * 2);
    // byte offsets of this lane's (i, j) 4-column group inside a chunk: row 16 i + r, column 32 w + 16 j + 4 g
    unsigned rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowoff[i] = (m0 + 16 * i + r < M) ? (unsigned)(((m0 + 16 * i + r) * Hd + 32 * w + 4 * g) * 2) : FC_OOB;
    // H image: this wave's columns lie in k-tile w >> 1 at k = 32 (w & 1) + 16 j + 4 g
    unsigned hoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) hoff[j] = (unsigned)(XB + (w >> 1) * ML_TILE + r * 128 + (((4 * (w & 1) + 2 * j + (g >> 1)) ^ (r >> 1)) << 4) + (g & 1) * 8);
    ml_read_a(A0, xa0);                                                     // piece 0: X image 0
    ml_read_a(A1, xa1);
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const int hsel = (c & 1) * 2 * ML_TILE;
      ML_STAMP(2 + 4 * c);
      float4 bias[2];
      uint2 gp[4][2];
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bias[j] = *(const float4*)(a.b1 + c * ML_HC + 32 * w + 16 * j + 4 * g);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsG, rowoff[i], (unsigned)((c * ML_HC + 16 * j) * 2), 0);
            gp[i][j] = *(uint2*)&v;
          }
      }
#define ML_STEP_A(P)                                                                                           \
  {                                                                                                            \
    constexpr int PN = ((P) + 1) % NT1;                                                                        \
    ml_mfma<0, true>(A0, Bq[(P) % ML_NBA], uacc, xa0 + PN * ML_TILE);                                          \
    ml_mfma<1, true>(A1, Bq[(P) % ML_NBA], uacc, xa1 + PN * ML_TILE);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (!ML_DBG(4)) ML_REFILL(P, ML_NBA, 0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
      static_assert(ND == 3, "the piece sequence below is written out for D = 384");
      ML_STEP_A(0) ML_STEP_A(1) ML_STEP_A(2) ML_STEP_A(3) ML_STEP_A(4) ML_STEP_A(5)
#undef ML_STEP_A
      ML_STAMP(3 + 4 * c);
      ml_barrier_x();                     // phase boundary: the matrix pipe goes to the other role (second product of chunk c - 1) ...
      // ---- ... while this one runs the activation on the VALU: registers -> H image c & 1 (and gelu' -> G image) in LDS; the other role
      // stores the images to global memory during the next chunk's first phase
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float hv[8], gv[8];
        if (!BWD) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float bj[4] = {bias[j].x, bias[j].y, bias[j].z, bias[j].w};
#pragma unroll
            for (int x = 0; x < 4; ++x) v[4 * j + x] = uacc[i][j][x] + bj[x];
          }
          if (!ML_DBG(1)) ml_gelu<8>(v, hv, gv);
          else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { hv[k] = v[k]; gv[k] = v[k]; }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const unsigned ga = gp[i][j].x, gb = gp[i][j].y;
            hv[4 * j + 0] = uacc[i][j][0] * __uint_as_float(ga << 16); hv[4 * j + 1] = uacc[i][j][1] * __uint_as_float(ga & 0xffff0000u);
            hv[4 * j + 2] = uacc[i][j][2] * __uint_as_float(gb << 16); hv[4 * j + 3] = uacc[i][j][3] * __uint_as_float(gb & 0xffff0000u);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!BWD) {
            uint2 gg = make_uint2(f2bf2(gv[4 * j], gv[4 * j + 1]), f2bf2(gv[4 * j + 2], gv[4 * j + 3]));
            *(uint2*)(smem + (GB - XB) + hoff[j] + hsel + i * 2048) = gg;
          }
          uint2 hh = make_uint2(f2bf2(hv[4 * j], hv[4 * j + 1]), f2bf2(hv[4 * j + 2], hv[4 * j + 3]));
          *(uint2*)(smem + hoff[j] + hsel + i * 2048) = hh;
          uacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      ML_STAMP(4 + 4 * c);
      lds_barrier();                      // H image c & 1 is complete; the other role has finished reading image (c + 1) & 1 (chunk c - 1)
      ML_STAMP(5 + 4 * c);
    }
    ml_barrier_x();                       // the other role's last chunk: its stores, then its products
    lds_barrier();
  } else {
    // ================================================== second product
#pragma unroll
    for (int k = 0; k < ML_NB; ++k) load_b(Bq[k], NT1 + k);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * ML_NB) : "memory");        // the X panel has landed (the weight loads are younger)
    ML_STAMP(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ML_STAMP(1);
    f32x4 yacc[ND][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nb = 0; nb < ND; ++nb) yacc[nb][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsAct = make_store_rsrc((void*)a.act, (long)M * Hd * 2);
    const __amdgpu_buffer_rsrc_t rsG = make_store_rsrc((void*)a.gsave, (long)M * Hd * 2);
    // The activation leaves for global memory from THIS role (it has the slack: no activation arithmetic), row-wise out of the LDS images,
    // behind the chunk's products: 16 lanes store one 256-byte row segment with 16-byte accesses.  (Stored straight from the accumulator layout -- 8 bytes per lane, 32-byte runs -- the sixteen dwordx2 stores per
    // wave and chunk took 1.6 us per chunk at issue, 19 us of a 51-us launch: profiles/r05/mlp_fused_v4a_ablate.txt.)
    auto store_images = [&](int cc) {
      const int hs = (cc & 1) * 2 * ML_TILE;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int item = it * 256 + (tid - 256), c8 = item & 7, t = (item >> 3) & 1, rho = item >> 4;
        const int lo = hs + t * ML_TILE + rho * 128 + ((c8 ^ ((rho >> 1) & 7)) << 4);
        const unsigned go = (m0 + rho < M) ? (unsigned)(((m0 + rho) * Hd + cc * ML_HC + t * 64 + c8 * 8) * 2) : FC_OOB;
        const v4u hv = *(const v4u*)(smem + XB + lo);
        __builtin_amdgcn_raw_buffer_store_b128(hv, rsAct, go, 0, 0);
        if (!BWD) {
          const v4u gv = *(const v4u*)(smem + GB + lo);
          __builtin_amdgcn_raw_buffer_store_b128(gv, rsG, go, 0, 0);
        }
      }
    };
    ml_barrier_x();
    lds_barrier();                        // H image 0 is complete
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const int hsel = XB + (c & 1) * 2 * ML_TILE;
      ML_STAMP(2 + 4 * c);
      const char* ha0 = xa0 + hsel;
      const char* ha1 = xa1 + hsel;
      store_images(c);                    // phase 1 (the other role runs the first product of chunk c + 1 on the matrix pipe): chunk c's images -> global
      ml_read_a(A0, ha0);                                                   // piece 0 of the half chunk: H image 0
      ml_read_a(A1, ha1);
      ML_STAMP(4 + 4 * c);
      ml_barrier_x();                     // phase 2: the matrix pipe is this role's
#define ML_STEP_B(P)                                                                                           \
  {                                                                                                            \
    ml_mfma<0, ((P) + 1 < NT1)>(A0, Bq[(P) % ML_NB], yacc[(P) >> 1], ha0 + (((P) + 1) & 1) * ML_TILE);         \
    ml_mfma<1, ((P) + 1 < NT1)>(A1, Bq[(P) % ML_NB], yacc[(P) >> 1], ha1 + (((P) + 1) & 1) * ML_TILE);         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (!ML_DBG(4)) ML_REFILL(P, ML_NB, 1);                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
      ML_STEP_B(0) ML_STEP_B(1) ML_STEP_B(2) ML_STEP_B(3) ML_STEP_B(4) ML_STEP_B(5)
#undef ML_STEP_B
      ML_STAMP(3 + 4 * c);
      lds_barrier();                      // done with H image c & 1; image (c + 1) & 1 is complete
      ML_STAMP(5 + 4 * c);
    }
    // every wave of the workgroup is past its last LDS read: y -> fp32 image [64][D + 4] over the LDS
    {
      float* Cw = (float*)smem;
#pragma unroll
      for (int nb = 0; nb < ND; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            *(float4*)(Cw + (16 * i + r) * (D + 4) + nb * 128 + 32 * w + 16 * j + 4 * g) = make_float4(yacc[nb][i][j][0], yacc[nb][i][j][1], yacc[nb][i][j][2], yacc[nb][i][j][3]);
    }
  }
#undef ML_REFILL
  // ---- output: the fp32 image [64][D + 4] over the LDS (written by the second role above) -> rows of 2 D bytes, by all 512 threads
  constexpr int CLD = D + 4;
  const float* Cs = (const float*)smem;
  ML_STAMP(60);
  lds_barrier();
  constexpr int CG = D / 8;                                                  // 8-column groups per row
#pragma unroll 2
  for (int it = 0; it < ML_ROWS * CG / 512; ++it) {
    const int item = tid + 512 * it, row = item / CG, cg = item % CG;
    const int m = m0 + row;
    if (m < M) {
      float v[8];
      const float4 x0 = *(const float4*)(Cs + row * CLD + cg * 8), x1 = *(const float4*)(Cs + row * CLD + cg * 8 + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      const size_t o = (size_t)m * D + cg * 8;
      if (!BWD) {
        float bb[8], rr[8];
        Vec8<float>::ld(a.b2 + cg * 8, bb);
        Vec8<bf16_t>::ld(a.res + o, rr);
        const float sc = a.rowscale ? a.rowscale[m / a.rps] : 1.0f;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          v[x] += bb[x];
          if (a.rowscale) v[x] *= sc;
          v[x] += rr[x];
        }
      }
      Vec8<bf16_t>::st(a.out + o, v);
    }
  }
  ML_STAMP(61);
}
#ifdef FC_PROBES
static long long* g_ml_stamps = nullptr;
extern "C" int fc_dbg_mlp_stamps(long long* out512) {      // tools build: the stamps of the last launch (call after a synchronise)
  if (!g_ml_stamps) return -1;
  return hipMemcpy(out512, g_ml_stamps, 512 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

static bool ml_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// Wp: this direction's packed stream (fc_mlp_pack).  Returns 1 when the shape is not covered (the caller runs the two GEMMs).
int fc_mlp_fused(int bwd, const void* X, const void* Wp, const float* b1, const float* b2, void* act, void* gsave, const void* res, const float* rowscale,
                 int rps, void* out, int M, int D, int Hd, hipStream_t s) {
  if (FC_ABLATED("gemm")) return 0;
  if (!fc_mlp_fused_ok(D, Hd) || M <= 0) return 1;
  if ((long)M * Hd * 2 >= 0x7fffffffL) return 1;                              // 32-bit byte offsets inside the buffer descriptors
  if (!ml_aligned16(X) || !ml_aligned16(Wp) || !ml_aligned16(act) || !ml_aligned16(gsave) || !ml_aligned16(out)) return 1;
  if (!bwd && (!ml_aligned16(b1) || !ml_aligned16(b2) || !ml_aligned16(res) || !b1 || !b2 || !res)) return 1;
  constexpr int ND = 3;
  const int lds = 2 * ND * ML_TILE + 8 * ML_TILE;                             // X panel + two H and two G buffers = 112 KB (covers the 99-KB fp32 output image)
  static_assert(2 * ND * ML_TILE + 8 * ML_TILE >= ML_ROWS * (128 * ND + 4) * 4, "LDS");
  MlpArgs a{(const bf16_t*)X, (const bf16_t*)Wp, b1, b2, (bf16_t*)act, (bf16_t*)gsave, (const bf16_t*)res, rowscale, (bf16_t*)out, M, Hd, rps > 0 ? rps : 1,
            fc_knob("FC_MLP_DBG", 0), nullptr};
#ifdef FC_PROBES
  if (fc_knob("FC_MLP_STAMPS", 0)) {
    if (!g_ml_stamps) { FC_CHECK_HIP(hipMalloc(&g_ml_stamps, 512 * sizeof(long long))); }
    FC_CHECK_HIP(hipMemsetAsync(g_ml_stamps, 0, 512 * sizeof(long long), s));
    a.stamps = g_ml_stamps;
  }
#endif
  const int grid = fc_cdiv(M, ML_ROWS);
#define ML_LAUNCH(AUX)                                                                                                             \
  do {                                                                                                                             \
    static bool done = false;                                                                                                      \
    if (!done) {                                                                                                                   \
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_mlp_fused<false, ND, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_mlp_fused<true, ND, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));  \
      done = true;                                                                                                                 \
    }                                                                                                                              \
    if (bwd) hipLaunchKernelGGL((k_mlp_fused<true, ND, AUX>), dim3(grid), dim3(512), lds, s, a);                                    \
    else hipLaunchKernelGGL((k_mlp_fused<false, ND, AUX>), dim3(grid), dim3(512), lds, s, a);                                       \
  } while (0)
#ifdef FC_PROBES
  static const int waux = fc_knob("FC_MLP_AUX", 0);      // cache policy of the weight loads: 0 default, 2 nt, 16 sc1, 17 sc0 sc1
  if (waux == 2) ML_LAUNCH(2); else if (waux == 16) ML_LAUNCH(16); else if (waux == 17) ML_LAUNCH(17); else ML_LAUNCH(0);
#else
  ML_LAUNCH(0);
#endif
#undef ML_LAUNCH
  FC_LAUNCH_CHECK();
  return 0;
}

#endif  // FC_PROBES
