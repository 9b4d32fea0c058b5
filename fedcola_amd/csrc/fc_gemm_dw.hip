// Wide-tile grouped weight-gradient GEMM for gfx950:  dW[out,in] (fp32) = dY[rows,out]^T . X[rows,in],  db = column sums of dY.
//
// Why a second tile shape.  The 128x128 kernel (fc_mfma.hip) reads every dY panel in/128 times and every X panel out/128 times; the
// step's counters (profiles/r02/step_pmc_FETCH_SIZE.txt) show the weight gradients fetching 4.9 GB per step through the fabric for
// 2.2 GB of operands -- the largest single consumer of the backward's traffic, which runs at ~4 TB/s of L2 misses.  Every linear
// of the model has in = 384 or a multiple of it, so a workgroup here owns 128 output rows x 384 input columns: with in = 384 the dY
// panel of a problem is read exactly once and X out/128 times (X is the small operand for qkv / fc1); 96 FLOP per staged byte
// instead of 64.  Measured in the client step (same box): weight-gradient fabric fetch 4.9 -> 2.4 GB per step, step 5.15 -> 4.98 ms.
//
// Structure: 512 threads = 8 waves (2 x 4), wave tile 64 x 96 = 4 x 6 v_mfma_f32_16x16x32_bf16 accumulators (swapped operands: a
// lane owns 4 consecutive output columns).  k-tiles of 64 rows: dY[64 x 128] + X[64 x 384] = 64 KB, staged through registers with two
// k-tiles in flight (as in the 128x128 kernel) into two LDS buffers of four [64 k][128 col] images (kr_off swizzle, fragments by
// ds_read_b64_tr_b16): 128 KB of LDS, one workgroup per CU, 2 waves per SIMD.  One workgroup walks the whole reduction: no split-K,
// no atomics, bitwise reproducible.  The optional AdamW epilogue is the one of fc_mfma.hip's grouped kernel.
#include "fc_kernels.h"
#include "fc_mfma_dev.h"

#define DW_BN 384
#define DW_LD 388                // padded fp32 row of the 64-row epilogue image
#define DW_STAGE 65536           // bytes per k-tile: 4 images of 16 KB (dY | X cols 0..127 | 128..255 | 256..383)

struct DwRegs { uint4 v[8]; };   // [2 p] of dY, [3 j][2 p] of X

__device__ __forceinline__ void frag_fence6(bf16x8 (&f)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5])::"memory");
}

struct DwOperand {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned kstride;              // bytes per k row
};
__device__ __forceinline__ DwOperand dw_operand(const bf16_t* P, long ld, int ncols, int K) {
  DwOperand o;
  unsigned long long base = (unsigned long long)P;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const void* up = (const void*)(((unsigned long long)hi << 32) | lo);
  unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)K - 1) * ld + ncols) * 2));
  o.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)up, 0, (int)bytes, 0x00020000);
  o.kstride = (unsigned)(ld * 2);
  return o;
}

// thread t stages chunk c = t & 15 (8 columns) of k rows (t >> 4) + 32 p of every image.  X has in % 384 == 0: its three images need no
// column check, their offsets ride on the scalar offset on top of the two per-lane offsets of image 0.
__device__ __forceinline__ void dw_load(DwRegs& R, const DwOperand& oa, const DwOperand& ob, const unsigned (&va)[2], const unsigned (&vb)[2], int k0) {
  const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)k0 * oa.kstride));
  const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)k0 * ob.kstride));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(oa.rsrc, va[p], sa, 0);
    R.v[p] = *(uint4*)&v;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      v4u v = __builtin_amdgcn_raw_buffer_load_b128(ob.rsrc, vb[p], sb + 256u * j, 0);   // image offset on the scalar side: no VGPR
      R.v[2 + 2 * j + p] = *(uint4*)&v;
    }
}
// sbase = kr_off(t >> 4, t & 15): the k + 32 row of the same chunk lies 8 KB further (the swizzle only looks at k & 3 and bit 3 of k)
// HALF 0: dY image + first X image, HALF 1: the other two X images -- one half goes out behind each k-step's MFMAs
template <int HALF>
__device__ __forceinline__ void dw_store(const DwRegs& R, char* buf, int sbase, float (&cs)[8], bool do_colsum) {
  char* sb = buf + sbase;
#pragma unroll
  for (int img = 2 * HALF; img < 2 * HALF + 2; ++img)
#pragma unroll
    for (int p = 0; p < 2; ++p) *(uint4*)(sb + 16384 * img + 8192 * p) = R.v[2 * img + p];
  if (HALF == 0 && do_colsum) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const bf16_t* hh = (const bf16_t*)&R.v[p];
#pragma unroll
      for (int i = 0; i < 8; ++i) cs[i] += bf2f(hh[i]);
    }
  }
}
// the four transposing reads of a fragment pair share one address: k-step 1 lies 8 KB, the second half of the 8-k group 1 KB further.
// The address is lane base ^ scalar key, formed inside the asm: written in C the compiler hoists all twenty (loop-invariant) addresses
// into VGPRs and spills.
template <int KS>
__device__ __forceinline__ bf16x8 dw_frag(unsigned lb, unsigned key) {
  s16x4 lo, hi;
  unsigned addr;
  asm volatile("v_xor_b32 %2, %3, %4\n\tds_read_b64_tr_b16 %0, %2 offset:%5\n\tds_read_b64_tr_b16 %1, %2 offset:%6"
               : "=&v"(lo), "=&v"(hi), "=&v"(addr)
               : "v"(lb), "s"(key), "n"(KS * 8192), "n"(KS * 8192 + 1024)
               : "memory");
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
__device__ __forceinline__ void frag_fence3(bf16x8 (&f)[3]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2])::"memory");
}
// X fragments in two groups of three, so that 28 fragment registers are live instead of 40 (the kernel sits at the 256-VGPR limit of
// 2 waves per SIMD); the second wave of the SIMD covers the second group's LDS latency
// Fragment addresses.  In the kr_off image the 32-B unit u of k row k sits at unit u ^ s(k), so the address of a lane's fragment is
//   [k*256 + (s << 5) + (chunk parity << 4) + byte]  ^  (u << 5)   +  image offset:
// ONE per-lane base (the bracket) and a wave-uniform key per fragment, (u << 5) | image offset | buffer offset, held in SGPRs --
// the bit fields do not overlap, so the sum is an XOR.  (Ten per-lane address registers would not fit next to 96 accumulators and
// two staging sets at 2 waves per SIMD.)
struct DwKeys { unsigned a[4], b[6]; };
template <int KS>
__device__ __forceinline__ void dw_kstep(unsigned lb, unsigned po, const DwKeys& kx, f32x4 (&acc)[4][6]) {
  bf16x8 af[4], b0[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = dw_frag<KS>(lb, kx.a[i] | po);
#pragma unroll
  for (int j = 0; j < 3; ++j) b0[j] = dw_frag<KS>(lb, kx.b[j] | po);
  frag_fence(af);
  frag_fence3(b0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], af[i], acc[i][j], 0, 0, 0);
  bf16x8 b1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1[j] = dw_frag<KS>(lb, kx.b[3 + j] | po);
  frag_fence3(b1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], af[i], acc[i][3 + j], 0, 0, 0);
}
// one k-tile: both k-steps of the tile in buffer `po`, with the register set `Rn` (a later k-tile) going to `nbuf` in two halves
// behind the MFMAs of each k-step.
// Measured (tools/kernel_bench.py under rocprofv3, one tile per CU): 300 us per tile = 1.52 us per k-tile, against 1 536 cycles of
// MFMA (2 waves x 48 per SIMD) + ~830 of LDS-write transfer + ~640 of fragment reads: the three run back to back, not overlapped.
// Tried: holding the SIMD-partner waves (w + 4) back with s_sleep after each barrier (+1.4 us per 64 cycles of sleep, no gain);
// giving them the stores first (31-357 spilled registers: hipcc merges the 96 accumulators through phi nodes at the joins); a
// 4-wave form with 64x192 wave tiles, one wave per SIMD, 192 accumulators in AGPRs and a whole k-step of fragments prefetched under
// the previous k-step's MFMAs (no spills, 372 us per tile, step 5.30 instead of 5.04 ms).
__device__ __forceinline__ void dw_tile(unsigned lb, unsigned po, const DwKeys& kx, f32x4 (&acc)[4][6], const DwRegs& Rn, char* nbuf, int sbase,
                                        float (&cs)[8], bool do_colsum) {
  dw_kstep<0>(lb, po, kx, acc);
  dw_store<0>(Rn, nbuf, sbase, cs, do_colsum);
  dw_kstep<1>(lb, po, kx, acc);
  dw_store<1>(Rn, nbuf, sbase, cs, do_colsum);
}

template <bool OPT>
__global__ void __launch_bounds__(512, 2) k_gemm_dw_wide(const FcTnProblem* __restrict__ probs, int nprob, FcAdamW o_) {
  const FcAdamW o = fc_adamw_resolve(o_);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int idx = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  while (pi + 1 < nprob && probs[pi + 1].tile_start <= idx) ++pi;
  const FcTnProblem P = probs[pi];
  const int local = idx - P.tile_start;
  const int tile_m = local / P.tiles_n, tile_n = local % P.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * DW_BN;
  const int M = P.M, N = P.N, K = P.K;
  const bool do_colsum = (tile_n == 0) && (P.bias_grad != nullptr);
  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) cs[i] = 0.f;

  const DwOperand oa = dw_operand(P.A, P.lda, M, K), ob = dw_operand(P.B, P.ldb, N, K);
  unsigned va[2], vb[2];
  {
    const int c8 = (tid & 15) * 8, k = tid >> 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      va[p] = (m0 + c8 < M) ? (unsigned)((((long)(k + 32 * p)) * P.lda + m0 + c8) * 2) : FC_OOB;
      vb[p] = (unsigned)((((long)(k + 32 * p)) * P.ldb + n0 + c8) * 2);
    }
  }
  // fragment addressing (see DwKeys): per-lane base + per-fragment scalar keys
  unsigned lb;
  DwKeys kx;
  {
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;   // 0: the kernel has no static LDS
    const int gq = lane >> 4, ii = lane & 15, q = ii >> 2, pp = ii & 3;
    const int sw = ((gq & 1) << 2) | q;                       // s(k) of k = 8 gq + q (+ 4, + 32: same s)
    lb = lds0 + (unsigned)((8 * gq + q) * 256 + (sw << 5) + ((pp >> 1) << 4) + (pp & 1) * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) kx.a[i] = (unsigned)__builtin_amdgcn_readfirstlane((wm * 4 + i) << 5);          // dY image: offset 0
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int u = wn * 6 + j;                               // 16-column block of the 384-wide X tile
      kx.b[j] = (unsigned)__builtin_amdgcn_readfirstlane(((u & 7) << 5) | (16384 * (1 + (u >> 3))));
    }
  }
  const int sbase = kr_off(tid >> 4, tid & 15);
  // rows past K fall outside the descriptors (zeros), so the loop is branch-free and odd tile counts end on an all-zero tile
  const int T = (K + BK - 1) / BK;
  char* buf0 = smem;
  char* buf1 = smem + DW_STAGE;
  DwRegs R0, R1;
  dw_load(R0, oa, ob, va, vb, 0);
  dw_load(R1, oa, ob, va, vb, BK);
  dw_store<0>(R0, buf0, sbase, cs, do_colsum);
  dw_store<1>(R0, buf0, sbase, cs, do_colsum);
  lds_barrier();
  for (int t = 0; t < T; t += 2) {
    dw_load(R0, oa, ob, va, vb, (t + 2) * BK);
    dw_tile(lb, 0u, kx, acc, R1, buf1, sbase, cs, do_colsum);
    lds_barrier();
    dw_load(R1, oa, ob, va, vb, (t + 3) * BK);
    dw_tile(lb, (unsigned)DW_STAGE, kx, acc, R0, buf0, sbase, cs, do_colsum);
    lds_barrier();
  }
  // ---- bias gradient: thread (c = tid & 15, k-group = tid >> 4) holds the sums of columns 8c .. 8c+7 over its k rows
  if (do_colsum) {
    float* Rd = (float*)smem;                    // [32 k-groups][128 cols]
#pragma unroll
    for (int i = 0; i < 8; ++i) Rd[(tid >> 4) * 128 + (tid & 15) * 8 + i] = cs[i];
    lds_barrier();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int kg = 0; kg < 32; ++kg) sum += Rd[kg * 128 + tid];
      if (m0 + tid < M) {
        P.bias_grad[m0 + tid] = sum;
        if (OPT) {
          const size_t ix = (size_t)(P.bias_grad + m0 + tid - o.g0);
          float pp = o.p[ix], mm = o.m[ix], vv = o.v[ix];
          fc_adamw_elem(pp, sum, mm, vv, o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
          o.p[ix] = pp; o.m[ix] = mm; o.v[ix] = vv;
          if (o.shadow) o.shadow[ix] = f2bf(pp);
        }
      }
    }
  }
  lds_barrier();
  // ---- output: the two 64-row halves of the tile go through a [64][384] fp32 image, rows leave as whole 1.5-KB segments
  float* Cs = (float*)smem;
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
          *(float4*)(Cs + (i * 16 + cl) * DW_LD + wn * 96 + j * 16 + 4 * g) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    lds_barrier();
#pragma unroll 4
    for (int q = 0; q < 12; ++q) {
      const int f = tid + 512 * q, row = f / 96, c4 = (f % 96) * 4;
      const int m = m0 + 64 * h + row, n = n0 + c4;
      if (m < M && n < N) {
        float* dst = P.C + (size_t)m * P.ldc + n;
        float4 gv = *(const float4*)(Cs + row * DW_LD + c4);
        *(float4*)dst = gv;
        if (OPT) {
          const size_t ix = (size_t)(dst - o.g0);
          float4 pp = *(const float4*)(o.p + ix), mm = *(const float4*)(o.m + ix), vv = *(const float4*)(o.v + ix);
          float* Pp = (float*)&pp; float* G = (float*)&gv; float* Mm = (float*)&mm; float* V = (float*)&vv;
          bf16_t sh[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            fc_adamw_elem(Pp[k], G[k], Mm[k], V[k], o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
            sh[k] = f2bf(Pp[k]);
          }
          *(float4*)(o.p + ix) = pp; *(float4*)(o.m + ix) = mm; *(float4*)(o.v + ix) = vv;
          if (o.shadow) *(uint2*)(o.shadow + ix) = *(const uint2*)sh;
        }
      }
    }
    lds_barrier();
  }
}

// ======================================================================== loader / consumer form (default; FC_DW_WIDE=1 selects the 8-wave form)
// Same tile, same image layout, 640 threads: waves 0-7 only read fragments and issue MFMAs, waves 8-9 only move data, by LDS-DMA
// (buffer_load ... lds: no staging registers, no ds_write), in stages of 32 k rows through a four-slot ring (4 x 32 KB): the loaders
// run up to three stages ahead and wait with a counted vmcnt(32), so DMA is in flight all the time; one barrier per stage.
//   Measured (tools/kernel_bench.py under rocprofv3, one tile per CU): 207 us per tile = 1.05 us per 64 k rows = 26 B/clk per CU, at
//   the ~30 B/clk vector-memory -> LDS ceiling, against 293 us for the 8-wave form (load, compute, store back to back); client step
//   5.07 -> 4.98 ms on the same box.  A first version with two 64-row buffers (ONE k-tile of DMA in flight, drained at every barrier)
//   was faster stand-alone (249 us) but SLOWER in the step (5.12-5.22 ms, also with every weight gradient in the un-overlapped
//   tail): what counts on the loaded chip is how many bytes stay in flight across the barriers.
//   LDS-DMA fills a 1-KB piece (4 k rows x 256 B) linearly, lane L -> byte 16 L, so the kr_off swizzle is applied to the SOURCE:
//   lane (kq = L >> 4, pc = L & 15) fetches logical chunk c = (((pc >> 1) ^ s) << 1) | (pc & 1) of row 4p + kq, s = s(4p + kq); s only
//   depends on kq and on bit 1 of p, so two per-lane offsets per operand serve all pieces of an image.
// The bias gradient comes from the dY fragments of the wn == 0 consumer waves: a lane holds 8 k values of one column,
// v_dot2c_f32_bf16 against (1, 1) adds two at a time.
typedef __attribute__((ext_vector_type(2))) __bf16 dws_bf16x2;
__device__ __forceinline__ float dws_sum8(const bf16x8& f, float acc) {
  const dws_bf16x2 one = __builtin_bit_cast(dws_bf16x2, 0x3F803F80u);
  const uint4 u = __builtin_bit_cast(uint4, f);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dws_bf16x2, u.x), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dws_bf16x2, u.y), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dws_bf16x2, u.z), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dws_bf16x2, u.w), one, acc, false);
  return acc;
}
template <int KS>
__device__ __forceinline__ void dws_kstep(unsigned lb, unsigned po, const DwKeys& kx, f32x4 (&acc)[4][6], float (&cs4)[4], bool colsum) {
  bf16x8 af[4], b0[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = dw_frag<KS>(lb, kx.a[i] | po);
#pragma unroll
  for (int j = 0; j < 3; ++j) b0[j] = dw_frag<KS>(lb, kx.b[j] | po);
  frag_fence(af);
  frag_fence3(b0);
  bf16x8 b1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1[j] = dw_frag<KS>(lb, kx.b[3 + j] | po);     // in flight under the first 12 MFMAs
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], af[i], acc[i][j], 0, 0, 0);
  if (colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cs4[i] = dws_sum8(af[i], cs4[i]);
  }
  frag_fence3(b1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], af[i], acc[i][3 + j], 0, 0, 0);
}

#define DWS_SLOT 32768   // one stage: 4 images of [32 k][128 cols]
// NSLOT: stages of the ring (4: 128 KB of LDS, three stages in flight; 3: 96 KB, two in flight -- leaves room for a 64-KB GEMM workgroup of a
// backward chain on the same CU)
// ---- flag-synchronised ring (FLAGS = true; round 5): no workgroup barrier inside the main loop.  A loader wave publishes "stage s has landed"
// by writing s + 1 to ready[slot][loader] (after its counted vmcnt), a consumer wave reports "done reading the slot" by adding 1 to done[slot];
// the loader refills a slot once its done counter has reached 8 x (uses so far).  The eight consumer waves are no longer forced through the
// read -> MFMA phases of a stage together: the two waves of a SIMD drift apart and one's fragment reads run under the other's MFMAs.
// LDS words behind the ring: ready[NSLOT][2], done[NSLOT].  All flag accesses are inline asm (a compiler-visible LDS access beside an LDS-DMA in
// flight gets an s_waitcnt vmcnt(0) in front of it).
__device__ __forceinline__ unsigned dwf_ld(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void dwf_ld2_issue(unsigned long long& v, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void dwf_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void dwf_add1(unsigned addr) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory"); }
// a k-step that also requests the NEXT stage's ready words beside its second half of fragment reads (they land under the first 12 MFMAs) and
// releases its slot as soon as its last fragment read has returned
template <int KS>
__device__ __forceinline__ void dws_kstep_flags(unsigned lb, unsigned po, const DwKeys& kx, f32x4 (&acc)[4][6], float (&cs4)[4], bool colsum,
                                                unsigned next_ready_addr, unsigned long long& next_ready, unsigned done_addr, bool lane0) {
  bf16x8 af[4], b0[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = dw_frag<KS>(lb, kx.a[i] | po);
#pragma unroll
  for (int j = 0; j < 3; ++j) b0[j] = dw_frag<KS>(lb, kx.b[j] | po);
  frag_fence(af);
  frag_fence3(b0);
  bf16x8 b1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1[j] = dw_frag<KS>(lb, kx.b[3 + j] | po);
  dwf_ld2_issue(next_ready, next_ready_addr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], af[i], acc[i][j], 0, 0, 0);
  if (colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cs4[i] = dws_sum8(af[i], cs4[i]);
  }
  frag_fence3(b1);                                           // lgkmcnt(0): every read of this slot (and the ready words) has returned
  asm volatile("" : "+v"(next_ready));
  if (lane0) dwf_add1(done_addr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], af[i], acc[i][3 + j], 0, 0, 0);
}
template <bool OPT, int NSLOT, bool FLAGS = false>
__global__ void __launch_bounds__(640) k_gemm_dw_spec(const FcTnProblem* __restrict__ probs, int nprob, FcAdamW o_) {
  const FcAdamW o = fc_adamw_resolve(o_);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 8;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int idx = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  while (pi + 1 < nprob && probs[pi + 1].tile_start <= idx) ++pi;
  const FcTnProblem P = probs[pi];
  const int local = idx - P.tile_start;
  const int tile_m = local / P.tiles_n, tile_n = local % P.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * DW_BN;
  const int M = P.M, N = P.N, K = P.K;
  const bool do_colsum = (tile_n == 0) && (P.bias_grad != nullptr);
  const int S = (K + 31) / 32;                                 // stages
  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs4[4] = {0.f, 0.f, 0.f, 0.f};
  if (loader) {
    const DwOperand oa = dw_operand(P.A, P.lda, M, K), ob = dw_operand(P.B, P.ldb, N, K);
    const int lw = wave - 8, kq = lane >> 4, pc = lane & 15;
    unsigned vA[2], vB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sw = (h << 2) | kq;                           // s(k) of k = 4p + kq with bit 1 of p = h
      const int col = ((((pc >> 1) ^ sw) << 1) | (pc & 1)) * 8;
      vA[h] = (m0 + col < M) ? (unsigned)(((long)kq * P.lda + m0 + col) * 2) : FC_OOB;
      vB[h] = (unsigned)(((long)kq * P.ldb + n0 + col) * 2);
    }
    // one stage = 32 pieces of 1 KB; loader wave lw moves images 2 lw and 2 lw + 1 (8 pieces each): a running scalar offset per image
    // (+ 4 rows per piece) instead of precomputed ones, which hipcc would hoist out of the loop and spill
    const unsigned rowA4 = (unsigned)__builtin_amdgcn_readfirstlane((int)(4u * oa.kstride));
    const unsigned rowB4 = (unsigned)__builtin_amdgcn_readfirstlane((int)(4u * ob.kstride));
#define DWS_ISSUE(st)                                                                                                        \
  do {                                                                                                                       \
    _Pragma("unroll") for (int im = 0; im < 2; ++im) {                                                                        \
      const int img = 2 * lw + im;                                                                                           \
      unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((st) * 32) * (img == 0 ? oa.kstride : ob.kstride))) + \
                    (img == 0 ? 0u : 256u * (unsigned)(img - 1));                                                            \
      char* dst = smem + ((st) % NSLOT) * DWS_SLOT + img * 8192;                                                              \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                                         \
        const int h = (p >> 1) & 1;                                                                                          \
        if (img == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(oa.rsrc, (lds_ptr_t)(dst + p * 1024), 16, vA[h], so, 0, 0);     \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ob.rsrc, (lds_ptr_t)(dst + p * 1024), 16, vB[h], so, 0, 0);             \
        so += (img == 0 ? rowA4 : rowB4);                                                                                    \
      }                                                                                                                      \
    }                                                                                                                        \
  } while (0)
    // rows past K read as zeros (descriptor bounds), so stages past S may be issued freely: the counted wait stays uniform
    if (FLAGS) {
      const unsigned fl0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem + NSLOT * DWS_SLOT;
      const unsigned ready0 = fl0 + 4u * (unsigned)lw, done0 = fl0 + 8u * NSLOT;
      __builtin_amdgcn_s_barrier();                            // the flag words are zero (all ten waves; the only barrier before the epilogue)
#pragma unroll 1
      for (int d = 0; d < NSLOT - 1; ++d) DWS_ISSUE(d);
      for (int st = 0; st < S; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * (NSLOT - 2)) : "memory");   // stage st has landed (NSLOT - 2 younger stages fly on)
        dwf_st(ready0 + 8u * (unsigned)(st % NSLOT), (unsigned)st + 1u);
        if (st >= 1) {                                         // the slot of stage st - 1 is free once all eight consumer waves have left it
          const unsigned want = 8u * (unsigned)((st - 1) / NSLOT + 1), da = done0 + 4u * (unsigned)((st - 1) % NSLOT);
          while (dwf_ld(da) < want) __builtin_amdgcn_s_sleep(1);
        }
        DWS_ISSUE(st + NSLOT - 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    DWS_ISSUE(0); DWS_ISSUE(1);
    if (NSLOT == 4) DWS_ISSUE(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * (NSLOT - 2)) : "memory");   // 16 per stage and wave: stage 0 has landed
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < S; ++st) {
      DWS_ISSUE(st + NSLOT - 1);                               // the slot of stage st - 1: its readers passed the barrier above
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * (NSLOT - 2)) : "memory");   // stage st + 1 has landed (the younger ones may still fly)
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the over-issued stages land before the epilogue reuses the LDS
    }
#undef DWS_ISSUE
  } else {
    unsigned lb;
    DwKeys kx;
    {
      const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
      const int gq = lane >> 4, ii = lane & 15, q = ii >> 2, pp = ii & 3;
      const int sw = ((gq & 1) << 2) | q;
      lb = lds0 + (unsigned)((8 * gq + q) * 256 + (sw << 5) + ((pp >> 1) << 4) + (pp & 1) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) kx.a[i] = (unsigned)__builtin_amdgcn_readfirstlane((wm * 4 + i) << 5);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int u = wn * 6 + j;
        kx.b[j] = (unsigned)__builtin_amdgcn_readfirstlane(((u & 7) << 5) | (8192 * (1 + (u >> 3))));
      }
    }
    const bool colsum = do_colsum && wn == 0;
    if (FLAGS) {
      const unsigned fl0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem + NSLOT * DWS_SLOT;
      if (tid < 3 * NSLOT) dwf_st(fl0 + 4u * (unsigned)tid, 0u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // flags zeroed
      const unsigned done0 = fl0 + 8u * NSLOT;
      unsigned long long rd = 0;
      for (int st = 0; st < S; ++st) {
        const unsigned slot = (unsigned)(st % NSLOT), want = (unsigned)st + 1u;
        // both loader waves' halves of this stage have landed?  (the words were requested during the previous stage; poll if not yet)
        unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rd), r1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rd >> 32));
        while (st == 0 || r0 < want || r1 < want) {
          r0 = dwf_ld(fl0 + 8u * slot); r1 = dwf_ld(fl0 + 8u * slot + 4u);
          if (r0 >= want && r1 >= want) break;
          __builtin_amdgcn_s_sleep(1);
        }
        dws_kstep_flags<0>(lb, slot * DWS_SLOT, kx, acc, cs4, colsum, fl0 + 8u * (unsigned)((st + 1) % NSLOT), rd, done0 + 4u * slot, lane == 0);
      }
    } else {
    __builtin_amdgcn_s_barrier();                              // stage 0 has landed
    for (int st = 0; st < S; ++st) {
      dws_kstep<0>(lb, (unsigned)((st % NSLOT) * DWS_SLOT), kx, acc, cs4, colsum);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of this slot is done before it is refilled
      __builtin_amdgcn_s_barrier();
    }
    }
  }
  lds_barrier();                                               // loaders: vmcnt(0) above; nobody touches the ring any more
  // ---- bias gradient: a wn == 0 consumer lane (g = lane >> 4, r = lane & 15) holds, per row block i, the sum over its k values of
  // dY column 64 wm + 16 i + r; the four g meet in LDS
  float* Rd = (float*)smem;                                    // [128 cols][4 g]
  if (!loader && do_colsum && wn == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Rd[(wm * 64 + i * 16 + (lane & 15)) * 4 + (lane >> 4)] = cs4[i];
  }
  lds_barrier();
  if (do_colsum && tid < 128 && m0 + tid < M) {
    const float sum = Rd[tid * 4] + Rd[tid * 4 + 1] + Rd[tid * 4 + 2] + Rd[tid * 4 + 3];
    P.bias_grad[m0 + tid] = sum;
    if (OPT) {
      const size_t ix = (size_t)(P.bias_grad + m0 + tid - o.g0);
      float pp = o.p[ix], mm = o.m[ix], vv = o.v[ix];
      fc_adamw_elem(pp, sum, mm, vv, o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
      o.p[ix] = pp; o.m[ix] = mm; o.v[ix] = vv;
      if (o.shadow) o.shadow[ix] = f2bf(pp);
    }
  }
  lds_barrier();
  // ---- output (consumer threads 0..511; the loader waves only keep the barrier count).  NSLOT == 4: the two 64-row halves of the tile go
  // through a [64][388] fp32 image (99 KB); NSLOT == 3 (96 KB of LDS): four passes of 32 rows through a [32][388] image
  float* Cs = (float*)smem;
  const int g = lane >> 4, cl = lane & 15;
  constexpr int PR = NSLOT == 4 ? 64 : 32, NPASS = 128 / PR, IB = PR / 16;      // rows per pass, passes, 16-row blocks of a wave per pass
#pragma unroll
  for (int h = 0; h < NPASS; ++h) {
    if (!loader && wm == (h * PR) / 64) {
      const int i0 = ((h * PR) % 64) / 16;
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f32x4 v = acc[NSLOT == 4 ? i : (h & 1) * 2 + i][j];
          (void)i0;
          *(float4*)(Cs + (i * 16 + cl) * DW_LD + wn * 96 + j * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    lds_barrier();
    if (!loader) {
#pragma unroll 4
      for (int q = 0; q < PR * 96 / 512; ++q) {
        const int f = tid + 512 * q, row = f / 96, c4 = (f % 96) * 4;
        const int m = m0 + PR * h + row, n = n0 + c4;
        if (m < M && n < N) {
          float* dst = P.C + (size_t)m * P.ldc + n;
          float4 gv = *(const float4*)(Cs + row * DW_LD + c4);
          *(float4*)dst = gv;
          if (OPT) {
            const size_t ix = (size_t)(dst - o.g0);
            float4 pp = *(const float4*)(o.p + ix), mm = *(const float4*)(o.m + ix), vv = *(const float4*)(o.v + ix);
            float* Pp = (float*)&pp; float* G = (float*)&gv; float* Mm = (float*)&mm; float* V = (float*)&vv;
            bf16_t sh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              fc_adamw_elem(Pp[k], G[k], Mm[k], V[k], o.decay, o.beta1, o.beta2, o.eps, o.step_size, o.inv_bc2_sqrt);
              sh[k] = f2bf(Pp[k]);
            }
            *(float4*)(o.p + ix) = pp; *(float4*)(o.m + ix) = mm; *(float4*)(o.v + ix) = vv;
            if (o.shadow) *(uint2*)(o.shadow + ix) = *(const uint2*)sh;
          }
        }
      }
    }
    lds_barrier();
  }
}

int fc_gemm_dw_wide_supported(const FcTnProblem& p) {
  static const int on = fc_knob("FC_DW_WIDE", 1);
  if (!on) return 0;
  // in (N) a multiple of the 384-wide tile: every linear of the ViT-S / ViT-B shaped models (384, 768, 1536, 3072)
  return (p.N % DW_BN) == 0 && !((p.M & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) ||
                                 ((uintptr_t)p.C & 15));
}
int fc_gemm_dw_wide_tiles(const FcTnProblem& p, int* tiles_n) {
  *tiles_n = fc_cdiv(p.N, DW_BN);
  return fc_cdiv(p.M, BM) * *tiles_n;
}
static bool dw_flags() {
  static const int v = fc_knob("FC_DW_FLAGS", 0);      // the flag-synchronised ring (no barrier in the main loop)
  return v != 0;
}
int fc_gemm_dw_wide(const FcTnProblem* probs_dev, int nprob, int total_tiles, hipStream_t s, const FcAdamW* opt, int form_arg) {
  if (nprob <= 0 || total_tiles <= 0) return 0;
  const int lds = 2 * DW_STAGE;
  static bool done = false;
  if (!done) {
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
  }
  static const int form_env = fc_knob("FC_DW_WIDE", 2);      // 1: 8 waves, 2: 8 consumer + 2 loader waves
  const int form = form_arg ? form_arg : form_env;
  if (form == 2) {
    static const int slots = fc_knob("FC_DW_SLOTS", 4);
    static bool done2 = false;
    if (!done2) {
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * DWS_SLOT));
      FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * DWS_SLOT));
      done2 = true;
    }
    if (slots == 3) {
      const int lds3 = 3 * DWS_SLOT;      // 96 KB: the epilogue goes through a [32][388] image in four passes
      static bool done3 = false;
      if (!done3) {
        FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
        FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
        done3 = true;
      }
      if (opt) hipLaunchKernelGGL((k_gemm_dw_spec<true, 3>), dim3(total_tiles), dim3(640), lds3, s, probs_dev, nprob, *opt);
      else hipLaunchKernelGGL((k_gemm_dw_spec<false, 3>), dim3(total_tiles), dim3(640), lds3, s, probs_dev, nprob, FcAdamW());
#ifdef FC_PROBES      // measured: 245.6 vs 230.9 us per launch in the step, step +0.5 % (profiles/r05/dw_flags_*.txt) -- the ring is ingest-bound, not lockstep-bound
    } else if (dw_flags()) {
      const int ldsf = lds + 64;      // + the ring's flag words
      static bool donef = false;
      if (!donef) {
        FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        FC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_dw_spec<true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        donef = true;
      }
      if (opt) hipLaunchKernelGGL((k_gemm_dw_spec<true, 4, true>), dim3(total_tiles), dim3(640), ldsf, s, probs_dev, nprob, *opt);
      else hipLaunchKernelGGL((k_gemm_dw_spec<false, 4, true>), dim3(total_tiles), dim3(640), ldsf, s, probs_dev, nprob, FcAdamW());
#endif
    } else {
      if (opt) hipLaunchKernelGGL((k_gemm_dw_spec<true, 4>), dim3(total_tiles), dim3(640), lds, s, probs_dev, nprob, *opt);
      else hipLaunchKernelGGL((k_gemm_dw_spec<false, 4>), dim3(total_tiles), dim3(640), lds, s, probs_dev, nprob, FcAdamW());
    }
    FC_LAUNCH_CHECK();
    return 0;
  }
  if (opt) hipLaunchKernelGGL(k_gemm_dw_wide<true>, dim3(total_tiles), dim3(512), lds, s, probs_dev, nprob, *opt);
  else hipLaunchKernelGGL(k_gemm_dw_wide<false>, dim3(total_tiles), dim3(512), lds, s, probs_dev, nprob, FcAdamW());
  FC_LAUNCH_CHECK();
  return 0;
}
