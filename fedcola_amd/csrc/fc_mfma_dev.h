// Device-side building blocks shared by the MFMA GEMM kernels (fc_mfma.hip, fc_gemm_dw.hip): LDS tile images and their
// swizzles, LDS-DMA staging through buffer descriptors, fragment reads, 8-wide epilogue loads / stores.
#pragma once
#include "fc_kernels.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global loads / stores.
// (__syncthreads() carries a release fence: with stores or loads in flight hipcc drains vmcnt(0) in front of every
// barrier, which serialises the register prefetch pipeline and makes each epilogue wait out its own HBM write latency.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
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

// ---- staging: each thread moves 4 x 16 B per operand per k-tile.
// Loads are raw buffer loads: one wave-uniform descriptor (SGPRs) + a 32-bit per-lane byte offset that is constant over
// the k loop + a scalar k offset, so the address arithmetic costs no VGPRs; out-of-range rows of the last tile fall
// outside the descriptor and read as zero in hardware.
typedef __attribute__((vector_size(16))) unsigned int v4u;
struct Operand {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[4];   // per-lane byte offsets of the 4 pieces (k-independent)
  unsigned kstride;   // bytes per k element (KC: 2) or per k row (KR: 2*ld)
  int c8;             // KC only: first k of this lane's 16-B chunk inside a tile
};
#define FC_OOB 0x80000000u
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
    // inline asm, not __builtin_amdgcn_ds_read_tr16_b64: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the builtin
    // whenever an LDS-DMA is in flight (it cannot prove the read disjoint from the DMA's destination), which serialised the
    // next k-tile's staging behind the MFMAs of the NN kernels.  The compiler does not count these reads: the caller follows
    // the fragment reads of a k-step with frag_fence() before the MFMAs use them.
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(lds + kr_off(k0, c) + cbyte)) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(lds + kr_off(k0 + 4, c) + cbyte)) : "memory");
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// all LDS reads issued so far have returned; ties the fragments so that no consumer can be scheduled above the wait
__device__ __forceinline__ void frag_fence(bf16x8 (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}
__device__ __forceinline__ void frag_fence(bf16x8 (&f)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1])::"memory");
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
    *(uint4*)p = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
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
// ---- direct-to-LDS staging (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write.  One wave-instruction fills 1 KB
// of LDS linearly (lane L -> base + 16 L), so the swizzle is applied to the per-lane SOURCE address instead: wave w owns
// the 1-KB pieces 4w .. 4w+3 of each 16-KB operand tile.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int MODE>
__device__ __forceinline__ Operand make_operand_glds(const bf16_t* P, long ld, int row0, int nrows, int K, int wave, int lane) {
  Operand o;
  unsigned long long base = (unsigned long long)P;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const void* up = (const void*)(((unsigned long long)hi << 32) | lo);
  long rows = (MODE == KC) ? nrows : K;
  unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(((rows - 1) * ld + ((MODE == KC) ? K : nrows)) * 2));
  o.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)up, 0, (int)bytes, 0x00020000);
  o.kstride = (MODE == KC) ? 2u : (unsigned)(ld * 2);
  o.c8 = 0;
  return o;
}
// NP: 1-KB pieces per wave and operand tile: 4 = a 128-row (KC) / 64-k-row (KR) tile, 2 = a 64-row KC tile (the 64 x 128 output tile)
template <int MODE, int NP = 4>
__device__ __forceinline__ void retarget_glds(Operand& o, long ld, int row0, int nrows, int wave, int lane, bool valid) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int piece = wave * NP + p;
    if (MODE == KC) {   // piece = 8 rows x 128 B; lane -> (row, physical chunk)
      int row = piece * 8 + (lane >> 3), pc = lane & 7;
      int c = pc ^ ((row >> 1) & 7);
      int r = row0 + row;
      if (p == 0) o.c8 = c * 8;                           // (c differs per piece only through the row swizzle; see stage_glds)
      o.voff[p] = (valid && r < nrows) ? (unsigned)((r * ld + c * 8) * 2) : FC_OOB;
    } else {            // piece = 4 k-rows x 256 B; lane -> (k row, physical 16-B chunk)
      int k = piece * 4 + (lane >> 4), pc = lane & 15;
      int sw = (((k >> 3) & 1) << 2) | (k & 3);
      int c = (((pc >> 1) ^ sw) << 1) | (pc & 1);
      int col = row0 + c * 8;
      o.voff[p] = (valid && col < nrows) ? (unsigned)((k * ld + col) * 2) : FC_OOB;
    }
  }
}
// issue the 4 pieces of one operand tile for k-tile k0 into `buf` (16 KB)
template <int MODE, int NP = 4>
__device__ __forceinline__ void stage_glds(const Operand& o, char* buf, int k0, int K, int wave, int lane) {
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)k0 * o.kstride));
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    unsigned vo = o.voff[p];
    if (MODE == KC) {  // k tail (K % 64 != 0): this lane's logical chunk within the tile
      int row = (wave * NP + p) * 8 + (lane >> 3);
      int c = (lane & 7) ^ ((row >> 1) & 7);
      vo = (k0 + c * 8 < K) ? vo : FC_OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(o.rsrc, (lds_ptr_t)(buf + (wave * NP + p) * 1024), 16, vo, soff, 0, 0);
  }
}
// buffer descriptor for the epilogue stores: lanes outside the matrix use an out-of-range offset, so every thread issues the
// SAME number of store instructions per tile and the main loop can use a counted s_waitcnt that skips them
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_store_rsrc(void* p, long bytes) {
  unsigned long long base = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  void* up = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(up, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
template <typename TC>
__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, size_t elem_off, bool ok, const float (&v)[8]) {
  if (sizeof(TC) == 2) {
    uint4 u = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
    __builtin_amdgcn_raw_buffer_store_b128(*(v4u*)&u, r, ok ? (unsigned)(elem_off * 2) : FC_OOB, 0, 0);
  } else {
    float4 a = make_float4(v[0], v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
    __builtin_amdgcn_raw_buffer_store_b128(*(v4u*)&a, r, ok ? (unsigned)(elem_off * 4) : FC_OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(*(v4u*)&b, r, ok ? (unsigned)(elem_off * 4 + 16) : FC_OOB, 0, 0);
  }
}
__device__ __forceinline__ int xcd_remap(int b, int nwg) {  // blocks b, b+8, ... share an XCD: give each XCD a contiguous id range
  int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
