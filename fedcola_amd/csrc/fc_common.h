// Common device/host helpers for the FedCola MI355X (gfx950) hot-path library.
// CDNA4 only: 64-wide wavefronts are hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#define FC_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 storage

// ---------------------------------------------------------------- error handling (no exceptions across the ABI)
void fc_set_error(const char* fmt, ...);
#define FC_CHECK_HIP(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fc_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                            \
    }                                                                                       \
  } while (0)
#define FC_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      fc_set_error(__VA_ARGS__);   \
      return -1;                   \
    }                              \
  } while (0)
#define FC_LAUNCH_CHECK() FC_CHECK_HIP(hipGetLastError())
// Measurement aids (skip a kernel family, GEMM phase ablations, in-kernel stamps) exist only in the tools build
// (-DFC_PROBES -> libfedcola_hip_probes.so, tools/build_probes.py); the product library contains none of them.
#ifdef FC_PROBES
#include <stdlib.h>
#include <string.h>
static inline bool fc_ablated(const char* what) {
  static const char* ab = getenv("FC_ABLATE");
  return ab && strstr(ab, what);
}
#define FC_ABLATED(what) fc_ablated(what)
#else
#define FC_ABLATED(what) false
#endif
// Tuning and experiment knobs (FC_MB_FIRST, FC_DW_FLUSH, FC_SCHEDULE, FC_FUSED_OPT, ...) exist only in the tools build: the product library
// reads no FC_* environment variable and always runs the defaults.
#ifdef FC_PROBES
static inline int fc_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* fc_knob_str(const char* name) { return getenv(name); }
#else
static inline int fc_knob(const char*, int dflt) { return dflt; }
static inline const char* fc_knob_str(const char*) { return nullptr; }
#endif
#define FC_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__host__ __device__ inline float bf2f(bf16_t u) {
  union { uint32_t i; float f; } c;
  c.i = ((uint32_t)u) << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  __bf16 h = (__bf16)f;                     // v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN stays NaN
  return *(bf16_t*)&h;
#else
  union { uint32_t i; float f; } c;
  c.f = f;
  uint32_t u = c.i;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // keep NaN a NaN
  return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);                // round to nearest even
#endif
}
// two floats -> packed bf16x2 (one v_cvt_pk_bf16_f32)
__device__ inline uint32_t f2bf2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return *(uint32_t*)&v;
}

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ inline float ld(const float* p, size_t i) { return p[i]; }
  static __device__ inline void st(float* p, size_t i, float v) { p[i] = v; }
  static __device__ inline float rt(float v) { return v; }              // the value as it reads back after a store
};
template <> struct Io<bf16_t> {
  static __device__ inline float ld(const bf16_t* p, size_t i) { return bf2f(p[i]); }
  static __device__ inline void st(bf16_t* p, size_t i, float v) { p[i] = f2bf(v); }
  static __device__ inline float rt(float v) { return bf2f(f2bf(v)); }
};

// ---------------------------------------------------------------- wave64 reductions (DPP/shuffle, no LDS)
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * __expf(-0.5f * x * x) * 0.39894228040143267794f;
}

// erf-GELU with the Abramowitz-Stegun 7.1.26 rational form of erf (|error| <= 1.5e-7, far below bf16 resolution): one
// v_exp + one v_rcp + a few FMAs instead of libm's erff (~40 VALU ops).  Used by the bf16 MFMA epilogues only; the fp32
// parity path keeps erff.
// (round 5: exp(-x^2/2) as v_exp_f32 of -(k x)^2 with k = sqrt(log2(e) / 2) -- one multiply instead of three -- and the argument of the
// reciprocal as one fma of |x|: nine VALU + two transcendental instructions per element instead of thirteen + two, and v_rcp_f32 instead
// of __frcp_rn, which expands to the ten-instruction IEEE division.  fc_mlp.hip's ml_gelu runs the same sequence stage by stage.)
#define FC_GELU_K 0.84932180028801904272f      /* sqrt(log2(e) / 2) */
#define FC_GELU_P 0.23164189045929018f         /* 0.3275911 / sqrt(2) */
__device__ inline void gelu_fast_parts(float x, float& cdf, float& pdf) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), FC_GELU_P, 1.0f));
  const float w = x * FC_GELU_K;
  const float ex = __builtin_amdgcn_exp2f(-w * w);         // = exp(-x^2/2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfz = 1.0f - poly * ex;                     // erf(|x|/sqrt 2)
  cdf = 0.5f + copysignf(0.5f * erfz, x);
  pdf = ex * 0.39894228040143267794f;
}
__device__ inline float gelu_fast(float x) { float c, p; gelu_fast_parts(x, c, p); return x * c; }
__device__ inline float gelu_fast_grad(float x) { float c, p; gelu_fast_parts(x, c, p); return c + x * p; }

static inline int fc_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- GEMM epilogue description (shared by generic + MFMA GEMMs)
struct GemmEpi {
  const float* bias = nullptr;      // [N] added to acc
  const void* res = nullptr;        // residual [M,N] in the OUTPUT element type (may alias C)
  const float* rowscale = nullptr;  // per-sample multiplier (drop-path) applied to (acc+bias) before the residual add
  int rows_per_sample = 1;          // rowscale index = m / rows_per_sample
  void* preact = nullptr;           // if set: store (acc+bias) here (output type) and write gelu(acc+bias) to C
  const void* gelu_in = nullptr;    // if set: C = acc * gelu'(gelu_in[m,n])   (output type)
  int gelu_saved_grad = 0;          // preact receives gelu'(acc+bias) instead of (acc+bias); gelu_in then already holds gelu'
  int accumulate = 0;               // C += result (fp32 C only)
  int out_zeroed = 0;               // caller guarantees C is zero-filled (split-K kernels skip their own memset)
  float alpha = 1.0f;               // result = alpha*acc (+bias...)
  int patch_rows = 0;               // >0: patch-embed remap: out row = m + m/patch_rows + 1, adds pos[1 + m%patch_rows]
  const float* pos = nullptr;       // [1+patch_rows, N] fp32
  int dbg = 0;                      // tools build only (FC_PROBES, FC_GEMM_DBG): 1 = no global loads in the loop, 2 = no epilogue, 4 = no MFMA
};
