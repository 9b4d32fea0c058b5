// Probe of gfx950 MFMA fragment maps and ds_read_b64_tr_b16 semantics (development aid, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

__global__ void k_mfma16(const float* A, const float* B, float* C) {  // A[16][32], B[32][16]
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)f2bf(A[r * 32 + 8 * g + j]); b[j] = (short)f2bf(B[(8 * g + j) * 16 + r]); }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(g * 4 + j) * 16 + r] = c[j];
}
__global__ void k_mfma32(const float* A, const float* B, float* C) {  // A[32][16], B[16][32]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)f2bf(A[r * 16 + 8 * h + j]); b[j] = (short)f2bf(B[(8 * h + j) * 32 + r]); }
  f32x16 c; for (int j = 0; j < 16; ++j) c[j] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 16; ++j) C[((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = c[j];
}
__global__ void k_mfma_f32(const float* A, const float* B, float* C) {  // 16x16x4 f32: A[16][4], B[4][16]
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * 4 + g], B[g * 16 + r], c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(g * 4 + j) * 16 + r] = c[j];
}
// tr read: LDS tile T[32][16] of shorts, value = 100*row + col. Each lane reads block rows 4*grp..4*grp+3
__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short T[32 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 32 * 16; i += 64) T[i] = (short)(100 * (i / 16) + (i % 16));
  __syncthreads();
  int grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const short* addr = &T[(4 * grp + q) * 16 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
  float *dA, *dB, *dC; short* dS;
  CK(hipMalloc(&dA, 4096)); CK(hipMalloc(&dB, 4096)); CK(hipMalloc(&dC, 4096)); CK(hipMalloc(&dS, 1024));
  std::vector<float> A(512), B(512), C(1024), R(1024);
  auto fill = [&]() { for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 13 - 6); B[i] = (float)((i * 5 + 1) % 11 - 5); } };
  fill();
  // 16x16x32
  CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
  k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  int bad = 0; for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[m * 32 + k] * B[k * 16 + n]; if (s != C[m * 16 + n]) ++bad; }
  printf("mfma16x16x32 bf16: %d mismatches\n", bad);
  k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
  bad = 0; for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n]; if (s != C[m * 32 + n]) ++bad; }
  printf("mfma32x32x16 bf16: %d mismatches\n", bad);
  k_mfma_f32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  bad = 0; for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 4; ++k) s += A[m * 4 + k] * B[k * 16 + n]; if (s != C[m * 16 + n]) ++bad; }
  printf("mfma16x16x4 f32: %d mismatches\n", bad);
  std::vector<short> S(256);
  k_tr<<<1, 64>>>(dS); CK(hipMemcpy(S.data(), dS, 512, hipMemcpyDeviceToHost));
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int expect = 100 * (4 * (l >> 4) + j) + (l & 15); if (S[l * 4 + j] != expect) ++bad; }
  printf("ds_read_tr16_b64: %d mismatches vs (lane i of group gets column i, element j = row j)\n", bad);
  if (bad) for (int l = 0; l < 64; l += 5) printf(" lane %d: %d %d %d %d\n", l, S[l*4], S[l*4+1], S[l*4+2], S[l*4+3]);
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s CUs %d clock %d MHz mem %zu GB lds/block %zu\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000, pr.totalGlobalMem >> 30, pr.sharedMemPerBlock);
  return 0;
}
