// Phase timing of the persistent LDS-DMA GEMM main loop (development aid, not product): the product's helpers are included
// verbatim and one instrumented copy of the NT / EPI_BIAS kernel stamps s_memtime at the points of every k-step:
//   [0] top of step  [1] after s_waitcnt vmcnt  [2] after s_barrier  [3] after the DMA issue  [4] after the MFMAs
// build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -I fedcola_amd/csrc -c tools/gemm_probe.hip -o /tmp/gemm_probe.o ; hipcc --offload-arch=gfx950 /tmp/gemm_probe.o \
//         <every fedcola_amd/build/*.hip.o except fc_mfma.hip.o> -ldl -o tools/gemm_probe      (fc_mfma.hip is #included, not linked)
// run:    tools/gemm_probe M N K [mode]     mode bit0: no DMA, bit1: no MFMA
#include "../fedcola_amd/csrc/fc_mfma.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

char g_fc_err_dummy;
#define NSTAMP 5
template <int AMODE, int BMODE>
__global__ void __launch_bounds__(256, 2)
k_probe(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bm, long ldb, bf16_t* C, long ldc, int M, int N, int K, int tiles_n, int ntiles,
        GemmEpi e, long long* stamps, int max_steps, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);
  const int T = (K + BK - 1) / BK;
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int lt = first, lk = 0, lb = 0;
  Operand oa = make_operand_glds<AMODE>(A, lda, 0, M, K, wave, lane);
  Operand ob = make_operand_glds<BMODE>(Bm, ldb, 0, N, K, wave, lane);
  retarget_glds<AMODE>(oa, lda, (lt / tiles_n) * BM, M, wave, lane, lt < ntiles);
  retarget_glds<BMODE>(ob, ldb, (lt % tiles_n) * BN, N, wave, lane, lt < ntiles);
#define ISSUE_NEXT()                                                                   \
  do {                                                                                 \
    char* dst = smem + lb * 32768;                                                     \
    if (!(mode & 1)) {                                                                 \
      stage_glds<AMODE>(oa, dst, lk * BK, K, wave, lane);                              \
      stage_glds<BMODE>(ob, dst + 16384, lk * BK, K, wave, lane);                      \
    }                                                                                  \
    lb ^= 1;                                                                           \
    if (++lk == T) {                                                                   \
      lk = 0;                                                                          \
      lt += G;                                                                         \
      retarget_glds<AMODE>(oa, lda, (lt / tiles_n) * BM, M, wave, lane, lt < ntiles);  \
      retarget_glds<BMODE>(ob, ldb, (lt % tiles_n) * BN, N, wave, lane, lt < ntiles);  \
    }                                                                                  \
  } while (0)
  ISSUE_NEXT();
  int cb = 0, gs = 0;
  const __amdgpu_buffer_rsrc_t crs = make_store_rsrc((void*)C, (long)M * ldc * 2);
  EpiRegs pre;
  const bool rec = (tid == 0);
  long long* my = stamps + (size_t)blockIdx.x * max_steps * NSTAMP;
#define STAMP(i) do { if (rec && gs < max_steps) my[gs * NSTAMP + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  for (int ct = first; ct < ntiles; ct += G) {
    const int m0 = (ct / tiles_n) * BM, n0 = (ct % tiles_n) * BN;
    for (int k = 0; k < T; ++k) {
      STAMP(0);
      if (k == 0 && ct != first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(2);
      ISSUE_NEXT();
      STAMP(3);
      if (k == T - 1) epi_prefetch<EPI_BIAS, bf16_t>(pre, ldc, m0, n0, M, N, e, tid);
      if (!(mode & 2)) tile_compute<AMODE, BMODE>(smem + cb * 32768, acc, wm, wn, lane);
      asm volatile("s_nop 0" ::: "memory");
      STAMP(4);
      cb ^= 1;
      ++gs;
    }
    float* Cs = (float*)(smem + (cb ^ 1) * 32768);
    lds_barrier();
    acc_to_lds_half<0>(Cs, acc, wm, wn, lane);
    lds_barrier();
    half_epilogue<EPI_BIAS, bf16_t, true>(Cs, C, ldc, m0, n0, M, N, e, tid, 0, crs, crs, pre.bias, pre.rin0, pre.sc0);
    lds_barrier();
    acc_to_lds_half<1>(Cs, acc, wm, wn, lane);
    lds_barrier();
    half_epilogue<EPI_BIAS, bf16_t, true>(Cs, C, ldc, m0, n0, M, N, e, tid, 1, crs, crs, pre.bias, pre.rin1, pre.sc1);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();
    if (rec && gs < max_steps) my[gs * NSTAMP] = -__builtin_amdgcn_s_memtime();   // marks an epilogue end (negative)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 6304, N = argc > 2 ? atoi(argv[2]) : 384, K = argc > 3 ? atoi(argv[3]) : 1536, mode = argc > 4 ? atoi(argv[4]) : 0;
  const int ring = argc > 5 ? atoi(argv[5]) : 1;     // > 1: every launch on another A / C out of `ring` copies (HBM-cold operand); the last launch is reported
  bf16_t *A, *W, *C;
  float* bias;
  const size_t a_el = ((size_t)M * K + 1023) & ~(size_t)1023, c_el = ((size_t)M * N + 1023) & ~(size_t)1023;
  hipMalloc(&A, a_el * 2 * ring); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, c_el * 2 * ring); hipMalloc(&bias, N * 4);
  hipMemset(A, 0x3c, a_el * 2 * ring); hipMemset(W, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, N * 4);
  int tiles_n = (N + 127) / 128, tiles = ((M + 127) / 128) * tiles_n, T = (K + 63) / 64;
  int grid = std::min(tiles, 512), max_steps = ((tiles + grid - 1) / grid) * T + 2;
  long long* st;
  hipMalloc(&st, (size_t)grid * max_steps * NSTAMP * 8);
  GemmEpi e{}; e.bias = bias; e.alpha = 1.f;
  auto kfn = k_probe<KC, KC>;
  hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 3 + (ring > 1 ? ring : 0); ++it) {
    hipMemset(st, 0, (size_t)grid * max_steps * NSTAMP * 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 65536, 0, A + (size_t)(it % ring) * a_el, (long)K, W, (long)K, C + (size_t)(it % ring) * c_el, (long)N, M, N, K, tiles_n, tiles, e, st,
                       max_steps, mode);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h((size_t)grid * max_steps * NSTAMP);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  printf("M %d N %d K %d mode %d ring %d: tiles %d grid %d T %d  kernel %.1f us\n", M, N, K, mode, ring, tiles, grid, T, ms * 1e3);
  // s_memtime counts shader-clock cycles here (24 k-steps of ~2 000 counts in a 27-us kernel: ~2.1 GHz)
  const char* names[4] = {"wait vmcnt", "barrier", "issue DMA", "MFMA+reads"};
  for (int ph = 0; ph < 4; ++ph) {
    std::vector<long long> d;
    for (int g = 0; g < grid; ++g)
      for (int s = 1; s < T && s < max_steps; ++s) {       // steps 1..T-1 of the first tile of every workgroup (steady state)
        long long a = h[((size_t)g * max_steps + s) * NSTAMP + ph], b = h[((size_t)g * max_steps + s) * NSTAMP + ph + 1];
        if (a > 0 && b > 0) d.push_back(b - a);
      }
    std::sort(d.begin(), d.end());
    if (!d.empty()) printf("  %-11s median %5lld  p10 %5lld  p90 %5lld cycles   n=%zu\n", names[ph], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.size());
  }
  std::vector<long long> step;
  for (int g = 0; g < grid; ++g)
    for (int s = 1; s + 1 < T && s + 1 < max_steps; ++s) {
      long long a = h[((size_t)g * max_steps + s) * NSTAMP], b = h[((size_t)g * max_steps + s + 1) * NSTAMP];
      if (a > 0 && b > 0) step.push_back(b - a);
    }
  std::sort(step.begin(), step.end());
  if (!step.empty()) printf("  whole k-step median %lld cycles\n", step[step.size() / 2]);
  long long t0 = 1LL << 62, t1 = 0;
  for (int g = 0; g < grid; ++g) { long long a = h[(size_t)g * max_steps * NSTAMP]; if (a > 0) { t0 = std::min(t0, a); } for (int s = 0; s < max_steps; ++s) { long long b = h[((size_t)g * max_steps + s) * NSTAMP + 4]; t1 = std::max(t1, b); } }
  printf("  first stamp -> last MFMA stamp: %lld cycles\n", t1 - t0);
  return 0;
}
