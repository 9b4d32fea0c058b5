// Probe (development aid, not product): a GEMM main loop that keeps MORE BYTES IN FLIGHT per CU than two 32-KB LDS stages allow, at the same
// 64 KB of LDS per workgroup.  The A operand (activations, [M][K] bf16) streams through a FOUR-stage LDS-DMA ring of 16-KB tiles (three k-tiles in
// flight); the B operand (weights) never touches LDS: it is read from a PRE-PACKED copy in MFMA-fragment order -- one fully coalesced 1-KB
// buffer_load per fragment, three k-tiles ahead, into registers.  Waves are laid out 1 x 4 (each wave: all 128 rows x 32 columns), so no two waves
// load the same B fragment.  Compared, on the same problem and the same cold ring of A / C buffers, with the product kernel (k_gemm_mfma, PLAIN).
// The probe stores its result straight from the accumulators (8-byte stores): the epilogue is not the question here.
// build: as tools/gemm_probe.hip (fc_mfma.hip is #included);  run: tools/gemm_probe2 M N K [ring]
#include "../fedcola_amd/csrc/fc_mfma.hip"
#include <cstdio>
#include <vector>

char g_fc_err_dummy2;
#define NS 4
// packed B: [tile_n][kt][wave][ks][j][lane] x 16 B  (tile_n: 128 columns, kt: 64 k, wave: 32 columns, ks: 32 k, j: 16 columns)
__global__ void k_pack_b(const bf16_t* __restrict__ W, uint4* __restrict__ Bp, int N, int K, int nt) {   // W[N][K] (NT form)
  const int T = K / 64;
  const long total = (long)((N + 127) / 128) * T * 4 * 2 * 2 * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int lane = idx & 63, j = (idx >> 6) & 1, ks = (idx >> 7) & 1, wave = (idx >> 8) & 3;
    const long r = idx >> 10;
    const int kt = (int)(r % T), tn = (int)(r / T);
    const int n = tn * 128 + wave * 32 + j * 16 + (lane & 15), k = kt * 64 + ks * 32 + (lane >> 4) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (n < N) v = *(const uint4*)(W + (size_t)n * K + k);
    Bp[idx] = v;
  }
}

template <int DA>
__global__ void __launch_bounds__(256, 2) k_probe2(const bf16_t* __restrict__ A, long lda, const uint4* __restrict__ Bp, bf16_t* __restrict__ C, long ldc, int M, int N,
                                                   int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NS x 16 KB of A
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (ct / tiles_n) * BM, tn = ct % tiles_n, n0 = tn * BN;
  const int T = K / BK;
  Operand oa = make_operand_glds<KC>(A, lda, 0, M, K, wave, lane);
  retarget_glds<KC>(oa, lda, m0, M, wave, lane, true);
  // this wave's packed B: 4 KB per k-tile, contiguous
  const char* bbase = (const char*)Bp + ((size_t)tn * T * 4 + wave) * 4096;
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, (int)((size_t)T * 4 * 4096), 0x00020000);
  const unsigned bl = (unsigned)lane * 16u;
  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  v4u b0[4], b1[4], b2[4];
// B loads are inline asm: hipcc's own s_waitcnt insertion merges the loop's entry states conservatively and drained the prefetch ring down to
// 4 operations in flight at every third k-step; the wait is the explicit counted one at the top of a step, tied to the registers it guards
#define LOAD_B(dst, kt)                                                                                                     \
  do {                                                                                                                      \
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((kt) < T ? (kt) * 16384 : 0x7ff00000);                     \
    asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen offset:1024\n\t"     \
                 "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:2048\n\tbuffer_load_dwordx4 %3, %4, %5, %6 offen offset:3072" \
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(bl), "s"(rb), "s"(so) : "memory");       \
  } while (0)
#define WAIT_B(bcur)                                                                                                                   \
  do {                                                                                                                                 \
    if (DA == 3) asm volatile("s_waitcnt vmcnt(16)" : "+v"(bcur[0]), "+v"(bcur[1]), "+v"(bcur[2]), "+v"(bcur[3])::"memory");            \
    else asm volatile("s_waitcnt vmcnt(4)" : "+v"(bcur[0]), "+v"(bcur[1]), "+v"(bcur[2]), "+v"(bcur[3])::"memory");                     \
  } while (0)
#define ISSUE_A(kt) stage_glds<KC>(oa, smem + ((kt) % (DA + 1)) * 16384, (kt) * BK, K, wave, lane)   /* past K: the descriptor zero-fills */
#define STEP(bcur, kt)                                                                                                      \
  do {                                                                                                                      \
    WAIT_B(bcur);                                       /* everything but the two younger k-tiles (4 DMA + 4 loads each) */  \
    __builtin_amdgcn_s_barrier();                                                                                           \
    asm volatile("" ::: "memory");                                                                                          \
    ISSUE_A((kt) + DA);                                                                                                     \
    const char* la = smem + ((kt) % (DA + 1)) * 16384;                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                      \
      bf16x8 af[8];                                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = frag_read<KC>(la, i * 16, ks, lane);                            \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&bcur[ks * 2 + j], af[i], acc[i][j], 0, 0, 0);        \
    }                                                                                                                       \
    LOAD_B(bcur, (kt) + 3);                                                                                                 \
  } while (0)
  if (DA == 3) {
    ISSUE_A(0); LOAD_B(b0, 0);
    ISSUE_A(1); LOAD_B(b1, 1);
    ISSUE_A(2); LOAD_B(b2, 2);
  } else {           // B three k-tiles ahead in registers, A one k-tile ahead in a two-stage ring: order = B0 B1 A0 B2, then per step A(k+1) ... B(k+3)
    LOAD_B(b0, 0); LOAD_B(b1, 1);
    ISSUE_A(0);
    LOAD_B(b2, 2);
  }
  for (int t = 0; t < T; t += 3) {            // T is a multiple of 3 for the model's K (384, 1152, 1536); a tail would run on zero tiles
    STEP(b0, t);
    if (t + 1 < T) STEP(b1, t + 1);
    if (t + 2 < T) STEP(b2, t + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + i * 16 + cl;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wave * 32 + j * 16 + 4 * g;
      if (n < N) *(uint2*)(C + (size_t)m * ldc + n) = make_uint2(f2bf2(acc[i][j][0], acc[i][j][1]), f2bf2(acc[i][j][2], acc[i][j][3]));
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 12608, N = argc > 2 ? atoi(argv[2]) : 384, K = argc > 3 ? atoi(argv[3]) : 1536, ring = argc > 4 ? atoi(argv[4]) : 40;
  bf16_t *A, *W, *C, *C2;
  uint4* Bp;
  const size_t a_el = ((size_t)M * K + 1023) & ~(size_t)1023, c_el = ((size_t)M * N + 1023) & ~(size_t)1023;
  const int tiles_n = (N + 127) / 128, tiles = ((M + 127) / 128) * tiles_n, T = K / 64;
  hipMalloc(&A, a_el * 2 * ring); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, c_el * 2 * ring); hipMalloc(&C2, c_el * 2 * ring);
  hipMalloc(&Bp, (size_t)tiles_n * T * 16384);
  std::vector<bf16_t> h(a_el);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (bf16_t)(0x3c00 + ((i * 2654435761u >> 20) & 0x3ff));      // values in [0.0078, 0.031): no overflow over K
  for (int r = 0; r < ring; ++r) hipMemcpy(A + (size_t)r * a_el, h.data(), a_el * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data() + 12345, (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_pack_b, dim3(512), dim3(256), 0, 0, W, Bp, N, K, tiles_n);
  hipFuncSetAttribute((const void*)k_probe2<3>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * 16384);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  GemmEpi e{}; e.alpha = 1.f;
  float ms_p = 0, ms_q = 0;
  const int reps = 20;
  for (int pass = 0; pass < 2; ++pass) {
    for (int it = 0; it < 3; ++it) fc_gemm_mfma(FC_GEMM_NT, FC_BF16, A + (size_t)(it % ring) * a_el, K, W, K, C + (size_t)(it % ring) * c_el, N, M, N, K, e, 0);
    hipEventRecord(e0);
    for (int it = 0; it < reps; ++it) fc_gemm_mfma(FC_GEMM_NT, FC_BF16, A + (size_t)((it + 3) % ring) * a_el, K, W, K, C + (size_t)((it + 3) % ring) * c_el, N, M, N, K, e, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_p, e0, e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_probe2<3>, dim3(tiles), dim3(256), NS * 16384, 0, A + (size_t)(it % ring) * a_el, (long)K, Bp, C2 + (size_t)(it % ring) * c_el, (long)N, M, N, K, tiles_n);
    hipEventRecord(e0);
    for (int it = 0; it < reps; ++it)
      hipLaunchKernelGGL(k_probe2<3>, dim3(tiles), dim3(256), NS * 16384, 0, A + (size_t)((it + 3) % ring) * a_el, (long)K, Bp, C2 + (size_t)((it + 3) % ring) * c_el, (long)N, M, N, K, tiles_n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_q, e0, e1);
  }
  float ms_r = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_probe2<1>, dim3(tiles), dim3(256), 2 * 16384, 0, A + (size_t)(it % ring) * a_el, (long)K, Bp, C2 + (size_t)(it % ring) * c_el, (long)N, M, N, K, tiles_n);
    hipEventRecord(e0);
    for (int it = 0; it < reps; ++it)
      hipLaunchKernelGGL(k_probe2<1>, dim3(tiles), dim3(256), 2 * 16384, 0, A + (size_t)((it + 3) % ring) * a_el, (long)K, Bp, C2 + (size_t)((it + 3) % ring) * c_el, (long)N, M, N, K, tiles_n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_r, e0, e1);
  }
  // same numbers?
  std::vector<bf16_t> c1((size_t)M * N), c2((size_t)M * N);
  hipMemcpy(c1.data(), C, c1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), C2, c2.size() * 2, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < c1.size(); ++i) bad += c1[i] != c2[i];
  const double fl = 2.0 * M * N * K;
  printf("M %6d N %5d K %5d (%4d tiles): product %7.1f us (%6.1f TFLOP/s)   A ring of 4 + packed B %7.1f us (x %.2f)   A ring of 2 (32 KB LDS) + packed B %7.1f us (x %.2f)   differing outputs %zu\n", M, N, K,
         tiles, ms_p / reps * 1e3, fl / (ms_p / reps) / 1e9, ms_q / reps * 1e3, ms_p / ms_q, ms_r / reps * 1e3, ms_p / ms_r, bad);
  return 0;
}
