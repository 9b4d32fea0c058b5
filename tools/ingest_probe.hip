// Probe (development aid, not product): how many bytes per clock a CU takes in when a GEMM's two operand tiles arrive by different
// paths.  The NT / NN kernels of fc_mfma.hip stage BOTH operand tiles by LDS-DMA (buffer_load ... lds) and their k-step time is the
// CU's L2 -> LDS ingest (~28 B/clk).  Question: do LDS-DMA and plain global -> VGPR loads share that limit, or
// can a kernel that takes A through LDS (shared by its waves) and B straight into MFMA-fragment registers (each wave its own 32
// columns: no sharing needed in a 1 x 4 wave grid) take in more?
//   mode 0: A by LDS-DMA + B by LDS-DMA   (the product kernel's k-step: 32 KB)
//   mode 1: A by LDS-DMA + B by fragment-shaped global loads into VGPRs (16 rows x 64 B per wave-instruction)
//   mode 2: A by LDS-DMA only (16 KB)          mode 3: B by fragment loads only (16 KB)
//   mode 4: B by full-line global loads into VGPRs (8 rows x 128 B per wave-instruction) only (16 KB)
//   mode 5: A by LDS-DMA + B by full-line global loads
//   mode 6: B (the L2-resident operand) by LDS-DMA only (16 KB)      mode 7: A (the cold operand) by full-line loads -> VGPR only
//   mode 8: B by full-line loads -> VGPR -> ds_write_b128 into LDS (register staging, 16 KB)
//   mode 9: A AND B by full-line loads -> VGPR -> ds_write_b128 (register staging of the whole k-step, 32 KB)
// A: [rows][K] bf16, one 128-row panel per workgroup (cold: 100 MB in all); W: [128][K], shared by all workgroups (L2 hits).
// build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip ; usage: ./ingest_probe [wgs_per_cu=1|2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((vector_size(16))) unsigned int v4u;
#define K 1536
#define KSTEPS (K / 64)

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_ingest(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned* sink, int rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int panel = blockIdx.x % (rows / 128);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)panel * 128 * K), 0, 128 * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 128 * K * 2, 0x00020000);
  unsigned va[4], vf[4], vl[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (wave * 4 + p) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    va[p] = (unsigned)((row * K + c * 8) * 2);                                  // DMA piece: 8 rows x 128 B
    vl[p] = (unsigned)((((wave * 4 + p) * 8 + (lane >> 3)) * K + (lane & 7) * 8) * 2);   // full-line register load: same shape
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) vf[j * 2 + ks] = (unsigned)(((wave * 32 + j * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8) * 2);   // MFMA B fragment
  unsigned acc = 0;
  v4u r[4];
  for (int rep = 0; rep < 4; ++rep)
    for (int t = 0; t < KSTEPS; ++t) {
      char* buf = smem + (t & 1) * 32768;
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(t * 128);
      if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5) {
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(buf + (wave * 4 + p) * 1024), 16, va[p], so, 0, 0);
      }
      if (MODE == 0 || MODE == 6) {
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(buf + 16384 + (wave * 4 + p) * 1024), 16, va[p], so, 0, 0);
      }
      v4u r2[4];
      if (MODE == 7 || MODE == 9) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r2[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, vl[q], so, 0);
      }
      if (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, vf[q], so, 0);
      }
      if (MODE == 4 || MODE == 5 || MODE == 8 || MODE == 9) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, vl[q], so, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= r[q][0] ^ r[q][3];
      }
      if (MODE == 7) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= r2[q][0] ^ r2[q][3];
      }
      if (MODE == 8 || MODE == 9) {                    // swizzled 16-byte chunks of 128-byte rows, as the product's k-contiguous image
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (wave * 4 + q) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
          *(v4u*)(buf + 16384 + row * 128 + c * 16) = r[q];
          if (MODE == 9) *(v4u*)(buf + row * 128 + c * 16) = r2[q];
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc + *(unsigned*)smem;
}

template <int MODE>
static double run(const unsigned short* A, const unsigned short* W, unsigned* sink, int rows, int grid, hipStream_t s) {
  hipFuncSetAttribute((const void*)k_ingest<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_ingest<MODE>, dim3(grid), dim3(256), 65536, s, A, W, sink, rows);
  hipEventRecord(e0, s);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_ingest<MODE>, dim3(grid), dim3(256), 65536, s, A, W, sink, rows);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const int cus = 256, grid = cus * per_cu, rows = 128 * 256;
  unsigned short *A, *W;
  unsigned* sink;
  hipMalloc(&A, (size_t)rows * K * 2);
  hipMalloc(&W, (size_t)128 * K * 2);
  hipMalloc(&sink, 4096 * 4);
  std::vector<unsigned short> h((size_t)rows * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(i * 2654435761u >> 16);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)128 * K * 2, hipMemcpyHostToDevice);
  hipStream_t s;
  hipStreamCreate(&s);
  const char* names[10] = {"A dma + B dma (product k-step)", "A dma + B fragment loads -> VGPR", "A dma only", "B fragment loads only", "B full-line loads -> VGPR only",
                           "A dma + B full-line loads -> VGPR", "B dma only (L2-resident operand)", "A full-line loads -> VGPR only (cold operand)",
                           "B full-line loads -> VGPR -> ds_write", "A + B full-line loads -> VGPR -> ds_write"};
  const double kb[10] = {32, 32, 16, 16, 16, 32, 16, 16, 16, 32};
  double ms[10];
  ms[0] = run<0>(A, W, sink, rows, grid, s); ms[1] = run<1>(A, W, sink, rows, grid, s); ms[2] = run<2>(A, W, sink, rows, grid, s);
  ms[3] = run<3>(A, W, sink, rows, grid, s); ms[4] = run<4>(A, W, sink, rows, grid, s); ms[5] = run<5>(A, W, sink, rows, grid, s);
  ms[6] = run<6>(A, W, sink, rows, grid, s); ms[7] = run<7>(A, W, sink, rows, grid, s); ms[8] = run<8>(A, W, sink, rows, grid, s);
  ms[9] = run<9>(A, W, sink, rows, grid, s);
  printf("# %d workgroup(s) of 256 threads per CU, %d k-steps x 4 repeats per launch, K = %d; per k-step one barrier + vmcnt(0)\n", per_cu, KSTEPS, K);
  for (int m = 0; m < 10; ++m) {
    const double bytes = kb[m] * 1024.0 * KSTEPS * 4 * grid;
    printf("mode %d  %-36s %8.1f us   %7.1f GB/s per CU   %6.2f TB/s chip   k-step %6.0f ns per workgroup\n", m, names[m], ms[m] * 1e3, bytes / (ms[m] * 1e-3) / cus / 1e9,
           bytes / (ms[m] * 1e-3) / 1e12, ms[m] * 1e6 / (KSTEPS * 4));
  }
  return 0;
}
