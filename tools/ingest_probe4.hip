// Probe (development aid, not product): the L2 -> LDS rate for DISTINCT, L2-RESIDENT data per workgroup.  ingest_probe measured a tile shared by every
// workgroup (32 TB/s chip-wide: mostly one hot set of lines) and per-workgroup panels that miss the L2 (6-7 TB/s: the fabric); a GEMM's operand reads are
// mostly L2 hits on data that only a few workgroups share.  Here every workgroup re-reads its OWN panel (128 rows x KB bytes: 48 or 96 KB, so that all
// panels of an XCD fit its 4-MB L2) REPS times by LDS-DMA, 16 KB per k-step, 1 or 3 k-steps in flight, one or two workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe4 ingest_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define REPS 32

template <int DEPTH>
__global__ void __launch_bounds__(256) k_l2(const unsigned short* __restrict__ A, unsigned* sink, int kel) {   // kel: k elements per row (row = kel * 2 bytes)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // (DEPTH + 1) x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)blockIdx.x * 128 * kel), 0, 128 * kel * 2, 0x00020000);
  unsigned va[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (wave * 4 + p) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    va[p] = (unsigned)((row * kel + c * 8) * 2);
  }
  const int ksteps = kel / 64, T = ksteps * REPS;
  auto issue = [&](int t) {
    char* buf = smem + (t % (DEPTH + 1)) * 16384;
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((t % ksteps) * 128);
#pragma unroll
    for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(buf + (wave * 4 + p) * 1024), 16, va[p], so, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  for (int t = 0; t < T; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink[0] == 0x12345678u) sink[blockIdx.x] = *(unsigned*)smem;
}

template <int DEPTH>
static void run(const unsigned short* A, unsigned* sink, int grid, int kel) {
  auto k = k_l2<DEPTH>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (DEPTH + 1) * 16384);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), (DEPTH + 1) * 16384, 0, A, sink, kel);
  (void)hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), (DEPTH + 1) * 16384, 0, A, sink, kel);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 16384.0 * (kel / 64) * REPS * grid;
  printf("panel %3d KB per workgroup, %3d workgroups (%5.1f MB in all, %4.1f MB per XCD), %d k-step(s) in flight: k-step %5.0f ns   %6.1f GB/s per CU   %5.2f TB/s chip\n", kel * 256 / 1024, grid,
         grid * kel * 256 / 1e6, grid * kel * 256 / 8e6, DEPTH, ms * 1e6 / ((kel / 64) * REPS), bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned short* A;
  unsigned* sink;
  (void)hipMalloc(&A, (size_t)64 << 20);
  (void)hipMalloc(&sink, 4096 * 4);
  (void)hipMemset(sink, 0, 4096 * 4);
  (void)hipMemset(A, 0x3c, (size_t)64 << 20);
  for (int kel : {384, 192}) {
    run<1>(A, sink, 256, kel); run<3>(A, sink, 256, kel);
    if (kel == 192) { run<1>(A, sink, 512, kel); run<3>(A, sink, 512, kel); }
  }
  run<1>(A, sink, 512, 384); run<3>(A, sink, 512, 384);      // 6 MB per XCD: does not fit the L2 any more
  return 0;
}
