// Probe (development aid, not product): is the staging rate of an HBM-cold operand a limit of the CHIP (fabric / HBM) or of the CU?
// Every workgroup streams its own 128-row panels (K = 1536, 4 panels one after the other = 1.5 MB) by LDS-DMA, no reuse anywhere, each
// launch on another slice of a 3-GB ring.  Variables: workgroups in the launch (32 .. 256 on 256 CUs: one per CU at most; 512: two per CU),
// k-steps in flight (1 / 3), the cache-policy bits of the load (aux: 1 = sc0, 2 = nt, 16 = sc1), LDS-DMA or loads into VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe3 ingest_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((vector_size(16))) unsigned int v4u;
#define K 1536
#define KSTEPS (K / 64)
#define PANELS 4

template <int DEPTH, int AUX, int VG>
__device__ __forceinline__ void issue_step(char* smem, __amdgpu_buffer_rsrc_t ra, const unsigned (&va)[4], int t_, int wave, v4u (&r)[4]) {
  char* buf = smem + (t_ % (DEPTH + 1)) * 16384;
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((t_ / KSTEPS) * (128 * K * 2) + (t_ % KSTEPS) * 128);
  if (VG) {
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = __builtin_amdgcn_raw_buffer_load_b128(ra, va[p], so, AUX);
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(buf + (wave * 4 + p) * 1024), 16, va[p], so, 0, AUX);
  }
}

template <int CFG>   // CFG = 1000 VG + 100 DEPTH + AUX
__global__ void __launch_bounds__(256) k_ingest(const unsigned short* __restrict__ A, unsigned* sink) {
  constexpr int VG = CFG / 1000, DEPTH = (CFG / 100) % 10, AUX = CFG % 100;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // (DEPTH + 1) x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)blockIdx.x * PANELS * 128 * K), 0, PANELS * 128 * K * 2, 0x00020000);
  unsigned va[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (wave * 4 + p) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    va[p] = (unsigned)((row * K + c * 8) * 2);
  }
  constexpr int T = KSTEPS * PANELS;
  v4u r[4] = {};
  unsigned acc = 0;
  if (VG) {                       // loads into registers: one k-step in flight, consumed after the wait
    for (int t = 0; t < T; ++t) {
      issue_step<DEPTH, AUX, 1>(smem, ra, va, t, wave, r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 4; ++p) acc ^= r[p][0] ^ r[p][3];
      __builtin_amdgcn_s_barrier();
    }
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_step<DEPTH, AUX, 0>(smem, ra, va, d, wave, r);
    for (int t = 0; t < T; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_step<DEPTH, AUX, 0>(smem, ra, va, t + DEPTH < T ? t + DEPTH : T - 1, wave, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (sink[0] == 0x12345678u) sink[blockIdx.x] = *(unsigned*)smem + acc;
}

typedef void (*kfn_t)(const unsigned short*, unsigned*);
static size_t g_ring_bytes;
static void run(kfn_t kfn, int VG, int DEPTH, int AUX, const unsigned short* A, unsigned* sink, int grid, hipStream_t s) {
  const int lds = (DEPTH + 1) * 16384;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const size_t set = (size_t)grid * PANELS * 128 * K * 2;
  const int nsets = (int)(g_ring_bytes / set);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int it = 0;
  for (int i = 0; i < 3; ++i, ++it) hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, A + (size_t)(it % nsets) * set / 2, sink);
  (void)hipEventRecord(e0, s);
  const int reps = 12;
  for (int i = 0; i < reps; ++i, ++it) hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, A + (size_t)(it % nsets) * set / 2, sink);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 16384.0 * KSTEPS * PANELS * grid;
  printf("%-4s in flight %d  aux %2d  workgroups %3d : k-step %6.0f ns   %6.1f GB/s per workgroup   %5.2f TB/s chip   (%d ring slices)\n", VG ? "vgpr" : "dma", DEPTH, AUX, grid,
         (ms * 1e6 - 3000) / (KSTEPS * PANELS), bytes / (ms * 1e-3 - 3e-6) / grid / 1e9, bytes / (ms * 1e-3 - 3e-6) / 1e12, nsets);
}

int main() {
  g_ring_bytes = (size_t)3 << 30;
  unsigned short* A;
  unsigned* sink;
  if (hipMalloc(&A, g_ring_bytes) != hipSuccess) { printf("no memory\n"); return 1; }
  (void)hipMalloc(&sink, 4096 * 4);
  (void)hipMemset(sink, 0, 4096 * 4);
  (void)hipMemset(A, 0x3c, g_ring_bytes);
  hipStream_t s;
  (void)hipStreamCreate(&s);
  printf("# launch overhead of ~3 us subtracted from the per-launch HIP-event time\n");
  const int grids[] = {32, 64, 128, 256, 512};
#define ROW(vg_, d_, aux_) for (int g : grids) run(k_ingest<1000 * vg_ + 100 * d_ + aux_>, vg_, d_, aux_, A, sink, g, s);
  ROW(0, 1, 0) ROW(0, 3, 0)
  ROW(0, 1, 1) ROW(0, 1, 2) ROW(0, 1, 3) ROW(0, 1, 16) ROW(0, 1, 17) ROW(0, 1, 18) ROW(0, 1, 19)
  ROW(0, 3, 2) ROW(0, 3, 17) ROW(0, 3, 19)
  ROW(1, 1, 0) ROW(1, 1, 2) ROW(1, 1, 17)
  return 0;
}
