// Probe (development aid, not product): what limits the LDS-DMA staging rate of one workgroup -- waves issuing, pieces in flight per wave,
// or the CU.  One k-step = A[128 x 64] (one panel per workgroup, Infinity-Cache resident) + W[128 x 64] (shared, L2 resident) = 32 pieces
// of 1 KB each per operand half... 16 + 16 pieces.  Variants: waves per workgroup NW in {4, 8, 16} (pieces per wave and k-step: 8 / 4 / 2),
// k-steps in flight DEPTH in {1, 2, 3} (counted vmcnt, ring of DEPTH + 1 stages), workgroups per CU in {1, 2}.
// build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe2 ingest_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define K 1536
#define KSTEPS (K / 64)

template <int NW, int DEPTH, int WHAT, int PP>
__device__ __forceinline__ void issue_step(char* smem, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, const unsigned (&va)[PP], int t_, int wave) {
  char* buf = smem + (t_ % (DEPTH + 1)) * 32768;
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((t_ % KSTEPS) * 128);
  if (WHAT != 2) {
#pragma unroll
    for (int p = 0; p < PP; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(buf + (wave * PP + p) * 1024), 16, va[p], so, 0, 0);
  }
  if (WHAT != 1) {
#pragma unroll
    for (int p = 0; p < PP; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(buf + 16384 + (wave * PP + p) * 1024), 16, va[p], so, 0, 0);
  }
}

template <int CFG>   // CFG = 100 NW + 10 DEPTH + WHAT;  WHAT: 0 = A + W, 1 = A only, 2 = W only
__global__ void __launch_bounds__(1024) k_ingest(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned* sink, int rows) {
  constexpr int NW = CFG / 100, DEPTH = (CFG / 10) % 10, WHAT = CFG % 10;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // (DEPTH + 1) x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int panel = blockIdx.x % (rows / 128);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)panel * 128 * K), 0, 128 * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 128 * K * 2, 0x00020000);
  constexpr int PP = 16 / NW;              // pieces per wave per operand per k-step
  unsigned va[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int row = (wave * PP + p) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    va[p] = (unsigned)((row * K + c * 8) * 2);
  }
  constexpr int PER = (WHAT == 0 ? 2 : 1) * PP;      // operations per wave per k-step
  constexpr int T = KSTEPS * 4;
#define ISSUE(tt) issue_step<NW, DEPTH, WHAT, PP>(smem, ra, rw, va, (tt), wave)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ISSUE(d);
  for (int t = 0; t < T; ++t) {
    // (stages past T are issued too: the wait count stays constant)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (DEPTH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    ISSUE(t + DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink[0] == 0x12345678u) sink[blockIdx.x] = *(unsigned*)smem;
}

typedef void (*kfn_t)(const unsigned short*, const unsigned short*, unsigned*, int);
static void run(kfn_t kfn, int NW, int DEPTH, int WHAT, const unsigned short* A, const unsigned short* W, unsigned* sink, int rows, int per_cu, hipStream_t s) {
  const int lds = (DEPTH + 1) * 32768, grid = 256 * per_cu;
  if (lds * per_cu > 163840) return;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW), lds, s, A, W, sink, rows);
  (void)hipEventRecord(e0, s);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW), lds, s, A, W, sink, rows);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double kb = WHAT == 0 ? 32 : 16, bytes = kb * 1024.0 * KSTEPS * 4 * grid;
  printf("%-6s waves %2d  in flight %d  wg/CU %d : k-step %6.0f ns   %6.1f GB/s per CU   %5.2f TB/s chip\n", WHAT == 0 ? "A+W" : (WHAT == 1 ? "A" : "W"), NW, DEPTH, per_cu,
         ms * 1e6 / (KSTEPS * 4), bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const int rows = 128 * 256;
  unsigned short *A, *W;
  unsigned* sink;
  (void)hipMalloc(&A, (size_t)rows * K * 2);
  (void)hipMalloc(&W, (size_t)128 * K * 2);
  (void)hipMalloc(&sink, 4096 * 4);
  (void)hipMemset(sink, 0, 4096 * 4);
  std::vector<unsigned short> h((size_t)rows * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(i * 2654435761u >> 16);
  (void)hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), (size_t)128 * K * 2, hipMemcpyHostToDevice);
  hipStream_t s;
  (void)hipStreamCreate(&s);
#define ROW(a_, b_, c_) run(k_ingest<100 * a_ + 10 * b_ + c_>, a_, b_, c_, A, W, sink, rows, 1, s); run(k_ingest<100 * a_ + 10 * b_ + c_>, a_, b_, c_, A, W, sink, rows, 2, s);
  ROW(4, 1, 0) ROW(4, 2, 0) ROW(4, 3, 0) ROW(4, 4, 0)
  ROW(8, 1, 0) ROW(8, 2, 0) ROW(8, 3, 0)
  ROW(16, 1, 0) ROW(16, 2, 0) ROW(16, 3, 0)
  ROW(4, 1, 1) ROW(4, 2, 1) ROW(4, 3, 1) ROW(8, 2, 1) ROW(16, 2, 1)
  ROW(4, 1, 2) ROW(4, 2, 2) ROW(4, 3, 2)
  return 0;
}
