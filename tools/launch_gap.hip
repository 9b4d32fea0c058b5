// Probe (development aid, not product): GPU-side gap between dependent kernel launches on one HIP stream.
// Each launch stamps s_memrealtime (100 MHz) at the start of block 0 and at the end of its last-finishing block (approximated by
// block 0's end for the 1-block form).  gap[i] = start[i+1] - end[i].   build: hipcc --offload-arch=gfx950 -O2 -o launch_gap launch_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ void k_stamp(long long* st, int i, int spin) {
  long long t0 = __builtin_amdgcn_s_memrealtime();
  if (spin) { long long t = t0; while (t - t0 < spin) t = __builtin_amdgcn_s_memrealtime(); }
  if (blockIdx.x == 0 && threadIdx.x == 0) { st[2 * i] = t0; st[2 * i + 1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void k_touch(float* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.f; }
int main(int argc, char** argv) {
  const int N = 400;
  long long* st; hipMalloc(&st, sizeof(long long) * 2 * N);
  float* buf; size_t nb = 64u << 20; hipMalloc(&buf, nb * 4);
  hipStream_t s; hipStreamCreate(&s);
  std::vector<long long> h(2 * N);
  for (int mode = 0; mode < 4; ++mode) {
    int blocks = (mode & 1) ? 512 : 1, spin = 300;           // 3 us kernels
    bool dirty = mode & 2;                                    // a 256-MB writer in between: dirty L2 lines at the kernel boundary
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_stamp, dim3(blocks), dim3(256), 0, s, st, i, spin);
        if (dirty && (i % 8) == 7) hipLaunchKernelGGL(k_touch, dim3((unsigned)(nb / 256)), dim3(256), 0, s, buf, nb);
      }
      hipStreamSynchronize(s);
    }
    hipMemcpy(h.data(), st, sizeof(long long) * 2 * N, hipMemcpyDeviceToHost);
    std::vector<double> gaps;
    for (int i = 100; i + 1 < N; ++i) if (!(dirty && (i % 8) == 7)) gaps.push_back((h[2 * (i + 1)] - h[2 * i + 1]) * 0.01);
    std::sort(gaps.begin(), gaps.end());
    printf("blocks %3d %s: gap end->start median %.2f us  p10 %.2f  p90 %.2f   (kernel %.2f us)\n", blocks, dirty ? "dirty" : "clean", gaps[gaps.size() / 2],
           gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], (h[2 * 200 + 1] - h[2 * 200]) * 0.01);
  }
  return 0;
}
