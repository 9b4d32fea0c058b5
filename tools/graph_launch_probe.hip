// Probe (development aid, not product): host cost of launching a chain of N small kernels eagerly vs as ONE captured hipGraph, and from
// two host threads on two streams at once.  build: hipcc --offload-arch=gfx950 -O2 -o graph_launch_probe graph_launch_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { float v[40]; };          // ~160 B of kernel arguments, like the GEMM launches
__global__ void k_small(float* p, Big b, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i & 63] += b.v[i & 31]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void eager(hipStream_t s, float* p, int chains, int n) {
  Big b{};
  for (int c = 0; c < chains; ++c)
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, p, b, i);
}
int main() {
  float* p; hipMalloc(&p, 256);
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const int chains = 400;
  for (int n : {7, 14}) {
    eager(s, p, 20, n); hipStreamSynchronize(s);
    double t0 = now(); eager(s, p, chains, n); double t1 = now(); hipStreamSynchronize(s); double t2 = now();
    printf("eager  : %2d kernels per chain: host %.2f us per kernel (%.1f us per chain), GPU-complete %.2f us per kernel\n", n, (t1 - t0) / (chains * n) * 1e6,
           (t1 - t0) / chains * 1e6, (t2 - t0) / (chains * n) * 1e6);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    eager(s, p, 1, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int c = 0; c < 20; ++c) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = now();
    for (int c = 0; c < chains; ++c) hipGraphLaunch(ge, s);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("graph  : %2d kernels per graph: host %.2f us per graph launch (%.2f us per kernel), GPU-complete %.2f us per kernel\n", n, (t1 - t0) / chains * 1e6,
           (t1 - t0) / (chains * n) * 1e6, (t2 - t0) / (chains * n) * 1e6);
    // two host threads, two streams
    t0 = now();
    std::thread th([&] { eager(s2, p + 64, chains, n); });
    eager(s, p, chains, n);
    th.join();
    t1 = now(); hipStreamSynchronize(s); hipStreamSynchronize(s2); t2 = now();
    printf("2 threads x eager: host %.2f us per kernel per thread (aggregate %.2f us per kernel), GPU-complete %.2f us per kernel\n", (t1 - t0) / (chains * n) * 1e6,
           (t1 - t0) / (2.0 * chains * n) * 1e6, (t2 - t0) / (2.0 * chains * n) * 1e6);
    // one thread alternating between two streams (what the library does today)
    Big b{};
    t0 = now();
    for (int c = 0; c < chains; ++c)
      for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, p, b, i); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, p + 64, b, i); }
    t1 = now(); hipStreamSynchronize(s); hipStreamSynchronize(s2);
    printf("1 thread alternating 2 streams: host %.2f us per kernel\n", (t1 - t0) / (2.0 * chains * n) * 1e6);
  }
  return 0;
}
