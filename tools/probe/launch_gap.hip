// Probe: stream-ordered back-to-back launches of (a) an empty kernel, (b) a 1-workgroup kernel that touches memory, (c) a 304-workgroup
// kernel reading 8.5 MB: the per-launch cost a chain of small dependent kernels pays on this part.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_small(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ __launch_bounds__(256) void k_read(const float4* in, float* out, int n4) {
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.f) out[0] = acc;
}
template <typename F> float run(F f, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < n; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
  float *buf, *big; (void)hipMalloc(&buf, 1 << 20); (void)hipMalloc(&big, 16 << 20); (void)hipMemset(big, 0, 16 << 20);
  printf("empty kernel, 1 block        : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 2000));
  printf("empty kernel, 1024 blocks    : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0); }, 2000));
  printf("1 store per block, 304 blocks: %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_small, dim3(304), dim3(256), 0, 0, buf); }, 2000));
  printf("read 8.5 MB, 304 blocks      : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_read, dim3(304), dim3(256), 0, 0, (const float4*)big, buf, (int)(8.5e6 / 16)); }, 2000));
  printf("read 8.5 MB, 1216 blocks     : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_read, dim3(1216), dim3(256), 0, 0, (const float4*)big, buf, (int)(8.5e6 / 16)); }, 2000));
  return 0;
}
