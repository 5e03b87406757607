// Probe: do MFMA and VALU work overlap on one SIMD (a) across two waves, (b) inside one wave's instruction stream?
// mode 0: every wave issues MFMAs only; 1: VALU only; 2: waves 0-3 MFMA, waves 4-7 VALU (wave w and w+4 share a SIMD);
// 3: every wave does both, MFMAs first then VALU (phases); 4: both, interleaved in the source (4 MFMAs, then a VALU slice, ...).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)lane}, b = {1, 1, 2, 2, 3, 3, 4, (short)wave};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = 0.001f * (lane + i);
  const bool do_mfma = MODE == 0 || MODE >= 3 || (MODE == 2 && wave < 4);
  const bool do_valu = MODE == 1 || MODE >= 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 8 * q; i < 8 * q + 8; ++i) {
          v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
          v[i] = __builtin_fmaf(v[i], 0.9999f, -0.5f);
          if ((i & 1) == 0) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.001f);
        }
      }
    } else {
      if (do_mfma) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
      if (do_valu) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
          v[i] = __builtin_fmaf(v[i], 0.9999f, -0.5f);
          if ((i & 1) == 0) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.001f);
        }
      }
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 2000;
  // per iteration and wave: 16 MFMAs 32x32x16 (32 cycles each = 512 cycles), 64 fma + 16 mul + 16 exp
  printf("mode0 MFMA only (8 waves/CU)        %8.1f us\n", run<0>(out, iters));
  printf("mode1 VALU only (8 waves/CU)        %8.1f us\n", run<1>(out, iters));
  printf("mode2 4 waves MFMA + 4 waves VALU   %8.1f us\n", run<2>(out, iters));
  printf("mode3 every wave: MFMA phase, VALU phase %8.1f us\n", run<3>(out, iters));
  printf("mode4 every wave: interleaved source     %8.1f us\n", run<4>(out, iters));
  hipFree(out);
  return 0;
}
