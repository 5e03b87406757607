// Diagnostic build of the wave-per-row LayerNorm BACKWARD kernel that round 1 removed (git d46d3fb~1: norm_bwd_wave_kernel<CPL>), with a
// per-lane trace, to find out WHY it was not reproducible beside a second stream (DESIGN.md "Run-to-run determinism").  Not part of the
// product library: built by tools/norm_bwd_wave_probe.py into tools/probe/libnormprobe.so.
//
// Trace, 8 dwords per (row, lane):  [0] s1 partial  [1] s2 partial  [2..5] xor-hash of the raw dwords this lane LOADED from x, dy, w, dx
//                                   [6] m1 as this lane got it from the wave reduction  [7] m2 likewise
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_bits;
typedef short bf16x8_bits __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__

DEV float bf2f(bf16_bits b) { return __uint_as_float(((uint32_t)b) << 16); }
DEV bf16_bits f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_bits)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_bits)(u >> 16);
}
DEV uint32_t load8h(const bf16_bits* p, float (&f)[8]) {
  const bf16x8_bits v = *reinterpret_cast<const bf16x8_bits*>(p);
  uint32_t h = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f[j] = bf2f((bf16_bits)v[j]);
    h = ((h << 5) | (h >> 27)) ^ (uint32_t)(uint16_t)v[j] ^ (0x9e3779b9u * (j + 1));
  }
  return h;
}
DEV void store8(bf16_bits* p, const float (&f)[8]) {
  bf16x8_bits v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(f[j]);
  *reinterpret_cast<bf16x8_bits*>(p) = v;
}
DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int CPL>
__global__ __launch_bounds__(256) void norm_bwd_wave_probe(const bf16_bits* __restrict__ x, const bf16_bits* __restrict__ dy,
                                                           const bf16_bits* __restrict__ w, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, bf16_bits* __restrict__ dx, uint32_t* __restrict__ dbg,
                                                           int rows, int dim, int is_rms, int dx_accum) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_bits* xr = x + row * dim;
  const bf16_bits* dyr = dy + row * dim;
  bf16_bits* dxr = dx + row * dim;
  const float mean = is_rms ? 0.f : mean_in[row];
  const float rstd = rstd_in[row];
  const int nchunk = dim >> 3;
  float xh[CPL][8], gw[CPL][8];
  float s1 = 0.f, s2 = 0.f;
  uint32_t hx = 0, hdy = 0, hw = 0, hdx = 0;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float g[8], wf[8];
      hx ^= load8h(xr + c * 8, xh[i]) * (2 * i + 1);
      hdy ^= load8h(dyr + c * 8, g) * (2 * i + 1);
      hw ^= load8h(w + c * 8, wf) * (2 * i + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gw[i][j] = g[j] * wf[j];
        xh[i][j] = (xh[i][j] - mean) * rstd;
        s1 += gw[i][j];
        s2 += gw[i][j] * xh[i][j];
      }
    }
  }
  const float m1 = is_rms ? 0.f : wave_sum(s1) / (float)dim;
  const float m2 = wave_sum(s2) / (float)dim;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float o[8];
      if (dx_accum) hdx ^= load8h(dxr + c * 8, o) * (2 * i + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rstd * (gw[i][j] - m1 - xh[i][j] * m2);
        o[j] = dx_accum ? o[j] + v : v;
      }
      store8(dxr + c * 8, o);
    }
  }
  uint32_t* t = dbg + (row * 64 + lane) * 8;      // host allocates rows * 64 * 8 dwords
  t[0] = __float_as_uint(s1); t[1] = __float_as_uint(s2); t[2] = hx; t[3] = hdy; t[4] = hw; t[5] = hdx;
  t[6] = __float_as_uint(m1); t[7] = __float_as_uint(m2);
}

extern "C" int probe_norm_bwd(const void* x, const void* dy, const void* w, const float* mean, const float* rstd, void* dx, uint32_t* dbg,
                              int rows, int dim, int is_rms, int dx_accum, void* stream) {
  if (!x || !dy || !w || !rstd || !dx || !dbg || rows <= 0 || dim <= 0 || (dim % 8) || dim > 1536) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (dim <= 1024)
    hipLaunchKernelGGL(norm_bwd_wave_probe<2>, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_bits*)x, (const bf16_bits*)dy, (const bf16_bits*)w, mean,
                       rstd, (bf16_bits*)dx, dbg, rows, dim, is_rms, dx_accum);
  else
    hipLaunchKernelGGL(norm_bwd_wave_probe<3>, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_bits*)x, (const bf16_bits*)dy, (const bf16_bits*)w, mean,
                       rstd, (bf16_bits*)dx, dbg, rows, dim, is_rms, dx_accum);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
