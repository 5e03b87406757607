// common.h -- shared device/host helpers for libovla_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ovla.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_bits;  // 8 bf16 bit patterns = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4_bits;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef unsigned short bf16_bits;

#define OVLA_DEV __device__ __forceinline__

OVLA_DEV float bf2f(bf16_bits u) { return __uint_as_float(((unsigned)u) << 16); }
// round-to-nearest-even via the hardware convert (NaN stays NaN; see MI355X_MICROARCH "Correctness boundaries")
OVLA_DEV bf16_bits f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_bits, b);
}
OVLA_DEV float bfround(float f) { return bf2f(f2bf(f)); }

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): ~12 instructions (v_rcp, v_exp, 5 FMAs) instead of libm's branchy
// erff (~100 executed instructions per wave with divergent lanes: it made the GELU epilogue of a ViT fc1 GEMM cost 40 % of the
// launch).  The GELU outputs are rounded to bf16 (2^-9 relative) right after, three orders of magnitude coarser.
OVLA_DEV float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}
OVLA_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
OVLA_DEV float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
OVLA_DEV float gelu_tanh(float x) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  return 0.5f * x * (1.0f + tanhf(k * (x + c * x * x * x)));
}
OVLA_DEV float gelu_tanh_grad(float x) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  const float t = tanhf(k * (x + c * x * x * x));
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k * (1.0f + 3.0f * c * x * x);
}
OVLA_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }
OVLA_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

OVLA_DEV float apply_act(float v, int act) {
  switch (act) {
    case OVLA_ACT_GELU: return gelu_erf(v);
    case OVLA_ACT_RELU: return v > 0.f ? v : 0.f;
    case OVLA_ACT_SILU: return silu(v);
    case OVLA_ACT_GELU_TANH: return gelu_tanh(v);
    default: return v;
  }
}
OVLA_DEV float act_grad(float z, int act) {
  switch (act) {
    case OVLA_ACT_GELU: return gelu_erf_grad(z);
    case OVLA_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case OVLA_ACT_SILU: { float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }
    case OVLA_ACT_GELU_TANH: return gelu_tanh_grad(z);
    default: return 1.f;
  }
}

// LayerNorm output element, y = (x - mean) * rstd * w + b in fp32 with ONE explicit fma (every kernel that normalises -- the row kernels of
// elementwise.hip and the fused head tail of head_optim.hip -- goes through this function, so they agree bit for bit by construction).
OVLA_DEV float ln_affine(float x, float mean, float rstd, float w, float b) { return __builtin_fmaf((x - mean) * rstd, w, b); }

OVLA_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
OVLA_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side -------------------------------------------------------------------------------------------------
void ovla_set_error(const char* fmt, ...);
#define OVLA_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      ovla_set_error(__VA_ARGS__);             \
      return OVLA_EINVAL;                      \
    }                                          \
  } while (0)
#define OVLA_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      ovla_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return OVLA_ELAUNCH;                                                        \
    }                                                                             \
  } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
