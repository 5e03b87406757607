// head_optim.hip -- the N = ACTION_DIM tail of the L1 / diffusion action head (forward + loss, backward) and the fused
// AdamW step.  Reference: prismatic/models/action_heads.py:69-81, vla-scripts/finetune.py:400,407,952.
#include "common.h"
#include <math.h>

namespace {

constexpr int MAX_ADIM = 16;

// one workgroup per row: pred[m, a] = bf16(x[m,:] . W[a,:] + b[a]);  loss_sum += | bf16(pred - target) |
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const bf16_bits* __restrict__ x, const bf16_bits* __restrict__ W,
                                                           const bf16_bits* __restrict__ bias, bf16_bits* __restrict__ pred,
                                                           const bf16_bits* __restrict__ target, float* __restrict__ loss_sum,
                                                           int dim, int adim, int mse) {
  __shared__ float red[4][MAX_ADIM];
  const int m = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[MAX_ADIM];
#pragma unroll
  for (int a = 0; a < MAX_ADIM; ++a) acc[a] = 0.f;
  for (int c = threadIdx.x * 8; c < dim; c += 256 * 8) {
    const bf16x8_bits xv = *reinterpret_cast<const bf16x8_bits*>(x + (int64_t)m * dim + c);
#pragma unroll
    for (int a = 0; a < MAX_ADIM; ++a) {
      if (a < adim) {
        const bf16x8_bits wv = *reinterpret_cast<const bf16x8_bits*>(W + (int64_t)a * dim + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[a] += bf2f((bf16_bits)xv[j]) * bf2f((bf16_bits)wv[j]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < MAX_ADIM; ++a) {
    const float s = wave_sum(acc[a]);
    if (lane == 0) red[wave][a] = s;
  }
  __syncthreads();
  if (threadIdx.x < adim) {
    const int a = threadIdx.x;
    float v = red[0][a] + red[1][a] + red[2][a] + red[3][a];
    v = bfround(v + (bias ? bf2f(bias[a]) : 0.f));
    pred[(int64_t)m * adim + a] = f2bf(v);
    if (target && loss_sum) {
      const float d = bfround(bf2f(target[(int64_t)m * adim + a]) - v);
      atomicAdd(loss_sum, mse ? bfround(d * d) : fabsf(d));
    }
  }
}

// column-parallel backward: thread owns column c.  dpred[m,a] = bf16(sign(pred-target) * scale) (L1) or
// bf16(2 (pred-target) * scale) (MSE);  dx[m,c] = sum_a dpred[m,a] W[a,c];  dW[a,c] += sum_m dpred[m,a] x[m,c]
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const bf16_bits* __restrict__ x, const bf16_bits* __restrict__ W,
                                                           const bf16_bits* __restrict__ pred, const bf16_bits* __restrict__ target,
                                                           const bf16_bits* __restrict__ dpred, float scale, int mse, bf16_bits* __restrict__ dx, float* __restrict__ dW,
                                                           float* __restrict__ db, int rows, int dim, int adim) {
  extern __shared__ float dp[];  // [rows][adim]
  for (int i = threadIdx.x; i < rows * adim; i += 256) {
    if (dpred) {
      dp[i] = bf2f(dpred[i]);
      continue;
    }
    const float d = bfround(bf2f(pred[i]) - bf2f(target[i]));
    float g;
    if (mse) g = bfround(2.f * d * scale);
    else g = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    dp[i] = bfround(g);
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < adim && db) {
    float s = 0.f;
    for (int m = 0; m < rows; ++m) s += dp[m * adim + threadIdx.x];
    db[threadIdx.x] += s;
  }
  if (c >= dim) return;
  float w[MAX_ADIM], gw[MAX_ADIM];
#pragma unroll
  for (int a = 0; a < MAX_ADIM; ++a) {
    w[a] = a < adim ? bf2f(W[(int64_t)a * dim + c]) : 0.f;
    gw[a] = 0.f;
  }
  for (int m = 0; m < rows; ++m) {
    const float xv = bf2f(x[(int64_t)m * dim + c]);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < MAX_ADIM; ++a) {
      if (a < adim) {
        const float g = dp[m * adim + a];
        s += g * w[a];
        gw[a] += g * xv;
      }
    }
    dx[(int64_t)m * dim + c] = f2bf(s);
  }
  if (dW) {
#pragma unroll
    for (int a = 0; a < MAX_ADIM; ++a)
      if (a < adim) dW[(int64_t)a * dim + c] += gw[a];
  }
}

struct AdamScalars {
  float decay, w1, beta2, w2, bc2_sqrt, eps, step_size, grad_scale;
};

// lerp as ATen's vectorised CPU kernel computes it (aten/src/ATen/native/cpu/LerpKernel.cpp):
// fmadd(|w| < 0.5 ? w : w - 1, b - a, |w| < 0.5 ? a : b)
OVLA_DEV float aten_lerp(float a, float b, float w) {
  return fabsf(w) < 0.5f ? __builtin_fmaf(w, b - a, a) : __builtin_fmaf(w - 1.f, b - a, b);
}

__global__ __launch_bounds__(256) void adamw_bf16_kernel(bf16_bits* __restrict__ p, bf16_bits* __restrict__ m, bf16_bits* __restrict__ v,
                                                         const float* __restrict__ g, int64_t n, AdamScalars s) {
#pragma clang fp contract(off)  // keep ATen's operation order and roundings: no fused multiply-adds the reference lacks
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float grad = bfround(g[i] * s.grad_scale);          // what autograd would have stored in the bf16 .grad
    float pp = bfround(bf2f(p[i]) * s.decay);                  // param.mul_(1 - lr*wd)
    const float mm = bfround(aten_lerp(bf2f(m[i]), grad, s.w1));  // exp_avg.lerp_(grad, 1-beta1)
    float vv = bfround(bf2f(v[i]) * s.beta2);                  // exp_avg_sq.mul_(beta2)
    vv = bfround(vv + (s.w2 * grad) * grad);                   //   .addcmul_(grad, grad, value=1-beta2): self + value*t1*t2
    float d = bfround(sqrtf(vv));                              // exp_avg_sq.sqrt()
    d = bfround(d / s.bc2_sqrt);                               //   / bias_correction2_sqrt
    d = bfround(d + s.eps);                                    //   .add_(eps)
    pp = bfround(pp + (s.step_size * mm) / d);                 // param.addcdiv_(exp_avg, denom, value=-step_size): self + value*t1/t2
    p[i] = f2bf(pp);
    m[i] = f2bf(mm);
    v[i] = f2bf(vv);
  }
}

__global__ __launch_bounds__(256) void adamw_f32_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ g, int64_t n, AdamScalars s) {
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float grad = g[i] * s.grad_scale;
    float pp = p[i] * s.decay;
    const float mm = aten_lerp(m[i], grad, s.w1);
    float vv = v[i] * s.beta2;
    vv = vv + (s.w2 * grad) * grad;
    const float d = sqrtf(vv) / s.bc2_sqrt + s.eps;
    pp = pp + (s.step_size * mm) / d;
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}


// One workgroup per row: max / argmax, sum of exponentials, loss, gradient.  (Rows are few -- B * (A + 1) -- and 64 KB each: the
// three passes re-read the row from L2.)
__global__ __launch_bounds__(256) void token_ce_kernel(const bf16_bits* __restrict__ logits, int64_t ld, const int64_t* __restrict__ targets,
                                                       float* __restrict__ loss_rows, int* __restrict__ argmax_out, bf16_bits* dlogits, int64_t ld_d,
                                                       int vocab, float grad_scale) {
  __shared__ float red_f[4];
  __shared__ int red_i[4];
  __shared__ float bc_f[2];
  __shared__ int bc_i;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_bits* x = logits + (int64_t)row * ld;
  int tgt = (int)targets[row];
  tgt = tgt < 0 ? 0 : (tgt >= vocab ? vocab - 1 : tgt);   // rows with an ignored label never get here (the caller gathers); no out-of-bounds read either way
  const float x_tgt = bf2f(x[tgt]);          // read before the barriers below: the gradient pass may overwrite the row in place
  float m = -INFINITY; int mi = 0x7fffffff;
  for (int j = tid; j < vocab; j += 256) {
    const float v = bf2f(x[j]);
    if (v > m) { m = v; mi = j; }            // j ascends per thread: the first maximum wins
  }
  // wave then block reduction of (max, lowest index of it)
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off); const int oi = __shfl_xor(mi, off);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if (lane == 0) { red_f[wave] = m; red_i[wave] = mi; }
  __syncthreads();
  if (tid == 0) {
    float bm = red_f[0]; int bi = red_i[0];
    for (int w = 1; w < 4; ++w) if (red_f[w] > bm || (red_f[w] == bm && red_i[w] < bi)) { bm = red_f[w]; bi = red_i[w]; }
    bc_f[0] = bm; bc_i = bi;
  }
  __syncthreads();
  m = bc_f[0];
  float s = 0.0f;
  for (int j = tid; j < vocab; j += 256) s += __expf(bf2f(x[j]) - m);
  s = wave_sum(s);
  if (lane == 0) red_f[wave] = s;
  __syncthreads();
  if (tid == 0) bc_f[1] = (red_f[0] + red_f[1]) + (red_f[2] + red_f[3]);
  __syncthreads();
  s = bc_f[1];
  if (tid == 0) {
    loss_rows[row] = (__logf(s) + m) - x_tgt;
    argmax_out[row] = bc_i;
  }
  if (dlogits) {
    const float inv = 1.0f / s;
    bf16_bits* d = dlogits + (int64_t)row * ld_d;
    for (int j = tid; j < vocab; j += 256) {
      const float p = __expf(bf2f(x[j]) - m) * inv;
      d[j] = f2bf((p - (j == tgt ? 1.0f : 0.0f)) * grad_scale);
    }
  }
}
}  // namespace

extern "C" int ovla_token_ce(const ovla_token_ce_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->logits && a->targets && a->loss_rows && a->argmax, "ovla_token_ce: null pointer");
  OVLA_REQUIRE(a->rows > 0 && a->vocab > 0 && a->ld >= a->vocab && (!a->dlogits || a->ld_d >= a->vocab), "ovla_token_ce: rows=%d vocab=%d ld=%lld", a->rows, a->vocab, (long long)a->ld);
  hipLaunchKernelGGL(token_ce_kernel, dim3(a->rows), dim3(256), 0, stream, (const bf16_bits*)a->logits, a->ld, a->targets, a->loss_rows, a->argmax,
                     (bf16_bits*)a->dlogits, a->ld_d, a->vocab, a->grad_scale);
  OVLA_CHECK_LAUNCH("ovla_token_ce");
  return OVLA_OK;
}

extern "C" int ovla_head_out_fwd(const ovla_head_out_fwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->W && a->pred, "ovla_head_out_fwd: null pointer");
  OVLA_REQUIRE(a->rows > 0 && a->dim > 0 && (a->dim % 8) == 0 && a->adim > 0 && a->adim <= MAX_ADIM, "ovla_head_out_fwd: rows=%d dim=%d adim=%d", a->rows, a->dim, a->adim);
  OVLA_REQUIRE(aligned16(a->x) && aligned16(a->W), "ovla_head_out_fwd: 16-byte alignment");
  hipLaunchKernelGGL(head_out_fwd_kernel, dim3(a->rows), dim3(256), 0, stream, (const bf16_bits*)a->x, (const bf16_bits*)a->W,
                     (const bf16_bits*)a->b, (bf16_bits*)a->pred, (const bf16_bits*)a->target, a->loss_sum, a->dim, a->adim, a->mse);
  OVLA_CHECK_LAUNCH("ovla_head_out_fwd");
  return OVLA_OK;
}

extern "C" int ovla_head_out_bwd(const ovla_head_out_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->W && a->dx && (a->dpred || (a->pred && a->target)), "ovla_head_out_bwd: null pointer");
  OVLA_REQUIRE(a->rows > 0 && a->dim > 0 && a->adim > 0 && a->adim <= MAX_ADIM, "ovla_head_out_bwd: rows=%d dim=%d adim=%d", a->rows, a->dim, a->adim);
  const size_t lds = (size_t)a->rows * a->adim * sizeof(float);
  OVLA_REQUIRE(lds <= 48 * 1024, "ovla_head_out_bwd: rows*adim too large for the LDS slab");
  hipLaunchKernelGGL(head_out_bwd_kernel, dim3(cdiv(a->dim, 256)), dim3(256), lds, stream, (const bf16_bits*)a->x, (const bf16_bits*)a->W,
                     (const bf16_bits*)a->pred, (const bf16_bits*)a->target, (const bf16_bits*)a->dpred, a->dloss_scale, a->mse, (bf16_bits*)a->dx, a->dW, a->db, a->rows,
                     a->dim, a->adim);
  OVLA_CHECK_LAUNCH("ovla_head_out_bwd");
  return OVLA_OK;
}

extern "C" int ovla_adamw(const ovla_adamw_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->param && a->exp_avg && a->exp_avg_sq && a->grad && a->n > 0 && a->step >= 1, "ovla_adamw: bad arguments");
  // scalars exactly as torch/optim/adamw.py computes them (Python floats = double), then narrowed to the fp32 opmath type
  const double bc1 = 1.0 - pow(a->beta1, (double)a->step);
  const double bc2 = 1.0 - pow(a->beta2, (double)a->step);
  AdamScalars s;
  s.decay = (float)(1.0 - a->lr * a->weight_decay);
  s.w1 = (float)(1.0 - a->beta1);
  s.beta2 = (float)a->beta2;
  s.w2 = (float)(1.0 - a->beta2);
  s.bc2_sqrt = (float)sqrt(bc2);
  s.eps = (float)a->eps;
  s.step_size = (float)(-(a->lr / bc1));
  s.grad_scale = a->grad_scale;
  int64_t blocks = (a->n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (a->is_bf16)
    hipLaunchKernelGGL(adamw_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_bits*)a->param, (bf16_bits*)a->exp_avg,
                       (bf16_bits*)a->exp_avg_sq, a->grad, a->n, s);
  else
    hipLaunchKernelGGL(adamw_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)a->param, (float*)a->exp_avg,
                       (float*)a->exp_avg_sq, a->grad, a->n, s);
  OVLA_CHECK_LAUNCH("ovla_adamw");
  return OVLA_OK;
}
