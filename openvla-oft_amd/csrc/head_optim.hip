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

// ---------------------------------------------------------------------------------------------------------------
// Fused head tail (ovla.h: ovla_head_tail_fwd).
struct HeadTailParams {
  const bf16_bits* x0;
  const bf16_bits *ln_w[2], *ln_b[2], *W[2], *bias[2];
  bf16_bits *hb[2], *zb[2], *xo[2];
  float *mean[2], *rstd[2];
  const bf16_bits *ln2_w, *ln2_b; bf16_bits* h2; float *mean2, *rstd2;
  const bf16_bits *W2, *b2; bf16_bits* pred; const bf16_bits* target; float* loss_sum;
  unsigned* sync;
  int R, rows_real, D, adim, mse, ksplit;
  float eps;
};

// One row of LayerNorm exactly as ovla_norm_fwd computes it (elementwise.hip): widths <= 1536 take norm_fwd_wave_kernel's form (one wave per
// row, the row in registers, one butterfly sum), wider rows norm_fwd_kernel's (256 threads per row, chunks t, t + 256, ..., block_sum_256).
// Writes y[row, :], optionally mean / rstd, and returns the thread's normalised values of chunks t and t + 256 (wide form) for the caller.
OVLA_DEV float head_block_sum(float v, float* red) {   // block_sum_256 of elementwise.hip
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

OVLA_DEV void head_ln_row_wide(const bf16_bits* __restrict__ xr, bf16_bits* __restrict__ yr, const bf16_bits* __restrict__ w, const bf16_bits* __restrict__ b,
                               float* mean_out, float* rstd_out, int D, float eps, float* red) {
  const int nchunk = D >> 3;
  float s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f((bf16_bits)q[j]);
  }
  const float mean = head_block_sum(s, red) / (float)D;
  float ss = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = bf2f((bf16_bits)q[j]) - mean;
      ss = __builtin_fmaf(d, d, ss);
    }
  }
  const float var = head_block_sum(ss, red) / (float)D;
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    if (mean_out) *mean_out = mean;
    if (rstd_out) *rstd_out = rstd;
  }
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
    const bf16x8_bits wq = *reinterpret_cast<const bf16x8_bits*>(w + c * 8), bq = *reinterpret_cast<const bf16x8_bits*>(b + c * 8);
    bf16x8_bits o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(ln_affine(bf2f((bf16_bits)q[j]), mean, rstd, bf2f((bf16_bits)wq[j]), bf2f((bf16_bits)bq[j])));
    *reinterpret_cast<bf16x8_bits*>(yr + c * 8) = o;
  }
}

OVLA_DEV void head_ln_row_wave(const bf16_bits* __restrict__ xr, bf16_bits* __restrict__ yr, const bf16_bits* __restrict__ w, const bf16_bits* __restrict__ b,
                               float* mean_out, float* rstd_out, int D, float eps, int lane) {
  const int nchunk = D >> 3;
  float s = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f((bf16_bits)q[j]);
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = bf2f((bf16_bits)q[j]) - mean;
      ss = __builtin_fmaf(d, d, ss);
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  if (lane == 0) {
    if (mean_out) *mean_out = mean;
    if (rstd_out) *rstd_out = rstd;
  }
  for (int c = lane; c < nchunk; c += 64) {
    const bf16x8_bits q = *reinterpret_cast<const bf16x8_bits*>(xr + c * 8);
    const bf16x8_bits wq = *reinterpret_cast<const bf16x8_bits*>(w + c * 8), bq = *reinterpret_cast<const bf16x8_bits*>(b + c * 8);
    bf16x8_bits o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(ln_affine(bf2f((bf16_bits)q[j]), mean, rstd, bf2f((bf16_bits)wq[j]), bf2f((bf16_bits)bq[j])));
    *reinterpret_cast<bf16x8_bits*>(yr + c * 8) = o;
  }
}

// LayerNorm of all R rows, rows dealt over the grid (a whole workgroup per row for wide rows, a wave per row for narrow ones)
OVLA_DEV void head_ln_rows(const bf16_bits* x, bf16_bits* y, const bf16_bits* w, const bf16_bits* b, float* mean, float* rstd, int R, int D, float eps, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (D > 1536) {
    for (int m = blockIdx.x; m < R; m += gridDim.x)
      head_ln_row_wide(x + (int64_t)m * D, y + (int64_t)m * D, w, b, mean ? mean + m : nullptr, rstd ? rstd + m : nullptr, D, eps, red);
  } else {
    for (int m = blockIdx.x * 4 + wave; m < R; m += gridDim.x * 4)
      head_ln_row_wave(x + (int64_t)m * D, y + (int64_t)m * D, w, b, mean ? mean + m : nullptr, rstd ? rstd + m : nullptr, D, eps, lane);
  }
}

// Grid-wide barrier, bounded: every workgroup's stores so far become visible to every workgroup's loads after it.  false = timed out.
OVLA_DEV bool head_grid_barrier(unsigned* sync, unsigned target, int* flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 21)) {                       // ~ seconds: a grid that is not fully resident ends here instead of hanging
        ok = 0;
        __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *flag = ok;
  }
  __syncthreads();
  return *flag != 0;
}

// Stages, separated by grid barriers:  LN rows of block 0 | GEMM 0 + epilogue | LN rows of block 1 | GEMM 1 + epilogue | LN 2 rows + fc2 + loss.
// LayerNorm phases deal ROWS over the workgroups (each row normalised once, by ovla_norm_fwd's own arithmetic) and write the normalised
// rows hb (the Linear's input, which the backward needs anyway); GEMM phases deal 16-column strips: a workgroup streams its 16 weight rows
// from HBM once and reads the R x dim normalised rows from L2, one wave per 16-row tile.
__global__ __launch_bounds__(256) void head_tail_kernel(const HeadTailParams p) {
  __shared__ float red[4][MAX_ADIM];
  __shared__ int bar_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = p.R, D = p.D;
  // wave -> (16-row tile mt, 16-column strip): with fewer than four row tiles the spare waves take further column strips of the same
  // workgroup (R = 16: four strips per workgroup, grid = dim / 64), so every wave streams weights whatever the batch size
  const int tiles_m = R / 16, strips = (tiles_m == 1) ? 4 : (tiles_m == 2 ? 2 : 1);
  const int mt = wave % tiles_m, ns = wave / tiles_m;
  const bool has_tile = ns < strips;
  const int n0 = (blockIdx.x * strips + ns) * 16;
  const unsigned G = gridDim.x;
  unsigned epoch = 0;
  auto barrier = [&]() -> bool {
    if (head_grid_barrier(p.sync, G * (++epoch), &bar_ok)) return true;
    // A timed-out barrier (the grid was not co-resident) must not look like a success, here or in the backward that reads the saved tensors:
    // every output and every saved buffer is filled with NaN (each workgroup its 16-column strips, workgroup 0 the small ones), and sync[1]
    // -- which no launch ever clears -- stays set until the host reads it (ActionHead.check_fused_tail, called where the step already syncs).
    const bf16_bits qnan = 0x7fc0;
    for (int b = 0; b < 2; ++b) {
      bf16_bits* bufs[3] = {p.hb[b], p.zb[b], p.xo[b]};
      for (int q = 0; q < 3; ++q)
        if (bufs[q])
          for (int i = tid; i < R * 16 * strips; i += 256) bufs[q][(int64_t)(i / (16 * strips)) * D + blockIdx.x * 16 * strips + i % (16 * strips)] = qnan;
    }
    for (int i = tid; i < R * 16 * strips; i += 256) p.h2[(int64_t)(i / (16 * strips)) * D + blockIdx.x * 16 * strips + i % (16 * strips)] = qnan;
    if (blockIdx.x == 0) {
      for (int i = tid; i < p.rows_real * p.adim; i += 256) p.pred[i] = qnan;
      if (tid == 0 && p.loss_sum) *p.loss_sum = __builtin_nanf("");
    }
    return false;
  };
  const bf16_bits* xin = p.x0;
  for (int b = 0; b < 2; ++b) {
    head_ln_rows(xin, p.hb[b], p.ln_w[b], p.ln_b[b], p.mean[b], p.rstd[b], R, D, p.eps, &red[0][0]);
    if (!barrier()) return;
    if (has_tile) {   // rows 16 mt .. +15 x columns n0 .. +15, K = D in `ksplit` halves summed in order
      const int row = mt * 16 + (lane & 15), kq = 8 * (lane >> 4);
      const bf16_bits* hr = p.hb[b] + (int64_t)row * D;
      const bf16_bits* wr = p.W[b] + (int64_t)(n0 + (lane & 15)) * D;
      const int ksteps = D / 32;
      const int half = p.ksplit > 1 ? ((D / 64 + 1) / 2) * 2 : ksteps;       // gemm_nt's split: ceil(T / 2) K tiles of 64 = 2 k-steps each
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      // the two halves are independent accumulation chains: they run interleaved (twice the loads in flight, the matrix pipe never waits
      // on one chain), each in its own k order, so the sums are those of the two split-K workgroups of the unfused GEMM
      const int n_hi = ksteps - half;              // k-steps of the second half (0 when ksplit == 1; <= half)
      // Weight rows and normalised rows stream straight into MFMA fragments (no LDS: every byte is used once per wave).  A register ring
      // keeps PD k-steps of both chains in flight per wave -- 4 x PD 16-byte loads: round 2's loop had 4-8 and sat at 0.07 of the HBM
      // roofline, one memory round trip per 4 KB.  The MFMA order of each chain is unchanged (k ascending), so the sums are bit-identical.
      constexpr int PD = 8;
      bf16x8_bits ra0[PD], rw0[PD], ra1[PD], rw1[PD];
      const int lim = half < ksteps ? half : ksteps;
      auto fetch = [&](int ks, int slot) {
        const int k = ks * 32 + kq;
        ra0[slot] = *reinterpret_cast<const bf16x8_bits*>(hr + k);
        rw0[slot] = *reinterpret_cast<const bf16x8_bits*>(wr + k);
        if (ks < n_hi) {
          ra1[slot] = *reinterpret_cast<const bf16x8_bits*>(hr + k + half * 32);
          rw1[slot] = *reinterpret_cast<const bf16x8_bits*>(wr + k + half * 32);
        }
      };
#pragma unroll
      for (int i = 0; i < PD; ++i)
        if (i < lim) fetch(i, i);
      for (int ks0 = 0; ks0 < lim; ks0 += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
          const int ks = ks0 + sl;
          if (ks < lim) {
            if (ks < n_hi) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rw1[sl], ra1[sl], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rw0[sl], ra0[sl], acc[0], 0, 0, 0);
            if (ks + PD < lim) fetch(ks + PD, sl);
          }
        }
      }
      // epilogue = epilogue_store of gemm_nt.hip after its split-K reduce: (0 + p0) + p1, + bias, round, save z, ReLU, round, + residual, round
      const int n = n0 + 4 * (lane >> 4);
      const f32x4 v = p.ksplit > 1 ? (acc[0] + acc[1]) : acc[0];
      const bf16x4_bits bb = *reinterpret_cast<const bf16x4_bits*>(p.bias[b] + n);
      const bf16x4_bits rr = *reinterpret_cast<const bf16x4_bits*>(xin + (int64_t)row * D + n);
      bf16x4_bits z, o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = bfround(v[j] + bf2f((bf16_bits)bb[j]));
        z[j] = (short)f2bf(t);
        t = bfround(t > 0.f ? t : 0.f);
        o[j] = (short)f2bf(bfround(t + bf2f((bf16_bits)rr[j])));
      }
      if (p.zb[b]) *reinterpret_cast<bf16x4_bits*>(p.zb[b] + (int64_t)row * D + n) = z;
      *reinterpret_cast<bf16x4_bits*>(p.xo[b] + (int64_t)row * D + n) = o;
    }
    if (!barrier()) return;
    xin = p.xo[b];
  }
  // ---- LayerNorm 2 -> h2 (rows over the grid), then fc2 + loss per row (ovla_head_out_fwd's arithmetic on the bf16 h2 the row's owner wrote) ----
  head_ln_rows(xin, p.h2, p.ln2_w, p.ln2_b, p.mean2, p.rstd2, R, D, p.eps, &red[0][0]);
  if (D <= 1536) {   // narrow rows were normalised one per WAVE, possibly by other workgroups: one more barrier before the row dot products
    if (!barrier()) return;
  } else {
    __syncthreads();  // wide rows: row m was written by THIS workgroup (rows are dealt m = blockIdx.x, + G, ... in both loops)
  }
  for (int m = blockIdx.x; m < p.rows_real; m += G) {
    float acc[MAX_ADIM];
#pragma unroll
    for (int a = 0; a < MAX_ADIM; ++a) acc[a] = 0.f;
    for (int c = tid * 8; c < D; c += 256 * 8) {
      const bf16x8_bits xv = *reinterpret_cast<const bf16x8_bits*>(p.h2 + (int64_t)m * D + c);
#pragma unroll
      for (int a = 0; a < MAX_ADIM; ++a) {
        if (a < p.adim) {
          const bf16x8_bits wv = *reinterpret_cast<const bf16x8_bits*>(p.W2 + (int64_t)a * D + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[a] += bf2f((bf16_bits)xv[j]) * bf2f((bf16_bits)wv[j]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MAX_ADIM; ++a) {
      const float sa = wave_sum(acc[a]);
      if (lane == 0) red[wave][a] = sa;
    }
    __syncthreads();
    if (tid < p.adim) {
      const int a = tid;
      float v = red[0][a] + red[1][a] + red[2][a] + red[3][a];
      v = bfround(v + (p.b2 ? bf2f(p.b2[a]) : 0.f));
      p.pred[(int64_t)m * p.adim + a] = f2bf(v);
      if (p.target && p.loss_sum) {
        const float d = bfround(bf2f(p.target[(int64_t)m * p.adim + a]) - v);
        atomicAdd(p.loss_sum, p.mse ? bfround(d * d) : fabsf(d));
      }
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

extern "C" int ovla_head_tail_resident_blocks(void) {
  static int cached = -1;
  if (cached < 0) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_tail_kernel, 256, 0) != hipSuccess)
      return 0;   // no device / query failed: nothing is known to be resident (not cached: a later call may succeed)
    cached = cus * per_cu;
  }
  return cached;
}

extern "C" int ovla_head_tail_fwd(const ovla_head_tail_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x0 && a->xo[0] && a->xo[1] && a->hb[0] && a->hb[1] && a->h2 && a->W2 && a->pred && a->sync && a->ln2_w && a->ln2_b, "ovla_head_tail_fwd: null pointer");
  for (int b = 0; b < 2; ++b)
    OVLA_REQUIRE(a->ln_w[b] && a->ln_b[b] && a->W[b] && a->bias[b], "ovla_head_tail_fwd: block %d needs LayerNorm weight / bias and Linear weight / bias", b);
  OVLA_REQUIRE(a->rows >= 16 && a->rows <= 64 && (a->rows % 16) == 0 && a->rows_real >= 1 && a->rows_real <= a->rows, "ovla_head_tail_fwd: rows=%d (multiple of 16, <= 64) rows_real=%d",
               a->rows, a->rows_real);
  OVLA_REQUIRE(a->dim >= 64 && (a->dim % 64) == 0 && a->dim / 16 <= 256 && (a->dim / 16) % 4 == 0, "ovla_head_tail_fwd: dim=%d must be a multiple of 64 and <= 4096 (one resident workgroup per 16 columns)", a->dim);
  OVLA_REQUIRE(a->adim > 0 && a->adim <= MAX_ADIM && (a->ksplit == 1 || a->ksplit == 2), "ovla_head_tail_fwd: adim=%d ksplit=%d", a->adim, a->ksplit);
  OVLA_REQUIRE(aligned16(a->x0) && aligned16(a->xo[0]) && aligned16(a->xo[1]) && aligned16(a->h2) && aligned16(a->W[0]) && aligned16(a->W[1]) && aligned16(a->W2) &&
               aligned16(a->ln_w[0]) && aligned16(a->ln_b[0]) && aligned16(a->ln_w[1]) && aligned16(a->ln_b[1]) && aligned16(a->ln2_w) && aligned16(a->ln2_b) &&
               (!a->hb[0] || aligned16(a->hb[0])) && (!a->hb[1] || aligned16(a->hb[1])), "ovla_head_tail_fwd: 16-byte alignment");
  OVLA_REQUIRE((((uintptr_t)a->bias[0] | (uintptr_t)a->bias[1]) & 7) == 0 && (!a->zb[0] || aligned16(a->zb[0])) && (!a->zb[1] || aligned16(a->zb[1])), "ovla_head_tail_fwd: bias / zb alignment");
  HeadTailParams p;
  p.x0 = (const bf16_bits*)a->x0;
  for (int b = 0; b < 2; ++b) {
    p.ln_w[b] = (const bf16_bits*)a->ln_w[b]; p.ln_b[b] = (const bf16_bits*)a->ln_b[b]; p.W[b] = (const bf16_bits*)a->W[b]; p.bias[b] = (const bf16_bits*)a->bias[b];
    p.hb[b] = (bf16_bits*)a->hb[b]; p.zb[b] = (bf16_bits*)a->zb[b]; p.xo[b] = (bf16_bits*)a->xo[b]; p.mean[b] = a->mean[b]; p.rstd[b] = a->rstd[b];
  }
  p.ln2_w = (const bf16_bits*)a->ln2_w; p.ln2_b = (const bf16_bits*)a->ln2_b; p.h2 = (bf16_bits*)a->h2; p.mean2 = a->mean2; p.rstd2 = a->rstd2;
  p.W2 = (const bf16_bits*)a->W2; p.b2 = (const bf16_bits*)a->b2; p.pred = (bf16_bits*)a->pred; p.target = (const bf16_bits*)a->target; p.loss_sum = a->loss_sum;
  p.sync = a->sync; p.R = a->rows; p.rows_real = a->rows_real; p.D = a->dim; p.adim = a->adim; p.mse = a->mse; p.ksplit = a->ksplit; p.eps = a->eps;
  const int tiles_m = a->rows / 16, strips = tiles_m == 1 ? 4 : (tiles_m == 2 ? 2 : 1);
  const int grid = a->dim / 16 / strips;
  // the kernel spins on software grid barriers: refuse the launch unless the whole grid can be resident at once on this device
  OVLA_REQUIRE(ovla_head_tail_resident_blocks() >= grid, "ovla_head_tail_fwd: %d workgroups cannot be co-resident (%d fit): use the unfused sequence", grid,
               ovla_head_tail_resident_blocks());
  // sync[0] = arrival counter of THIS launch; sync[1] = sticky timeout flag: never cleared here, the host reads (and clears) it
  if (hipMemsetAsync(a->sync, 0, sizeof(uint32_t), stream) != hipSuccess) { ovla_set_error("ovla_head_tail_fwd: hipMemsetAsync failed"); return OVLA_ELAUNCH; }
  hipLaunchKernelGGL(head_tail_kernel, dim3(grid), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_head_tail_fwd");
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
