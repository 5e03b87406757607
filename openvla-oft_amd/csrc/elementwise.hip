// elementwise.hip -- HBM-bound kernels of the OpenVLA-OFT path: normalisation, RoPE, SwiGLU, activation backward,
// multimodal assembly, row gathers, ViT embedding glue, casts/transposes.  All loads/stores are 8- or 16-byte vectors
// of bf16 (cdna_hip_programming.md Guideline 13); reductions are wave shuffles + one LDS hop.
#include "common.h"

namespace {

OVLA_DEV float block_sum_256(float v, float* red /* >= 4 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

OVLA_DEV void load8(const bf16_bits* p, float (&f)[8]) {
  const bf16x8_bits v = *reinterpret_cast<const bf16x8_bits*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = bf2f((bf16_bits)v[j]);
}
OVLA_DEV void store8(bf16_bits* p, const float (&f)[8]) {
  bf16x8_bits v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(f[j]);
  *reinterpret_cast<bf16x8_bits*>(p) = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Norm forward: one 256-thread workgroup per row; the row is re-read from L2 for the second/third pass.
// RMS (transformers LlamaRMSNorm): y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))
// LN  (torch layer_norm, fp32 math): y = bf16((x - mean) * rstd * w + b)
__global__ __launch_bounds__(256) void norm_fwd_kernel(const bf16_bits* __restrict__ x, bf16_bits* __restrict__ y,
                                                       const bf16_bits* __restrict__ w, const bf16_bits* __restrict__ b,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int dim, float eps, int is_rms) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const bf16_bits* xr = x + row * dim;
  bf16_bits* yr = y + row * dim;
  const int nchunk = dim >> 3;
  float s = 0.f;
  float mean = 0.f;
  if (!is_rms) {
    for (int c = threadIdx.x; c < nchunk; c += 256) {
      float f[8];
      load8(xr + c * 8, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[j];
    }
    mean = block_sum_256(s, red) / (float)dim;
  }
  float ss = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    float f[8];
    load8(xr + c * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = f[j] - mean;
      ss = __builtin_fmaf(d, d, ss);      // explicit fma: the fused head-tail kernel (head_optim.hip) repeats this arithmetic bit for bit
    }
  }
  const float var = block_sum_256(ss, red) / (float)dim;
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    float f[8], wf[8], o[8];
    load8(xr + c * 8, f);
    load8(w + c * 8, wf);
    if (is_rms) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = wf[j] * bfround(f[j] * rstd);
    } else {
      float bfv[8];
      if (b) load8(b + c * 8, bfv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ln_affine(f[j], mean, rstd, wf[j], b ? bfv[j] : 0.f);
    }
    store8(yr + c * 8, o);
  }
}

// Norm backward: g = dy * w; xhat = (x - mean) * rstd
//   RMS: dx = rstd * (g - xhat * mean(g * xhat));   LN: dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
__global__ __launch_bounds__(256) void norm_bwd_kernel(const bf16_bits* __restrict__ x, const bf16_bits* __restrict__ dy,
                                                       const bf16_bits* __restrict__ w, const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in, bf16_bits* __restrict__ dx,
                                                       float* __restrict__ dw, float* __restrict__ db, int dim, int is_rms,
                                                       int dx_accum) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const bf16_bits* xr = x + row * dim;
  const bf16_bits* dyr = dy + row * dim;
  bf16_bits* dxr = dx + row * dim;
  const float mean = is_rms ? 0.f : mean_in[row];
  const float rstd = rstd_in[row];
  const int nchunk = dim >> 3;
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    float f[8], g[8], wf[8];
    load8(xr + c * 8, f);
    load8(dyr + c * 8, g);
    load8(w + c * 8, wf);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gj = g[j] * wf[j];
      s1 += gj;
      s2 += gj * (f[j] - mean) * rstd;
    }
  }
  const float m1 = is_rms ? 0.f : block_sum_256(s1, red) / (float)dim;
  const float m2 = block_sum_256(s2, red) / (float)dim;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    float f[8], g[8], wf[8], o[8];
    load8(xr + c * 8, f);
    load8(dyr + c * 8, g);
    load8(w + c * 8, wf);
    if (dx_accum) load8(dxr + c * 8, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xhat = (f[j] - mean) * rstd;
      const float v = rstd * (g[j] * wf[j] - m1 - xhat * m2);
      o[j] = dx_accum ? o[j] + v : v;
      if (dw) atomicAdd(dw + c * 8 + j, g[j] * xhat);
      if (db) atomicAdd(db + c * 8 + j, g[j]);
    }
    store8(dxr + c * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-per-row variants for dim <= 1536 (every LayerNorm of the two ViTs; at the Llama width the workgroup-per-row kernels above are as
// fast or faster in the step): the row lives in registers (ONE global read instead of three passes), reductions are wave shuffles (no LDS,
// no barriers), 4 rows per 256-thread workgroup.  rocprof, per launch: forward 12.6 -> 7-8.5 us, backward 13.1 -> 8.3-9.7 us.  Same
// arithmetic and rounding points as the kernels above; only the order of the fp32 sums differs.
template <int CPL>   // 16-byte chunks per lane: dim <= 512 * CPL
__global__ __launch_bounds__(256) void norm_fwd_wave_kernel(const bf16_bits* __restrict__ x, bf16_bits* __restrict__ y,
                                                            const bf16_bits* __restrict__ w, const bf16_bits* __restrict__ b,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int dim, float eps, int is_rms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_bits* xr = x + row * dim;
  bf16_bits* yr = y + row * dim;
  const int nchunk = dim >> 3;
  float f[CPL][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      load8(xr + c * 8, f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    }
  }
  const float mean = is_rms ? 0.f : wave_sum(s) / (float)dim;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i)
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = f[i][j] - mean;
        ss = __builtin_fmaf(d, d, ss);
      }
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)dim + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float wf[8], o[8];
      load8(w + c * 8, wf);
      if (is_rms) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wf[j] * bfround(f[i][j] * rstd);
      } else {
        float bfv[8];
        if (b) load8(b + c * 8, bfv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ln_affine(f[i][j], mean, rstd, wf[j], b ? bfv[j] : 0.f);
      }
      store8(yr + c * 8, o);
    }
  }
}

// (A wave-per-row BACKWARD kernel like norm_fwd_wave_kernel existed in round 1.  Beside the second vision tower's kernels on another stream a
// few waves per launch dropped one term of a per-lane running sum in lanes 48-63: the compiler's packed-FP32 code (v_pk_add_f32 fed by
// v_mov_b32) under that co-issue pattern, root-caused in round 2 with tools/norm_bwd_wave_probe.py -- DESIGN.md "Run-to-run determinism".
// The library is now built without packed FP32; the workgroup-per-row norm_bwd_kernel below stays because it costs the same.)

// ---------------------------------------------------------------------------------------------------------------
// RoPE tables and in-place rotation.  HF convention: cos/sin are fp32, cast to bf16; q' = bf16(bf16(q*c) + bf16(rot(q)*s)).
__global__ void rope_table_kernel(bf16_bits* cosT, bf16_bits* sinT, int S, int half, float theta) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * half) return;
  const int pos = idx / half, i = idx % half;
  const float inv_freq = 1.0f / powf(theta, (float)(2 * i) / (float)(2 * half));
  const float ang = (float)pos * inv_freq;
  cosT[idx] = f2bf(cosf(ang));
  sinT[idx] = f2bf(sinf(ang));
}

__global__ __launch_bounds__(256) void rope_kernel(bf16_bits* __restrict__ qk, int64_t ld, int rows, int S, int n_heads,
                                                   int head_dim, const bf16_bits* __restrict__ cosT,
                                                   const bf16_bits* __restrict__ sinT, int inverse) {
  const int half = head_dim >> 1;
  const int cpr = half >> 2;  // 4-element chunks per half head
  const int64_t total = (int64_t)rows * n_heads * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cpr);
    const int h = (int)((idx / cpr) % n_heads);
    const int64_t row = idx / ((int64_t)cpr * n_heads);
    const int pos = (int)(row % S);
    bf16_bits* base = qk + row * ld + (int64_t)h * head_dim + c * 4;
    const bf16x4_bits lo = *reinterpret_cast<const bf16x4_bits*>(base);
    const bf16x4_bits hi = *reinterpret_cast<const bf16x4_bits*>(base + half);
    const bf16x4_bits cs = *reinterpret_cast<const bf16x4_bits*>(cosT + (int64_t)pos * half + c * 4);
    const bf16x4_bits sn = *reinterpret_cast<const bf16x4_bits*>(sinT + (int64_t)pos * half + c * 4);
    bf16x4_bits olo, ohi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf2f((bf16_bits)lo[j]), b = bf2f((bf16_bits)hi[j]);
      const float cc = bf2f((bf16_bits)cs[j]);
      const float s = inverse ? -bf2f((bf16_bits)sn[j]) : bf2f((bf16_bits)sn[j]);
      // forward: lo' = a*c - b*s ; hi' = b*c + a*s     (rotate_half(x) = [-x_hi, x_lo])
      olo[j] = (short)f2bf(bfround(a * cc) + bfround(-b * s));
      ohi[j] = (short)f2bf(bfround(b * cc) + bfround(a * s));
    }
    *reinterpret_cast<bf16x4_bits*>(base) = olo;
    *reinterpret_cast<bf16x4_bits*>(base + half) = ohi;
  }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_bits* __restrict__ gu, bf16_bits* __restrict__ h,
                                                         int rows, int F) {
  const int cpr = F >> 3;
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / cpr;
    const int c = (int)(idx % cpr);
    float g[8], u[8], o[8];
    load8(gu + row * 2 * F + c * 8, g);
    load8(gu + row * 2 * F + F + c * 8, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bfround(silu(g[j])) * u[j];
    store8(h + row * F + c * 8, o);
  }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_bits* __restrict__ gu, const bf16_bits* __restrict__ dh,
                                                         bf16_bits* __restrict__ dgu, int rows, int F) {
  const int cpr = F >> 3;
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / cpr;
    const int c = (int)(idx % cpr);
    float g[8], u[8], d[8], dg[8], du[8];
    load8(gu + row * 2 * F + c * 8, g);
    load8(gu + row * 2 * F + F + c * 8, u);
    load8(dh + row * F + c * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = sigmoidf_(g[j]);
      const float sl = g[j] * s;
      du[j] = d[j] * bfround(sl);
      dg[j] = d[j] * u[j] * (s * (1.f + g[j] * (1.f - s)));
    }
    store8(dgu + row * 2 * F + c * 8, dg);
    store8(dgu + row * 2 * F + F + c * 8, du);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_bits* __restrict__ z, const bf16_bits* __restrict__ dh,
                                                      bf16_bits* __restrict__ dz, int64_t nchunk, int act) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    float zf[8], d[8], o[8];
    load8(z + c * 8, zf);
    load8(dh + c * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = d[j] * act_grad(zf[j], act);
    store8(dz + c * 8, o);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_bits* __restrict__ a, const bf16_bits* __restrict__ b,
                                                  bf16_bits* __restrict__ out, int64_t nchunk) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    float x[8], y[8];
    load8(a + c * 8, x);
    if (b) {
      load8(b + c * 8, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += y[j];
    }
    store8(out + c * 8, x);
  }
}

__global__ __launch_bounds__(256) void colscale_kernel(const bf16_bits* __restrict__ x, const bf16_bits* __restrict__ s,
                                                       bf16_bits* __restrict__ out, int rows, int dim) {
  const int cpr = dim >> 3;
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cpr);
    float f[8], sf[8];
    load8(x + idx * 8, f);
    load8(s + c * 8, sf);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= sf[j];
    store8(out + idx * 8, f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// im2col for patch embedding: out[((b, img), gy, gx), (c, py, px)], K padded with zeros up to ldo.  Image `img` of
// batch row b lives in channels [c0 + img*img_cstride, +3) of the channel-stacked pixel tensor.
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_bits* __restrict__ px, bf16_bits* __restrict__ out, int64_t ldo,
                                                     int B, int C_total, int c0, int H, int W, int patch, int n_img, int img_cstride) {
  const int gh = H / patch, gw = W / patch;
  const int64_t total = (int64_t)B * n_img * gh * gw * ldo;
  const int kreal = 3 * patch * patch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % ldo);
    const int64_t r = idx / ldo;
    bf16_bits v = 0;
    if (k < kreal) {
      const int c = k / (patch * patch), py = (k / patch) % patch, pxx = k % patch;
      const int gx = (int)(r % gw), gy = (int)((r / gw) % gh);
      const int64_t bi = r / ((int64_t)gw * gh);
      const int64_t b = bi / n_img;
      const int img = (int)(bi % n_img);
      v = px[((b * C_total + c0 + img * img_cstride + c) * H + gy * patch + py) * (int64_t)W + gx * patch + pxx];
    }
    out[idx] = v;
  }
}

__global__ __launch_bounds__(256) void vit_embed_kernel(const bf16_bits* __restrict__ patches, const bf16_bits* __restrict__ pos,
                                                        const bf16_bits* __restrict__ prefix, bf16_bits* __restrict__ tokens,
                                                        int B, int n_patches, int n_prefix, int dim) {
  const int cpr = dim >> 3;
  const int ntok = n_patches + n_prefix;
  const int64_t total = (int64_t)B * ntok * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cpr);
    const int t = (int)((idx / cpr) % ntok);
    const int64_t b = idx / ((int64_t)cpr * ntok);
    float f[8];
    if (t < n_prefix) {
      load8(prefix + (int64_t)t * dim + c * 8, f);
    } else {
      float pf[8];
      load8(patches + ((b * n_patches) + (t - n_prefix)) * (int64_t)dim + c * 8, f);
      load8(pos + (int64_t)(t - n_prefix) * dim + c * 8, pf);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += pf[j];
    }
    store8(tokens + idx * 8, f);
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_bits* __restrict__ src, bf16_bits* __restrict__ dst, int B,
                                                        int rows, int dim, int64_t sbs, int64_t sr0, int64_t sld, int64_t dbs,
                                                        int64_t dr0, int64_t dld, int64_t dc0, int accumulate) {
  const int cpr = dim >> 3;
  const int64_t total = (int64_t)B * rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cpr);
    const int r = (int)((idx / cpr) % rows);
    const int64_t b = idx / ((int64_t)cpr * rows);
    float f[8];
    load8(src + b * sbs + (sr0 + r) * sld + c * 8, f);
    bf16_bits* d = dst + b * dbs + (dr0 + r) * dld + dc0 + c * 8;
    if (accumulate) {
      float g[8];
      load8(d, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j];
    }
    store8(d, f);
  }
}

// out[b, :] = mean over rows i with row_mask[b,i] != 0 of x[b, i, :]   (FiLM average language embedding)
__global__ __launch_bounds__(256) void masked_mean_kernel(const bf16_bits* __restrict__ x, const uint8_t* __restrict__ mask,
                                                          bf16_bits* __restrict__ out, int B, int L, int dim) {
  const int b = blockIdx.y;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= dim) return;
  float s = 0.f;
  int cnt = 0;
  for (int i = 0; i < L; ++i) {
    if (mask[(int64_t)b * L + i]) {
      s += bf2f(x[((int64_t)b * L + i) * dim + col]);
      ++cnt;
    }
  }
  out[(int64_t)b * dim + col] = f2bf(s / (float)(cnt > 0 ? cnt : 1));
}

// out[b, :] = mean over text positions i with labels[b,i] <= action_token_begin of embed[ids[b,i], :]   (FiLM conditioning vector)
__global__ __launch_bounds__(256) void language_average_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ labels,
                                                               const bf16_bits* __restrict__ embed, bf16_bits* __restrict__ out, int L, int D,
                                                               int vocab, int64_t action_token_begin) {
  const int b = blockIdx.y;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= D) return;
  float s = 0.f;
  int cnt = 0;
  for (int i = 0; i < L; ++i) {
    if (labels[(int64_t)b * L + i] <= action_token_begin) {
      int64_t id = ids[(int64_t)b * L + i];
      id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);      // same clamp as the assembly kernel: never read outside the table
      s += bf2f(embed[id * D + col]);
      ++cnt;
    }
  }
  out[(int64_t)b * D + col] = f2bf(s / (float)(cnt > 0 ? cnt : 1));
}

// FiLM backward: one thread per (batch, column), walking the batch's rows (coalesced across the 256 columns of a block).
__global__ __launch_bounds__(256) void film_bwd_kernel(bf16_bits* __restrict__ dy, const bf16_bits* __restrict__ x_pre,
                                                       const bf16_bits* __restrict__ gamma, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int rows, int dim, int row_chunk) {
  const int b = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
  if (col >= dim) return;
  const int r0 = blockIdx.z * row_chunk, r1 = (r0 + row_chunk) < rows ? (r0 + row_chunk) : rows;
  const float one_plus = bfround(1.0f + bf2f(gamma[(int64_t)b * dim + col]));
  float sg = 0.f, sb = 0.f;
  for (int r = r0; r < r1; ++r) {
    const int64_t idx = ((int64_t)b * rows + r) * dim + col;
    const float d = bf2f(dy[idx]);
    sg += d * bf2f(x_pre[idx]);
    sb += d;
    dy[idx] = f2bf(d * one_plus);
  }
  atomicAdd(dgamma + (int64_t)b * dim + col, sg);
  atomicAdd(dbeta + (int64_t)b * dim + col, sb);
}

// ---------------------------------------------------------------------------------------------------------------
// Multimodal assembly.  One workgroup per output row (b, s).  The action mask follows train_utils.py:8-39 exactly:
// cumsum(labels != IGNORE) >= 1 and labels > ACTION_TOKEN_BEGIN_IDX.
__global__ __launch_bounds__(256) void assemble_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ labels,
                                                       const bf16_bits* __restrict__ table, const bf16_bits* __restrict__ patches,
                                                       const bf16_bits* __restrict__ noisy, bf16_bits* __restrict__ out,
                                                       int32_t* __restrict__ action_pos, int B, int L, int P, int D, int A,
                                                       int vocab, int ignore_index, int action_begin) {
  __shared__ int sh_is_action, sh_slot;
  const int S = P + L;
  const int b = blockIdx.x / S, s = blockIdx.x % S;
  bf16_bits* o = out + ((int64_t)b * S + s) * D;
  const bf16_bits* src = nullptr;
  bool zero = false;
  if (s >= 1 && s <= P) {
    src = patches + ((int64_t)b * P + (s - 1)) * D;
  } else {
    const int i = (s == 0) ? 0 : s - P;  // text index
    if (threadIdx.x == 0) {
      const int64_t* lab = labels + (int64_t)b * L;
      int cum = 0, slot = 0, is_action = 0;
      for (int j = 0; j <= i; ++j) {
        cum += (lab[j] != ignore_index) ? 1 : 0;
        const int act = (cum >= 1 && lab[j] > action_begin) ? 1 : 0;
        if (j < i) slot += act; else is_action = act;
      }
      sh_is_action = is_action;
      sh_slot = slot;
      if (is_action && action_pos && slot < A) action_pos[(int64_t)b * A + slot] = b * S + P + i - 1;  // row whose hidden state predicts this slot
    }
    __syncthreads();
    if (sh_is_action) {
      if (noisy && sh_slot < A) src = noisy + ((int64_t)b * A + sh_slot) * D;
      else zero = true;
    } else {
      int64_t id = ids[(int64_t)b * L + i];
      if (id < 0) id = 0;
      if (id >= vocab) id = vocab - 1;
      src = table + id * D;
    }
  }
  for (int c = threadIdx.x; c < (D >> 3); c += 256) {
    bf16x8_bits v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!zero) v = *reinterpret_cast<const bf16x8_bits*>(src + c * 8);
    *reinterpret_cast<bf16x8_bits*>(o + c * 8) = v;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_bits* __restrict__ src, const int32_t* __restrict__ index,
                                                          bf16_bits* __restrict__ dst, int n, int dim, int64_t src_ld,
                                                          int64_t dst_ld, int scatter_add) {
  const int i = blockIdx.x;
  const int64_t r = index[i];
  for (int c = threadIdx.x; c < (dim >> 3); c += 256) {
    if (!scatter_add) {
      *reinterpret_cast<bf16x8_bits*>(dst + (int64_t)i * dst_ld + c * 8) =
          *reinterpret_cast<const bf16x8_bits*>(src + r * src_ld + c * 8);
    } else {  // dst[index[i]] += src[i]   (indices are unique per launch on this path)
      float a[8], bb[8];
      load8(src + (int64_t)i * src_ld + c * 8, a);
      load8(dst + r * dst_ld + c * 8, bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += bb[j];
      store8(dst + r * dst_ld + c * 8, a);
    }
  }
}

__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float* __restrict__ src, bf16_bits* __restrict__ dst, int64_t n,
                                                           float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f2bf(src[i] * scale);
}
__global__ __launch_bounds__(256) void cvt_bf16_f32_kernel(const bf16_bits* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                           float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = bf2f(src[i]) * scale;
}

// 64x64 tile transpose through LDS (padded rows: no bank conflicts)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_bits* __restrict__ src, bf16_bits* __restrict__ dst, int rows,
                                                        int cols, int64_t lds_, int64_t ldd) {
  __shared__ bf16_bits tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(int64_t)(r0 + r) * lds_ + c0 + c] : (bf16_bits)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) dst[(int64_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
  }
}

// One block per 64x64 tile of ANY entry: `tile_start[i]` (exclusive prefix sum of the entries' tile counts, n+1 values,
// stored right after the descriptor table) locates the entry by binary search.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const ovla_transpose_args* __restrict__ table, const int* __restrict__ tile_start,
                                                                int n) {
  __shared__ bf16_bits tile[64][66];
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const ovla_transpose_args d = table[lo];
  const int t = b - tile_start[lo];
  const int tiles_x = (d.cols + 63) / 64;
  const int r0 = (t / tiles_x) * 64, c0 = (t % tiles_x) * 64;
  const bf16_bits* src = (const bf16_bits*)d.src;
  bf16_bits* dst = (bf16_bits*)d.dst;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < d.rows && c0 + c < d.cols) ? src[(int64_t)(r0 + r) * d.lds + c0 + c] : (bf16_bits)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < d.rows && c0 + c < d.cols) dst[(int64_t)(c0 + c) * d.ldd + r0 + r] = tile[r][c];
  }
}

// column sums: out[n] += sum_m X[m,n]; grid (N/64 col blocks, row chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_bits* __restrict__ X, int64_t ldx, float* __restrict__ out, int M,
                                                     int N, int rows_per_block) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = (r0 + rows_per_block) < M ? (r0 + rows_per_block) : M;
  float s = 0.f;
  if (col < N)
    for (int r = r0 + sub; r < r1; r += 4) s += bf2f(X[(int64_t)r * ldx + col]);
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && col < N) atomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// image preparation (center crop via bilinear crop_and_resize -> uint8 -> dual normalisation), one thread per output pixel.
// Arithmetic mirrors the numpy restatement operation by operation (each fp32 op rounded on its own).
namespace {
struct ImagePrepParams { const uint8_t* src; bf16_bits* dst; int n_img, H, W, out, crop; float ybase, ystep, xbase, xstep; float mean[6], stdv[6]; };

__global__ __launch_bounds__(256) void image_prep_kernel(const ImagePrepParams p) {
#pragma clang fp contract(off)  // one rounding per multiply / add / subtract, as numpy and TF's kernel do (plain operators: the
                                // pragma does not reach HIP's inlined __fmul_rn & co.)
  const int64_t total = (int64_t)p.n_img * p.out * p.out;
  const float inv255 = 1.0f / 255.0f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % p.out), oy = (int)((idx / p.out) % p.out), img = (int)(idx / ((int64_t)p.out * p.out));
    const uint8_t* im = p.src + (int64_t)img * p.H * p.W * 3;
    uint8_t q[3];
    if (p.crop) {
      const float ty = (float)oy * p.ystep, tx = (float)ox * p.xstep;
      const float ys = p.ybase + ty, xs = p.xbase + tx;
      const float fy = floorf(ys), fx = floorf(xs);
      const int y0 = (int)fy, x0 = (int)fx, y1 = (int)ceilf(ys), x1 = (int)ceilf(xs);
      const float wy = ys - fy, wx = xs - fx;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float tl = (float)im[((int64_t)y0 * p.W + x0) * 3 + c] * inv255, tr = (float)im[((int64_t)y0 * p.W + x1) * 3 + c] * inv255;
        const float bl = (float)im[((int64_t)y1 * p.W + x0) * 3 + c] * inv255, br = (float)im[((int64_t)y1 * p.W + x1) * 3 + c] * inv255;
        const float dt = tr - tl, db = br - bl;
        const float pt = dt * wx, pb = db * wx;
        const float top = tl + pt, bot = bl + pb;
        const float dv = bot - top;
        const float pv = dv * wy;
        float v = top + pv;
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        const float scaled = v * 255.5f;
        q[c] = (uint8_t)scaled;                // truncation: TF convert_image_dtype(float -> uint8, saturate)
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) q[c] = im[((int64_t)oy * p.W + ox) * 3 + c];
    }
    const int64_t plane = (int64_t)p.out * p.out;
    bf16_bits* o = p.dst + (int64_t)img * 6 * plane + (int64_t)oy * p.out + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = (float)q[c] / 255.0f;
      const float d0 = x - p.mean[c], d1 = x - p.mean[3 + c];
      o[c * plane] = f2bf(d0 / p.stdv[c]);
      o[(3 + c) * plane] = f2bf(d1 / p.stdv[3 + c]);
    }
  }
}

// ---- training-time augmentation (ovla.h: ovla_image_augment) --------------------------------------------------------------------------
constexpr int AUG_BLOCKS = 49;   // blocks per image in pass A (1024 pixels each at 224 x 224); partial sums are combined in this fixed order
struct ImageAugParams {
  const uint8_t* src; bf16_bits* dst; const float* params; float* tmp; double* partial;
  int n_img, H, W, out, mask; float mean[6], stdv[6];
};

// pass A: u8 -> float, crop-and-resize, brightness; fp32 intermediate + per-block channel sums (fp64, fixed tree order)
__global__ __launch_bounds__(256) void image_aug_a_kernel(const ImageAugParams p) {
#pragma clang fp contract(off)
  __shared__ double red[3][256];
  const int img = blockIdx.y, blk = blockIdx.x;
  const int npix = p.out * p.out, per_blk = (npix + AUG_BLOCKS - 1) / AUG_BLOCKS;
  const float* prm = p.params + (int64_t)img * 8;
  const uint8_t* im = p.src + (int64_t)img * p.H * p.W * 3;
  const float y1 = prm[0], x1 = prm[1], y2 = prm[2], x2 = prm[3], bdelta = prm[4];
  const float hy = (float)(p.H - 1), hx = (float)(p.W - 1), od = (float)(p.out - 1);
  const float sy0 = y2 - y1, sx0 = x2 - x1;
  const float sy1 = sy0 * hy, sx1 = sx0 * hx;
  const float ystep = sy1 / od, xstep = sx1 / od;
  const float ybase = y1 * hy, xbase = x1 * hx;
  double acc[3] = {0.0, 0.0, 0.0};
  const int end = (blk + 1) * per_blk < npix ? (blk + 1) * per_blk : npix;
  for (int pix = blk * per_blk + threadIdx.x; pix < end; pix += 256) {
    const int oy = pix / p.out, ox = pix - oy * p.out;
    float v[3];
    if (p.mask & 1) {
      const float ty = (float)oy * ystep, tx = (float)ox * xstep;
      const float ys = ybase + ty, xs = xbase + tx;
      if (ys < 0.0f || ys > hy || xs < 0.0f || xs > hx) {
        v[0] = v[1] = v[2] = 0.0f;    // extrapolation_value
      } else {
        const float fy = floorf(ys), fx = floorf(xs);
        const int y0 = (int)fy, x0 = (int)fx, yb = (int)ceilf(ys), xr = (int)ceilf(xs);
        const float wy = ys - fy, wx = xs - fx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float tl = (float)im[((int64_t)y0 * p.W + x0) * 3 + c] / 255.0f, tr = (float)im[((int64_t)y0 * p.W + xr) * 3 + c] / 255.0f;
          const float bl = (float)im[((int64_t)yb * p.W + x0) * 3 + c] / 255.0f, br = (float)im[((int64_t)yb * p.W + xr) * 3 + c] / 255.0f;
          const float dt = tr - tl, db = br - bl;
          const float pt = dt * wx, pb = db * wx;
          const float top = tl + pt, bot = bl + pb;
          const float dv = bot - top;
          const float pv = dv * wy;
          v[c] = top + pv;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = v[c] < 0.0f ? 0.0f : (v[c] > 1.0f ? 1.0f : v[c]);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)im[((int64_t)oy * p.W + ox) * 3 + c] / 255.0f;
    }
    if (p.mask & 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float t = v[c] + bdelta;
        v[c] = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
      }
    }
    float* o = p.tmp + ((int64_t)img * npix + pix) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = v[c]; acc[c] += (double)v[c]; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) p.partial[((int64_t)img * AUG_BLOCKS + blk) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// adjust_saturation_op.cc (CPU kernel): internal::rgb_to_hsv / hsv_to_rgb, operation for operation (its `2.0 / 6.0` are double constants)
__device__ __forceinline__ void tf_rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
#pragma clang fp contract(off)
  const float vv = fmaxf(r, fmaxf(g, b));
  const float range = vv - fminf(r, fminf(g, b));
  s = vv > 0.0f ? range / vv : 0.0f;
  const float six_range = 6.0f * range;
  const float norm = 1.0f / six_range;
  float hh;
  if (r == vv) { const float d = g - b; hh = norm * d; }
  else if (g == vv) { const float d = b - r; const float t = norm * d; hh = (float)((double)t + 2.0 / 6.0); }
  else { const float d = r - g; const float t = norm * d; hh = (float)((double)t + 4.0 / 6.0); }
  if (range <= 0.0f) hh = 0.0f;
  if (hh < 0.0f) hh = hh + 1.0f;
  v = vv; h = hh;
}
__device__ __forceinline__ void tf_hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
#pragma clang fp contract(off)
  const float c = s * v;
  const float m = v - c;
  const float dh = h * 6.0f;
  const int cat = (int)dh;
  float fmodu = dh;
  while (fmodu <= 0.0f) fmodu += 2.0f;
  while (fmodu >= 2.0f) fmodu -= 2.0f;
  const float a1 = fmodu - 1.0f;
  const float a2 = 1.0f - fabsf(a1);
  const float x = c * a2;
  float rr = 0.0f, gg = 0.0f, bb = 0.0f;
  switch (cat) {
    case 0: rr = c; gg = x; break;
    case 1: rr = x; gg = c; break;
    case 2: gg = c; bb = x; break;
    case 3: gg = x; bb = c; break;
    case 4: rr = x; bb = c; break;
    case 5: rr = c; bb = x; break;
    default: break;
  }
  r = rr + m; g = gg + m; b = bb + m;
}
// adjust_hue_op.cc (CPU kernel): rgb_to_hv_range / hv_range_to_rgb
__device__ __forceinline__ void tf_adjust_hue(float& r, float& g, float& b, float delta_h) {
#pragma clang fp contract(off)
  float v_min, v_mid, v_max; int cat;
  if (r < g) {
    if (b < r) { v_max = g; v_mid = r; v_min = b; cat = 1; }
    else if (b > g) { v_max = b; v_mid = g; v_min = r; cat = 3; }
    else { v_max = g; v_mid = b; v_min = r; cat = 2; }
  } else {
    if (b < g) { v_max = r; v_mid = g; v_min = b; cat = 0; }
    else if (b > r) { v_max = b; v_mid = r; v_min = g; cat = 4; }
    else { v_max = r; v_mid = b; v_min = g; cat = 5; }
  }
  float h;
  if (v_max == v_min) {
    h = 0.0f;
  } else {
    const float num = v_mid - v_min, den = v_max - v_min;
    const float ratio = num / den;
    const float one_m = 1.0f - ratio;
    h = (float)cat + ((cat & 1) == 0 ? ratio : one_m);
  }
  const float dd = delta_h * 6.0f;
  h = h + dd;
  while (h < 0.0f) h += 6.0f;
  while (h >= 6.0f) h -= 6.0f;
  const int c2 = (int)h;
  float ratio = h - (float)c2;
  if ((c2 & 1) != 0) ratio = 1.0f - ratio;
  const float span = v_max - v_min;
  const float pr = ratio * span;
  const float mid = v_min + pr;
  switch (c2) {
    case 0: r = v_max; g = mid; b = v_min; break;
    case 1: r = mid; g = v_max; b = v_min; break;
    case 2: r = v_min; g = v_max; b = mid; break;
    case 3: r = v_min; g = mid; b = v_max; break;
    case 4: r = mid; g = v_min; b = v_max; break;
    default: r = v_max; g = v_min; b = mid; break;
  }
}

// pass B: contrast (needs the image mean), saturation, hue, uint8 re-quantisation, both normalisations
__global__ __launch_bounds__(256) void image_aug_b_kernel(const ImageAugParams p) {
#pragma clang fp contract(off)
  __shared__ float mean_s[3];
  const int img = blockIdx.y;
  const int npix = p.out * p.out;
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < AUG_BLOCKS; ++k) s += p.partial[((int64_t)img * AUG_BLOCKS + k) * 3 + threadIdx.x];
    mean_s[threadIdx.x] = (float)(s / (double)npix);
  }
  __syncthreads();
  const float* prm = p.params + (int64_t)img * 8;
  const float cf = prm[5], sf = prm[6], hd = prm[7];
  const int64_t plane = (int64_t)npix;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npix; pix += gridDim.x * 256) {
    const float* t = p.tmp + ((int64_t)img * npix + pix) * 3;
    float v[3] = {t[0], t[1], t[2]};
    if (p.mask & 4) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = v[c] - mean_s[c];
        const float e = d * cf;
        const float f = e + mean_s[c];
        v[c] = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
      }
    }
    if (p.mask & 8) {
      float h, s, vv;
      tf_rgb_to_hsv(v[0], v[1], v[2], h, s, vv);
      const float s2 = s * sf;
      s = fminf(1.0f, fmaxf(0.0f, s2));
      tf_hsv_to_rgb(h, s, vv, v[0], v[1], v[2]);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = v[c] < 0.0f ? 0.0f : (v[c] > 1.0f ? 1.0f : v[c]);
    }
    if (p.mask & 16) {
      tf_adjust_hue(v[0], v[1], v[2], hd);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = v[c] < 0.0f ? 0.0f : (v[c] > 1.0f ? 1.0f : v[c]);
    }
    bf16_bits* o = p.dst + (int64_t)img * 6 * plane + pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float scaled = v[c] * 255.0f;
      const uint8_t q = (uint8_t)scaled;                  // tf.cast(image * 255, tf.uint8): truncation
      const float x = (float)q / 255.0f;
      const float d0 = x - p.mean[c], d1 = x - p.mean[3 + c];
      o[c * plane] = f2bf(d0 / p.stdv[c]);
      o[(3 + c) * plane] = f2bf(d1 / p.stdv[3 + c]);
    }
  }
}
}  // namespace

namespace {
struct ImageResizeParams {
  const uint8_t* src; uint8_t* dst; float* tmp; const int* row_starts; const float* row_weights; const int* col_starts; const float* col_weights;
  int n_img, H, W, out_h, out_w, row_span, col_span;
};
__global__ __launch_bounds__(256) void image_resize_rows_kernel(const ImageResizeParams p) {
#pragma clang fp contract(off)
  const int64_t row_elems = (int64_t)p.W * 3, total = (int64_t)p.n_img * p.out_h * row_elems;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int e = (int)(idx % row_elems), y = (int)((idx / row_elems) % p.out_h), img = (int)(idx / (row_elems * p.out_h));
    const int st = p.row_starts[y];
    const int real = (st + p.row_span < p.H ? st + p.row_span : p.H) - st;
    const uint8_t* in = p.src + ((int64_t)img * p.H + st) * row_elems + e;
    const float* w = p.row_weights + (int64_t)y * p.row_span;
    float acc = 0.0f;
    for (int k = 0; k < real; ++k) {
      const float t = (float)in[(int64_t)k * row_elems] * w[k];
      acc = acc + t;
    }
    p.tmp[idx] = acc;
  }
}
__global__ __launch_bounds__(256) void image_resize_cols_kernel(const ImageResizeParams p) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)p.n_img * p.out_h * p.out_w * 3;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % 3), x = (int)((idx / 3) % p.out_w);
    const int64_t row = idx / ((int64_t)3 * p.out_w);     // img * out_h + y
    const int st = p.col_starts[x];
    const int real = (st + p.col_span < p.W ? st + p.col_span : p.W) - st;
    const float* in = p.tmp + (row * p.W + st) * 3 + c;
    const float* w = p.col_weights + (int64_t)x * p.col_span;
    float acc = 0.0f;
    for (int k = 0; k < real; ++k) {
      const float t = in[k * 3] * w[k];
      acc = acc + t;
    }
    float r = rintf(acc);                                   // tf.round: half to even
    r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
    p.dst[idx] = (uint8_t)r;
  }
}
}  // namespace

extern "C" int64_t ovla_image_resize_workspace_bytes(int32_t n_img, int32_t W, int32_t out_h) {
  return (n_img > 0 && W > 0 && out_h > 0) ? (int64_t)n_img * out_h * W * 3 * 4 : 0;
}

extern "C" int ovla_image_resize(const ovla_image_resize_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->workspace && a->row_starts && a->row_weights && a->col_starts && a->col_weights, "ovla_image_resize: null pointer");
  OVLA_REQUIRE(a->n_img > 0 && a->H > 0 && a->W > 0 && a->out_h > 0 && a->out_w > 0, "ovla_image_resize: bad shape %d x %d x %d -> %d x %d", a->n_img, a->H, a->W, a->out_h, a->out_w);
  OVLA_REQUIRE(a->row_span > 0 && a->row_span <= a->H && a->col_span > 0 && a->col_span <= a->W, "ovla_image_resize: span sizes %d / %d exceed the image", a->row_span, a->col_span);
  OVLA_REQUIRE(a->workspace_bytes >= ovla_image_resize_workspace_bytes(a->n_img, a->W, a->out_h), "ovla_image_resize: needs a workspace of %lld bytes",
               (long long)ovla_image_resize_workspace_bytes(a->n_img, a->W, a->out_h));
  ImageResizeParams p;
  p.src = (const uint8_t*)a->src; p.dst = (uint8_t*)a->dst; p.tmp = (float*)a->workspace;
  p.row_starts = a->row_starts; p.row_weights = a->row_weights; p.col_starts = a->col_starts; p.col_weights = a->col_weights;
  p.n_img = a->n_img; p.H = a->H; p.W = a->W; p.out_h = a->out_h; p.out_w = a->out_w; p.row_span = a->row_span; p.col_span = a->col_span;
  hipLaunchKernelGGL(image_resize_rows_kernel, dim3(grid_for((int64_t)a->n_img * a->out_h * a->W * 3)), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_image_resize(rows)");
  hipLaunchKernelGGL(image_resize_cols_kernel, dim3(grid_for((int64_t)a->n_img * a->out_h * a->out_w * 3)), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_image_resize(cols)");
  return OVLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// JPEG encode -> decode round trip (ovla.h: ovla_jpeg_roundtrip).  libjpeg-turbo's arithmetic, restated: jccolor.c / jcsample.c /
// jfdctint.c / jcdctmgr.c / jidctint.c / jdsample.c / jdcolor.c (published algorithms; the library is not in the reference tree).
namespace {
struct JpegParams {
  const uint8_t* src; uint8_t* dst; uint8_t* planes;     // planes: Y [n, H16, W16] | Cb [n, H16/2, W16/2] | Cr [n, H16/2, W16/2]
  int n_img, H, W, H16, W16;
  int16_t ql[64], qc[64];                                 // quantisation tables, natural order
};
constexpr int J_CONST_BITS = 13, J_PASS1_BITS = 2;
constexpr int JF_0_298631336 = 2446, JF_0_390180644 = 3196, JF_0_541196100 = 4433, JF_0_765366865 = 6270, JF_0_899976223 = 7373, JF_1_175875602 = 9633,
              JF_1_501321110 = 12299, JF_1_847759065 = 15137, JF_1_961570560 = 16069, JF_2_053119869 = 16819, JF_2_562915447 = 20995, JF_3_072711026 = 25172;
OVLA_DEV int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
constexpr int jfix(double x) { return (int)(x * 65536.0 + 0.5); }

OVLA_DEV void jpeg_ycc(const uint8_t* px, int& y, int& cb, int& cr) {       // jccolor.c rgb_ycc_convert
  const int r = px[0], g = px[1], b = px[2];
  y = (jfix(0.29900) * r + jfix(0.58700) * g + jfix(0.11400) * b + 32768) >> 16;
  cb = (-jfix(0.16874) * r - jfix(0.33126) * g + jfix(0.50000) * b + (128 << 16) + 32767) >> 16;
  cr = (jfix(0.50000) * r - jfix(0.41869) * g - jfix(0.08131) * b + (128 << 16) + 32767) >> 16;
}

// one 1-D pass of jfdctint.c over 8 values with stride `st`
OVLA_DEV void jpeg_fdct8(int* d, int st, bool first) {
  const int d0 = d[0], d1 = d[st], d2 = d[2 * st], d3 = d[3 * st], d4 = d[4 * st], d5 = d[5 * st], d6 = d[6 * st], d7 = d[7 * st];
  const int tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  const int sh = first ? J_CONST_BITS - J_PASS1_BITS : J_CONST_BITS + J_PASS1_BITS;
  d[0] = first ? (tmp10 + tmp11) << J_PASS1_BITS : jdescale(tmp10 + tmp11, J_PASS1_BITS);
  d[4 * st] = first ? (tmp10 - tmp11) << J_PASS1_BITS : jdescale(tmp10 - tmp11, J_PASS1_BITS);
  int z1 = (tmp12 + tmp13) * JF_0_541196100;
  d[2 * st] = jdescale(z1 + tmp13 * JF_0_765366865, sh);
  d[6 * st] = jdescale(z1 + tmp12 * (-JF_1_847759065), sh);
  z1 = tmp4 + tmp7;
  int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
  const int z5 = (z3 + z4) * JF_1_175875602;
  const int t4 = tmp4 * JF_0_298631336, t5 = tmp5 * JF_2_053119869, t6 = tmp6 * JF_3_072711026, t7 = tmp7 * JF_1_501321110;
  z1 *= -JF_0_899976223; z2 *= -JF_2_562915447; z3 = z3 * (-JF_1_961570560) + z5; z4 = z4 * (-JF_0_390180644) + z5;
  d[7 * st] = jdescale(t4 + z1 + z3, sh);
  d[5 * st] = jdescale(t5 + z2 + z4, sh);
  d[3 * st] = jdescale(t6 + z2 + z3, sh);
  d[st] = jdescale(t7 + z1 + z4, sh);
}

// one 1-D pass of jidctint.c
OVLA_DEV void jpeg_idct8(int* d, int st, bool first) {
  const int i0 = d[0], i1 = d[st], i2 = d[2 * st], i3 = d[3 * st], i4 = d[4 * st], i5 = d[5 * st], i6 = d[6 * st], i7 = d[7 * st];
  int z1 = (i2 + i6) * JF_0_541196100;
  const int tmp2 = z1 + i6 * (-JF_1_847759065), tmp3 = z1 + i2 * JF_0_765366865;
  const int tmp0 = (i0 + i4) << J_CONST_BITS, tmp1 = (i0 - i4) << J_CONST_BITS;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int t0 = i7, t1 = i5, t2 = i3, t3 = i1;
  z1 = t0 + t3;
  int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
  const int z5 = (z3 + z4) * JF_1_175875602;
  t0 *= JF_0_298631336; t1 *= JF_2_053119869; t2 *= JF_3_072711026; t3 *= JF_1_501321110;
  z1 *= -JF_0_899976223; z2 *= -JF_2_562915447; z3 = z3 * (-JF_1_961570560) + z5; z4 = z4 * (-JF_0_390180644) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  const int sh = first ? J_CONST_BITS - J_PASS1_BITS : J_CONST_BITS + J_PASS1_BITS + 3;
  d[0] = jdescale(tmp10 + t3, sh); d[7 * st] = jdescale(tmp10 - t3, sh);
  d[st] = jdescale(tmp11 + t2, sh); d[6 * st] = jdescale(tmp11 - t2, sh);
  d[2 * st] = jdescale(tmp12 + t1, sh); d[5 * st] = jdescale(tmp12 - t1, sh);
  d[3 * st] = jdescale(tmp13 + t0, sh); d[4 * st] = jdescale(tmp13 - t0, sh);
}

OVLA_DEV int jpeg_range_limit(int x) {          // sample_range_limit + CENTERJSAMPLE indexed with x & RANGE_MASK
  const int m = x & 1023;
  return m < 128 ? m + 128 : (m < 512 ? 255 : (m < 896 ? 0 : m - 896));
}

// One workgroup per MCU (16 x 16 pixels = 4 Y blocks + 1 Cb + 1 Cr block of 8 x 8).
__global__ __launch_bounds__(256) void jpeg_codec_kernel(const JpegParams p) {
  __shared__ int ws[6][64];
  const int tid = threadIdx.x, img = blockIdx.z;
  const int y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
  const uint8_t* src = p.src + (int64_t)img * p.H * p.W * 3;
  {   // luma: thread (ty, tx) of the MCU; edges replicate the last column / row (expand_right_edge / expand_bottom_edge)
    const int ty = tid >> 4, tx = tid & 15;
    const int gy = min(y0 + ty, p.H - 1), gx = min(x0 + tx, p.W - 1);
    int y, cb, cr;
    jpeg_ycc(src + ((int64_t)gy * p.W + gx) * 3, y, cb, cr);
    ws[(ty >> 3) * 2 + (tx >> 3)][(ty & 7) * 8 + (tx & 7)] = y - 128;
  }
  if (tid < 64) {   // chroma: h2v2_downsample; the input is widened by column replication and made even-height by row replication, rows of the
                    // last iMCU row beyond the real data replicate the last DOWNSAMPLED row
    const int cy = tid >> 3, cx = tid & 7;
    const int hc = (p.H + 1) >> 1;
    const int rc = min(blockIdx.y * 8 + cy, hc - 1), c = blockIdx.x * 8 + cx;
    const int r0 = 2 * rc, r1 = min(2 * rc + 1, p.H - 1), c0 = min(2 * c, p.W - 1), c1 = min(2 * c + 1, p.W - 1);
    int sb = 0, sr = 0, y, cb, cr;
    jpeg_ycc(src + ((int64_t)r0 * p.W + c0) * 3, y, cb, cr); sb += cb; sr += cr;
    jpeg_ycc(src + ((int64_t)r0 * p.W + c1) * 3, y, cb, cr); sb += cb; sr += cr;
    jpeg_ycc(src + ((int64_t)r1 * p.W + c0) * 3, y, cb, cr); sb += cb; sr += cr;
    jpeg_ycc(src + ((int64_t)r1 * p.W + c1) * 3, y, cb, cr); sb += cb; sr += cr;
    const int bias = (c & 1) ? 2 : 1;                 // alternating 1, 2 along the output row (the MCU starts at an even column)
    ws[4][cy * 8 + cx] = ((sb + bias) >> 2) - 128;
    ws[5][cy * 8 + cx] = ((sr + bias) >> 2) - 128;
  }
  __syncthreads();
  const int blk = tid >> 3, k = tid & 7;              // 48 (block, row / column) tasks
  if (tid < 48) jpeg_fdct8(&ws[blk][k * 8], 1, true);                 // FDCT pass 1: rows
  __syncthreads();
  if (tid < 48) {                                                     // FDCT pass 2 on column k, quantise, dequantise, IDCT pass 1 on the same column
    int* col = &ws[blk][k];
    jpeg_fdct8(col, 8, false);
    const int16_t* q = blk < 4 ? p.ql : p.qc;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int v = col[r * 8], qv = q[r * 8 + k], div = qv << 3;
      const int a = v < 0 ? -v : v;
      const int quant = (a + (div >> 1)) / div;
      col[r * 8] = (v < 0 ? -quant : quant) * qv;
    }
    jpeg_idct8(col, 8, true);
  }
  __syncthreads();
  if (tid < 48) jpeg_idct8(&ws[blk][k * 8], 1, false);                // IDCT pass 2: rows (still scaled; range-limited at the store)
  __syncthreads();
  {
    const int ty = tid >> 4, tx = tid & 15;
    uint8_t* Y = p.planes + (int64_t)img * p.H16 * p.W16;
    Y[(int64_t)(y0 + ty) * p.W16 + x0 + tx] = (uint8_t)jpeg_range_limit(ws[(ty >> 3) * 2 + (tx >> 3)][(ty & 7) * 8 + (tx & 7)]);
  }
  if (tid < 128) {
    const int comp = tid >> 6, cy = (tid >> 3) & 7, cx = tid & 7;
    const int hc16 = p.H16 >> 1, wc16 = p.W16 >> 1;
    uint8_t* C = p.planes + (int64_t)p.n_img * p.H16 * p.W16 + ((int64_t)comp * p.n_img + img) * hc16 * wc16;
    C[(int64_t)(blockIdx.y * 8 + cy) * wc16 + blockIdx.x * 8 + cx] = (uint8_t)jpeg_range_limit(ws[4 + comp][cy * 8 + cx]);
  }
}

// jdsample.c h2v2_fancy_upsample at one output position of the REAL chroma plane [hc, wc] (row stride ld)
OVLA_DEV int jpeg_fancy(const uint8_t* C, int ld, int hc, int wc, int y, int x) {
  const int r = y >> 1, c = x >> 1;
  const int rn = (y & 1) ? min(r + 1, hc - 1) : max(r - 1, 0);
  const int cur = 3 * C[r * ld + c] + C[rn * ld + c];
  if (x & 1) {
    if (c == wc - 1) return (cur * 4 + 7) >> 4;
    return (cur * 3 + 3 * C[r * ld + c + 1] + C[rn * ld + c + 1] + 7) >> 4;
  }
  if (c == 0) return (cur * 4 + 8) >> 4;
  return (cur * 3 + 3 * C[r * ld + c - 1] + C[rn * ld + c - 1] + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_upsample_color_kernel(const JpegParams p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)p.H * p.W;
  if (idx >= per * p.n_img) return;
  const int img = (int)(idx / per), y = (int)((idx % per) / p.W), x = (int)(idx % p.W);
  const int hc16 = p.H16 >> 1, wc16 = p.W16 >> 1, hc = (p.H + 1) >> 1, wc = (p.W + 1) >> 1;
  const uint8_t* Y = p.planes + (int64_t)img * p.H16 * p.W16;
  const uint8_t* Cb = p.planes + (int64_t)p.n_img * p.H16 * p.W16 + (int64_t)img * hc16 * wc16;
  const uint8_t* Cr = Cb + (int64_t)p.n_img * hc16 * wc16;
  const int yy = Y[(int64_t)y * p.W16 + x];
  const int xb = jpeg_fancy(Cb, wc16, hc, wc, y, x) - 128, xr = jpeg_fancy(Cr, wc16, hc, wc, y, x) - 128;
  const int r = yy + ((jfix(1.40200) * xr + 32768) >> 16);                                   // jdcolor.c ycc_rgb_convert
  const int g = yy + ((-jfix(0.34414) * xb + 32768 - jfix(0.71414) * xr) >> 16);
  const int b = yy + ((jfix(1.77200) * xb + 32768) >> 16);
  uint8_t* o = p.dst + idx * 3;
  o[0] = (uint8_t)min(max(r, 0), 255); o[1] = (uint8_t)min(max(g, 0), 255); o[2] = (uint8_t)min(max(b, 0), 255);
}
}  // namespace

extern "C" int64_t ovla_jpeg_roundtrip_workspace_bytes(int32_t n_img, int32_t H, int32_t W) {
  if (n_img <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t H16 = (H + 15) / 16 * 16, W16 = (W + 15) / 16 * 16;
  return (int64_t)n_img * (H16 * W16 + 2 * (H16 / 2) * (W16 / 2));
}

extern "C" int ovla_jpeg_roundtrip(const ovla_jpeg_roundtrip_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->workspace, "ovla_jpeg_roundtrip: null pointer");
  OVLA_REQUIRE(a->n_img > 0 && a->H > 0 && a->W > 0 && a->H <= 16384 && a->W <= 16384, "ovla_jpeg_roundtrip: bad shape %d x %d x %d", a->n_img, a->H, a->W);
  OVLA_REQUIRE(a->quality >= 1 && a->quality <= 100, "ovla_jpeg_roundtrip: quality %d outside 1..100", a->quality);
  OVLA_REQUIRE(a->workspace_bytes >= ovla_jpeg_roundtrip_workspace_bytes(a->n_img, a->H, a->W), "ovla_jpeg_roundtrip: needs a workspace of %lld bytes",
               (long long)ovla_jpeg_roundtrip_workspace_bytes(a->n_img, a->H, a->W));
  static const int kLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
  static const int kChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
  JpegParams p;
  p.src = (const uint8_t*)a->src; p.dst = (uint8_t*)a->dst; p.planes = (uint8_t*)a->workspace;
  p.n_img = a->n_img; p.H = a->H; p.W = a->W; p.H16 = (a->H + 15) / 16 * 16; p.W16 = (a->W + 15) / 16 * 16;
  const int scale = a->quality < 50 ? 5000 / a->quality : 200 - 2 * a->quality;     // jpeg_quality_scaling; jpeg_add_quant_table(force_baseline)
  for (int i = 0; i < 64; ++i) {
    int l = (kLuma[i] * scale + 50) / 100, c = (kChroma[i] * scale + 50) / 100;
    p.ql[i] = (int16_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
    p.qc[i] = (int16_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
  }
  hipLaunchKernelGGL(jpeg_codec_kernel, dim3(p.W16 / 16, p.H16 / 16, p.n_img), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_jpeg_roundtrip(codec)");
  hipLaunchKernelGGL(jpeg_upsample_color_kernel, dim3((unsigned)(((int64_t)p.n_img * p.H * p.W + 255) / 256)), dim3(256), 0, stream, p);   // one thread per pixel, not grid-stride
  OVLA_CHECK_LAUNCH("ovla_jpeg_roundtrip(upsample)");
  return OVLA_OK;
}

extern "C" int64_t ovla_image_augment_workspace_bytes(int32_t n_img, int32_t out) {
  if (n_img <= 0 || out <= 0) return 0;
  const int64_t tmp = (((int64_t)n_img * out * out * 3 * 4) + 15) / 16 * 16;
  return tmp + (int64_t)n_img * AUG_BLOCKS * 3 * 8;
}

extern "C" int ovla_image_augment(const ovla_image_augment_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->params && a->workspace, "ovla_image_augment: null pointer");
  OVLA_REQUIRE(a->n_img > 0 && a->n_img <= 65535 && a->H > 1 && a->W > 1 && a->out > 1, "ovla_image_augment: bad shape %d x %d x %d -> %d", a->n_img, a->H, a->W, a->out);
  OVLA_REQUIRE((a->ops_mask & ~31) == 0, "ovla_image_augment: ops_mask %d has unknown bits", a->ops_mask);
  OVLA_REQUIRE((a->ops_mask & 1) || (a->H == a->out && a->W == a->out), "ovla_image_augment: without crop-and-resize the input must already be %d x %d", a->out, a->out);
  OVLA_REQUIRE(aligned16(a->workspace) && a->workspace_bytes >= ovla_image_augment_workspace_bytes(a->n_img, a->out),
               "ovla_image_augment: needs a 16-byte aligned workspace of %lld bytes", (long long)ovla_image_augment_workspace_bytes(a->n_img, a->out));
  ImageAugParams p;
  p.src = (const uint8_t*)a->src; p.dst = (bf16_bits*)a->dst; p.params = a->params;
  p.tmp = (float*)a->workspace;
  p.partial = (double*)((char*)a->workspace + (((int64_t)a->n_img * a->out * a->out * 3 * 4) + 15) / 16 * 16);
  p.n_img = a->n_img; p.H = a->H; p.W = a->W; p.out = a->out; p.mask = a->ops_mask;
  for (int i = 0; i < 6; ++i) {
    OVLA_REQUIRE(a->std[i] != 0.0f, "ovla_image_augment: std[%d] == 0", i);
    p.mean[i] = a->mean[i]; p.stdv[i] = a->std[i];
  }
  hipLaunchKernelGGL(image_aug_a_kernel, dim3(AUG_BLOCKS, a->n_img), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_image_augment(a)");
  hipLaunchKernelGGL(image_aug_b_kernel, dim3(cdiv(a->out * a->out, 1024), a->n_img), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_image_augment(b)");
  return OVLA_OK;
}

extern "C" int ovla_norm_fwd(const ovla_norm_fwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->y && a->weight, "ovla_norm_fwd: null pointer");
  OVLA_REQUIRE(a->rows > 0 && a->dim > 0 && (a->dim % 8) == 0, "ovla_norm_fwd: rows=%d dim=%d (dim %% 8 == 0)", a->rows, a->dim);
  OVLA_REQUIRE(aligned16(a->x) && aligned16(a->y) && aligned16(a->weight) && (!a->bias || aligned16(a->bias)), "ovla_norm_fwd: 16-byte alignment");
#define OVLA_NORM_FWD_WAVE(CPL)                                                                                                             \
  hipLaunchKernelGGL(norm_fwd_wave_kernel<CPL>, dim3(cdiv(a->rows, 4)), dim3(256), 0, stream, (const bf16_bits*)a->x, (bf16_bits*)a->y, \
                     (const bf16_bits*)a->weight, (const bf16_bits*)a->bias, a->mean, a->rstd, a->rows, a->dim, a->eps, a->is_rms)
  if (a->dim <= 1024) OVLA_NORM_FWD_WAVE(2);
  else if (a->dim <= 1536) OVLA_NORM_FWD_WAVE(3);
  else
    hipLaunchKernelGGL(norm_fwd_kernel, dim3(a->rows), dim3(256), 0, stream, (const bf16_bits*)a->x, (bf16_bits*)a->y,
                       (const bf16_bits*)a->weight, (const bf16_bits*)a->bias, a->mean, a->rstd, a->dim, a->eps, a->is_rms);
#undef OVLA_NORM_FWD_WAVE
  OVLA_CHECK_LAUNCH("ovla_norm_fwd");
  return OVLA_OK;
}

extern "C" int ovla_norm_bwd(const ovla_norm_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->dy && a->weight && a->rstd && a->dx, "ovla_norm_bwd: null pointer");
  OVLA_REQUIRE(a->is_rms || a->mean, "ovla_norm_bwd: LayerNorm needs mean");
  OVLA_REQUIRE(a->rows > 0 && a->dim > 0 && (a->dim % 8) == 0, "ovla_norm_bwd: rows=%d dim=%d", a->rows, a->dim);
  OVLA_REQUIRE(aligned16(a->x) && aligned16(a->dy) && aligned16(a->dx) && aligned16(a->weight), "ovla_norm_bwd: 16-byte alignment");
  hipLaunchKernelGGL(norm_bwd_kernel, dim3(a->rows), dim3(256), 0, stream, (const bf16_bits*)a->x, (const bf16_bits*)a->dy,
                     (const bf16_bits*)a->weight, a->mean, a->rstd, (bf16_bits*)a->dx, a->dweight, a->dbias, a->dim, a->is_rms,
                     a->dx_accum);
  OVLA_CHECK_LAUNCH("ovla_norm_bwd");
  return OVLA_OK;
}

extern "C" int ovla_rope_table(void* cos_table, void* sin_table, int32_t S, int32_t head_dim, float theta, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(cos_table && sin_table && S > 0 && head_dim > 0 && (head_dim % 8) == 0, "ovla_rope_table: bad arguments");
  const int half = head_dim / 2;
  hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv((int64_t)S * half, 256)), dim3(256), 0, stream, (bf16_bits*)cos_table,
                     (bf16_bits*)sin_table, S, half, theta);
  OVLA_CHECK_LAUNCH("ovla_rope_table");
  return OVLA_OK;
}

extern "C" int ovla_rope(const ovla_rope_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->qk && a->cos_table && a->sin_table, "ovla_rope: null pointer");
  OVLA_REQUIRE(a->rows > 0 && a->S > 0 && a->n_heads > 0 && (a->head_dim % 8) == 0 && (a->ld % 4) == 0, "ovla_rope: bad shape");
  OVLA_REQUIRE((((uintptr_t)a->qk) & 7) == 0, "ovla_rope: qk must be 8-byte aligned");
  const int64_t work = (int64_t)a->rows * a->n_heads * (a->head_dim / 8);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(work)), dim3(256), 0, stream, (bf16_bits*)a->qk, a->ld, a->rows, a->S, a->n_heads,
                     a->head_dim, (const bf16_bits*)a->cos_table, (const bf16_bits*)a->sin_table, a->inverse);
  OVLA_CHECK_LAUNCH("ovla_rope");
  return OVLA_OK;
}

extern "C" int ovla_swiglu_fwd(const ovla_swiglu_fwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->gu && a->h && a->rows > 0 && a->F > 0 && (a->F % 8) == 0, "ovla_swiglu_fwd: bad arguments");
  OVLA_REQUIRE(aligned16(a->gu) && aligned16(a->h), "ovla_swiglu_fwd: 16-byte alignment");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((int64_t)a->rows * a->F / 8)), dim3(256), 0, stream, (const bf16_bits*)a->gu,
                     (bf16_bits*)a->h, a->rows, a->F);
  OVLA_CHECK_LAUNCH("ovla_swiglu_fwd");
  return OVLA_OK;
}
extern "C" int ovla_swiglu_bwd(const ovla_swiglu_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->gu && a->dh && a->dgu && a->rows > 0 && a->F > 0 && (a->F % 8) == 0, "ovla_swiglu_bwd: bad arguments");
  OVLA_REQUIRE(aligned16(a->gu) && aligned16(a->dh) && aligned16(a->dgu), "ovla_swiglu_bwd: 16-byte alignment");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((int64_t)a->rows * a->F / 8)), dim3(256), 0, stream, (const bf16_bits*)a->gu,
                     (const bf16_bits*)a->dh, (bf16_bits*)a->dgu, a->rows, a->F);
  OVLA_CHECK_LAUNCH("ovla_swiglu_bwd");
  return OVLA_OK;
}
extern "C" int ovla_act_bwd(const ovla_act_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->z && a->dh && a->dz && a->n > 0 && (a->n % 8) == 0, "ovla_act_bwd: bad arguments");
  OVLA_REQUIRE(aligned16(a->z) && aligned16(a->dh) && aligned16(a->dz), "ovla_act_bwd: 16-byte alignment");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(a->n / 8)), dim3(256), 0, stream, (const bf16_bits*)a->z, (const bf16_bits*)a->dh,
                     (bf16_bits*)a->dz, a->n / 8, a->act);
  OVLA_CHECK_LAUNCH("ovla_act_bwd");
  return OVLA_OK;
}
extern "C" int ovla_add_bf16(const ovla_add_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->a && a->out && a->n > 0 && (a->n % 8) == 0, "ovla_add_bf16: bad arguments");
  OVLA_REQUIRE(aligned16(a->a) && aligned16(a->out) && (!a->b || aligned16(a->b)), "ovla_add_bf16: 16-byte alignment");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(a->n / 8)), dim3(256), 0, stream, (const bf16_bits*)a->a, (const bf16_bits*)a->b,
                     (bf16_bits*)a->out, a->n / 8);
  OVLA_CHECK_LAUNCH("ovla_add_bf16");
  return OVLA_OK;
}
extern "C" int ovla_colscale_bf16(const ovla_colscale_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->scale && a->out && a->rows > 0 && a->dim > 0 && (a->dim % 8) == 0, "ovla_colscale_bf16: bad arguments");
  hipLaunchKernelGGL(colscale_kernel, dim3(grid_for((int64_t)a->rows * a->dim / 8)), dim3(256), 0, stream, (const bf16_bits*)a->x,
                     (const bf16_bits*)a->scale, (bf16_bits*)a->out, a->rows, a->dim);
  OVLA_CHECK_LAUNCH("ovla_colscale_bf16");
  return OVLA_OK;
}
extern "C" int ovla_image_prep(const ovla_image_prep_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst, "ovla_image_prep: null pointer");
  OVLA_REQUIRE(a->n_img > 0 && a->H > 1 && a->W > 1 && a->out > 1, "ovla_image_prep: bad shape %d x %d x %d -> %d", a->n_img, a->H, a->W, a->out);
  OVLA_REQUIRE(a->crop || (a->H == a->out && a->W == a->out), "ovla_image_prep: without the crop the input must already be %d x %d", a->out, a->out);
  OVLA_REQUIRE(!a->crop || a->crop_scale > 0.0f, "ovla_image_prep: crop_scale %f must be positive", (double)a->crop_scale);
  ImagePrepParams p;
  p.src = (const uint8_t*)a->src; p.dst = (bf16_bits*)a->dst; p.n_img = a->n_img; p.H = a->H; p.W = a->W; p.out = a->out; p.crop = a->crop;
  {   // the box and the sampling grid in fp32, one rounding per operation, as TF computes them (volatile: no host-side contraction)
    volatile float side = sqrtf(a->crop_scale);
    side = side < 0.0f ? 0.0f : (side > 1.0f ? 1.0f : side);
    volatile float o1 = (1.0f - side) / 2.0f;
    volatile float o2 = o1 + side;
    volatile float span = o2 - o1;
    volatile float sy = span * (float)(a->H - 1), sx = span * (float)(a->W - 1);
    p.ystep = sy / (float)(a->out - 1); p.xstep = sx / (float)(a->out - 1);
    volatile float by = o1 * (float)(a->H - 1), bx = o1 * (float)(a->W - 1);
    p.ybase = by; p.xbase = bx;
  }
  for (int i = 0; i < 6; ++i) {
    OVLA_REQUIRE(a->std[i] != 0.0f, "ovla_image_prep: std[%d] == 0", i);
    p.mean[i] = a->mean[i]; p.stdv[i] = a->std[i];
  }
  hipLaunchKernelGGL(image_prep_kernel, dim3(grid_for((int64_t)a->n_img * a->out * a->out)), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_image_prep");
  return OVLA_OK;
}
extern "C" int ovla_im2col(const ovla_im2col_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->pixels && a->out, "ovla_im2col: null pointer");
  OVLA_REQUIRE(a->B > 0 && a->patch > 0 && a->H % a->patch == 0 && a->W % a->patch == 0, "ovla_im2col: image %dx%d not divisible by patch %d", a->H, a->W, a->patch);
  const int n_img = a->n_img > 0 ? a->n_img : 1;
  OVLA_REQUIRE(a->c0 >= 0 && a->c0 + (n_img - 1) * a->img_cstride + 3 <= a->C_total && a->ldo >= 3 * a->patch * a->patch, "ovla_im2col: channel range / ldo");
  const int64_t total = (int64_t)a->B * n_img * (a->H / a->patch) * (a->W / a->patch) * a->ldo;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_bits*)a->pixels, (bf16_bits*)a->out, a->ldo,
                     a->B, a->C_total, a->c0, a->H, a->W, a->patch, n_img, a->img_cstride);
  OVLA_CHECK_LAUNCH("ovla_im2col");
  return OVLA_OK;
}
extern "C" int ovla_vit_embed(const ovla_vit_embed_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->patches && a->pos && a->tokens && (a->n_prefix == 0 || a->prefix), "ovla_vit_embed: null pointer");
  OVLA_REQUIRE(a->B > 0 && a->n_patches > 0 && (a->dim % 8) == 0, "ovla_vit_embed: bad shape");
  const int64_t total = (int64_t)a->B * (a->n_patches + a->n_prefix) * (a->dim / 8);
  hipLaunchKernelGGL(vit_embed_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_bits*)a->patches, (const bf16_bits*)a->pos,
                     (const bf16_bits*)a->prefix, (bf16_bits*)a->tokens, a->B, a->n_patches, a->n_prefix, a->dim);
  OVLA_CHECK_LAUNCH("ovla_vit_embed");
  return OVLA_OK;
}
extern "C" int ovla_copy_rows(const ovla_copy_rows_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->B > 0 && a->rows > 0 && a->dim > 0 && (a->dim % 8) == 0, "ovla_copy_rows: bad arguments");
  OVLA_REQUIRE((a->src_ld % 8) == 0 && (a->dst_ld % 8) == 0 && (a->dst_col0 % 8) == 0 && (a->src_batch_stride % 8) == 0 && (a->dst_batch_stride % 8) == 0,
               "ovla_copy_rows: strides must be multiples of 8 elements");
  OVLA_REQUIRE(aligned16(a->src) && aligned16(a->dst), "ovla_copy_rows: 16-byte alignment");
  const int64_t total = (int64_t)a->B * a->rows * (a->dim / 8);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_bits*)a->src, (bf16_bits*)a->dst, a->B,
                     a->rows, a->dim, a->src_batch_stride, a->src_row0, a->src_ld, a->dst_batch_stride, a->dst_row0, a->dst_ld,
                     a->dst_col0, a->accumulate);
  OVLA_CHECK_LAUNCH("ovla_copy_rows");
  return OVLA_OK;
}
extern "C" int ovla_film_bwd(const ovla_film_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->dy && a->x_pre && a->gamma && a->dgamma && a->dbeta && a->B > 0 && a->rows_per_batch > 0 && a->dim > 0, "ovla_film_bwd: bad arguments");
  const int chunk = 32;
  hipLaunchKernelGGL(film_bwd_kernel, dim3(cdiv(a->dim, 256), a->B, cdiv(a->rows_per_batch, chunk)), dim3(256), 0, stream, (bf16_bits*)a->dy,
                     (const bf16_bits*)a->x_pre, (const bf16_bits*)a->gamma, a->dgamma, a->dbeta, a->rows_per_batch, a->dim, chunk);
  OVLA_CHECK_LAUNCH("ovla_film_bwd");
  return OVLA_OK;
}
extern "C" int ovla_masked_mean(const ovla_masked_mean_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->x && a->row_mask && a->out && a->B > 0 && a->L > 0 && a->dim > 0, "ovla_masked_mean: bad arguments");
  hipLaunchKernelGGL(masked_mean_kernel, dim3(cdiv(a->dim, 256), a->B), dim3(256), 0, stream, (const bf16_bits*)a->x, a->row_mask,
                     (bf16_bits*)a->out, a->B, a->L, a->dim);
  OVLA_CHECK_LAUNCH("ovla_masked_mean");
  return OVLA_OK;
}
extern "C" int ovla_language_average(const ovla_language_average_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->ids && a->labels && a->embed_table && a->out, "ovla_language_average: null pointer");
  OVLA_REQUIRE(a->B > 0 && a->L > 0 && a->D > 0 && a->vocab > 0, "ovla_language_average: bad shape B=%d L=%d D=%d vocab=%d", a->B, a->L, a->D, a->vocab);
  hipLaunchKernelGGL(language_average_kernel, dim3(cdiv(a->D, 256), a->B), dim3(256), 0, stream, a->ids, a->labels, (const bf16_bits*)a->embed_table,
                     (bf16_bits*)a->out, a->L, a->D, a->vocab, a->action_token_begin);
  OVLA_CHECK_LAUNCH("ovla_language_average");
  return OVLA_OK;
}
extern "C" int ovla_assemble_multimodal(const ovla_assemble_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->ids && a->labels && a->embed_table && a->patches && a->out, "ovla_assemble_multimodal: null pointer");
  OVLA_REQUIRE(a->B > 0 && a->L > 0 && a->P > 0 && a->D > 0 && (a->D % 8) == 0 && a->vocab > 0, "ovla_assemble_multimodal: bad shape");
  OVLA_REQUIRE(aligned16(a->embed_table) && aligned16(a->patches) && aligned16(a->out), "ovla_assemble_multimodal: 16-byte alignment");
  hipLaunchKernelGGL(assemble_kernel, dim3(a->B * (a->P + a->L)), dim3(256), 0, stream, a->ids, a->labels, (const bf16_bits*)a->embed_table,
                     (const bf16_bits*)a->patches, (const bf16_bits*)a->noisy, (bf16_bits*)a->out, a->action_pos, a->B, a->L, a->P, a->D,
                     a->A, a->vocab, a->ignore_index, a->action_token_begin);
  OVLA_CHECK_LAUNCH("ovla_assemble_multimodal");
  return OVLA_OK;
}
namespace {
// one wave per row; lane j owns slot j, j + 64, ... (64 bf16 = 128 bytes = 8 16-byte loads per slot)
__global__ __launch_bounds__(256) void row_sumsq_kernel(const bf16_bits* __restrict__ x, int64_t ld, float* __restrict__ out, int rows, int slots) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  for (int j = lane; j < slots; j += 64) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f[8];
      load8(x + (int64_t)m * ld + j * 64 + c * 8, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = __builtin_fmaf(f[e], f[e], s);
    }
    out[(int64_t)m * slots + j] = s;
  }
}
}  // namespace
extern "C" int ovla_row_sumsq(const void* x, int64_t ld, float* out, int32_t rows, int32_t dim, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(x && out && rows > 0 && dim > 0 && (dim % 64) == 0 && (ld % 8) == 0 && ld >= dim && aligned16(x), "ovla_row_sumsq: rows=%d dim=%d (multiple of 64) ld=%lld", rows, dim, (long long)ld);
  hipLaunchKernelGGL(row_sumsq_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, (const bf16_bits*)x, ld, out, rows, dim / 64);
  OVLA_CHECK_LAUNCH("ovla_row_sumsq");
  return OVLA_OK;
}
extern "C" int ovla_gather_rows(const ovla_gather_rows_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->index && a->dst && a->n > 0 && a->dim > 0 && (a->dim % 8) == 0, "ovla_gather_rows: bad arguments");
  OVLA_REQUIRE((a->src_ld % 8) == 0 && (a->dst_ld % 8) == 0 && aligned16(a->src) && aligned16(a->dst), "ovla_gather_rows: alignment");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(a->n), dim3(256), 0, stream, (const bf16_bits*)a->src, a->index, (bf16_bits*)a->dst, a->n,
                     a->dim, a->src_ld, a->dst_ld, a->scatter_add);
  OVLA_CHECK_LAUNCH("ovla_gather_rows");
  return OVLA_OK;
}
extern "C" int ovla_cvt_f32_to_bf16(const ovla_cvt_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->n > 0, "ovla_cvt_f32_to_bf16: bad arguments");
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3(grid_for(a->n)), dim3(256), 0, stream, a->src, (bf16_bits*)a->dst, a->n, a->scale);
  OVLA_CHECK_LAUNCH("ovla_cvt_f32_to_bf16");
  return OVLA_OK;
}
extern "C" int ovla_cvt_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(src && dst && n > 0, "ovla_cvt_bf16_to_f32: bad arguments");
  hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const bf16_bits*)src, dst, n, scale);
  OVLA_CHECK_LAUNCH("ovla_cvt_bf16_to_f32");
  return OVLA_OK;
}
extern "C" int ovla_transpose_bf16(const ovla_transpose_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->src && a->dst && a->rows > 0 && a->cols > 0 && a->lds >= a->cols && a->ldd >= a->rows, "ovla_transpose_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(a->cols, 64), cdiv(a->rows, 64)), dim3(256), 0, stream, (const bf16_bits*)a->src,
                     (bf16_bits*)a->dst, a->rows, a->cols, a->lds, a->ldd);
  OVLA_CHECK_LAUNCH("ovla_transpose_bf16");
  return OVLA_OK;
}
extern "C" int ovla_transpose_batched(const ovla_transpose_args* table, const int32_t* tile_start, int32_t n, int32_t total_tiles, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(table && tile_start && n > 0 && total_tiles > 0, "ovla_transpose_batched: bad arguments");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, stream, table, tile_start, n);
  OVLA_CHECK_LAUNCH("ovla_transpose_batched");
  return OVLA_OK;
}
extern "C" int ovla_colsum_bf16(const ovla_colsum_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->X && a->out && a->M > 0 && a->N > 0, "ovla_colsum_bf16: bad arguments");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(a->N, 64), cdiv(a->M, rpb)), dim3(256), 0, stream, (const bf16_bits*)a->X, a->ldx, a->out,
                     a->M, a->N, rpb);
  OVLA_CHECK_LAUNCH("ovla_colsum_bf16");
  return OVLA_OK;
}
