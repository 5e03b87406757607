// lora_bwd.hip -- ONE pass over dy for the two LoRA backward products that contract or stream it (gfx950):
//
//     dt_g  = scale * dy_g . B_g            [M, r]    (input of the data-gradient K-extension  dx = dy W + dt A  and of  dA += dt^T x)
//     dB_g += dy_g^T . t_g                  [gn, r]   (fp32, atomically accumulated like every weight gradient)
//
// for the G fused groups of an adapted Linear (q|k|v: 3, gate|up: 2, else 1; peft LoRA r = 32, vla-scripts/finetune.py:862-871).  Before this
// kernel the two were a block-diagonal skinny NT GEMM (+ its split-K reduce) and a TN GEMM, each streaming dy -- 40 .. 214 MB per Linear at
// the fine-tune shapes -- from HBM / the Infinity Cache once: 0.83 GB per decoder layer for 0.1 % of its FLOPs.  Here a workgroup owns a
// (row range, 256-column chunk) of dy_g and walks the rows in steps of 64; each step's [64 x 256] tile is staged once in LDS (row-major,
// coalesced 16-byte loads, register-staged two steps ahead) and feeds BOTH products:
//   * dt: wave w takes rows 16 w .. 16 w + 15 of the step; mfma_f32_16x16x32_bf16 with the operands swapped (B_g^T fragment as A operand, the
//     dy fragment -- a plain 16-byte LDS row read -- as B operand), so a lane owns 4 consecutive r-columns of one row; the 16 B_g^T fragments of
//     the chunk are loop-invariant and live in registers.  The chunk's partial sums go to an fp32 slab [chunk][M][r] with plain 16-byte stores;
//     lora_bwd_finish_kernel adds the chunks IN A FIXED ORDER, scales and rounds to bf16 -- dt feeds the data-gradient chain, whose run-to-run
//     bit reproducibility (tests/test_fullsize_gpu.py::test_full_size_step_is_reproducible) must not depend on an atomic order;
//   * dB: mfma_f32_32x32x16_bf16, dy^T and t fragments fetched with ds_read_b64_tr_b16 from the same LDS image (the contraction index is the
//     ROW of both operands), accumulators persistent over the row walk, one fp32 atomicAdd pass at the end (one accumulator register = two
//     128-byte row segments of dB: the shape float atomics run at full rate with, gemm_tn.hip).
// HBM-bound: algorithmic bytes = M * G * gn * 2 (dy once) + the slab round trip (gn / 256 * M * r * 8).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int LB_ROWS = 64;            // rows of dy per step
constexpr int LB_COLS = 256;           // columns of dy_g per workgroup
constexpr int LB_R = 32;               // LoRA rank handled by this kernel
constexpr int LB_SD = LB_COLS + 32;    // LDS row stride of the dy tile: 576 bytes = 64 (mod 256): conflict-free transposing reads
constexpr int LB_ST = LB_R;            // LDS row stride of the t tile: 64 bytes

struct LoraBwdParams {
  const bf16_bits *dy, *Bt, *t;
  int64_t ld_dy, ld_bt, ld_t, ld_db;
  float* dB;
  float* slab;            // [G][chunks][M][r] fp32 partial dt
  int M, gn, G, chunks, row_splits, m_chunk;
};

OVLA_DEV bf16x4_bits lb_tr16(const bf16_bits* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_bits*)p);
}

// 32x32x16 operand fragment read transposed from a row-major [m][col] tile (gemm_tn.hip: tr_frag32):
// lane l (r = l & 31, hh = l >> 5) receives tile[m0 + 8 hh + jj][c0 + r], jj = 0..7.
OVLA_DEV bf16x8_bits lb_tr_frag32(const bf16_bits* tile, int m0, int c0, int stride, int lane) {
  const int i = lane & 15, pc = (lane >> 4) & 1, hh = lane >> 5;
  const bf16_bits* a0 = tile + (m0 + 8 * hh + (i >> 2)) * stride + c0 + 16 * pc + 4 * (i & 3);
  const bf16x4_bits lo = lb_tr16(a0);
  const bf16x4_bits hi = lb_tr16(a0 + 4 * stride);
  return bf16x8_bits{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) void lora_bwd_kernel(const LoraBwdParams p) {
  __shared__ __attribute__((aligned(16))) bf16_bits s_dy[LB_ROWS * LB_SD];
  __shared__ __attribute__((aligned(16))) bf16_bits s_t[LB_ROWS * LB_ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int rs = b % p.row_splits; b /= p.row_splits;
  const int chunk = b % p.chunks;
  const int g = b / p.chunks;
  const int c0 = chunk * LB_COLS;                       // first column of the chunk inside group g
  const int m_begin = rs * p.m_chunk;
  const int m_end = (m_begin + p.m_chunk) < p.M ? (m_begin + p.m_chunk) : p.M;
  const int nsteps = (m_end - m_begin + LB_ROWS - 1) / LB_ROWS;
  const bf16_bits* dy = p.dy + (int64_t)g * p.gn + c0;  // column origin of this chunk
  const bf16_bits* tg = p.t + (int64_t)g * LB_R;
  const bf16x8_bits zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // loop-invariant B_g^T fragments: bt[jh][ks] = Bt[g r + 16 jh + (lane & 15)][c0 + 32 ks + 8 (lane >> 4) .. + 8]
  bf16x8_bits bt[2][LB_COLS / 32];
#pragma unroll
  for (int jh = 0; jh < 2; ++jh)
#pragma unroll
    for (int ks = 0; ks < LB_COLS / 32; ++ks) {
      const int c = c0 + 32 * ks + 8 * (lane >> 4);
      bt[jh][ks] = (c < p.gn) ? *reinterpret_cast<const bf16x8_bits*>(p.Bt + (int64_t)(g * LB_R + 16 * jh + (lane & 15)) * p.ld_bt + c) : zero8;
    }

  f32x16 accB[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[a][r] = 0.f;

  // register staging, TWO steps ahead (two register sets used alternately: the loads of step s + 2 are issued while step s computes, so a
  // workgroup keeps two 36 KB tiles in flight and a CU's two resident workgroups four -- one tile ahead left the loop waiting ~1.5 us per
  // step for its own loads: 2.3-2.6 TB/s)
  bf16x8_bits rdy[2][8], rt[2];
  auto load = [&](int m0, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i, rr = id >> 5, ch = id & 31;
      const int m = m0 + rr, c = c0 + ch * 8;
      rdy[S][i] = (m < m_end && c < p.gn) ? *reinterpret_cast<const bf16x8_bits*>(dy + (int64_t)m * p.ld_dy + ch * 8) : zero8;
    }
    const int rr = tid >> 2, ch = tid & 3;
    rt[S] = (m0 + rr < m_end) ? *reinterpret_cast<const bf16x8_bits*>(tg + (int64_t)(m0 + rr) * p.ld_t + ch * 8) : zero8;
  };
  auto store = [&](auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i;
      *reinterpret_cast<bf16x8_bits*>(s_dy + (id >> 5) * LB_SD + (id & 31) * 8) = rdy[S][i];
    }
    *reinterpret_cast<bf16x8_bits*>(s_t + (tid >> 2) * LB_ST + (tid & 3) * 8) = rt[S];
  };

  float* slab = p.slab + ((int64_t)(g * p.chunks + chunk) * p.M) * LB_R;
  auto step = [&](int st, auto set_tag) {
    const int m0 = m_begin + st * LB_ROWS;
    __syncthreads();            // the previous step's fragment reads are done
    store(set_tag);
    __syncthreads();
    if (st + 2 < nsteps) load(m0 + 2 * LB_ROWS, set_tag);
    // ---- dt partial of rows m0 + 16 wave .. + 15 over this chunk's 256 columns ----
    f32x4 accT[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bf16_bits* arow = s_dy + (16 * wave + (lane & 15)) * LB_SD + 8 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < LB_COLS / 32; ++ks) {
      const bf16x8_bits a = *reinterpret_cast<const bf16x8_bits*>(arow + 32 * ks);
      accT[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bt[0][ks], a, accT[0], 0, 0, 0);
      accT[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bt[1][ks], a, accT[1], 0, 0, 0);
    }
    {   // swapped operands: the lane owns row m0 + 16 wave + (lane & 15), columns 16 jh + 4 (lane >> 4) .. + 3
      const int m = m0 + 16 * wave + (lane & 15);
      if (m < m_end) {
        float* dst = slab + (int64_t)m * LB_R + 4 * (lane >> 4);
        *reinterpret_cast<f32x4*>(dst) = accT[0];
        *reinterpret_cast<f32x4*>(dst + 16) = accT[1];
      }
    }
    // ---- dB partial: columns 64 wave .. + 63 of the chunk (two 32-column blocks), contraction over the step's 64 rows ----
#pragma unroll
    for (int ks = 0; ks < LB_ROWS / 16; ++ks) {
      const bf16x8_bits yf = lb_tr_frag32(s_t, ks * 16, 0, LB_ST, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8_bits xf = lb_tr_frag32(s_dy, ks * 16, (2 * wave + a) * 32, LB_SD, lane);
        accB[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, accB[a], 0, 0, 0);
      }
    }
  };
  if (nsteps > 0) load(m_begin, std::integral_constant<int, 0>{});
  if (nsteps > 1) load(m_begin + LB_ROWS, std::integral_constant<int, 1>{});
  for (int st = 0; st < nsteps; st += 2) {
    step(st, std::integral_constant<int, 0>{});
    if (st + 1 < nsteps) step(st + 1, std::integral_constant<int, 1>{});
  }
  // D[n][j]: col j = lane & 31, row n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = c0 + (2 * wave + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < p.gn) atomicAdd(p.dB + (int64_t)(g * p.gn + n) * p.ld_db + (lane & 31), accB[a][r]);
    }
}

// dt[m][g r + j] = bf16(scale * sum over the chunks, in chunk order, of slab[g][chunk][m][j])
__global__ __launch_bounds__(256) void lora_bwd_finish_kernel(const float* __restrict__ slab, bf16_bits* __restrict__ dt, int64_t ld_dt, int M, int G,
                                                              int chunks, float scale) {
  const int64_t quads = (int64_t)M * G * (LB_R / 4);
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
    const int j4 = (int)(q % (LB_R / 4)) * 4;
    const int64_t mg = q / (LB_R / 4);
    const int g = (int)(mg % G);
    const int m = (int)(mg / G);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < chunks; ++c) v += *reinterpret_cast<const f32x4*>(slab + (((int64_t)(g * chunks + c) * M) + m) * LB_R + j4);
    bf16x4_bits o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (short)f2bf(v[e] * scale);
    *reinterpret_cast<bf16x4_bits*>(dt + (int64_t)m * ld_dt + g * LB_R + j4) = o;
  }
}

}  // namespace

extern "C" int64_t ovla_lora_bwd_workspace_bytes(int32_t M, int32_t gn, int32_t G) {
  return (int64_t)G * cdiv(gn, LB_COLS) * M * LB_R * 4;
}

extern "C" int ovla_lora_bwd(const ovla_lora_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->dy && a->Bt && a->t && a->dt && a->dB, "ovla_lora_bwd: null pointer");
  OVLA_REQUIRE(a->M > 0 && a->gn > 0 && a->G >= 1 && a->r == LB_R, "ovla_lora_bwd: M=%d gn=%d G=%d r=%d (the kernel handles rank 32)", a->M, a->gn, a->G, a->r);
  OVLA_REQUIRE((a->gn % 8) == 0 && (a->ld_dy % 8) == 0 && (a->ld_bt % 8) == 0 && (a->ld_t % 8) == 0 && (a->ld_dt % 4) == 0,
               "ovla_lora_bwd: gn, ld_dy, ld_bt, ld_t must be multiples of 8, ld_dt of 4");
  OVLA_REQUIRE(a->ld_dy >= (int64_t)a->G * a->gn && a->ld_bt >= a->gn && a->ld_t >= (int64_t)a->G * LB_R && a->ld_dt >= (int64_t)a->G * LB_R && a->ld_db >= LB_R,
               "ovla_lora_bwd: leading dimension smaller than extent");
  OVLA_REQUIRE(aligned16(a->dy) && aligned16(a->Bt) && aligned16(a->t) && (((uintptr_t)a->dt) & 7) == 0 && (((uintptr_t)a->dB) & 3) == 0,
               "ovla_lora_bwd: dy / Bt / t need 16-byte, dt 8-byte alignment");
  const int64_t need = ovla_lora_bwd_workspace_bytes(a->M, a->gn, a->G);
  OVLA_REQUIRE(a->workspace && aligned16(a->workspace) && a->workspace_bytes >= need, "ovla_lora_bwd: needs a 16-byte aligned workspace of %lld bytes", (long long)need);
  LoraBwdParams p;
  p.dy = (const bf16_bits*)a->dy; p.Bt = (const bf16_bits*)a->Bt; p.t = (const bf16_bits*)a->t;
  p.ld_dy = a->ld_dy; p.ld_bt = a->ld_bt; p.ld_t = a->ld_t; p.ld_db = a->ld_db;
  p.dB = a->dB; p.slab = (float*)a->workspace;
  p.M = a->M; p.gn = a->gn; p.G = a->G;
  p.chunks = cdiv(a->gn, LB_COLS);
  // row splits: enough workgroups to stream from every CU (>= ~3 per CU), at least 256 rows each (dB atomics <= 1/8 of the bytes read)
  int rs = cdiv(768, p.chunks * p.G);
  const int max_rs = cdiv(a->M, 256);
  if (rs > max_rs) rs = max_rs;
  if (rs < 1) rs = 1;
  p.m_chunk = cdiv(cdiv(a->M, rs), LB_ROWS) * LB_ROWS;
  p.row_splits = cdiv(a->M, p.m_chunk);
  hipLaunchKernelGGL(lora_bwd_kernel, dim3((unsigned)(p.G * p.chunks * p.row_splits)), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_lora_bwd");
  const int64_t quads = (int64_t)a->M * a->G * (LB_R / 4);
  int blocks = cdiv(quads, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(lora_bwd_finish_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)a->workspace, (bf16_bits*)a->dt, a->ld_dt, a->M, a->G, p.chunks,
                     a->scale);
  OVLA_CHECK_LAUNCH("ovla_lora_bwd(finish)");
  return OVLA_OK;
}
