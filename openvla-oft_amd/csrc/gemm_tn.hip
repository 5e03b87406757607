// gemm_tn.hip -- "TN" GEMM for weight gradients on gfx950:  C[P,Q] (+)= alpha * X[M,P]^T . Y[M,Q]
//
// Replaces autograd's weight-gradient matmuls for the trainable tensors of the OpenVLA-OFT fine-tune
// (LoRA A/B, action head, proprio / noisy-action projectors, FiLM Linears; vla-scripts/finetune.py:862-932).
//
// Both operands are contracted over their ROW index, so neither is K-contiguous: X and Y tiles are staged row-major in
// LDS (coalesced 16-byte global loads, register-staged one tile ahead) and the MFMA fragments are fetched with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read.  mfma_f32_32x32x16_bf16 is used unswapped so that one accumulator
// register is two 128-byte row segments of C: the shape global float atomics run at full rate with
// (MI355X_MICROARCH.md "Global float atomics").  The M range is split over workgroups only in atomic mode.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int BMK = 64;  // rows of M per LDS stage

struct TnParams {
  const bf16_bits *X, *Y;
  int64_t ldx, ldy, ldc;
  void* C;
  int M, P, Q, tiles_p, tiles_q, m_chunk, out_mode;
  float alpha;
};

OVLA_DEV bf16x4_bits lds_tr16(const bf16_bits* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_bits*)p);
}

// 32x32x16 operand fragment read transposed from a row-major [m][col] tile:
// lane l (r = l&31, hh = l>>5) receives tile[m0 + 8hh + jj][c0 + r], jj = 0..7.
OVLA_DEV bf16x8_bits tr_frag32(const bf16_bits* tile, int m0, int c0, int stride, int lane) {
  const int i = lane & 15, pc = (lane >> 4) & 1, hh = lane >> 5;
  const bf16_bits* a0 = tile + (m0 + 8 * hh + (i >> 2)) * stride + c0 + 16 * pc + 4 * (i & 3);
  const bf16x4_bits lo = lds_tr16(a0);
  const bf16x4_bits hi = lds_tr16(a0 + 4 * stride);
  return bf16x8_bits{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int COLS>
struct TnStage {
  static constexpr int CH = COLS / 8;
  static constexpr int TOTAL = BMK * CH;
  static constexpr int PER_THREAD = (TOTAL + 255) / 256;
  static constexpr int STRIDE = COLS + (COLS == 32 ? 0 : 32);  // row bytes == 64 or 192 (mod 256): conflict-free tr reads
  bf16x8_bits r[PER_THREAD];
  OVLA_DEV void load(const bf16_bits* G, int64_t ld, int m0, int m_end, int c0, int ncols, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + 256 * i;
      const int rr = id / CH, ch = id % CH;
      const int m = m0 + rr, c = c0 + ch * 8;
      if (id < TOTAL && m < m_end && c < ncols) r[i] = *reinterpret_cast<const bf16x8_bits*>(G + (int64_t)m * ld + c);
      else r[i] = bf16x8_bits{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  OVLA_DEV void store(bf16_bits* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + 256 * i;
      if (id < TOTAL) *reinterpret_cast<bf16x8_bits*>(tile + (id / CH) * STRIDE + (id % CH) * 8) = r[i];
    }
  }
};

template <int BP, int BQ, int WP, int WQ>
OVLA_DEV void gemm_tn_body(const TnParams& p, bf16_bits* smem, int tile, int msplit) {
  static_assert(WP * WQ == 4, "4 waves");
  constexpr int TP = BP / WP / 32, TQ = BQ / WQ / 32;
  constexpr int SX = TnStage<BP>::STRIDE, SY = TnStage<BQ>::STRIDE;
  bf16_bits* Xs = smem;
  bf16_bits* Ys = smem + BMK * SX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / WQ, wq = wave % WQ;
  const int tp = tile / p.tiles_q, tq = tile % p.tiles_q;
  const int p0 = tp * BP, q0 = tq * BQ;
  const int m_begin = msplit * p.m_chunk;
  const int m_end = (m_begin + p.m_chunk) < p.M ? (m_begin + p.m_chunk) : p.M;
  const int nsteps = (m_end - m_begin + BMK - 1) / BMK;

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int a = 0; a < TP; ++a)
#pragma unroll
    for (int b = 0; b < TQ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  TnStage<BP> xs;
  TnStage<BQ> ys;
  if (nsteps > 0) {
    xs.load(p.X, p.ldx, m_begin, m_end, p0, p.P, tid);
    ys.load(p.Y, p.ldy, m_begin, m_end, q0, p.Q, tid);
  }
  for (int t = 0; t < nsteps; ++t) {
    __syncthreads();
    xs.store(Xs, tid);
    ys.store(Ys, tid);
    __syncthreads();
    if (t + 1 < nsteps) {
      xs.load(p.X, p.ldx, m_begin + (t + 1) * BMK, m_end, p0, p.P, tid);
      ys.load(p.Y, p.ldy, m_begin + (t + 1) * BMK, m_end, q0, p.Q, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BMK / 16; ++ks) {
      bf16x8_bits xf[TP], yf[TQ];
#pragma unroll
      for (int a = 0; a < TP; ++a) xf[a] = tr_frag32(Xs, ks * 16, (wp * TP + a) * 32, SX, lane);
#pragma unroll
      for (int b = 0; b < TQ; ++b) yf[b] = tr_frag32(Ys, ks * 16, (wq * TQ + b) * 32, SY, lane);
#pragma unroll
      for (int a = 0; a < TP; ++a)
#pragma unroll
        for (int b = 0; b < TQ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a], yf[b], acc[a][b], 0, 0, 0);
    }
  }
  // D[p][q]: col q = lane & 31, row p = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < TP; ++a)
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
      const int qq = q0 + (wq * TQ + b) * 32 + (lane & 31);
      if (qq >= p.Q) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = p0 + (wp * TP + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (pp >= p.P) continue;
        const float v = acc[a][b][r] * p.alpha;
        const int64_t off = (int64_t)pp * p.ldc + qq;
        if (p.out_mode == 0) atomicAdd(reinterpret_cast<float*>(p.C) + off, v);
        else if (p.out_mode == 1) reinterpret_cast<float*>(p.C)[off] = v;
        else reinterpret_cast<bf16_bits*>(p.C)[off] = f2bf(v);
      }
    }
}

constexpr int TN_LDS_ELEMS = BMK * (128 + 32) * 2;   // largest X + Y tile pair (128x128 config)

template <int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnParams p) {
  __shared__ __attribute__((aligned(16))) bf16_bits smem[TN_LDS_ELEMS];
  gemm_tn_body<BP, BQ, WP, WQ>(p, smem, blockIdx.x, blockIdx.y);
}

struct TnGroup {
  TnParams prob[OVLA_TN_MAX_GROUP];
  int cfg[OVLA_TN_MAX_GROUP];      // 0: <32,128>  1: <128,32>  2: <128,128>
  int blocks[OVLA_TN_MAX_GROUP];   // tiles * m-splits of each problem
  int n;
};

__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const TnGroup g) {
  __shared__ __attribute__((aligned(16))) bf16_bits smem[TN_LDS_ELEMS];
  int b = blockIdx.x, i = 0;
  while (i < g.n - 1 && b >= g.blocks[i]) {
    b -= g.blocks[i];
    ++i;
  }
  const TnParams& p = g.prob[i];
  const int tiles = p.tiles_p * p.tiles_q;
  const int tile = b % tiles, msplit = b / tiles;
  if (g.cfg[i] == 0) gemm_tn_body<32, 128, 1, 4>(p, smem, tile, msplit);
  else if (g.cfg[i] == 1) gemm_tn_body<128, 32, 4, 1>(p, smem, tile, msplit);
  else gemm_tn_body<128, 128, 2, 2>(p, smem, tile, msplit);
}

static int plan_tn(TnParams& p, int bp, int bq, int target_blocks) {
  p.tiles_p = cdiv(p.P, bp);
  p.tiles_q = cdiv(p.Q, bq);
  const int tiles = p.tiles_p * p.tiles_q;
  int splits = 1;
  if (p.out_mode == 0) {
    splits = target_blocks / tiles;
    const int max_splits = cdiv(p.M, 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  p.m_chunk = cdiv(cdiv(p.M, splits), BMK) * BMK;
  splits = cdiv(p.M, p.m_chunk);
  return tiles * splits;
}

template <int BP, int BQ, int WP, int WQ>
int launch_tn(TnParams& p, hipStream_t stream) {
  const int blocks = plan_tn(p, BP, BQ, 768);
  const int tiles = p.tiles_p * p.tiles_q;
  hipLaunchKernelGGL((gemm_tn_kernel<BP, BQ, WP, WQ>), dim3(tiles, blocks / tiles), dim3(256), 0, stream, p);
  OVLA_CHECK_LAUNCH("ovla_gemm_tn_bf16");
  return OVLA_OK;
}

int fill_params(const ovla_gemm_tn_args* a, TnParams& p, const char* who) {
  if (!(a && a->X && a->Y && a->C)) { ovla_set_error("%s: null pointer", who); return OVLA_EINVAL; }
  if (!(a->M > 0 && a->P > 0 && a->Q > 0)) { ovla_set_error("%s: empty problem M=%d P=%d Q=%d", who, a->M, a->P, a->Q); return OVLA_EINVAL; }
  if ((a->P % 8) || (a->Q % 8) || (a->ldx % 8) || (a->ldy % 8)) { ovla_set_error("%s: P, Q, ldx, ldy must be multiples of 8", who); return OVLA_EINVAL; }
  if (!aligned16(a->X) || !aligned16(a->Y)) { ovla_set_error("%s: X/Y need 16-byte alignment", who); return OVLA_EINVAL; }
  if (a->ldx < a->P || a->ldy < a->Q || a->ldc < a->Q) { ovla_set_error("%s: leading dimension smaller than extent", who); return OVLA_EINVAL; }
  if (a->out_mode < 0 || a->out_mode > 2) { ovla_set_error("%s: out_mode %d", who, a->out_mode); return OVLA_EINVAL; }
  p.X = (const bf16_bits*)a->X; p.Y = (const bf16_bits*)a->Y; p.C = a->C;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldc = a->ldc; p.M = a->M; p.P = a->P; p.Q = a->Q; p.alpha = a->alpha; p.out_mode = a->out_mode;
  return OVLA_OK;
}

}  // namespace

extern "C" int ovla_gemm_tn_bf16(const ovla_gemm_tn_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TnParams p;
  if (int rc = fill_params(a, p, "ovla_gemm_tn_bf16")) return rc;
  if (a->P <= 32) return launch_tn<32, 128, 1, 4>(p, stream);
  if (a->Q <= 32) return launch_tn<128, 32, 4, 1>(p, stream);
  return launch_tn<128, 128, 2, 2>(p, stream);
}

extern "C" int ovla_gemm_tn_grouped(const ovla_gemm_tn_args* problems, int32_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(problems && n >= 1 && n <= OVLA_TN_MAX_GROUP, "ovla_gemm_tn_grouped: 1..%d problems", OVLA_TN_MAX_GROUP);
  TnGroup g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    if (int rc = fill_params(problems + i, g.prob[i], "ovla_gemm_tn_grouped")) return rc;
    const TnParams& p = g.prob[i];
    g.cfg[i] = p.P <= 32 ? 0 : (p.Q <= 32 ? 1 : 2);
    const int bp = g.cfg[i] == 0 ? 32 : 128, bq = g.cfg[i] == 1 ? 32 : 128;
    g.blocks[i] = plan_tn(g.prob[i], bp, bq, 768 / n);
    total += g.blocks[i];
  }
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total), dim3(256), 0, stream, g);
  OVLA_CHECK_LAUNCH("ovla_gemm_tn_grouped");
  return OVLA_OK;
}
