// gemm_nt.hip -- bf16 "NT" GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T (+ A2[M,K2] . B2[N,K2]^T))
//
// Replaces every nn.Linear forward and (through the resident W^T copies) every data-gradient matmul on the
// OpenVLA-OFT action-chunk path; the (A2,B2) K-extension carries peft's LoRA update inside the same accumulator.
// See include/ovla.h for the reference call sites.
//
// Structure (CDNA4):
//   * one workgroup = BM x BN output tile, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile of 16x16x32 bf16 MFMAs;
//   * A/B K-tiles (BK = 64) go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), double buffered:
//     tile t+1 is in flight while tile t feeds the MFMAs, one barrier per K-tile;
//   * LDS image: 128-byte rows, 16-byte chunk index XOR-swizzled with (row>>1)&7 -> ds_read_b128 fragment reads are
//     bank-conflict free; LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address;
//   * K tails / the LoRA rank (K2 = 32) read a 16-byte zero chunk instead of branching;
//   * operands are fed to the MFMA swapped (B fragment as A operand) so each lane owns 4 consecutive output columns of
//     one row: the epilogue does 8-byte loads/stores;
//   * block ids are remapped so each XCD (private L2) owns a contiguous band of tiles, grouped 8 tile-rows deep.
#include "common.h"
#include <stdarg.h>

namespace {

constexpr int BK = 64;

__device__ __attribute__((aligned(16))) bf16_bits g_ovla_zero_chunk[8];

struct GemmParams {
  const bf16_bits *A, *B, *A2, *B2;
  int64_t lda, ldb, lda2, ldb2;
  bf16_bits *C, *Cpre;
  int64_t ldc;
  const bf16_bits *bias, *colscale, *residual;
  int64_t ldr;
  const bf16_bits *film_gamma, *film_beta;
  int film_rows;
  int M, N, K, K2, k2_group_n, act, split_k;
  float alpha;
  float* ws;
  int tiles_m, tiles_n, T1, T2;
};

// Stage ROWS x 64 bf16 of a row-major [rows, ld] matrix into an LDS tile (swizzled 128-byte rows) with LDS-DMA.
template <int ROWS, int NW>
OVLA_DEV void stage_tile(const bf16_bits* __restrict__ G, int64_t ld, int row0, int row_last, int k0, int K,
                         bf16_bits* lds_tile, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 8 / NW;
  static_assert(PER_WAVE * 8 * NW == ROWS, "tile rows must split evenly over the waves");
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rbase = (wave * PER_WAVE + i) * 8;  // wave-uniform: one instruction writes rows rbase..rbase+7
    const int r = rbase + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);    // logical 8-element chunk that lands in physical slot lane&7
    const int kk = k0 + c * 8;
    int gr = row0 + r;
    gr = gr < row_last ? gr : row_last;
    const bf16_bits* src = (kk < K) ? (G + (int64_t)gr * ld + kk) : g_ovla_zero_chunk;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + rbase * BK), 16, 0, 0);
  }
}

OVLA_DEV bf16x8_bits lds_frag(const bf16_bits* tile, int row, int chunk) {
  const int phys = chunk ^ ((row >> 1) & 7);
  return *reinterpret_cast<const bf16x8_bits*>(tile + row * BK + phys * 8);
}

// Epilogue on 4 consecutive columns n..n+3 of row m.  Every step rounds to bf16, as the reference's separate ops do.
OVLA_DEV void epilogue_store(const GemmParams& p, int m, int n, f32x4 v) {
  v *= p.alpha;
  if (p.bias) {
    const bf16x4_bits b = *reinterpret_cast<const bf16x4_bits*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j] + bf2f((bf16_bits)b[j]));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j]);
  }
  if (p.Cpre) {
    bf16x4_bits o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (short)f2bf(v[j]);
    *reinterpret_cast<bf16x4_bits*>(p.Cpre + (int64_t)m * p.ldc + n) = o;
  }
  if (p.act != OVLA_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(apply_act(v[j], p.act));
  }
  if (p.colscale) {
    const bf16x4_bits s = *reinterpret_cast<const bf16x4_bits*>(p.colscale + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j] * bf2f((bf16_bits)s[j]));
  }
  if (p.residual) {
    const bf16x4_bits r = *reinterpret_cast<const bf16x4_bits*>(p.residual + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j] + bf2f((bf16_bits)r[j]));
  }
  if (p.film_gamma) {
    const int64_t off = (int64_t)(m / p.film_rows) * p.N + n;
    const bf16x4_bits g = *reinterpret_cast<const bf16x4_bits*>(p.film_gamma + off);
    const bf16x4_bits b = *reinterpret_cast<const bf16x4_bits*>(p.film_beta + off);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float one_plus = bfround(1.0f + bf2f((bf16_bits)g[j]));
      v[j] = bfround(bfround(v[j] * one_plus) + bf2f((bf16_bits)b[j]));
    }
  }
  bf16x4_bits o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (short)f2bf(v[j]);
  *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int TILE_ELEMS = (BM + BN) * BK;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_bits* smem = reinterpret_cast<bf16_bits*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> (split, tile_m, tile_n): XCD-contiguous bands, then 8-deep row groups ------------------------
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int split = bid / tiles_mn;
  const int t_mn = bid - split * tiles_mn;
  constexpr int GROUP = 8;
  const int group_sz = GROUP * p.tiles_n;
  const int gid = t_mn / group_sz;
  const int first_m = gid * GROUP;
  const int gm = (p.tiles_m - first_m) < GROUP ? (p.tiles_m - first_m) : GROUP;
  const int in_group = t_mn - gid * group_sz;
  const int tm = first_m + in_group % gm;
  const int tn = in_group / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int T = p.T1 + p.T2;
  int t_begin = 0, t_end = T;
  if (p.split_k > 1) {
    const int chunk = (T + p.split_k - 1) / p.split_k;
    t_begin = split * chunk;
    t_end = t_begin + chunk < T ? t_begin + chunk : T;
  }
  const int a2_col0 = p.k2_group_n > 0 ? (n0 / p.k2_group_n) * p.K2 : 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int t, int buf) {
    bf16_bits* sA = smem + buf * TILE_ELEMS;
    bf16_bits* sB = sA + BM * BK;
    if (t < p.T1) {
      const int k0 = t * BK;
      stage_tile<BM, NW>(p.A, p.lda, m0, p.M - 1, k0, p.K, sA, wave, lane);
      stage_tile<BN, NW>(p.B, p.ldb, n0, p.N - 1, k0, p.K, sB, wave, lane);
    } else {
      const int k0 = (t - p.T1) * BK;
      stage_tile<BM, NW>(p.A2 + a2_col0, p.lda2, m0, p.M - 1, k0, p.K2, sA, wave, lane);
      stage_tile<BN, NW>(p.B2, p.ldb2, n0, p.N - 1, k0, p.K2, sB, wave, lane);
    }
  };

  if (t_begin < t_end) stage(t_begin, 0);
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA for tile t has landed
    __syncthreads();                                   // ... everyone's has; and buf^1 is no longer being read
    if (t + 1 < t_end) stage(t + 1, buf ^ 1);
    const bf16_bits* sA = smem + buf * TILE_ELEMS;
    const bf16_bits* sB = sA + BM * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_bits a[MT], b[NT];
      const int chunk = s * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = lds_frag(sA, wm * WTM + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = lds_frag(sB, wn * WTN + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // MFMA layout (operands swapped): lane owns C[m][n..n+3] with m = tile row (lane&15), n = 4*(lane>>4).
  if (p.split_k > 1) {  // raw fp32 partials; the reduce kernel applies the epilogue
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WTM + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 16 + 4 * (lane >> 4);
        if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(p.ws + ((int64_t)split * p.M + m) * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  // Accumulators go through a wave-private fp32 LDS slab (32 rows at a time, row stride WTN+4 floats: conflict-free
  // ds_write_b128) so that ONE rolled loop applies the epilogue and writes whole 128-byte row segments.
  constexpr int LDSW = WTN + 4;
  constexpr int RM = MT < 2 ? MT : 2;  // m-tiles per round
  constexpr int QUADS = WTN / 4;       // 4-column groups per sub-tile row
  float* wstage = reinterpret_cast<float*>(smem_raw) + wave * (RM * 16 * LDSW);
#pragma unroll
  for (int round = 0; round < MT / RM; ++round) {
    __syncthreads();  // main-loop LDS reads (round 0) / previous round's read-back are done
#pragma unroll
    for (int ii = 0; ii < RM; ++ii)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<f32x4*>(wstage + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[round * RM + ii][j];
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < RM * 16 * QUADS / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / QUADS, c4 = idx % QUADS;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c4 * 4);
      const int m = m0 + wm * WTM + round * RM * 16 + row;
      const int n = n0 + wn * WTN + c4 * 4;
      if (m < p.M && n < p.N) epilogue_store(p, m, n, v);
    }
  }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmParams p) {
  const int64_t quads = (int64_t)p.M * (p.N / 4);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(q / (p.N / 4));
    const int n = (int)(q % (p.N / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split_k; ++s) v += *reinterpret_cast<const f32x4*>(p.ws + ((int64_t)s * p.M + m) * p.N + n);
    epilogue_store(p, m, n, v);
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(GemmParams& p, hipStream_t stream) {
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.N, BN);
  if (p.k2_group_n > 0 && (p.k2_group_n % BN) != 0) {
    ovla_set_error("ovla_gemm_bf16: k2_group_n=%d is not a multiple of the N tile %d", p.k2_group_n, BN);
    return OVLA_EINVAL;
  }
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16_bits);
  auto kern = gemm_nt_kernel<BM, BN, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * splits));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, stream, p);
  OVLA_CHECK_LAUNCH("ovla_gemm_bf16");
  if (splits > 1) {
    const int64_t quads = (int64_t)p.M * (p.N / 4);
    int blocks = cdiv(quads, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(split-k reduce)");
  }
  return OVLA_OK;
}

}  // namespace

extern "C" int64_t ovla_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  return split_k > 1 ? (int64_t)split_k * M * N * 4 : 0;
}

extern "C" int ovla_gemm_bf16(const ovla_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a != nullptr, "ovla_gemm_bf16: null args");
  OVLA_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "ovla_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  OVLA_REQUIRE(a->A && a->B && a->C, "ovla_gemm_bf16: null A/B/C");
  OVLA_REQUIRE((a->K % 8) == 0 && (a->N % 8) == 0, "ovla_gemm_bf16: K=%d and N=%d must be multiples of 8", a->K, a->N);
  OVLA_REQUIRE((a->lda % 8) == 0 && (a->ldb % 8) == 0 && (a->ldc % 4) == 0, "ovla_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
  OVLA_REQUIRE(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "ovla_gemm_bf16: leading dimension smaller than extent");
  OVLA_REQUIRE(aligned16(a->A) && aligned16(a->B) && (((uintptr_t)a->C) & 7) == 0, "ovla_gemm_bf16: A/B need 16-byte, C 8-byte alignment");
  if (a->K2 > 0) {
    OVLA_REQUIRE(a->A2 && a->B2, "ovla_gemm_bf16: K2>0 but A2/B2 null");
    OVLA_REQUIRE((a->K2 % 8) == 0 && (a->lda2 % 8) == 0 && (a->ldb2 % 8) == 0, "ovla_gemm_bf16: K2/lda2/ldb2 must be multiples of 8");
    OVLA_REQUIRE(aligned16(a->A2) && aligned16(a->B2), "ovla_gemm_bf16: A2/B2 need 16-byte alignment");
    OVLA_REQUIRE(a->ldb2 >= a->K2, "ovla_gemm_bf16: ldb2 < K2");
  }
  if (a->residual) OVLA_REQUIRE((a->ldr % 4) == 0 && (((uintptr_t)a->residual) & 7) == 0, "ovla_gemm_bf16: residual alignment");
  if (a->bias) OVLA_REQUIRE((((uintptr_t)a->bias) & 7) == 0, "ovla_gemm_bf16: bias alignment");
  if (a->colscale) OVLA_REQUIRE((((uintptr_t)a->colscale) & 7) == 0, "ovla_gemm_bf16: colscale alignment");
  if (a->film_gamma) OVLA_REQUIRE(a->film_beta && a->film_rows > 0, "ovla_gemm_bf16: FiLM needs beta and film_rows");
  if (a->split_k > 1) OVLA_REQUIRE(a->workspace != nullptr && aligned16(a->workspace), "ovla_gemm_bf16: split_k needs a 16-byte aligned workspace");

  GemmParams p;
  p.A = (const bf16_bits*)a->A; p.B = (const bf16_bits*)a->B;
  p.A2 = (const bf16_bits*)a->A2; p.B2 = (const bf16_bits*)a->B2;
  p.lda = a->lda; p.ldb = a->ldb; p.lda2 = a->lda2; p.ldb2 = a->ldb2;
  p.C = (bf16_bits*)a->C; p.Cpre = (bf16_bits*)a->C_pre; p.ldc = a->ldc;
  p.bias = (const bf16_bits*)a->bias; p.colscale = (const bf16_bits*)a->colscale;
  p.residual = (const bf16_bits*)a->residual; p.ldr = a->ldr;
  p.film_gamma = (const bf16_bits*)a->film_gamma; p.film_beta = (const bf16_bits*)a->film_beta; p.film_rows = a->film_rows;
  p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->K2 > 0 ? a->K2 : 0;
  p.k2_group_n = a->k2_group_n; p.act = a->act; p.split_k = a->split_k > 1 ? a->split_k : 1;
  p.ws = (float*)a->workspace;
  p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
  p.T1 = cdiv(p.K, BK); p.T2 = p.K2 > 0 ? cdiv(p.K2, BK) : 0;
  if (p.split_k > p.T1 + p.T2) p.split_k = p.T1 + p.T2;

  int tile = a->tile;
  if (tile == 0) tile = (a->M <= 64) ? 2 : 1;
  switch (tile) {
    case 1: return launch_cfg<128, 128, 2, 2>(p, stream);
    case 2: return launch_cfg<64, 128, 1, 4>(p, stream);
    case 3: return launch_cfg<256, 128, 4, 2>(p, stream);
    default: ovla_set_error("ovla_gemm_bf16: unknown tile id %d", tile); return OVLA_EINVAL;
  }
}
