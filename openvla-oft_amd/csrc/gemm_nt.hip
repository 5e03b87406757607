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
#include <type_traits>
#include <cstdlib>
#include <algorithm>
#include <stdarg.h>

namespace {

constexpr int BK = 64;

__device__ __attribute__((aligned(16))) bf16_bits g_ovla_zero_chunk[8];

// Timing ablations are not part of the product kernel: in the default build OVLA_DBG(bit) is the constant 0 and every ablation branch
// folds away; `build.sh ablate` compiles a separate library with -DOVLA_GEMM_ABLATE where it tests GemmParams::dbg (= args.tile / 1000).
#ifdef OVLA_GEMM_ABLATE
#define OVLA_DBG(bit) (p.dbg & (bit))
#else
#define OVLA_DBG(bit) 0
#endif

struct GemmParams {
  const bf16_bits *A, *B, *A2, *B2;
  int64_t lda, ldb, lda2, ldb2;
  bf16_bits *C, *Cpre;
  int64_t ldc;
  const bf16_bits *bias, *colscale, *residual;
  int64_t ldr;
  const bf16_bits *film_gamma, *film_beta;
  int film_rows;
  int M, N, K, K2, k2_group_n, a_group_n, act, split_k;
  float alpha;
  float* ws;
  int tiles_m, tiles_n, T1, T2;
  int fast_addr;  // 1: every staged byte offset fits in 32 bits (host-checked)
  const bf16_bits* dact_src; int64_t ld_dact; int dact_mode, dact_act;   // backward epilogues (ovla.h)
  const bf16_bits *rope_cos, *rope_sin; int rope_S, rope_cols;           // forward RoPE on columns [0, rope_cols), head_dim 128
  int fast_epi;  // host: no FiLM / backward epilogue / RoPE and 16-byte aligned operands -> the unrolled read-back path applies
  int fast_swiglu_bwd;  // host: dact_mode 2 with 16-byte aligned [M, 2N] operands and nothing else in the epilogue -> SwiGLU' in the unrolled read-back
  int dbg;  // timing ablations, compiled in ONLY with -DOVLA_GEMM_ABLATE (build.sh ablate -> libovla_hip_ablate.so, tools/gemm_ablate.py): bit0 = stage only the first two K tiles, bit1 = read fragments once, bit2 = every workgroup stages tile (0,0): all L2 hits, bit3 = no epilogue, bit4 = epilogue without its stores, bit5 = nontemporal stores, bit6 = force the LDS-staged epilogue, bit7 / bit8 / bit9 = no (A and B) / B / A fragment reads after the first K tile
  int full_tiles, rem_tiles, rem_splits;  // hybrid schedule: tiles >= full_tiles are split rem_splits ways along K
  // RMSNorm fold (ovla.h): producer writes per-row sums of squares of its output's 64-column groups; consumer scales the accumulator by rstd[m]
  float* rowsq_out;
  const float* rowscale_part; int rowscale_slots; float rowscale_eps; float* rowscale_r;
  int hyb_cnt_n;
  unsigned* hyb_cnt;   // per-remainder-tile arrival counters (all zero between launches) for the in-launch reduce; nullptr: separate reduce kernel
};

// sum of squares of 4 bf16-rounded outputs, reduced over the 16 lanes that hold one row's 64-column group (lanes aligned to 16)
OVLA_DEV float rowsq16(f32x4 v) {
  float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
  return s;
}

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

// Loop-invariant part of the staging addresses: per-lane 32-bit BYTE offsets (row clamp and chunk swizzle baked in).
// Inside the K loop a stage is then `uniform base + k0` (scalar) plus these offsets: no vector address math per tile.
template <int ROWS, int NW>
OVLA_DEV void stage_offsets(uint32_t (&off)[ROWS / 8 / NW], int64_t ld, int row0, int row_last, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 8 / NW;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int r = (wave * PER_WAVE + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_last ? gr : row_last;
    off[i] = (uint32_t)(((int64_t)gr * ld + c * 8) * 2);
  }
}

template <int ROWS, int NW>
OVLA_DEV void stage_tile_fast(const char* __restrict__ base_k /* uniform: matrix base + 2*k0 */, const uint32_t (&off)[ROWS / 8 / NW],
                              bf16_bits* lds_tile, int wave) {
  constexpr int PER_WAVE = ROWS / 8 / NW;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rbase = (wave * PER_WAVE + i) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_k + off[i]),
                                     (__attribute__((address_space(3))) void*)(lds_tile + rbase * BK), 16, 0, 0);
  }
}

template <int N, int I = 0, typename F>
OVLA_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

OVLA_DEV bf16x8_bits lds_frag(const bf16_bits* tile, int row, int chunk) {
  const int phys = chunk ^ ((row >> 1) & 7);
  return *reinterpret_cast<const bf16x8_bits*>(tile + row * BK + phys * 8);
}
#ifdef OVLA_GEMM_ABLATE
// timing ablations on the fragment reads (bit7: no A or B fragment read after the first K tile; bit8: no B fragment read; bit9: no A fragment read):
// what the LDS read traffic costs the main loop
#define OVLA_FRAG(var, expr, bits) do { if (!(p.dbg & (bits)) || t == t_begin) var = (expr); } while (0)
#else
#define OVLA_FRAG(var, expr, bits) var = (expr)
#endif

// Epilogue on 4 consecutive columns n..n+3 of row m.  Every step rounds to bf16, as the reference's separate ops do.
OVLA_DEV f32x4 epilogue_store(const GemmParams& p, int m, int n, f32x4 v) {   // returns the stored (bf16-rounded) values
  v *= p.alpha;
  if (p.bias) {
    const bf16x4_bits b = *reinterpret_cast<const bf16x4_bits*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j] + bf2f((bf16_bits)b[j]));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j]);
  }
  if (p.Cpre && !p.film_gamma) {   // value before the activation (saved for the backward)
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
    if (p.Cpre) {   // with FiLM, C_pre receives the value BEFORE the modulation (needed for d gamma)
      bf16x4_bits o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (short)f2bf(v[j]);
      *reinterpret_cast<bf16x4_bits*>(p.Cpre + (int64_t)m * p.ldc + n) = o;
    }
    const int64_t off = (int64_t)(m / p.film_rows) * p.N + n;
    const bf16x4_bits g = *reinterpret_cast<const bf16x4_bits*>(p.film_gamma + off);
    const bf16x4_bits b = *reinterpret_cast<const bf16x4_bits*>(p.film_beta + off);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float one_plus = bfround(1.0f + bf2f((bf16_bits)g[j]));
      v[j] = bfround(bfround(v[j] * one_plus) + bf2f((bf16_bits)b[j]));
    }
  }
  if (p.dact_mode == 1) {          // dz = dh * act'(z): same arithmetic as act_bwd_kernel on the bf16-rounded dh
    const bf16x4_bits z = *reinterpret_cast<const bf16x4_bits*>(p.dact_src + (int64_t)m * p.ld_dact + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bfround(v[j]) * act_grad(bf2f((bf16_bits)z[j]), p.dact_act);
  } else if (p.dact_mode == 2) {   // SwiGLU backward (swiglu_bwd_kernel's arithmetic): two outputs per element
    const bf16x4_bits g4 = *reinterpret_cast<const bf16x4_bits*>(p.dact_src + (int64_t)m * p.ld_dact + n);
    const bf16x4_bits u4 = *reinterpret_cast<const bf16x4_bits*>(p.dact_src + (int64_t)m * p.ld_dact + p.N + n);
    bf16x4_bits du;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = bfround(v[j]), g = bf2f((bf16_bits)g4[j]), u = bf2f((bf16_bits)u4[j]);
      const float s = sigmoidf_(g);
      du[j] = (short)f2bf(d * bfround(g * s));
      v[j] = d * u * (s * (1.f + g * (1.f - s)));
    }
    *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + p.N + n) = du;
  }
  bf16x4_bits o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = (short)f2bf(v[j]); v[j] = bf2f((bf16_bits)o[j]); }
  if (OVLA_DBG(16) && v[0] != 123.456f) return v;          // timing ablation: the whole epilogue except the store
  if (OVLA_DBG(32)) {                                          // timing ablation: write-through store that does not stay in this XCD's L2
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + n));
    return v;
  }
  *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
  return v;
}

// RoPE epilogue (head_dim 128): v = this lane's 4 accumulator sums at columns n..n+3 of row m, vp = the sums at the rotation
// partner columns n +- 64 (same head).  y = bf16(acc) is the q|k projection output the reference rotates; arithmetic and rounding
// points are rope_kernel's (elementwise.hip): lo' = bf16(a c) + bf16(-b s), hi' = bf16(b c) + bf16(a s).
OVLA_DEV void rope_store(const GemmParams& p, int m, int n, f32x4 v, f32x4 vp) {
  const int c = n & 127;                  // column within the head
  const bool lo = c < 64;
  const int pos = m % p.rope_S;
  const bf16x4_bits cs = *reinterpret_cast<const bf16x4_bits*>(p.rope_cos + (int64_t)pos * 64 + (c & 63));
  const bf16x4_bits sn = *reinterpret_cast<const bf16x4_bits*>(p.rope_sin + (int64_t)pos * 64 + (c & 63));
  bf16x4_bits o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x = bfround(v[j]), y = bfround(vp[j]);
    const float cc = bf2f((bf16_bits)cs[j]), sv = bf2f((bf16_bits)sn[j]);
    o[j] = (short)f2bf(lo ? bfround(x * cc) + bfround(-y * sv) : bfround(x * cc) + bfround(y * sv));
  }
  *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
}

template <int BM, int BN>
OVLA_DEV void hybrid_reduce_quads(const GemmParams& p, int rt, int q0, int qstride);

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int TILE_ELEMS = (BM + BN) * BK;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_bits* smem = reinterpret_cast<bf16_bits*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
#ifdef OVLA_GEMM_STAMPS   // in-kernel timeline (100 MHz wall clock) of a plain launch into the unused workspace: tools/gemm_stamps.py
#define OVLA_STAMP(k) do { if (tid == 0) reinterpret_cast<unsigned long long*>(p.ws)[(int64_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define OVLA_STAMP(k) do { } while (0)
#endif
  OVLA_STAMP(0);

  // ---- block -> work unit ------------------------------------------------------------------------------------
  // Plain / split-K launch: blocks [0, tiles*splits) ; block ids are remapped so each XCD (private L2) owns a contiguous
  // band of tiles.  Hybrid launch (rem_tiles > 0): blocks [0, full_tiles) are whole tiles (whole rounds of the chip);
  // the remaining tiles -- a partial round that would leave most CUs idle -- are split rem_splits ways along K, one block
  // per (tile, K part), partial sums going to the fp32 workspace for gemm_hybrid_reduce_kernel.
  int bid = blockIdx.x;
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int remap_n = p.rem_tiles > 0 ? p.full_tiles : (int)gridDim.x;
  int split = 0, t_mn, rem_unit = -1;
  if (bid < remap_n) {
#ifdef OVLA_XCD_ROT   // experiment: give physical XCD x the band of XCD (x + ROT) & 7 -- does a slow band follow the silicon or the data?
    const int xcd = ((bid & 7) + OVLA_XCD_ROT) & 7, q = remap_n >> 3, r = remap_n & 7;
#else
    const int xcd = bid & 7, q = remap_n >> 3, r = remap_n & 7;
#endif
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    split = bid / tiles_mn;
    t_mn = bid - split * tiles_mn;
  } else {
    // remainder units, split-major and XCD-banded like the full tiles: the workgroups an XCD runs together are the SAME
    // K part of neighbouring tiles, so they share A / B panels through that XCD's L2 (tile-major order put the K parts of
    // one tile on 8 different XCDs: nothing shared).  Slab index stays (tile, split) for the reduce kernel.
    int u = bid - p.full_tiles;
    const int nrem = p.rem_tiles * p.rem_splits;
    const int xcd = u & 7, q = nrem >> 3, r = nrem & 7;
    u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (u >> 3);
    split = u / p.rem_tiles;
    const int rt = u - split * p.rem_tiles;
    t_mn = p.full_tiles + rt;
    rem_unit = rt * p.rem_splits + split;
  }
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
  const int nsplit = rem_unit >= 0 ? p.rem_splits : p.split_k;
  if (nsplit > 1) {
    const int chunk = (T + nsplit - 1) / nsplit;
    t_begin = split * chunk;
    t_end = t_begin + chunk < T ? t_begin + chunk : T;
  }
  const int a2_col0 = p.k2_group_n > 0 ? (n0 / p.k2_group_n) * p.K2 : 0;
  const bf16_bits* Ablk = p.A + (p.a_group_n > 0 ? (int64_t)(n0 / p.a_group_n) * p.K : 0);   // block-diagonal: group's A columns

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // RMSNorm fold, consumer side (128x128 config only): rstd of this tile's 128 rows from the producer's per-64-column sums of squares, into
  // LDS behind the K-tile buffers / epilogue slabs.  Two lanes per row, 8 16-byte loads each; they are in flight with the first K tile's
  // LDS-DMA and waited for by the same vmcnt(0), and the K loop's first barrier publishes s_rstd: the prologue costs no round trip of its own.
  constexpr bool NORMFOLD = (BM == 128 && BN == 128 && WM == 2 && WN == 2);
  constexpr size_t NF_KB = (size_t)2 * TILE_ELEMS * sizeof(bf16_bits), NF_EB = (size_t)NW * (MT < 2 ? MT : 2) * 16 * (WTN + 4) * sizeof(float);
  float* s_rstd = reinterpret_cast<float*>(smem_raw + (NF_KB > NF_EB ? NF_KB : NF_EB));
  bool rowscale = false;
  uint32_t offA[BM / 8 / NW], offB[BN / 8 / NW];
  stage_offsets<BM, NW>(offA, p.lda, OVLA_DBG(4) ? 0 : m0, p.M - 1, wave, lane);
  stage_offsets<BN, NW>(offB, p.ldb, OVLA_DBG(4) ? 0 : n0, p.N - 1, wave, lane);
  const int t_fast = p.fast_addr ? p.K / BK : 0;   // K tiles that lie entirely inside [0, K): no zero-chunk select needed

  auto stage = [&](int t, int buf) {
    bf16_bits* sA = smem + buf * TILE_ELEMS;
    bf16_bits* sB = sA + BM * BK;
    if (t < t_fast) {
      stage_tile_fast<BM, NW>(reinterpret_cast<const char*>(Ablk) + (int64_t)t * (BK * 2), offA, sA, wave);
      stage_tile_fast<BN, NW>(reinterpret_cast<const char*>(p.B) + (int64_t)t * (BK * 2), offB, sB, wave);
    } else if (t < p.T1) {
      const int k0 = t * BK;
      stage_tile<BM, NW>(Ablk, p.lda, m0, p.M - 1, k0, p.K, sA, wave, lane);
      stage_tile<BN, NW>(p.B, p.ldb, n0, p.N - 1, k0, p.K, sB, wave, lane);
    } else {
      const int k0 = (t - p.T1) * BK;
      stage_tile<BM, NW>(p.A2 + a2_col0, p.lda2, m0, p.M - 1, k0, p.K2, sA, wave, lane);
      stage_tile<BN, NW>(p.B2, p.ldb2, n0, p.N - 1, k0, p.K2, sB, wave, lane);
    }
  };

  // MFMA rows r = (k-substep s, m-tile i).  Fragment reads run one row ahead of the MFMAs, and the LAST row of every K
  // tile is deferred across the barrier: its fragments stay in registers (a_def, b1) and its NT MFMAs issue right after
  // the next tile's first fragment reads, so the matrix pipe has work while those reads (and the LDS-DMA issue) are in
  // flight instead of every wave draining its pipeline at each barrier.
  constexpr int BPR = (NT + MT - 1) / MT;  // next-substep B fragments fetched per row
  // Column map of the wave's NT accumulator tiles.  Plain: tile j = columns wn*WTN + 16 j.  RoPE tiles of the 2x4 layout (the q | k
  // heads of the fused q|k|v projection, head_dim 128 = two 64-column wave strips): wave (h = wn >> 1, w2 = wn & 1) takes columns
  // h*128 + 32 w2 + [0, 32) AND their rotation partners h*128 + 64 + 32 w2 + [0, 32), i.e. tile j sits at 16 (j & 1) + 64 (j >> 1): the
  // partner of tile j is tile j ^ 2 of the SAME wave, 32 floats away in the same row of its own epilogue slab -- the rotation needs no
  // other wave's data and no extra barrier, and 8 consecutive slab columns are still 8 consecutive output columns (16-byte stores).
  bool rope_tile = false;
  // (the same map serves the 128x128 tile on 2x2 waves: one head per tile, wave wn takes 32 wn + [0, 32) and + 64)
  constexpr bool ROPE_LAYOUT = WTN == 64 && ((WN == 4 && BN == 256) || (WN == 2 && BN == 128));
  if constexpr (ROPE_LAYOUT) rope_tile = p.rope_cos != nullptr && n0 < p.rope_cols;
  int bcol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bcol[j] = rope_tile ? ((wn >> 1) * 128 + (wn & 1) * 32 + (j & 1) * 16 + (j >> 1) * 64) : (wn * WTN + j * 16);
  const int arow = wm * WTM + (lane & 15), brl = lane & 15, cq = lane >> 4;
  bf16x8_bits b0[NT], b1[NT];
  bf16x8_bits a_def = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NT; ++j) b1[j] = a_def;

  // LDS-DMA issue is expensive for the issuing wave (~60-150 cycles per 1 KiB piece), so the next tile's pieces are not
  // issued as one burst after the barrier (all 8 waves stuck in issue, matrix pipe idle) but spread over the first MT row
  // slots, PPS per slot, where the SIMD partner's MFMAs cover them.  Only K tiles on the fast-address path are spread
  // (`SPREAD` loop); the K tail and the LoRA K-extension tiles keep the simple stage-after-barrier scheme.
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, PP = PA + PB;
  constexpr int PPS = (PP + MT - 1) / MT;
  // SPREAD2 (OVLA_GEMM_SPREAD2, experiment): when the pieces are no more than the m-tiles (256x256 on 8 waves: 8 and 8), deal ONE piece to every
  // SECOND of the 2 MT row slots of a K tile instead of one to each of the first MT: 8 MFMAs (128 matrix-pipe cycles) between two pieces
#ifdef OVLA_GEMM_SPREAD2
  constexpr bool SPREAD2 = (PP <= MT);
#else
  constexpr bool SPREAD2 = false;
#endif
  auto n_pieces_at = [](int slot) constexpr -> int {   // pieces issued at row slot `slot` (0 .. 2 MT - 1)
    if (SPREAD2) return (slot % 2 == 0 && slot / 2 < PP) ? 1 : 0;
    if (slot >= MT) return 0;
    return ((slot + 1) * PPS <= PP) ? PPS : ((slot * PPS < PP) ? PP - slot * PPS : 0);
  };
  auto tile_body = [&](const int t, auto spread_tag) {
    constexpr bool SPREAD = decltype(spread_tag)::value;
    const int buf = (t - t_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA for tile t has landed
    __syncthreads();                                   // ... everyone's has; and buf^1 is no longer being read
#ifdef OVLA_GEMM_STAMPS
    if (t == t_begin) OVLA_STAMP(2);
#endif
    if constexpr (!SPREAD) {
      if (t + 1 < t_end && !(OVLA_DBG(1) && t > t_begin)) stage(t + 1, buf ^ 1);
    }
    const bf16_bits* sA = smem + (OVLA_DBG(2) ? 0 : buf) * TILE_ELEMS;
    const bf16_bits* sB = sA + BM * BK;
    bf16_bits* nA = smem + (buf ^ 1) * TILE_ELEMS;
    bf16_bits* nB = nA + BM * BK;
    const char* gA = reinterpret_cast<const char*>(Ablk) + (int64_t)(t + 1) * (BK * 2);
    const char* gB = reinterpret_cast<const char*>(p.B) + (int64_t)(t + 1) * (BK * 2);
    auto pieces = [&](auto slot_tag) {
      constexpr int SLOT = decltype(slot_tag)::value;
      constexpr int Q0 = SPREAD2 ? SLOT / 2 : SLOT * PPS, Q1 = SPREAD2 ? (SLOT % 2 == 0 ? SLOT / 2 + 1 : SLOT / 2) : (SLOT + 1) * PPS;
#pragma unroll
      for (int q = Q0; q < Q1 && q < PP; ++q) {
        if (q < PA)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + offA[q < PA ? q : 0]),
                                           (__attribute__((address_space(3))) void*)(nA + (wave * PA + q) * 8 * BK), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + offB[q >= PA ? q - PA : 0]),
                                           (__attribute__((address_space(3))) void*)(nB + (wave * PB + (q - PA)) * 8 * BK), 16, 0, 0);
      }
    };
    constexpr int NP0 = n_pieces_at(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) OVLA_FRAG(b0[j], lds_frag(sB, bcol[j] + brl, cq), 128 | 256);
    bf16x8_bits a_cur = a_def;
    OVLA_FRAG(a_cur, lds_frag(sA, arow, cq), 128 | 512);
    if constexpr (SPREAD) pieces(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_setprio(1);
    // deferred last row of the previous tile (zeros on the first pass)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      acc[MT - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a_def, acc[MT - 1][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
    if constexpr (SPREAD) __builtin_amdgcn_sched_group_barrier(0x020, NP0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    auto row = [&](auto r_tag) {
      constexpr int r = decltype(r_tag)::value;
      constexpr int sub = r / MT, i = r % MT;
      bf16x8_bits a_nxt = a_cur;
      OVLA_FRAG(a_nxt, lds_frag(sA, arow + ((r + 1) % MT) * 16, ((r + 1) / MT) * 4 + cq), 128 | 512);
      if constexpr (sub == 0) {
#pragma unroll
        for (int jj = 0; jj < BPR; ++jj)
          if (i * BPR + jj < NT) OVLA_FRAG(b1[i * BPR + jj], lds_frag(sB, bcol[i * BPR + jj] + brl, 4 + cq), 128 | 256);
      }
      constexpr int NPR = n_pieces_at(r + 1);
      if constexpr (SPREAD && NPR > 0) pieces(std::integral_constant<int, r + 1>{});
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sub == 0 ? b0[j] : b1[j], a_cur, acc[i][j], 0, 0, 0);
      if constexpr (sub == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1 + BPR, 0);  // DS reads of the next row
      else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if constexpr (SPREAD && NPR > 0) __builtin_amdgcn_sched_group_barrier(0x020, NPR, 0);  // LDS-DMA pieces of the next tile
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);                       // this row's MFMAs
      a_cur = a_nxt;
    };
    static_for<2 * MT - 1>(row);
    a_def = a_cur;   // row 2*MT-1 (substep 1, m-tile MT-1): issued after the next barrier
    __builtin_amdgcn_s_setprio(0);
  };

  OVLA_STAMP(1);
  if (t_begin < t_end) stage(t_begin, 0);
  // (the fold's prologue loads are issued AFTER the first K tile's LDS-DMA, so the two latencies overlap: issued first, their wait delayed the DMA by a
  // whole memory round trip -- measured: the removed norm launches' time came back inside the GEMMs)
  if constexpr (NORMFOLD) {
    rowscale = p.rowscale_part != nullptr;
    if (rowscale) {
      const int row = tid >> 1, half = tid & 1, m = m0 + row;
      float ssum = 0.f;
      if (m < p.M) {
        const int per = p.rowscale_slots >> 1;             // slots this lane adds, in slot order (rowscale_slots % 8 == 0: host-checked)
        const float* src = p.rowscale_part + (int64_t)m * p.rowscale_slots + half * per;
        for (int j0 = 0; j0 < per; j0 += 32) {   // batches of eight independent 16-byte loads (all in flight together), added in slot order
          f32x4 q[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) q[u] = (j0 + 4 * u < per) ? *reinterpret_cast<const f32x4*>(src + j0 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 8; ++u) { ssum += q[u][0]; ssum += q[u][1]; ssum += q[u][2]; ssum += q[u][3]; }
        }
      }
      ssum += __shfl_xor(ssum, 1, 64);
      const float r = m < p.M ? rsqrtf(ssum / (float)(p.rowscale_slots * 64) + p.rowscale_eps) : 0.f;
      s_rstd[row] = r;
      if (half == 0 && m < p.M && tn == 0 && split == 0 && p.rowscale_r) p.rowscale_r[m] = r;   // for the hybrid-remainder reduce kernel
    }
  }

  int t = t_begin;
  {
    const int lim = t_end < t_fast ? t_end : t_fast;   // spread loop: tile t+1 exists and is a fast-address tile
    if constexpr (NW >= 8) {   // 4-wave configs run 2 workgroups per CU, which already interleave; spreading only costs them
      if (!OVLA_DBG(1))
        for (; t + 1 < lim; ++t) tile_body(t, std::true_type{});
    }
  }
  for (; t < t_end; ++t) tile_body(t, std::false_type{});
#pragma unroll
  for (int j = 0; j < NT; ++j)
    acc[MT - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a_def, acc[MT - 1][j], 0, 0, 0);

  OVLA_STAMP(3);
  if (OVLA_DBG(8)) {   // timing ablation: no epilogue at all (one store per lane keeps the accumulators alive)
    if (acc[0][0][0] == 123.456f) p.C[0] = 1;
    return;
  }
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // MFMA layout (operands swapped): lane owns C[m][n..n+3] with m = tile row (lane&15), n = 4*(lane>>4).
  if (rem_unit >= 0) {  // hybrid remainder unit: tile-local fp32 slab [rem_unit][BM][BN]
    float* slab = p.ws + (int64_t)rem_unit * (BM * BN);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<f32x4*>(slab + (wm * WTM + i * 16 + (lane & 15)) * BN + bcol[j] + 4 * (lane >> 4)) = acc[i][j];
    if (p.hyb_cnt == nullptr) return;          // the separate gemm_hybrid_reduce_kernel launch does the rest
    // In-launch reduce: the LAST of a tile's rem_splits units to arrive adds the slabs (its own included, from memory, in slab order: the same
    // bits as the reduce kernel) and runs the epilogue.  Arrival = agent-scope release of the workgroup's slab stores + one relaxed atomic on the
    // tile's counter; no unit ever waits for another (no co-residency requirement, no deadlock); the last arriver re-zeroes the counter, so the
    // caller's counter array stays all-zero between launches.
    // (the flag lives in the first word of the K-tile buffers, free after the barrier below: a second __shared__ object would move the dynamic LDS
    // base and push the 256x256 config's LDS-DMA destinations across the 128 KiB line)
    volatile int* s_last_p = reinterpret_cast<volatile int*>(smem_raw);
    __syncthreads();                           // every wave's slab stores have completed (the barrier's vmcnt(0)) and nobody reads the K tiles any more ...
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // ... and are written back where other XCDs' CUs will read them
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int rt0 = rem_unit / p.rem_splits;
      const unsigned old = __hip_atomic_fetch_add(p.hyb_cnt + rt0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)(p.rem_splits - 1);
      if (last) {
        __hip_atomic_store(p.hyb_cnt + rt0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      *s_last_p = last;
    }
    __syncthreads();
    if (!*s_last_p) return;
    hybrid_reduce_quads<BM, BN>(p, rem_unit / p.rem_splits, tid, 64 * NW);
    return;
  }
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
  // ds_write_b128) and are read back row-major, so that every store instruction writes whole contiguous row segments (a
  // direct 8-byte-per-lane store from the MFMA layout writes four 32-byte pieces per 128-byte line: measured 8-10 % of a
  // 256x256 tile's time).  The slab is private to the wave: ONE workgroup barrier (the slabs alias the K-tile buffers other
  // waves may still be reading), then only LDS waits.
  constexpr int LDSW = WTN + 4;
  constexpr int RM = MT < 2 ? MT : 2;  // m-tiles per round
  constexpr int QUADS = WTN / 4;       // 4-column groups per sub-tile row
  float* wstage = reinterpret_cast<float*>(smem_raw) + wave * (RM * 16 * LDSW);
  // Fast path: interior tile, every epilogue term except FiLM / the backward epilogues / RoPE (alpha, bias, pre-activation save,
  // activation, LayerScale, residual: every Llama and ViT projection, forward and data-gradient).  Fully unrolled read-back, 8 columns
  // = one 16-byte store per lane and step, no bounds checks.  (tools/gemm_ablate.py, tile + 8000: the rolled general loop below cost
  // 11-17 % of a 256x256 tile's time and 44 % of a ViT fc1 launch: bias + GELU + pre-activation save.)
  // (the 256x256 configs hold 128 accumulator registers: unrolling the activation code there spills, and no 256x256-tiled GEMM of
  // this model has an activation -- those keep the alpha / bias / residual subset)
  if ((WTN % 8) == 0 && p.fast_swiglu_bwd && m0 + BM <= p.M && n0 + BN <= p.N) {
    // SwiGLU backward in the unrolled read-back (swiglu_bwd_kernel's arithmetic on the bf16-rounded d h, as epilogue_store's dact_mode 2): this
    // tile of d h = dy . W_down yields d gate AND d up for its columns; gate / up come from the saved [M, 2N] projection output.
    constexpr int OCT = WTN / 8;
    constexpr int STEPS = RM * 16 * OCT / 64;
    const int mbase = m0 + wm * WTM, nbase = n0 + wn * WTN;
    __syncthreads();
#pragma unroll
    for (int round = 0; round < MT / RM; ++round) {
#pragma unroll
      for (int ii = 0; ii < RM; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          *reinterpret_cast<f32x4*>(wstage + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[round * RM + ii][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int idx = st * 64 + lane, row = idx / OCT, c8 = idx % OCT;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c8 * 8), hi = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c8 * 8 + 4);
        const int m = mbase + round * RM * 16 + row, n = nbase + c8 * 8;
        const bf16x8_bits g8 = *reinterpret_cast<const bf16x8_bits*>(p.dact_src + (int64_t)m * p.ld_dact + n);
        const bf16x8_bits u8 = *reinterpret_cast<const bf16x8_bits*>(p.dact_src + (int64_t)m * p.ld_dact + p.N + n);
        bf16x8_bits dg, du;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = bfround((e < 4 ? lo[e] : hi[e - 4]) * p.alpha), g = bf2f((bf16_bits)g8[e]), u = bf2f((bf16_bits)u8[e]);
          const float sg = sigmoidf_(g);
          du[e] = (short)f2bf(d * bfround(g * sg));
          dg[e] = (short)f2bf(d * u * (sg * (1.f + g * (1.f - sg))));
        }
        *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = dg;
        *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + p.N + n) = du;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  if constexpr (ROPE_LAYOUT) {
    if (rope_tile) {   // RoPE in the unrolled read-back (host guarantees whole column tiles, alpha only, 16-byte aligned tables; rows past M are skipped)
      constexpr int STEPS = RM * 16 * 8 / 64;
      const int mbase = m0 + wm * WTM, nhead = n0 + (wn >> 1) * 128, chalf = (wn & 1) * 32;
      __syncthreads();
#pragma unroll
      for (int round = 0; round < MT / RM; ++round) {
#pragma unroll
        for (int ii = 0; ii < RM; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4*>(wstage + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[round * RM + ii][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const int idx = st * 64 + lane, row = idx >> 3, c8 = idx & 7;
          const float* xs = wstage + row * LDSW + c8 * 8;
          const float* ys = wstage + row * LDSW + (c8 ^ 4) * 8;          // rotation partner: slab column +- 32 = output column +- 64
          const f32x4 xlo = *reinterpret_cast<const f32x4*>(xs), xhi = *reinterpret_cast<const f32x4*>(xs + 4);
          const f32x4 ylo = *reinterpret_cast<const f32x4*>(ys), yhi = *reinterpret_cast<const f32x4*>(ys + 4);
          const int cin = chalf + ((c8 & 3) << 3);                         // column within the 64-wide half of the head
          const bool upper = c8 >= 4;
          const int m = mbase + round * RM * 16 + row, n = nhead + cin + (upper ? 64 : 0);
          float ra = p.alpha;
          if constexpr (NORMFOLD) { if (rowscale) ra *= s_rstd[wm * WTM + round * RM * 16 + row]; }
          const int pos = m % p.rope_S;
          const bf16x8_bits cs = *reinterpret_cast<const bf16x8_bits*>(p.rope_cos + (int64_t)pos * 64 + cin);
          const bf16x8_bits sn = *reinterpret_cast<const bf16x8_bits*>(p.rope_sin + (int64_t)pos * 64 + cin);
          bf16x8_bits o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {    // rope_kernel's arithmetic on y = bf16(acc): lo' = bf16(a c) + bf16(-b s), hi' = bf16(b c) + bf16(a s)
            const float x = bfround((e < 4 ? xlo[e] : xhi[e - 4]) * ra), y = bfround((e < 4 ? ylo[e] : yhi[e - 4]) * ra);
            const float cc = bf2f((bf16_bits)cs[e]), sv = bf2f((bf16_bits)sn[e]);
            o[e] = (short)f2bf(upper ? bfround(x * cc) + bfround(y * sv) : bfround(x * cc) + bfround(-y * sv));
          }
          if (m < p.M) *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
  }
  constexpr bool FAST_ACT = MT * NT <= 16;
  if (p.fast_epi && (FAST_ACT || (p.act == OVLA_ACT_NONE && !p.Cpre && !p.colscale)) && !OVLA_DBG(64) && m0 + BM <= p.M && n0 + BN <= p.N &&
      (WTN % 8) == 0) {
    constexpr int OCT = WTN / 8;                 // 8-column groups per slab row
    constexpr int STEPS = RM * 16 * OCT / 64;    // read-back steps per round
    const int mbase = m0 + wm * WTM, nbase = n0 + wn * WTN;
    __syncthreads();
#pragma unroll
    for (int round = 0; round < MT / RM; ++round) {
#pragma unroll
      for (int ii = 0; ii < RM; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          *reinterpret_cast<f32x4*>(wstage + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[round * RM + ii][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own slab writes have landed
      constexpr int GRP = STEPS < 2 ? STEPS : 2;   // read-back steps in flight (the accumulators of later rounds are still live)
#pragma unroll
      for (int st0 = 0; st0 < STEPS; st0 += GRP) {
      f32x4 lo[GRP], hi[GRP];
#pragma unroll
      for (int s2 = 0; s2 < GRP; ++s2) {
        const int idx = (st0 + s2) * 64 + lane, row = idx / OCT, c8 = idx % OCT;
        lo[s2] = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c8 * 8);
        hi[s2] = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c8 * 8 + 4);
      }
#pragma unroll
      for (int s2 = 0; s2 < GRP; ++s2) {
        const int st = st0 + s2;
        const int idx = st * 64 + lane, row = idx / OCT, c8 = idx % OCT;
        const int m = mbase + round * RM * 16 + row, n = nbase + c8 * 8;
        float x[8];
        float ra = p.alpha;
        if constexpr (NORMFOLD) { if (rowscale) ra *= s_rstd[wm * WTM + round * RM * 16 + row]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = lo[s2][e] * ra; x[4 + e] = hi[s2][e] * ra; }
        if (p.bias) {
          const bf16x8_bits b8 = *reinterpret_cast<const bf16x8_bits*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = x[e] + bf2f((bf16_bits)b8[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = bfround(x[e]);
        if constexpr (FAST_ACT) {
          if (p.Cpre) {        // value before the activation, saved for the backward
            bf16x8_bits z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (short)f2bf(x[e]);
            *reinterpret_cast<bf16x8_bits*>(p.Cpre + (int64_t)m * p.ldc + n) = z;
          }
          if (p.act != OVLA_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = bfround(apply_act(x[e], p.act));
          }
          if (p.colscale) {
            const bf16x8_bits c8 = *reinterpret_cast<const bf16x8_bits*>(p.colscale + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = bfround(x[e] * bf2f((bf16_bits)c8[e]));
          }
        }
        if (p.residual) {
          const bf16x8_bits r8 = *reinterpret_cast<const bf16x8_bits*>(p.residual + (int64_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = bfround(x[e] + bf2f((bf16_bits)r8[e]));
        }
        bf16x8_bits o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(x[e]);
        if constexpr (NORMFOLD) {
          if (p.rowsq_out) {   // producer side: this wave's 64-column group of row m = 8 lanes x 8 stored values (x is already bf16-rounded here)
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = bf2f((bf16_bits)o[e]); sq += f * f; }
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            if (c8 == 0) p.rowsq_out[(int64_t)m * (p.N >> 6) + (nbase >> 6)] = sq;
          }
        }
        if (OVLA_DBG(32)) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n));
        else *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
      }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (reads consumed above; keeps the next round's writes behind them)
    }
#ifdef OVLA_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have been acknowledged
    OVLA_STAMP(4);
#endif
    return;
  }
  // General path (activations, pre-activation save, LayerScale, FiLM, backward epilogues, RoPE, edge tiles): one rolled loop.
  __syncthreads();  // every wave is done reading the K-tile buffers the slabs alias; from here on a wave touches only its own slab
#pragma unroll
  for (int round = 0; round < MT / RM; ++round) {
#pragma unroll
    for (int ii = 0; ii < RM; ++ii)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<f32x4*>(wstage + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[round * RM + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < RM * 16 * QUADS / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / QUADS, c4 = idx % QUADS;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + c4 * 4);
      const int m = m0 + wm * WTM + round * RM * 16 + row;
      const int n = n0 + wn * WTN + c4 * 4;
      if constexpr (WTN == 128) {   // one wave slab = one 128-wide head: the rotation partner quad is 16 quads away in the same slab row
        if (p.rope_cos && n < p.rope_cols) {
          const f32x4 vp = *reinterpret_cast<const f32x4*>(wstage + row * LDSW + (c4 ^ 16) * 4);
          if (m < p.M) rope_store(p, m, n, v, vp);
          continue;
        }
      }
      if constexpr (NORMFOLD) {
        // (WTN = 64: a slab row is one 64-column group, read back by 16 consecutive lanes; the host guarantees N % 64 == 0, so the 16 lanes of
        // a row are inside or outside the matrix together and the shuffles run in uniform control flow)
        f32x4 vs = v;
        if (rowscale) vs *= s_rstd[wm * WTM + round * RM * 16 + row];
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
        const bool inside = m < p.M && n < p.N;
        if (inside) st = epilogue_store(p, m, n, vs);
        if (p.rowsq_out) {
          const float sq = rowsq16(st);
          if (inside && c4 == 0) p.rowsq_out[(int64_t)m * (p.N >> 6) + (n >> 6)] = sq;
        }
        continue;
      }
      if (m < p.M && n < p.N) epilogue_store(p, m, n, v);
    }
  }
}

// =====================================================================================================================
// The 256x256 tile on FOUR waves (one per SIMD, 128x128 per wave, 256 accumulator registers in AGPRs), register-staged operands, hand-scheduled K loop.
// `tile = 0` takes it instead of the 8-wave kernel above for K >= 4096 in whole K tiles, a LoRA K-extension of 0 / 32 / 64 / 96 columns and the alpha / bias /
// residual or RoPE epilogue (ovla_gemm_bf16): every decoder projection of the fine-tune step, forward and data-gradient; launches 4-9 % faster there.
//   * The 8-wave kernel reads (128 + 64) fragment rows per wave and K step from LDS (192 KB per K tile and CU) and pays 60-150 cycles of issue per LDS-DMA
//     piece.  Four 128x128 waves read 128 KB per K tile; operands come by plain global_load_dwordx4 into registers (a few cycles of issue each) and go to LDS
//     by ds_write_b128 one K tile later -- one load and one write per MFMA row.
//   * With one wave per SIMD nothing but that wave's own instruction stream fills the MFMA shadows, and the wave issues in order: WHERE the other
//     instructions sit decides the speed.  As a block between two rows of eight MFMAs they let the matrix pipe run dry (1.38 us per K tile); one after each
//     of a row's first five MFMAs, with the fragment reads two rows ahead, they cost nothing extra.
//   * The compiler cannot hold 256 accumulators + 170 live VGPRs through its own scheduling (its listing: 736 v_accvgpr moves and 267 scratch stores per K
//     tile), so the K loop is inline asm, instruction by instruction: the accumulators are tied AGPR operands ("+a"), every wait count is explicit (all
//     memory operations of the loop are in the asm, in program order, so the counts are constants).
//   * Traps of this style, each hit once (tests/test_abi.py disassembles the shipped kernels and checks for them): an asm load whose result is never used
//     is a DEAD output -- the compiler hands its destination registers to the next live value while the load is in flight (w4_keep, STAGE = false);
//     an SGPR base the compiler produced with v_readfirstlane needs 5 wait states before an asm vector load reads it (tile_base); a compiler v_mov into a
//     fragment register right before the first asm MFMA is not separated from it by the hazard recognizer, which cannot see into the asm (s_nop 4 before
//     every MFMA sequence that follows compiler code); a branch around asm that updates the 64 accumulators makes the compiler merge them through scratch
//     (straight-line code, template parameters instead of branches); one dynamically indexed use keeps the accumulator array in scratch memory.
// Timing ablations (csrc/build.sh ablate, tools/gemm_w4_ablate.py): MFMAs + barriers alone 0.915 us per K tile (2.35 PFLOP/s), this loop 1.37.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
OVLA_DEV f32x4 w4_mfma(f32x4 acc, bf16x8_bits x, bf16x8_bits y) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y)); return acc; }
OVLA_DEV f32x4 w4_mfma0(bf16x8_bits x, bf16x8_bits y) { f32x4 acc; asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(x), "v"(y)); return acc; }
template <int OFF> OVLA_DEV bf16x8_bits w4_ds_read(uint32_t addr) { bf16x8_bits dst; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); return dst; }
template <int OFF> OVLA_DEV void w4_ds_write(uint32_t addr, u32x4 src) { asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(src), "n"(OFF)); }
OVLA_DEV f32x4 w4_pin(f32x4 v) { asm volatile("" : "+a"(v)); return v; }
OVLA_DEV void w4_keep(u32x4 v) { asm volatile("" : : "v"(v)); }
OVLA_DEV u32x4 w4_gload(uint32_t voff, const char* sbase) { u32x4 dst; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase)); return dst; }
// lgkmcnt a row of the 4-wave K loop needs before its first MFMA: LDS operations in program order are
//   top: b0[0 .. NT-1], a(0), a(1), [w]        row k: a(k + 2) (k <= 13), b1[k] (k < NT), [w] (k < NPR)        ([w] = a staging ds_write, only with staging)
// and everything NEWER than the youngest fragment row r uses -- a(r), and for the first row of the second k substep also b1[NT - 1] -- may stay in flight.
constexpr int w4_newer(int r, bool stage, int NT, int NPR, bool TOPW) {
  int ev[96] = {};
  int n = 0;
  for (int j = 0; j < NT; ++j) ev[n++] = 1;
  ev[n++] = 100; ev[n++] = 101;
  if (TOPW && stage) ev[n++] = 300;
  for (int kk = 0; kk < r; ++kk) {
    if (kk + 2 <= 15) ev[n++] = 100 + kk + 2;
    if (kk < NT) ev[n++] = 200 + kk;
    if (stage && kk < NPR) ev[n++] = 300;
  }
  int last = -1;
  for (int e = 0; e < n; ++e)
    if (ev[e] == 100 + r || (r == 8 && ev[e] == 200 + NT - 1)) last = e;
  return n - 1 - last;
}
template <int KEXT, int ABL = 0, int WNW = 2, bool RMAP = false, bool GMAP = false>   // GMAP: the SwiGLU pair map of the 128x256 configuration (act = OVLA_ACT_SWIGLU, below);  WNW = waves along N: 2 -> the 256x256 tile (2 x 2 waves of 128x128), 4 -> a 128x256 tile (1 x 4 waves of 128x64: batch-1 shapes; RMAP: its column map for RoPE launches, below);  KEXT: 32-wide k-steps of the LoRA K-extension (0 .. 3);  ABL: timing-only ablations (OVLA_GEMM_ABLATE builds), bits: 1 = no staging after the prologue, 2 = no fragment reads, 4 = no lgkmcnt waits in the rows, 8 = no vmcnt waits in the rows
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(const GemmParams p) {
  constexpr int WMW = 4 / WNW, BM = 128 * WMW, BN = 256, WTM = 128, WTN = BN / WNW, MT = 8, NT = WTN / 16;
  constexpr int PA = BM / 32, PB = BN / 32, NP = PA + PB;   // 1-KiB staging pieces (8 rows x 128 bytes) per wave and K tile: of A, of B, together (16 / 12)
  constexpr bool TOPW = NP == 16;                            // 16 pieces: one per MFMA row, the deferred row's slot at the top of a body included; 12: rows 0 .. 11
  constexpr int NPR = TOPW ? NP - 1 : NP;                    // pieces carried by rows 0 .. NPR - 1
  constexpr int TILE_BYTES = (BM + BN) * BK * 2;             // 65536 / 49152
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  // Column map of the wave's n-tiles.  Plain: wave wn owns columns wn * WTN + 16 j.  RMAP (128x256 configuration, RoPE launches: head_dim 128 = two 64-column
  // wave strips): wave (h = wn >> 1, w2 = wn & 1) takes columns h * 128 + 32 w2 + [0, 32) AND their rotation partners + 64, i.e. tile j sits at
  // 16 (j & 1) + 64 (j >> 1): the partner of a slab column is 32 slab columns away in the same row of the wave's own slab.
  // GMAP (act = OVLA_ACT_SWIGLU: B = [gate; up] stacked, 2 F rows): a workgroup takes 128 gate rows n0 + [0, 128) AND the 128 up rows F + n0 + [0, 128) as its
  // B tile; wave wn owns gate columns (WTN / 2) wn + [0, WTN / 2) (the first half of its n-tiles) and the SAME up columns (the second half): its slab row holds
  // WTN / 2 gate values and, WTN / 2 slab columns on, their up partners -- silu(g) * u is computed in the read-back and C is [M, F] (C_pre, if given, receives
  // the [M, 2 F] projection output itself: the fine-tune step keeps it for the backward).
  static_assert(!RMAP || WNW == 4, "the RoPE column map belongs to the 128x256 configuration");
  static_assert(!(RMAP && GMAP) && (!GMAP || KEXT <= 1), "one map at a time; the SwiGLU map takes a K-extension of 0 or 32 columns");
  constexpr int NH = NT / 2;
  const int cb = RMAP ? (wn >> 1) * 128 + (wn & 1) * 32 : GMAP ? wn * (WTN / 2) : wn * WTN;     // first column of the wave inside the tile
  auto cj = [](int j) constexpr { return RMAP ? 16 * (j & 1) + 64 * (j >> 1) : GMAP ? 16 * (j % NH) + 128 * (j / NH) : 16 * j; };   // column (B-tile row) of n-tile j relative to cb
  auto scol = [](int sc) constexpr { return RMAP ? (sc & 31) + 64 * (sc >> 5) : sc; };      // column (relative to cb) of slab column sc = 16 j + c
  OVLA_STAMP(0);

  int bid = blockIdx.x;
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int remap_n = p.rem_tiles > 0 ? p.full_tiles : (int)gridDim.x;
  int split = 0, t_mn, rem_unit = -1;
  if (bid < remap_n) {
    const int xcd = bid & 7, q = remap_n >> 3, r = remap_n & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    split = bid / tiles_mn;
    t_mn = bid - split * tiles_mn;
  } else {
    int u = bid - p.full_tiles;
    const int nrem = p.rem_tiles * p.rem_splits;
    const int xcd = u & 7, q = nrem >> 3, r = nrem & 7;
    u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (u >> 3);
    split = u / p.rem_tiles;
    const int rt = u - split * p.rem_tiles;
    t_mn = p.full_tiles + rt;
    rem_unit = rt * p.rem_splits + split;
  }
  constexpr int GROUP = 8;
  const int group_sz = GROUP * p.tiles_n;
  const int gid = t_mn / group_sz;
  const int first_m = gid * GROUP;
  const int gm = (p.tiles_m - first_m) < GROUP ? (p.tiles_m - first_m) : GROUP;
  const int in_group = t_mn - gid * group_sz;
  const int tm = first_m + in_group % gm;
  const int tn = in_group / gm;
  const int m0 = tm * BM, n0 = tn * (GMAP ? 128 : BN);   // (GMAP: 128 output = gate columns per tile; the host sets tiles_n = F / 128)

  const int T = p.T1;
  int t_begin = 0, t_end = T;
  const int nsplit = rem_unit >= 0 ? p.rem_splits : p.split_k;
  if (nsplit > 1) {
    const int chunk = (T + nsplit - 1) / nsplit;
    t_begin = split * chunk;
    t_end = t_begin + chunk < T ? t_begin + chunk : T;
  }

  f32x4 acc[MT][NT];
  if constexpr (KEXT == 0)   // (with a K-extension its MFMAs are the first to touch every accumulator and take the constant 0 as their C operand)
    static_for<MT * NT>([&](auto e_tag) { constexpr int e = decltype(e_tag)::value; acc[e / NT][e % NT] = w4_pin(f32x4{0.f, 0.f, 0.f, 0.f}); });   // pinned: left to itself the compiler keeps the zeros as constants and merges them into the loop through scratch

  // staging: piece i of this wave = rows (wave * 8 + i) * 8 .. + 7 of the A (i < 8) / B (i >= 8) tile; lane -> row + (lane >> 3), 16-byte chunk lane & 7
  // (eight lanes read one row's 128 contiguous bytes); the LDS image is the 8-wave kernel's (chunk XOR (row >> 1) & 7), applied on the WRITE address.
  uint32_t offA[PA], offB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int ga = m0 + (wave * PA + i) * 8 + (lane >> 3);
    ga = ga < p.M - 1 ? ga : p.M - 1;
    offA[i] = (uint32_t)(((int64_t)ga * p.lda + (lane & 7) * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int gb = n0 + (wave * PB + i) * 8 + (lane >> 3);
    if constexpr (GMAP) { const int br = (wave * PB + i) * 8 + (lane >> 3); gb = br < 128 ? n0 + br : (p.N >> 1) + n0 + (br - 128); }   // gate rows, then the up rows F + the same
    gb = gb < p.N - 1 ? gb : p.N - 1;
    offB[i] = (uint32_t)(((int64_t)gb * p.ldb + (lane & 7) * 8) * 2);
  }
  // LDS byte addresses (VGPRs); the piece / m-tile index goes into the instruction's offset field
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;   // LDS addresses are 32-bit (the pointer's low word)
  const int wx = (lane & 7) ^ (lane >> 4);               // (r >> 1) & 7 = (4 (i & 1) + (lane >> 4)) & 7 for r = 8 (8 wave + i) + (lane >> 3)
  const uint32_t w_even = lds0 + (lane >> 3) * 128 + wx * 16, w_odd = lds0 + (lane >> 3) * 128 + (wx ^ 4) * 16;
  const uint32_t wa = wave * PA * 1024, wb = BM * 128 + wave * PB * 1024;   // this wave's pieces of the A / B tile ((wave * P + i) & 1 = i & 1: PA and PB are even)
  uint32_t wrA[2][2] = {{w_even + wa, w_odd + wa}, {w_even + wa + TILE_BYTES, w_odd + wa + TILE_BYTES}};   // [buffer][piece parity]; + 1024 i in the offset field
  uint32_t wrB[2][2] = {{w_even + wb, w_odd + wb}, {w_even + wb + TILE_BYTES, w_odd + wb + TILE_BYTES}};
  const int arow = wm * WTM + (lane & 15), brow = cb + (lane & 15), cq = lane >> 4;
  const uint32_t ra0 = lds0 + arow * 128 + ((cq ^ ((arow >> 1) & 7)) * 16), rb0 = lds0 + BM * 128 + brow * 128 + ((cq ^ ((brow >> 1) & 7)) * 16);
  uint32_t ra[2][2] = {{ra0, ra0 ^ 64}, {ra0 + TILE_BYTES, (ra0 ^ 64) + TILE_BYTES}};   // [buffer][k substep]: + 2048 i per m-tile in the offset field
  uint32_t rb[2][2] = {{rb0, rb0 ^ 64}, {rb0 + TILE_BYTES, (rb0 ^ 64) + TILE_BYTES}};

  u32x4 g[NP];
  bf16x8_bits b0[NT], b1[NT];
  bf16x8_bits a_def = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NT; ++j) b1[j] = a_def;

  auto clampt = [&](int t) { return t < t_end ? t : t_end - 1; };
  // staging piece q of this wave: q < PA -> A piece q, else B piece q - PA
  auto piece_write = [&](auto buf_tag, auto q_tag) {
    constexpr int BUF = decltype(buf_tag)::value, q = decltype(q_tag)::value < NP ? decltype(q_tag)::value : 0;   // (rows >= NP instantiate this in a discarded branch)
    if constexpr (q < PA) w4_ds_write<q * 1024>(wrA[BUF][q & 1], g[q]);
    else w4_ds_write<(q - PA) * 1024>(wrB[BUF][(q - PA) & 1], g[q]);
  };
  auto piece_load = [&](auto q_tag, const char* bA, const char* bB) {
    constexpr int q = decltype(q_tag)::value < NP ? decltype(q_tag)::value : 0;
    if constexpr (q < PA) g[q] = w4_gload(offA[q], bA);
    else g[q] = w4_gload(offB[q - PA], bB);
  };
  auto tile_base = [&](const bf16_bits* base, int t) {   // wave-uniform by construction; said explicitly, because the asm loads take it as an SGPR pair ("s")
    const uint64_t a = reinterpret_cast<uint64_t>(base) + (uint64_t)((int64_t)t * (BK * 2));
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    // Both halves are materialised in SGPRs HERE, five wait states before any asm load can use them: when the compiler takes the tile index for
    // divergent (seen with the stamp build's `if (tid == 0)` stores) the halves come from v_readfirstlane, and an SGPR written by a VALU instruction
    // needs 5 wait states before a vector memory instruction reads it -- a hazard the compiler cannot see inside inline asm (it faulted the GPU once).
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  // prologue.  (1) K-extension (LoRA: K2 = 32 or 64, one or two MFMA k-steps): its fragments come straight from global memory (16 bytes per lane and
  // fragment, rows 64 / 128 bytes wide), issued FIRST.  (2) K tile t_begin goes to LDS buffer 0 by LDS-DMA (the only DMA of the kernel: no register round
  // trip for the tile everything waits on; source chunk = slot XOR (row >> 1) & 7 as in the 8-wave kernel).  (3) The K-extension's 64 MFMAs per k-step run
  // while that tile is on its way.  (4) Tile t_begin + 1 into the staging registers, in the loop's order.  The compiler counts the vector memory operations
  // it knows (1, 2) when it waits for (1) before (3); the asm loads of (4) are invisible to it, so none of them may be issued before (3).
  // RMSNorm fold, consumer side (128x256 configuration; ovla.h: rowscale_part): rstd of this tile's 128 rows from the producer's per-64-column sums of squares,
  // into LDS behind the K-tile buffers.  Two lanes per row; the loads are issued first and waited for after the LDS-DMA of the first K tile has been issued.
  float* s_rstd = reinterpret_cast<float*>(smem_raw + 2 * TILE_BYTES);
  bool rowscale = false;
  f32x4 rs_q[WNW == 4 ? 8 : 1];
  if constexpr (WNW == 4) {
    rowscale = p.rowscale_part != nullptr;
    if (rowscale) {   // (rowscale_slots = K / 64 is a multiple of 8 and at most 64 here: host-checked)
      const int m = m0 + (tid >> 1), per = p.rowscale_slots >> 1;
      const float* src = p.rowscale_part + (int64_t)(m < p.M ? m : p.M - 1) * p.rowscale_slots + (tid & 1) * per;
#pragma unroll
      for (int u = 0; u < 8; ++u) rs_q[u] = (4 * u < per) ? *reinterpret_cast<const f32x4*>(src + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  OVLA_STAMP(1);
  // (K2 = 32 KEXT, host-checked.  Straight-line code: a branch around asm that updates 64 accumulators makes the compiler merge them through scratch.  Two
  // fragment sets: steps 0 and 1 are requested up front, step 2 into set 0 once step 0's MFMAs have read it.)
  bf16x8_bits a2f[(KEXT > 1 || GMAP) ? 2 : 1][KEXT ? MT : 1], b2f[KEXT > 1 ? 2 : 1][KEXT ? NT : 1];   // (GMAP: a2f[0] / a2f[1] = the gate / the up group's columns of A2)
  const bf16_bits* a2p = nullptr;
  const bf16_bits* b2p = nullptr;
  auto kext_load = [&](auto set_tag, int ks) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) { int m = m0 + arow + i * 16; m = m < p.M - 1 ? m : p.M - 1; a2f[S][i] = *reinterpret_cast<const bf16x8_bits*>(a2p + (int64_t)m * p.lda2 + ks * 32); }
#pragma unroll
    for (int j = 0; j < NT; ++j) { int n = n0 + brow + cj(j); if constexpr (GMAP) n = n0 + brow + (cj(j) & 127) + (cj(j) >= 128 ? (p.N >> 1) : 0); n = n < p.N - 1 ? n : p.N - 1; b2f[S][j] = *reinterpret_cast<const bf16x8_bits*>(b2p + (int64_t)n * p.ldb2 + ks * 32); }
  };
  if constexpr (KEXT > 0) {
    const int a2_col0 = p.k2_group_n > 0 ? (n0 / p.k2_group_n) * p.K2 : 0;
    a2p = p.A2 + a2_col0 + 8 * cq;
    b2p = p.B2 + 8 * cq;
    kext_load(std::integral_constant<int, 0>{}, 0);
    if constexpr (KEXT > 1) kext_load(std::integral_constant<int, 1>{}, 1);
    if constexpr (GMAP) {   // the up rows' LoRA group (k2_group_n = F for the fused gate | up linear: group 1; ungrouped: the same columns again)
      const int up_col0 = p.k2_group_n > 0 ? (((p.N >> 1) + n0) / p.k2_group_n) * p.K2 : 0;
#pragma unroll
      for (int i = 0; i < MT; ++i) { int m = m0 + arow + i * 16; m = m < p.M - 1 ? m : p.M - 1; a2f[1][i] = *reinterpret_cast<const bf16x8_bits*>(p.A2 + up_col0 + 8 * cq + (int64_t)m * p.lda2); }
    }
  }
  {
    const char* gA = tile_base(p.A, clampt(t_begin));
    const char* gB = tile_base(p.B, clampt(t_begin));
    const int sw = lane >> 4, c = lane & 7;
    const int d_even = ((c ^ sw) - c) * 16, d_odd = ((c ^ sw ^ 4) - c) * 16;   // source chunk of LDS slot (lane & 7) in piece i: slot XOR (4 (i & 1) + (lane >> 4))
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + (int64_t)offA[i] + ((i & 1) ? d_odd : d_even)),
                                       (__attribute__((address_space(3))) void*)(smem_raw + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + (int64_t)offB[i] + ((i & 1) ? d_odd : d_even)),
                                       (__attribute__((address_space(3))) void*)(smem_raw + BM * 128 + (wave * PB + i) * 1024), 16, 0, 0);
  }
  if constexpr (WNW == 4) {
    if (rowscale) {
      float ssum = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) { ssum += rs_q[u][0]; ssum += rs_q[u][1]; ssum += rs_q[u][2]; ssum += rs_q[u][3]; }
      ssum += __shfl_xor(ssum, 1, 64);
      const int m = m0 + (tid >> 1);
      const float r = m < p.M ? rsqrtf(ssum / (float)(p.rowscale_slots * 64) + p.rowscale_eps) : 0.f;
      s_rstd[tid >> 1] = r;
      if ((tid & 1) == 0 && m < p.M && tn == 0 && split == 0 && p.rowscale_r) p.rowscale_r[m] = r;   // for the hybrid-remainder reduce kernel
    }
  }
  if constexpr (KEXT > 0) {
    const bool mine = t_begin == 0;   // a later K part of a split tile: the extension belongs to the first part only (zero fragments, same instruction stream)
    static_for<KEXT>([&](auto s_tag) {
      constexpr int st = decltype(s_tag)::value, S = st & 1;
      if (!mine) {
#pragma unroll
        for (int i = 0; i < MT; ++i) { a2f[S][i] = bf16x8_bits{0, 0, 0, 0, 0, 0, 0, 0}; if constexpr (GMAP) a2f[1][i] = a2f[S][i]; }
      }
      // The compiler may have written a fragment register with a VALU instruction right here (the zeroing above, or a v_mov restoring a register it borrowed
      // -- seen in the listing, and the first MFMA then read the stale value: the hazard recognizer cannot know that the asm below is an MFMA and inserts
      // nothing).  Five wait states between compiler code and the first MFMA of every asm sequence; tests/test_abi.py checks the disassembly for it.
      asm volatile("s_nop 4");
      static_for<MT * NT>([&](auto e_tag) {
        constexpr int i = decltype(e_tag)::value / NT, j = decltype(e_tag)::value % NT;
        if constexpr (st == 0) acc[i][j] = w4_mfma0(b2f[S][j], a2f[GMAP ? (j < NH ? 0 : 1) : S][i]);
        else acc[i][j] = w4_mfma(acc[i][j], b2f[S][j], a2f[S][i]);
      });
      if constexpr (st + 2 < KEXT) kext_load(std::integral_constant<int, S>{}, st + 2);
    });
  }
  {
    const char* hA = tile_base(p.A, clampt(t_begin + 1));
    const char* hB = tile_base(p.B, clampt(t_begin + 1));
    if constexpr (TOPW) piece_load(std::integral_constant<int, NP - 1>{}, hA, hB);   // the loop's order: (piece 15 at the top of a body,) then 0 .. NPR - 1 with the rows
    static_for<NPR>([&](auto q_tag) { piece_load(q_tag, hA, hB); });
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NP) : "memory");   // the NP LDS-DMA pieces of tile t_begin have landed (the NP register loads behind them may still fly)
  }

  // body t (PAR = parity of t - t_begin = LDS buffer holding tile t): 16 MFMA rows r = (k substep, m-tile), NT MFMAs each; row 15 of tile t - 1 is deferred
  // across the barrier and issued at the top, under the first fragment reads of tile t.  One memory instruction sits after each of a row's first MFMAs (not as
  // a block between two rows: the wave issues in order, and six other instructions after a row's last MFMA let the matrix pipe run dry for ~10 of every 138
  // cycles -- 1.38 us per K tile where this placement, with the fragment reads TWO rows ahead (three A fragments live), needs 1.37 at full speed of the rest):
  //   ds_read a(r + 2) | ds_read b1[i] (first k substep) | vmcnt(NP - 1) | ds_write piece r of tile t + 1 into the other buffer | global_load piece r of tile t + 2
  // NP loads are in flight per wave at all times, so `vmcnt(NP - 1)` is exactly "the load issued NP loads ago has landed"; the lgkmcnt a row needs before its
  // first MFMA comes from the LDS operations' program order (w4_newer).
  // (STAGE = false: the odd last tile, after which nothing is staged any more.  A staging load whose result is never used would be a DEAD asm output: the
  // compiler then hands its destination registers to the next live value while the load is still in flight -- seen in the listing, half a K tile lost.)
  // (SHIFT: the instructions may start SHIFT MFMA gaps later in a row; always 0 -- see the note at the k_loop call.)
  auto body = [&](const int t, auto par_tag, auto stage_tag, auto shift_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int SHIFT = decltype(shift_tag)::value;
    constexpr bool STAGE = decltype(stage_tag)::value && !(ABL & 1);
    constexpr bool FRAG = !(ABL & 2);
    const char* gA = tile_base(p.A, clampt(t + 2));
    const char* gB = tile_base(p.B, clampt(t + 2));
    bf16x8_bits af[3];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier");   // tile t is in LDS (everyone's ds_writes), nobody reads the other buffer any more
    if constexpr (FRAG) {
    static_for<NT>([&](auto j_tag) { constexpr int j = decltype(j_tag)::value; b0[j] = w4_ds_read<cj(j) * 128>(rb[PAR][0]); });
    af[0] = w4_ds_read<0>(ra[PAR][0]);
    af[1] = w4_ds_read<2048>(ra[PAR][0]);
    } else { af[0] = af[1] = af[2] = a_def; }
    asm volatile("s_setprio 1");
    // deferred row of tile t - 1, with the top's staging piece between its MFMAs
    static_for<NT>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      acc[MT - 1][j] = w4_mfma(acc[MT - 1][j], b1[j], a_def);
      if constexpr (STAGE && TOPW && j == SHIFT) { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NP - 1)); piece_write(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, NP - 1>{}); }
      if constexpr (STAGE && TOPW && j == SHIFT + 1) piece_load(std::integral_constant<int, NP - 1>{}, gA, gB);
    });
    static_for<2 * MT - 1>([&](auto r_tag) {
      constexpr int r = decltype(r_tag)::value;
      constexpr int sub = r / MT, i = r % MT;
      constexpr int newer = w4_newer(r, STAGE, NT, NPR, TOPW);   // (256x256: 2 / 4 / 5 ... 5 / 1 / 4 / 3 ... 3 with staging)
      if constexpr (FRAG && !(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(newer));
      auto mf = [&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        if constexpr (sub == 0) acc[i][j] = w4_mfma(acc[i][j], b0[j], af[r % 3]);
        else acc[i][j] = w4_mfma(acc[i][j], b1[j], af[r % 3]);
      };
      // one memory instruction per MFMA gap: a-fragment read, b-fragment read, vmcnt, ds_write, global load after MFMAs SHIFT .. SHIFT + 4 (with four MFMAs per
      // row -- the 128x256 configuration -- the wait and the write share a gap)
      static_for<NT>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value, kk = j - SHIFT;
        constexpr int K_V = 2, K_W = NT >= 5 ? 3 : 2, K_G = NT >= 5 ? 4 : 3;
        mf(j_tag);
        if constexpr (kk == 0 && FRAG && r + 2 <= 2 * MT - 1) af[(r + 2) % 3] = w4_ds_read<((r + 2) % MT) * 2048>(ra[PAR][(r + 2) / MT]);
        if constexpr (kk == 1 && FRAG && sub == 0 && i < NT) b1[i] = w4_ds_read<cj(i) * 128>(rb[PAR][1]);
        if constexpr (kk == K_V && STAGE && r < NPR && !(ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NP - 1));
        if constexpr (kk == K_W && STAGE && r < NPR) piece_write(std::integral_constant<int, PAR ^ 1>{}, r_tag);
        if constexpr (kk == K_G && STAGE && r < NPR) piece_load(r_tag, gA, gB);
      });
    });
    a_def = af[(2 * MT - 1) % 3];
    asm volatile("s_setprio 0");
  };
  OVLA_STAMP(2);
  auto k_loop = [&](auto shift_tag) {
    int t = t_begin;
    for (; t + 1 < t_end; t += 2) {
      body(t, std::integral_constant<int, 0>{}, std::true_type{}, shift_tag);
      body(t + 1, std::integral_constant<int, 1>{}, std::true_type{}, shift_tag);
    }
    if (t < t_end) body(t, std::integral_constant<int, 0>{}, std::false_type{}, shift_tag);
  };
  // (One loop for all four waves.  Staggering them was tried both ways and dropped: four code variants with the memory instructions SHIFT gaps later, picked
  // by `wave` -- a four-way branch around asm that updates 64 tied accumulators makes the compiler merge them through AGPR moves and scratch inside the loops,
  // 140 moves per K tile -- and a time skew of w x 8..96 cycles after the barrier, which only added its own length to every K tile.)
  k_loop(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  static_for<NP>([&](auto q_tag) { w4_keep(g[decltype(q_tag)::value]); });   // the staging registers stay allocated until their last loads have landed
  asm volatile("s_nop 4");   // (compiler code may sit between the loop and the last deferred row's MFMAs: same hazard as in the prologue)
  static_for<NT>([&](auto j_tag) { constexpr int j = decltype(j_tag)::value; acc[MT - 1][j] = w4_mfma(acc[MT - 1][j], b1[j], a_def); });
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the compiler's hazard recognizer does not see the asm MFMAs' AGPR writes
  OVLA_STAMP(3);

  // (compile-time indices everywhere: one dynamically indexed use would keep the accumulator array in scratch memory, written through after every MFMA)
  if (rem_unit >= 0) {
    float* slab = p.ws + (int64_t)rem_unit * (BM * BN);
    static_for<MT * NT>([&](auto e_tag) {
      constexpr int i = decltype(e_tag)::value / NT, j = decltype(e_tag)::value % NT;
      *reinterpret_cast<f32x4*>(slab + (wm * WTM + i * 16 + (lane & 15)) * BN + cb + cj(j) + 4 * (lane >> 4)) = acc[i][j];
    });
    return;
  }
  if (p.split_k > 1) {
    static_for<MT * NT>([&](auto e_tag) {
      constexpr int i = decltype(e_tag)::value / NT, j = decltype(e_tag)::value % NT;
      const int m = m0 + wm * WTM + i * 16 + (lane & 15), n = n0 + cb + cj(j) + 4 * (lane >> 4);
      if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(p.ws + ((int64_t)split * p.M + m) * p.N + n) = acc[i][j];
    });
    return;
  }
  // accumulators -> wave-private fp32 LDS slab (two m-tiles = 32 rows x (128 + 4) per round, four rounds) -> read back in row order, 16 lanes per row,
  // one 16-byte store per lane and step.  With one wave per SIMD nothing else hides the LDS and memory latencies: a round's eight read-back steps are all in
  // flight together and its residual rows / RoPE table rows are requested BEFORE the slab round trip.  (Stores straight from the registers -- n-tiles
  // permuted so that a lane's tile pair is 8 consecutive columns, 16 rows x 64 bytes per store instruction -- measured 7 % slower per tile than this.)
  constexpr int LDSW = WTN + 4;
  float* slab = reinterpret_cast<float*>(smem_raw) + wave * (32 * LDSW);
  const int mbase = m0 + wm * WTM, nbase = n0 + cb;   // (RMAP: a slab column sc is output column nbase + scol(sc))
  __syncthreads();   // nobody reads the K tiles any more; from here on a wave touches only its own slab
  constexpr int OCT = WTN / 8, STEPS = 32 * OCT / 64;   // 8-column groups per slab row; read-back steps per round (8 / 4)
  auto to_slab = [&](auto rd_tag) {
    constexpr int rd = decltype(rd_tag)::value;
    static_for<2 * NT>([&](auto e_tag) {
      constexpr int ii = decltype(e_tag)::value / NT, j = decltype(e_tag)::value % NT;
      *reinterpret_cast<f32x4*>(slab + (ii * 16 + (lane & 15)) * LDSW + j * 16 + 4 * (lane >> 4)) = acc[rd * 2 + ii][j];
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
  if (WNW == 2 && p.rope_cos && nbase < p.rope_cols) {   // RoPE, 256x256 configuration (head_dim 128 = this wave's 128 columns): the rotation partner of octet c8 is octet c8 ^ 8 of the same slab row
    static_for<MT / 2>([&](auto rd_tag) {
      constexpr int rd = decltype(rd_tag)::value;
      bf16x8_bits csv[8], snv[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int idx = st * 64 + lane, row = idx >> 4, c8 = idx & 15;
        const int m = mbase + rd * 32 + row, cin = (c8 & 7) * 8;
        const int pos = (m < p.M ? m : p.M - 1) % p.rope_S;
        csv[st] = *reinterpret_cast<const bf16x8_bits*>(p.rope_cos + (int64_t)pos * 64 + cin);
        snv[st] = *reinterpret_cast<const bf16x8_bits*>(p.rope_sin + (int64_t)pos * 64 + cin);
      }
      to_slab(rd_tag);
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int idx = st * 64 + lane, row = idx >> 4, c8 = idx & 15;
        const float* xs = slab + row * LDSW + c8 * 8;
        const float* ys = slab + row * LDSW + (c8 ^ 8) * 8;
        const f32x4 xlo = *reinterpret_cast<const f32x4*>(xs), xhi = *reinterpret_cast<const f32x4*>(xs + 4);
        const f32x4 ylo = *reinterpret_cast<const f32x4*>(ys), yhi = *reinterpret_cast<const f32x4*>(ys + 4);
        const int m = mbase + rd * 32 + row, n = nbase + c8 * 8;
        const bool upper = c8 >= 8;
        const bf16x8_bits cs = csv[st], sn = snv[st];
        bf16x8_bits o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {    // rope_kernel's arithmetic on y = bf16(acc): lo' = bf16(a c) + bf16(-b s), hi' = bf16(b c) + bf16(a s)
          const float x = bfround((e < 4 ? xlo[e] : xhi[e - 4]) * p.alpha), y = bfround((e < 4 ? ylo[e] : yhi[e - 4]) * p.alpha);
          const float cc = bf2f((bf16_bits)cs[e]), sv = bf2f((bf16_bits)sn[e]);
          o[e] = (short)f2bf(upper ? bfround(x * cc) + bfround(y * sv) : bfround(x * cc) + bfround(-y * sv));
        }
        if (m < p.M && n < p.N) *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    return;
  }
  if constexpr (GMAP) {   // SwiGLU in the read-back (swiglu_fwd_kernel's arithmetic on the bf16-rounded projection outputs): h = bf16(bf16(silu(g)) * u)
    constexpr int HO = OCT / 2, GST = 32 * HO / 64;   // output octets per slab row (gate octet c8 and its up partner HO octets on); read-back steps per round (4 / 2)
    const int F = p.N >> 1;
    static_for<MT / 2>([&](auto rd_tag) {
      constexpr int rd = decltype(rd_tag)::value;
      to_slab(rd_tag);
      f32x4 gl[GST], gh[GST], ul[GST], uh[GST];
#pragma unroll
      for (int st = 0; st < GST; ++st) {
        const int idx = st * 64 + lane, row = idx / HO, c8 = idx % HO;
        const float* gs = slab + row * LDSW + c8 * 8;
        gl[st] = *reinterpret_cast<const f32x4*>(gs); gh[st] = *reinterpret_cast<const f32x4*>(gs + 4);
        ul[st] = *reinterpret_cast<const f32x4*>(gs + WTN / 2); uh[st] = *reinterpret_cast<const f32x4*>(gs + WTN / 2 + 4);
      }
#pragma unroll
      for (int st = 0; st < GST; ++st) {
        const int idx = st * 64 + lane, row = idx / HO, c8 = idx % HO;
        const int m = mbase + rd * 32 + row, n = n0 + cb + c8 * 8;
        float ra = p.alpha;
        if constexpr (WNW == 4) { if (rowscale) ra *= s_rstd[rd * 32 + row]; }
        bf16x8_bits o, og, ou;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          og[e] = (short)f2bf((e < 4 ? gl[st][e] : gh[st][e - 4]) * ra); ou[e] = (short)f2bf((e < 4 ? ul[st][e] : uh[st][e - 4]) * ra);
          o[e] = (short)f2bf(bfround(silu(bf2f((bf16_bits)og[e]))) * bf2f((bf16_bits)ou[e]));
        }
        if (m < p.M && n < F) {
          *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
          if (p.Cpre) {   // the projection output itself, [M, 2 F] with row stride N
            *reinterpret_cast<bf16x8_bits*>(p.Cpre + (int64_t)m * p.N + n) = og;
            *reinterpret_cast<bf16x8_bits*>(p.Cpre + (int64_t)m * p.N + F + n) = ou;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    return;
  }
  if constexpr (RMAP) {
    if (p.rope_cos && n0 + (wn >> 1) * 128 < p.rope_cols) {   // RoPE, 128x256 configuration: slab columns [0, 32) are the lower-half columns 32 w2 + c of the head, [32, 64) their partners + 64
      static_for<MT / 2>([&](auto rd_tag) {
        constexpr int rd = decltype(rd_tag)::value;
        bf16x8_bits csv[STEPS], snv[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const int idx = st * 64 + lane, row = idx / OCT, c8 = idx % OCT;
          const int m = mbase + rd * 32 + row, cin = (wn & 1) * 32 + (c8 & 3) * 8;
          const int pos = (m < p.M ? m : p.M - 1) % p.rope_S;
          csv[st] = *reinterpret_cast<const bf16x8_bits*>(p.rope_cos + (int64_t)pos * 64 + cin);
          snv[st] = *reinterpret_cast<const bf16x8_bits*>(p.rope_sin + (int64_t)pos * 64 + cin);
        }
        to_slab(rd_tag);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const int idx = st * 64 + lane, row = idx / OCT, c8 = idx % OCT;
          const float* xs = slab + row * LDSW + c8 * 8;
          const float* ys = slab + row * LDSW + (c8 ^ 4) * 8;
          const f32x4 xlo = *reinterpret_cast<const f32x4*>(xs), xhi = *reinterpret_cast<const f32x4*>(xs + 4);
          const f32x4 ylo = *reinterpret_cast<const f32x4*>(ys), yhi = *reinterpret_cast<const f32x4*>(ys + 4);
          const int m = mbase + rd * 32 + row, n = nbase + scol(c8 * 8);
          const bool upper = c8 >= 4;
          const float ra = p.alpha * (rowscale ? s_rstd[rd * 32 + row] : 1.f);
          const bf16x8_bits cs = csv[st], sn = snv[st];
          bf16x8_bits o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {    // rope_kernel's arithmetic on y = bf16(acc): lo' = bf16(a c) + bf16(-b s), hi' = bf16(b c) + bf16(a s)
            const float x = bfround((e < 4 ? xlo[e] : xhi[e - 4]) * ra), y = bfround((e < 4 ? ylo[e] : yhi[e - 4]) * ra);
            const float cc = bf2f((bf16_bits)cs[e]), sv = bf2f((bf16_bits)sn[e]);
            o[e] = (short)f2bf(upper ? bfround(x * cc) + bfround(y * sv) : bfround(x * cc) + bfround(-y * sv));
          }
          if (m < p.M && n < p.N) *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      });
      return;
    }
  }
  if (p.fast_epi && interior && p.act == OVLA_ACT_NONE && !p.Cpre && !p.colscale) {   // alpha, bias, residual: every Llama projection, forward and data-gradient
    static_for<MT / 2>([&](auto rd_tag) {
      constexpr int rd = decltype(rd_tag)::value;
      constexpr int GRP = STEPS;
      bf16x8_bits r8[GRP];
      if (p.residual) {
#pragma unroll
        for (int s2 = 0; s2 < GRP; ++s2) {
          const int idx = s2 * 64 + lane, row = idx / OCT, c8 = idx % OCT;
          r8[s2] = *reinterpret_cast<const bf16x8_bits*>(p.residual + (int64_t)(mbase + rd * 32 + row) * p.ldr + nbase + scol(c8 * 8));
        }
      }
      to_slab(rd_tag);
      f32x4 lo[GRP], hi[GRP];
#pragma unroll
      for (int s2 = 0; s2 < GRP; ++s2) {
        const int idx = s2 * 64 + lane, row = idx / OCT, c8 = idx % OCT;
        lo[s2] = *reinterpret_cast<const f32x4*>(slab + row * LDSW + c8 * 8);
        hi[s2] = *reinterpret_cast<const f32x4*>(slab + row * LDSW + c8 * 8 + 4);
      }
#pragma unroll
      for (int s2 = 0; s2 < GRP; ++s2) {
        const int idx = s2 * 64 + lane, row = idx / OCT, c8 = idx % OCT;
        const int m = mbase + rd * 32 + row, n = nbase + scol(c8 * 8);
        float x[8];
        float ra = p.alpha;
        if constexpr (WNW == 4) { if (rowscale) ra *= s_rstd[rd * 32 + row]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = lo[s2][e] * ra; x[4 + e] = hi[s2][e] * ra; }
        if (p.bias) {
          const bf16x8_bits b8 = *reinterpret_cast<const bf16x8_bits*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = x[e] + bf2f((bf16_bits)b8[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = bfround(x[e]);
        if (p.residual) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = bfround(x[e] + bf2f((bf16_bits)r8[s2][e]));
        }
        bf16x8_bits o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(x[e]);
        if constexpr (WNW == 4 && !RMAP) {
          if (p.rowsq_out) {   // RMSNorm fold, producer side: this wave's slab row IS one 64-column group: 8 lanes x 8 stored (bf16-rounded) values
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = bf2f((bf16_bits)o[e]); sq += f * f; }
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            if (c8 == 0) p.rowsq_out[(int64_t)m * (p.N >> 6) + (nbase >> 6)] = sq;
          }
        }
        *reinterpret_cast<bf16x8_bits*>(p.C + (int64_t)m * p.ldc + n) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
#ifdef OVLA_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OVLA_STAMP(4);
#endif
    return;
  }
  // general path (edge tiles, activations, pre-activation save, LayerScale, FiLM, backward epilogues): one rolled loop per round
  static_for<MT / 2>([&](auto rd_tag) {
    constexpr int rd = decltype(rd_tag)::value;
    to_slab(rd_tag);
#pragma unroll 1
    for (int it = 0; it < WTN / 8; ++it) {
      const int idx = it * 64 + lane, row = idx / (WTN / 4), c4 = idx % (WTN / 4);
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * LDSW + c4 * 4);
      const int m = mbase + rd * 32 + row, n = nbase + scol(c4 * 4);
      if constexpr (WNW == 4) {   // (one slab row = one 64-column group, read back by 16 consecutive lanes; N % 64 == 0 with the fold: the 16 lanes are inside or outside together)
        if (rowscale) v *= s_rstd[rd * 32 + row];
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
        const bool inside = m < p.M && n < p.N;
        if (inside) st = epilogue_store(p, m, n, v);
        if (!RMAP && p.rowsq_out) {
          const float sq = rowsq16(st);
          if (inside && c4 == 0) p.rowsq_out[(int64_t)m * (p.N >> 6) + (n >> 6)] = sq;
        }
        continue;
      }
      if (m < p.M && n < p.N) epilogue_store(p, m, n, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}


// =====================================================================================================================
// Skinny-N, short-K GEMM in ONE launch (the ViT LoRA projections t = s x A^T and dt = s dy B: N = 32, K 1k..4k, M ~ 4k): the tiled
// split-K path needs two launches (GEMM + reduce), ~20 us of mostly launch / ramp latency for 9 MB of input.  Here one workgroup
// owns 32 rows x all N columns, its 4 waves split K and reduce through LDS.  Both MFMA operands are K-contiguous in memory, so
// every lane loads its 16-byte fragments straight from global (A streams from HBM once; B, N x K <= 280 KB, stays in L2).
// Used only where it wins (N == 32, K <= 3072: 13.3 vs 16.4 us back to back; beyond that the split-K pair is faster, and at N = 96
// the B re-reads from L2, once per 32 rows, made it slower at every K).
template <int NT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmParams p) {
  constexpr int LDW = NT * 16 + 4;
  __shared__ __attribute__((aligned(16))) float red[4][32][LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 32;
  const int ksteps = (p.K + 31) / 32;
  const int per_wave = (ksteps + 3) / 4;
  const int ks0 = wave * per_wave, ks1 = (ks0 + per_wave) < ksteps ? (ks0 + per_wave) : ksteps;
  const int kq = 8 * (lane >> 4);
  const bf16_bits* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + i * 16 + (lane & 15);
    m = m < p.M ? m : p.M - 1;
    arow[i] = p.A + (int64_t)m * p.lda;
  }
  const bf16_bits* brow = p.B + (int64_t)(lane & 15) * p.ldb;
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8_bits zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int ks = ks0; ks < ks1; ++ks) {
    const int k = ks * 32 + kq;
    const bool kin = k < p.K;
    bf16x8_bits a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = kin ? *reinterpret_cast<const bf16x8_bits*>(arow[i] + k) : zero;
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = kin ? *reinterpret_cast<const bf16x8_bits*>(brow + (int64_t)j * 16 * p.ldb + k) : zero;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  // cross-wave K reduction through LDS; lane owns C[m = i*16 + (lane&15)][n = j*16 + 4*(lane>>4) .. +3]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * 16 + (lane & 15)][j * 16 + 4 * (lane >> 4)]) = acc[i][j];
  __syncthreads();
  constexpr int QUADS = NT * 4;
  for (int idx = tid; idx < 32 * QUADS; idx += 256) {
    const int row = idx / QUADS, c4 = (idx % QUADS) * 4;
    const int m = m0 + row;
    if (m >= p.M) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][row][c4]);
    v += *reinterpret_cast<const f32x4*>(&red[1][row][c4]);
    v += *reinterpret_cast<const f32x4*>(&red[2][row][c4]);
    v += *reinterpret_cast<const f32x4*>(&red[3][row][c4]);
    bf16x4_bits o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (short)f2bf(v[j] * p.alpha);
    *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)m * p.ldc + c4) = o;
  }
}

// =====================================================================================================================
// Deep-pipelined variant (large tiles, ~1 workgroup per CU).
//   * BK = 32 stages in a STAGES-deep LDS ring: STAGES-1 stages of LDS-DMA stay in flight across the per-stage barrier
//     (counted s_waitcnt vmcnt, raw s_barrier -- __syncthreads() would drain the DMA queue, cdna_hip_programming.md
//     "Pipelining across barriers"), so HBM/L2 latency is covered by (STAGES-1) x one stage of MFMA time;
//   * 64-byte LDS rows, 16-byte chunk index XOR g[(row>>2)&3], g = {0,3,2,1}: conflict-free ds_read_b128 fragments;
//   * the LoRA rank-32 K-extension is exactly one extra stage.
constexpr int PK = 32;

template <int ROWS, int NW>
OVLA_DEV void stage_tile32(const bf16_bits* __restrict__ G, int64_t ld, int row0, int row_last, int k0, int K,
                           bf16_bits* lds_tile, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 16 / NW;
  static_assert(PER_WAVE * 16 * NW == ROWS, "tile rows must split evenly over the waves");
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rbase = (wave * PER_WAVE + i) * 16;   // one instruction writes rows rbase..rbase+15 (64 B each)
    const int r = rbase + (lane >> 2);
    const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
    const int kk = k0 + c * 8;
    int gr = row0 + r;
    gr = gr < row_last ? gr : row_last;
    const bf16_bits* src = (kk < K) ? (G + (int64_t)gr * ld + kk) : g_ovla_zero_chunk;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + rbase * PK), 16, 0, 0);
  }
}

template <int ROWS, int NW>
OVLA_DEV void stage_offsets32(uint32_t (&off)[ROWS / 16 / NW], int64_t ld, int row0, int row_last, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 16 / NW;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int r = (wave * PER_WAVE + i) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
    int gr = row0 + r;
    gr = gr < row_last ? gr : row_last;
    off[i] = (uint32_t)(((int64_t)gr * ld + c * 8) * 2);
  }
}

template <int ROWS, int NW>
OVLA_DEV void stage_tile32_fast(const char* __restrict__ base_k, const uint32_t (&off)[ROWS / 16 / NW], bf16_bits* lds_tile, int wave) {
  constexpr int PER_WAVE = ROWS / 16 / NW;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rbase = (wave * PER_WAVE + i) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_k + off[i]),
                                     (__attribute__((address_space(3))) void*)(lds_tile + rbase * PK), 16, 0, 0);
  }
}

OVLA_DEV bf16x8_bits lds_frag32(const bf16_bits* tile, int row, int chunk) {
  const int phys = chunk ^ ((4 - ((row >> 2) & 3)) & 3);
  return *reinterpret_cast<const bf16x8_bits*>(tile + row * PK + phys * 8);
}

template <int N>
OVLA_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

template <int BM, int BN, int WM, int WN, int STAGES, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_pipe_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int STAGE_ELEMS = (BM + BN) * PK;
  constexpr int G = (BM + BN) / 16 / NW;   // LDS-DMA instructions per wave per stage
  constexpr int D = STAGES - 1;            // stages in flight
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_bits* smem = reinterpret_cast<bf16_bits*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int split = bid / tiles_mn;
  const int t_mn = bid - split * tiles_mn;
  constexpr int GROUP = 4;
  const int group_sz = GROUP * p.tiles_n;
  const int gid = t_mn / group_sz;
  const int first_m = gid * GROUP;
  const int gm = (p.tiles_m - first_m) < GROUP ? (p.tiles_m - first_m) : GROUP;
  const int in_group = t_mn - gid * group_sz;
  const int tm = first_m + in_group % gm;
  const int tn = in_group / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int T1 = (p.K + PK - 1) / PK, T2 = p.K2 > 0 ? (p.K2 + PK - 1) / PK : 0;
  const int T = T1 + T2;
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

  uint32_t offA[BM / 16 / NW], offB[BN / 16 / NW];
  stage_offsets32<BM, NW>(offA, p.lda, m0, p.M - 1, wave, lane);
  stage_offsets32<BN, NW>(offB, p.ldb, n0, p.N - 1, wave, lane);
  const int t_fast = p.fast_addr ? p.K / PK : 0;

  auto stage = [&](int t) {
    bf16_bits* sA = smem + ((t - t_begin) % STAGES) * STAGE_ELEMS;
    bf16_bits* sB = sA + BM * PK;
    if (t < t_fast) {
      stage_tile32_fast<BM, NW>(reinterpret_cast<const char*>(p.A) + (int64_t)t * (PK * 2), offA, sA, wave);
      stage_tile32_fast<BN, NW>(reinterpret_cast<const char*>(p.B) + (int64_t)t * (PK * 2), offB, sB, wave);
    } else if (t < T1) {
      const int k0 = t * PK;
      stage_tile32<BM, NW>(p.A, p.lda, m0, p.M - 1, k0, p.K, sA, wave, lane);
      stage_tile32<BN, NW>(p.B, p.ldb, n0, p.N - 1, k0, p.K, sB, wave, lane);
    } else {
      const int k0 = (t - T1) * PK;
      stage_tile32<BM, NW>(p.A2 + a2_col0, p.lda2, m0, p.M - 1, k0, p.K2, sA, wave, lane);
      stage_tile32<BN, NW>(p.B2, p.ldb2, n0, p.N - 1, k0, p.K2, sB, wave, lane);
    }
  };

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (t_begin + s < t_end) stage(t_begin + s);

  if constexpr (MODE == 1) {
    // Round-3 variant of the ring: the BK = 64 loop's structure on BK = 32 stages.  (1) The LAST m-tile row of every stage is deferred across
    // the barrier: its fragments stay in registers (a_def, the previous stage's B fragments) and its NT MFMAs issue right after the next
    // stage's first fragment reads, so the matrix pipe has work while those reads are in flight instead of draining at every barrier -- with a
    // barrier per 32 K elements that drain was the ring's 17 %.  (2) The refill of the freed slot is not one burst after the barrier: its G
    // LDS-DMA pieces go out one per row slot.  Stages alternate between two B-fragment register sets (no copies).
    static_assert(MT >= 2 && G <= MT - 1, "tuned ring: one DMA piece per row slot");
    constexpr int PA = BM / 16 / NW;
    bf16x8_bits bfr[2][NT];
    bf16x8_bits a_def = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[1][j] = a_def;
    const int chunk = lane >> 4;
    const int arow = wm * WTM + (lane & 15), brow = wn * WTN + (lane & 15);
    auto body = [&](const int t, auto par_tag) {
      constexpr int PAR = decltype(par_tag)::value;
      const int newer = (t_end - 1 - t) < (D - 1) ? (t_end - 1 - t) : (D - 1);
      if (newer >= 3) wait_vmcnt<3 * G>();
      else if (newer == 2) wait_vmcnt<2 * G>();
      else if (newer == 1) wait_vmcnt<G>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const bf16_bits* sA = smem + ((t - t_begin) % STAGES) * STAGE_ELEMS;
      const bf16_bits* sB = sA + BM * PK;
      const int tn = t + D;                                  // the stage that refills the slot stage t - 1 used
      const bool refill = tn < t_end, fast = tn < t_fast;
      bf16_bits* nA = smem + ((tn - t_begin) % STAGES) * STAGE_ELEMS;
      bf16_bits* nB = nA + BM * PK;
      const char* gA = reinterpret_cast<const char*>(p.A) + (int64_t)tn * (PK * 2);
      const char* gB = reinterpret_cast<const char*>(p.B) + (int64_t)tn * (PK * 2);
      if (refill && !fast) stage(tn);                        // K tail / LoRA K-extension stages: the simple burst
      auto piece = [&](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        if (refill && fast) {
          if constexpr (q < PA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + offA[q]),
                                             (__attribute__((address_space(3))) void*)(nA + (wave * PA + q) * 16 * PK), 16, 0, 0);
          else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + offB[q - PA]),
                                             (__attribute__((address_space(3))) void*)(nB + (wave * (G - PA) + (q - PA)) * 16 * PK), 16, 0, 0);
        }
      };
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[PAR][j] = lds_frag32(sB, brow + j * 16, chunk);
      bf16x8_bits a_cur = lds_frag32(sA, arow, chunk);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < NT; ++j)      // deferred last row of the previous stage (zeros on the first pass)
        acc[MT - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[PAR ^ 1][j], a_def, acc[MT - 1][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
      static_for<MT - 1>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const bf16x8_bits a_nxt = lds_frag32(sA, arow + (i + 1) * 16, chunk);
        if constexpr (i < G) piece(std::integral_constant<int, i>{});
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[PAR][j], a_cur, acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (i < G) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        a_cur = a_nxt;
      });
      a_def = a_cur;
      __builtin_amdgcn_s_setprio(0);
    };
    int t = t_begin;
    for (; t + 1 < t_end; t += 2) {
      body(t, std::integral_constant<int, 0>{});
      body(t + 1, std::integral_constant<int, 1>{});
    }
    int last_par = 1;
    if (t < t_end) { body(t, std::integral_constant<int, 0>{}); last_par = 0; }
    // the last stage's deferred row
    if (last_par == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[MT - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], a_def, acc[MT - 1][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[MT - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][j], a_def, acc[MT - 1][j], 0, 0, 0);
    }
  } else
  for (int t = t_begin; t < t_end; ++t) {
    // stage t has landed once at most (stages issued after it) x G of this wave's DMAs are still outstanding
    const int newer = (t_end - 1 - t) < (D - 1) ? (t_end - 1 - t) : (D - 1);
    if (newer >= 3) wait_vmcnt<3 * G>();
    else if (newer == 2) wait_vmcnt<2 * G>();
    else if (newer == 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's DMAs for stage t landed; everyone finished reading stage t-1
    if (t + D < t_end) stage(t + D);  // refills the buffer stage t-1 used
    const bf16_bits* sA = smem + ((t - t_begin) % STAGES) * STAGE_ELEMS;
    const bf16_bits* sB = sA + BM * PK;
    // fragment reads are software-pipelined against the MFMAs inside the stage: row i+1's A fragment is in flight
    // while row i's NT MFMAs issue (sched_group_barrier pins the 1 ds_read : NT mfma interleave)
    bf16x8_bits b[NT];
    const int chunk = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = lds_frag32(sB, wn * WTN + j * 16 + (lane & 15), chunk);
    bf16x8_bits a_cur = lds_frag32(sA, wm * WTM + (lane & 15), chunk);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      bf16x8_bits a_nxt = a_cur;
      if (i + 1 < MT) a_nxt = lds_frag32(sA, wm * WTM + (i + 1) * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a_cur, acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // NT MFMA
      a_cur = a_nxt;
    }
    __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (p.split_k > 1) {
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
  constexpr int LDSW = WTN + 4;
  constexpr int RM = MT < 2 ? MT : 2;
  constexpr int QUADS = WTN / 4;
  float* wstage = reinterpret_cast<float*>(smem_raw) + wave * (RM * 16 * LDSW);
#pragma unroll
  for (int round = 0; round < MT / RM; ++round) {
    __syncthreads();
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

// Sums the rem_splits partial slabs of remainder tile `rt` IN SLAB ORDER and applies the epilogue, for the quads q = q0, q0 + qstride, ...:
// the body of gemm_hybrid_reduce_kernel and of the GEMM kernel's in-launch reduce (the last unit of a tile to arrive), so both give the same bits.
template <int BM, int BN>
OVLA_DEV void hybrid_reduce_quads(const GemmParams& p, int rt, int q0, int qstride) {
  const int t_mn = p.full_tiles + rt;
  constexpr int GROUP = 8;
  const int group_sz = GROUP * p.tiles_n;
  const int gid = t_mn / group_sz;
  const int first_m = gid * GROUP;
  const int gm = (p.tiles_m - first_m) < GROUP ? (p.tiles_m - first_m) : GROUP;
  const int in_group = t_mn - gid * group_sz;
  const int m0 = (first_m + in_group % gm) * BM, n0 = (in_group / gm) * BN;
  const float* slab0 = p.ws + (int64_t)rt * p.rem_splits * (BM * BN);
  for (int q = q0; q < BM * BN / 4; q += qstride) {
    const int lm = q / (BN / 4), ln = (q % (BN / 4)) * 4;
    if (p.act == OVLA_ACT_SWIGLU) {   // SwiGLU pair map (4-wave configs): tile-local columns [0, 128) are gate columns tn * 128 + ln, [128, 256) their up partners
      if (ln >= 128) continue;
      const int F = p.N >> 1, mg = m0 + lm, ng = (in_group / gm) * 128 + ln;
      if (mg >= p.M || ng >= F) continue;
      f32x4 vg = {0.f, 0.f, 0.f, 0.f}, vu = {0.f, 0.f, 0.f, 0.f};
      for (int sidx = 0; sidx < p.rem_splits; ++sidx) {
        vg += *reinterpret_cast<const f32x4*>(slab0 + (int64_t)sidx * (BM * BN) + lm * BN + ln);
        vu += *reinterpret_cast<const f32x4*>(slab0 + (int64_t)sidx * (BM * BN) + lm * BN + 128 + ln);
      }
      const float ra = p.alpha * (p.rowscale_part ? p.rowscale_r[mg] : 1.f);
      bf16x4_bits o, og, ou;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        og[e] = (short)f2bf(vg[e] * ra); ou[e] = (short)f2bf(vu[e] * ra);
        o[e] = (short)f2bf(bfround(silu(bf2f((bf16_bits)og[e]))) * bf2f((bf16_bits)ou[e]));
      }
      *reinterpret_cast<bf16x4_bits*>(p.C + (int64_t)mg * p.ldc + ng) = o;
      if (p.Cpre) {
        *reinterpret_cast<bf16x4_bits*>(p.Cpre + (int64_t)mg * p.N + ng) = og;
        *reinterpret_cast<bf16x4_bits*>(p.Cpre + (int64_t)mg * p.N + F + ng) = ou;
      }
      continue;
    }
    const int m = m0 + lm, n = n0 + ln;
    if (m >= p.M || n >= p.N) continue;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int sidx = 0; sidx < p.rem_splits; ++sidx) v += *reinterpret_cast<const f32x4*>(slab0 + (int64_t)sidx * (BM * BN) + lm * BN + ln);
    const float rs = p.rowscale_part ? p.rowscale_r[m] : 1.f;      // RMSNorm fold, consumer side: rstd written by the GEMM kernel's prologue
    if (p.rope_cos && n < p.rope_cols) {   // BN is a multiple of 128 here (launch_cfg checks): the partner quad is in the same tile
      f32x4 vp = {0.f, 0.f, 0.f, 0.f};
      for (int sidx = 0; sidx < p.rem_splits; ++sidx) vp += *reinterpret_cast<const f32x4*>(slab0 + (int64_t)sidx * (BM * BN) + lm * BN + (ln ^ 64));
      rope_store(p, m, n, v * rs, vp * rs);
      continue;
    }
    const f32x4 st = epilogue_store(p, m, n, v * rs);
    if (p.rowsq_out) {   // producer side: 16 consecutive threads hold one row's 64-column group (rows / columns outside the matrix skipped above, whole groups at a time)
      const float sq = rowsq16(st);
      if (((ln >> 2) & 15) == 0) p.rowsq_out[(int64_t)m * (p.N >> 6) + (n >> 6)] = sq;
    }
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_hybrid_reduce_kernel(const GemmParams p) {
  hybrid_reduce_quads<BM, BN>(p, blockIdx.y, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

static int g_num_cus = 0;
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

// Hybrid schedule of one tile config: whole rounds of the chip as full tiles, the partial last round split along K.
// ONE cost model (seconds) picks the split and, in auto mode, the tile config itself.  Per config (calibrated on MI355X with
// tools/gemm_sweep.py, gemm_m608.py, gemm_small_m.py): `rate` = sustained in-loop FLOP/s with every CU saturated, `t0` = fixed
// per-workgroup cost (prologue latency + epilogue), `tk1` = time per K tile of a workgroup that has its CU to itself (its own
// latency chain: a lone 4-wave 128x128 workgroup needs 0.86 us per K tile, two sharing a CU 0.95 us for both).  A round with u
// units on C CUs puts w = min(bpc, ceil(u / C)) workgroups on a CU and takes  Tk * max(tk1, w * tkc) + t0.
struct TileCfg { int id, BM, BN, bpc; double rate, t0, tk1; };
static const TileCfg kTileCfgs[] = {
    {17, 256, 256, 1, 1.30e15, 7e-6, 0.0},
    {1, 128, 128, 2, 1.13e15, 4e-6, 0.86e-6},
    {2, 64, 128, 3, 0.90e15, 3e-6, 0.57e-6},
    {5, 128, 32, 4, 0.60e15, 3e-6, 0.53e-6},
};
static const TileCfg& tile_cfg(int BM, int BN) {
  for (const TileCfg& c : kTileCfgs)
    if (c.BM == BM && c.BN == BN) return c;
  static const TileCfg generic = {0, 128, 128, 1, 0.9e15, 5e-6, 0.0};   // experimental configs: no hybrid tuning
  return generic;
}

struct HybridPlan { int full_tiles, rem_tiles, rem_splits; double est; };

static HybridPlan plan_hybrid(int M, int N, int T, const TileCfg& c, int64_t ws_floats) {
  const int C = num_cus(), slots = C * c.bpc;
  const double tkc = 2.0 * c.BM * c.BN * BK / (c.rate / C);
  auto round_time = [&](int units, int tk) {
    const int w = std::min(c.bpc, cdiv(units, C));
    return tk * std::max(c.tk1, w * tkc) + c.t0;
  };
  const int tiles = cdiv(M, c.BM) * cdiv(N, c.BN);
  HybridPlan pl{tiles, 0, 1, 0.0};
  const int full_rounds = tiles / slots, rem = tiles % slots;
  pl.est = full_rounds * round_time(slots, T);
  if (rem == 0) return pl;
  double best_t = round_time(rem, T);   // unsplit: one more (partial) round
  int best = 1;
  if (ws_floats > 0) {
    for (int sp = 2; sp <= 8 && sp * 4 <= T; ++sp) {
      if ((int64_t)rem * sp * c.BM * c.BN > ws_floats) break;
      const int units = rem * sp, tk = cdiv(T, sp);
      const double t = (units <= slots ? round_time(units, tk) : cdiv(units, slots) * round_time(slots, tk)) +
                       1.0 * rem * sp * (4.0 * c.BM * c.BN) / 5e12 + 5e-6;   // slab writes overlap; the reduce kernel reads them
      if (t < best_t) { best_t = t; best = sp; }
    }
  }
  pl.est += best_t;
  if (best > 1) { pl.full_tiles = tiles - rem; pl.rem_tiles = rem; pl.rem_splits = best; }
  return pl;
}

static int pick_tile(int M, int N, int T, int k2_group_n, int64_t ws_floats, HybridPlan* out) {
  double best = 1e30;
  int tile = 1;
  for (const TileCfg& c : kTileCfgs) {
    if (k2_group_n > 0 && k2_group_n % c.BN != 0) continue;
    const HybridPlan pl = plan_hybrid(M, N, T, c, ws_floats);
    if (pl.est < best) {
      best = pl.est; tile = c.id;
      if (out) *out = pl;
    }
  }
  return tile;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(GemmParams& p, hipStream_t stream, int64_t ws_bytes = 0, bool hybrid = false) {
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.N, BN);
  if (p.k2_group_n > 0 && (p.k2_group_n % BN) != 0) {
    ovla_set_error("ovla_gemm_bf16: k2_group_n=%d is not a multiple of the N tile %d", p.k2_group_n, BN);
    return OVLA_EINVAL;
  }
  const int splits = p.split_k > 1 ? p.split_k : 1;
  size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16_bits);
  const size_t epi = (size_t)WM * WN * 32 * (BN / WN + 4) * sizeof(float);   // epilogue staging slabs reuse the same LDS
  if (epi > lds) lds = epi;
  if (BM == 128 && BN == 128) lds += BM * sizeof(float);                       // s_rstd of the RMSNorm fold, behind both
  if ((p.rowsq_out || p.rowscale_part) && !(BM == 128 && BN == 128 && WM == 2 && WN == 2)) {
    ovla_set_error("ovla_gemm_bf16: the RMSNorm fold (rowsq_out / rowscale_part) runs on the 128x128 tile only; this problem resolved to %dx%d", BM, BN);
    return OVLA_EINVAL;
  }
  auto kern = gemm_nt_kernel<BM, BN, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles = p.tiles_m * p.tiles_n;
  p.full_tiles = tiles;
  p.rem_tiles = 0;
  p.rem_splits = 1;
  if (hybrid && splits == 1 && p.ws != nullptr) {
    static_assert((size_t)WM * WN * 32 * (BN / WN + 4) * sizeof(float) <= 160 * 1024, "epilogue slabs exceed LDS");
    const HybridPlan pl = plan_hybrid(p.M, p.N, p.T1 + p.T2, tile_cfg(BM, BN), ws_bytes / 4);
    p.full_tiles = pl.full_tiles; p.rem_tiles = pl.rem_tiles; p.rem_splits = pl.rem_splits;
  }
  if (p.rem_tiles == 0 || p.rem_tiles > p.hyb_cnt_n) p.hyb_cnt = nullptr;    // no remainder, or more remainder tiles than counters: separate reduce launch
  const unsigned nblk = p.rem_tiles > 0 ? (unsigned)(p.full_tiles + p.rem_tiles * p.rem_splits) : (unsigned)(tiles * splits);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WM * WN), lds, stream, p);
  OVLA_CHECK_LAUNCH("ovla_gemm_bf16");
  if (p.rem_tiles > 0 && p.hyb_cnt == nullptr) {
    hipLaunchKernelGGL((gemm_hybrid_reduce_kernel<BM, BN>), dim3(BM * BN / 4 / 256 / 4, p.rem_tiles), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(hybrid reduce)");
  } else if (p.rem_tiles > 0) {
    // reduced inside the GEMM launch by each tile's last-arriving unit
  } else if (splits > 1) {
    const int64_t quads = (int64_t)p.M * (p.N / 4);
    int blocks = cdiv(quads, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(split-k reduce)");
  }
  return OVLA_OK;
}

template <int BM, int BN, int WM, int WN, int STAGES, int MODE = 0>
int launch_pipe(GemmParams& p, hipStream_t stream) {
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.N, BN);
  if (p.k2_group_n > 0 && (p.k2_group_n % BN) != 0) {
    ovla_set_error("ovla_gemm_bf16: k2_group_n=%d is not a multiple of the N tile %d", p.k2_group_n, BN);
    return OVLA_EINVAL;
  }
  const int T = cdiv(p.K, PK) + (p.K2 > 0 ? cdiv(p.K2, PK) : 0);
  if (p.split_k > T) p.split_k = T;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  size_t lds = (size_t)STAGES * (BM + BN) * PK * sizeof(bf16_bits);
  const size_t epi = (size_t)WM * WN * 32 * (BN / WN + 4) * sizeof(float);
  if (epi > lds) lds = epi;
  auto kern = gemm_nt_pipe_kernel<BM, BN, WM, WN, STAGES, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n * splits)), dim3(64 * WM * WN), lds, stream, p);
  OVLA_CHECK_LAUNCH("ovla_gemm_bf16(pipe)");
  if (splits > 1) {
    const int64_t quads = (int64_t)p.M * (p.N / 4);
    int blocks = cdiv(quads, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(split-k reduce)");
  }
  return OVLA_OK;
}

template <int KEXT, int ABL = 0, int WNW = 2, bool RMAP = false, bool GMAP = false>
int launch_w4(GemmParams& p, hipStream_t stream, int64_t ws_bytes, bool hybrid) {
  constexpr int BM = 128 * (4 / WNW), BN = 256;
  if (p.K % BK != 0 || p.K2 != 32 * KEXT || !p.fast_addr || p.a_group_n > 0 || (p.k2_group_n > 0 && (p.k2_group_n % 256) != 0)) {
    ovla_set_error("ovla_gemm_bf16: the 4-wave configs need K %% 64 == 0, a K-extension of 0, 32, 64 or 96 columns and no block-diagonal mode");
    return OVLA_EINVAL;
  }
  if ((p.rowsq_out || p.rowscale_part) && (WNW != 4 || (p.rowscale_part && p.rowscale_slots > 64) || (p.rowsq_out && RMAP))) {
    ovla_set_error("ovla_gemm_bf16: among the 4-wave configs the RMSNorm fold runs on the 128x256 one only (rowscale_part: K <= 4096)");
    return OVLA_EINVAL;
  }
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = GMAP ? (p.N / 2) / 128 : cdiv(p.N, BN);
  if (GMAP) {
    if ((p.N % 256) != 0 || p.split_k > 1 || p.rowsq_out || p.bias || p.residual || p.colscale || p.film_gamma || p.dact_src || p.rope_cos ||
        (p.Cpre && ((((uintptr_t)p.Cpre) & 15) != 0 || (p.N % 8) != 0)) || (p.k2_group_n > 0 && p.k2_group_n != p.N / 2)) {
      ovla_set_error("ovla_gemm_bf16: act = OVLA_ACT_SWIGLU needs N = 2 F with F %% 128 == 0, a K-extension grouped by F (or not at all) and nothing else in the epilogue but alpha, "
                     "the RMSNorm-fold row scale and C_pre (the [M, 2 F] projection output)");
      return OVLA_EINVAL;
    }
  }
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16_bits) + (WNW == 4 ? 128 * sizeof(float) : 0);   // + s_rstd of the RMSNorm fold
  auto kern = gemm_nt_w4_kernel<KEXT, ABL, WNW, RMAP, GMAP>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles = p.tiles_m * p.tiles_n;
  p.full_tiles = tiles; p.rem_tiles = 0; p.rem_splits = 1;
  if (hybrid && splits == 1 && p.ws != nullptr) {
    static const TileCfg cfg128 = {22, 128, 256, 1, 1.2e15, 5e-6, 0.0};   // (not in kTileCfgs: tile = 0 does not pick this configuration by itself)
    const HybridPlan pl = plan_hybrid(p.M, p.N, p.T1 + p.T2, WNW == 2 ? tile_cfg(256, 256) : cfg128, ws_bytes / 4);   // (256x256: same plan as the 8-wave config; the K-extension is not a K tile here)
    p.full_tiles = pl.full_tiles; p.rem_tiles = pl.rem_tiles; p.rem_splits = pl.rem_splits;
  }
  p.hyb_cnt = nullptr;
  const unsigned nblk = p.rem_tiles > 0 ? (unsigned)(p.full_tiles + p.rem_tiles * p.rem_splits) : (unsigned)(tiles * splits);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, stream, p);
  OVLA_CHECK_LAUNCH("ovla_gemm_bf16(w4)");
  if (p.rem_tiles > 0) {
    hipLaunchKernelGGL((gemm_hybrid_reduce_kernel<BM, BN>), dim3(BM * BN / 4 / 256 / 4, p.rem_tiles), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(w4 hybrid reduce)");
  } else if (splits > 1) {
    const int64_t quads = (int64_t)p.M * (p.N / 4);
    int blocks = cdiv(quads, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    OVLA_CHECK_LAUNCH("ovla_gemm_bf16(w4 split-k reduce)");
  }
  return OVLA_OK;
}

}  // namespace

extern "C" int ovla_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t K2, int32_t k2_group_n, int64_t workspace_bytes, int32_t* tile,
                              int32_t* full_tiles, int32_t* rem_tiles, int32_t* rem_splits, double* est_seconds) {
  OVLA_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0 && tile, "ovla_gemm_plan: bad arguments");
  HybridPlan pl{0, 0, 1, 0.0};
  *tile = pick_tile(M, N, cdiv(K, BK) + (K2 > 0 ? cdiv(K2, BK) : 0), k2_group_n, workspace_bytes / 4, &pl);
  if (full_tiles) *full_tiles = pl.full_tiles;
  if (rem_tiles) *rem_tiles = pl.rem_tiles;
  if (rem_splits) *rem_splits = pl.rem_splits;
  if (est_seconds) *est_seconds = pl.est;
  return OVLA_OK;
}

extern "C" int64_t ovla_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  return split_k > 1 ? (int64_t)split_k * M * N * 4 : 0;
}

static thread_local int32_t* g_resolve_only = nullptr;   // ovla_gemm_resolved_tile: run the argument checks and the schedule decision, launch nothing

extern "C" int ovla_gemm_resolved_tile(const ovla_gemm_args* a, int32_t* tile) {
  OVLA_REQUIRE(tile != nullptr, "ovla_gemm_resolved_tile: null output");
  *tile = -1;
  g_resolve_only = tile;
  const int rc = ovla_gemm_bf16(a, nullptr);
  g_resolve_only = nullptr;
  return rc;
}

extern "C" int ovla_gemm_bf16(const ovla_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a != nullptr, "ovla_gemm_bf16: null args");
  OVLA_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "ovla_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  OVLA_REQUIRE(a->A && a->B && a->C, "ovla_gemm_bf16: null A/B/C");
  OVLA_REQUIRE((a->K % 8) == 0 && (a->N % 8) == 0, "ovla_gemm_bf16: K=%d and N=%d must be multiples of 8", a->K, a->N);
  OVLA_REQUIRE((a->lda % 8) == 0 && (a->ldb % 8) == 0 && (a->ldc % 4) == 0, "ovla_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
  OVLA_REQUIRE(a->lda >= a->K && a->ldb >= a->K && a->ldc >= (a->act == OVLA_ACT_SWIGLU ? a->N / 2 : a->N), "ovla_gemm_bf16: leading dimension smaller than extent");
  OVLA_REQUIRE(aligned16(a->A) && aligned16(a->B) && (((uintptr_t)a->C) & 7) == 0, "ovla_gemm_bf16: A/B need 16-byte, C 8-byte alignment");
  if (a->a_group_n > 0) {
    OVLA_REQUIRE(a->K2 <= 0 && (a->N % a->a_group_n) == 0 && (a->a_group_n == 32 || a->a_group_n % 128 == 0) && a->lda >= (int64_t)(a->N / a->a_group_n) * a->K,
                 "ovla_gemm_bf16: block-diagonal mode needs N %% a_group_n == 0, a_group_n 32 or a multiple of 128, lda >= groups*K, no K-extension");
    OVLA_REQUIRE(a->tile == 0 || a->tile == 5 || a->tile == 2 || a->tile == 1, "ovla_gemm_bf16: block-diagonal mode runs on the BK=64 kernel tiles only");
  }
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
  if (a->rope_cos || a->rope_sin) {
    OVLA_REQUIRE(a->rope_cos && a->rope_sin && a->rope_S > 0 && a->rope_cols > 0 && a->rope_cols <= a->N && (a->rope_cols % 128) == 0,
                 "ovla_gemm_bf16: RoPE epilogue needs both tables, rope_S > 0 and rope_cols a multiple of the head dim 128 (<= N)");
    OVLA_REQUIRE(!a->bias && !a->residual && !a->colscale && !a->film_gamma && !a->C_pre && !a->dact_src && a->act == OVLA_ACT_NONE &&
                     (a->alpha == 0.f || a->alpha == 1.f),
                 "ovla_gemm_bf16: the RoPE epilogue excludes the other epilogues");
    OVLA_REQUIRE((((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin) & 7) == 0, "ovla_gemm_bf16: RoPE tables need 8-byte alignment");
  }
  if (a->dact_src) {
    OVLA_REQUIRE(a->dact_mode == 1 || a->dact_mode == 2, "ovla_gemm_bf16: dact_mode %d (1 = activation derivative, 2 = SwiGLU)", a->dact_mode);
    OVLA_REQUIRE((a->ld_dact % 4) == 0 && (((uintptr_t)a->dact_src) & 7) == 0, "ovla_gemm_bf16: dact_src alignment");
    OVLA_REQUIRE(!a->bias && !a->residual && !a->colscale && !a->film_gamma && !a->C_pre && a->act == OVLA_ACT_NONE,
                 "ovla_gemm_bf16: a backward epilogue excludes the forward ones");
    if (a->dact_mode == 2) OVLA_REQUIRE(a->ldc >= 2 * (int64_t)a->N && a->ld_dact >= 2 * (int64_t)a->N, "ovla_gemm_bf16: SwiGLU backward writes/reads [M, 2N]");
  }
  if (a->split_k > 1)
    OVLA_REQUIRE(a->workspace != nullptr && aligned16(a->workspace) && a->workspace_bytes >= ovla_gemm_workspace_bytes(a->M, a->N, a->split_k),
                 "ovla_gemm_bf16: split_k=%d needs a 16-byte aligned workspace of %lld bytes", a->split_k, (long long)ovla_gemm_workspace_bytes(a->M, a->N, a->split_k));

  GemmParams p;
  p.A = (const bf16_bits*)a->A; p.B = (const bf16_bits*)a->B;
  p.A2 = (const bf16_bits*)a->A2; p.B2 = (const bf16_bits*)a->B2;
  p.lda = a->lda; p.ldb = a->ldb; p.lda2 = a->lda2; p.ldb2 = a->ldb2;
  p.C = (bf16_bits*)a->C; p.Cpre = (bf16_bits*)a->C_pre; p.ldc = a->ldc;
  p.bias = (const bf16_bits*)a->bias; p.colscale = (const bf16_bits*)a->colscale;
  p.residual = (const bf16_bits*)a->residual; p.ldr = a->ldr;
  p.film_gamma = (const bf16_bits*)a->film_gamma; p.film_beta = (const bf16_bits*)a->film_beta; p.film_rows = a->film_rows;
  p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->K2 > 0 ? a->K2 : 0;
  p.k2_group_n = a->k2_group_n; p.a_group_n = a->a_group_n; p.act = a->act; p.split_k = a->split_k > 1 ? a->split_k : 1;
  p.ws = (float*)a->workspace;
  p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
  p.dact_src = (const bf16_bits*)a->dact_src; p.ld_dact = a->ld_dact; p.dact_mode = a->dact_src ? a->dact_mode : 0; p.dact_act = a->dact_act;
  p.rope_cos = nullptr; p.rope_sin = nullptr; p.rope_S = a->rope_S; p.rope_cols = a->rope_cols;   // enabled below for the config that fuses it
  p.hyb_cnt = (unsigned*)a->hybrid_counters; p.hyb_cnt_n = a->hybrid_counters ? a->n_hybrid_counters : 0;
  p.rowsq_out = a->rowsq_out; p.rowscale_part = a->rowscale_part; p.rowscale_slots = a->rowscale_slots; p.rowscale_eps = a->rowscale_eps; p.rowscale_r = a->rowscale_r;
  if (a->rowsq_out) OVLA_REQUIRE((a->N % 64) == 0 && a->split_k <= 1 && !a->dact_src && !a->film_gamma && (((uintptr_t)a->rowsq_out) & 3) == 0,
                                 "ovla_gemm_bf16: rowsq_out needs N %% 64 == 0, no split_k and a forward epilogue");
  if (a->rowscale_part)
    OVLA_REQUIRE(a->rowscale_slots > 0 && (a->rowscale_slots % 8) == 0 && a->rowscale_slots * 64 == a->K && a->rowscale_r && aligned16(a->rowscale_part) && a->split_k <= 1 &&
                     a->a_group_n == 0, "ovla_gemm_bf16: rowscale_part needs rowscale_slots * 64 == K (multiple of 8 slots), a rowscale_r scratch and no split_k");
  p.T1 = cdiv(p.K, BK); p.T2 = p.K2 > 0 ? cdiv(p.K2, BK) : 0;
  if (p.split_k > p.T1 + p.T2) p.split_k = p.T1 + p.T2;
  p.full_tiles = 0; p.rem_tiles = 0; p.rem_splits = 1;
#ifdef OVLA_GEMM_ABLATE
  p.dbg = (a->tile >= 1000) ? (a->tile / 1000) : 0;
#else
  OVLA_REQUIRE(a->tile < 1000, "ovla_gemm_bf16: tile %d selects a timing ablation; this library was built without OVLA_GEMM_ABLATE", a->tile);
  p.dbg = 0;
#endif
  p.fast_swiglu_bwd = a->dact_src && a->dact_mode == 2 && !a->bias && !a->residual && !a->colscale && !a->film_gamma && !a->C_pre && !a->rope_cos &&
                      a->act == OVLA_ACT_NONE && (((uintptr_t)a->dact_src | (uintptr_t)a->C) & 15) == 0 && (a->ld_dact % 8) == 0 && (a->ldc % 8) == 0 && (a->N % 8) == 0;
  const bool rope_plain = !a->bias && !a->C_pre && !a->colscale && !a->residual && !a->film_gamma && !a->dact_src && a->act == OVLA_ACT_NONE;
  p.fast_epi = !a->film_gamma && !a->dact_src && !a->rope_cos && (!a->C_pre || (((uintptr_t)a->C_pre) & 15) == 0) &&
               (!a->colscale || (((uintptr_t)a->colscale) & 15) == 0) &&
               (!a->residual || ((((uintptr_t)a->residual) & 15) == 0 && (a->ldr % 8) == 0)) && (!a->bias || (((uintptr_t)a->bias) & 15) == 0);
  p.fast_addr = ((int64_t)p.M * p.lda * 2 < (int64_t)4e9 && (int64_t)p.N * p.ldb * 2 < (int64_t)4e9) ? 1 : 0;

  int tile = a->tile % 1000;
  const int64_t wsb = a->workspace ? a->workspace_bytes : 0;
  if (a->act == OVLA_ACT_SWIGLU)
    OVLA_REQUIRE((tile == 0 || tile == 18 || tile == 118 || tile == 22 || tile == 122) && a->ldc >= a->N / 2 && (a->K % BK) == 0,
                 "ovla_gemm_bf16: act = OVLA_ACT_SWIGLU (C [M, N / 2] = silu(gate) * up of the stacked [gate; up] projection) runs on the 4-wave configurations only (tile 0 / 18 / 118 / 22 / 122, K %% 64 == 0)");
  bool hybrid = false;
  if (tile == 0) {
    // auto schedule.  Skinny outputs (LoRA t / dt, N <= 128) and small M (action head) are HBM-bound weight/activation
    // streams: split K so that >= ~256 workgroups stream concurrently.  Large problems take the 256x256 tile (in-kernel
    // ~1.3 PFLOP/s) with the hybrid remainder schedule; mid-size ones the 128x128 tile (2 workgroups per CU).
    const int T = p.T1 + p.T2;
    auto want_split = [&](int tiles) {
      int sp = 1;
      while (sp < 8 && tiles * sp < 256 && (sp * 2) * 8 <= T && ovla_gemm_workspace_bytes(p.M, p.N, sp * 2) <= wsb) sp *= 2;
      return sp;
    };
    static const bool skinny_on = []() { const char* e = getenv("OVLA_SKINNY"); return !(e && e[0] == '0'); }();   // A/B switch
    if (skinny_on && p.N == 32 && p.a_group_n == 0 && p.K <= 3072 && p.M >= 512 && p.K2 == 0 && p.split_k <= 1 && !a->bias && !a->C_pre && !a->colscale &&
        !a->residual && !a->film_gamma && !a->dact_src && !a->rope_cos && a->act == OVLA_ACT_NONE) {
      if (g_resolve_only) { *g_resolve_only = 6; return OVLA_OK; }
      hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3((unsigned)cdiv(p.M, 32)), dim3(256), 0, stream, p);
      OVLA_CHECK_LAUNCH("ovla_gemm_bf16(skinny)");
      return OVLA_OK;
    }
    if (a->act == OVLA_ACT_SWIGLU) { tile = 18; hybrid = true; }
    else if (p.N <= 32 || p.a_group_n == 32) { tile = 5; if (p.split_k <= 1) p.split_k = want_split(cdiv(p.M, 128) * cdiv(p.N, 32)); }
    else if (p.a_group_n > 0) { tile = 1; }
    else if (p.M <= 64 || p.N <= 128) { tile = 2; if (p.split_k <= 1) p.split_k = want_split(cdiv(p.M, 64) * cdiv(p.N, 128)); }
    else {   // 256x256 / 128x128 / 64x128 / 128x32 tiles, whichever the hybrid-schedule cost model predicts fastest
      hybrid = true;
      tile = pick_tile(p.M, p.N, T, p.k2_group_n, wsb / 4, nullptr);
      // The 4-wave config of the 256x256 tile (hand-scheduled K loop) where it measured ahead of the 8-wave one (tools/gemm_w4_probe.py: +4...9 % per launch on
      // the decoder shapes): K >= 4096 in whole K tiles, a LoRA K-extension of 0 / 32 / 64 / 96 columns, the alpha / bias / residual or the RoPE epilogue.  OVLA_GEMM_W4=0 switches it off.
      static const bool w4_on = []() { const char* e = getenv("OVLA_GEMM_W4"); return !(e && e[0] == '0'); }();
      static const int w4_min_k = []() { const char* e = getenv("OVLA_GEMM_W4_MINK"); return e ? atoi(e) : 4096; }();   // tuning switch
      if (w4_on && tile == 17 && p.K >= w4_min_k && (p.K % BK) == 0 && (p.K2 == 0 || p.K2 == 32 || p.K2 == 64 || p.K2 == 96) && (p.k2_group_n % 256) == 0 && p.fast_addr &&
          (p.fast_epi || (a->rope_cos && rope_plain)) && p.act == OVLA_ACT_NONE && !a->C_pre && !a->colscale && !a->rowsq_out && !a->rowscale_part && p.split_k <= 1)
        tile = 18;
    }
  }
  if (g_resolve_only) { *g_resolve_only = tile; return OVLA_OK; }
  if (a->rope_cos) {
    // fused only where one wave slab is one head (256x256 tile, 4x2 waves) and the epilogue runs in-kernel or in the hybrid reduce;
    // every other schedule computes the plain projection and rotates it with one ovla_rope launch (same arithmetic)
    // ... or, on the default 2x4 layout, where every q | k tile is an interior tile whose waves take the columns together with their
    // rotation partners (gemm_nt_kernel: rope_tile): M, N and rope_cols multiples of 256, nothing but alpha in the epilogue
    const bool fused17 = (tile == 17 || tile == 117) && p.split_k <= 1 && rope_plain && (p.M % 256) == 0 && (p.N % 256) == 0 && (a->rope_cols % 256) == 0 &&
                         (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin | (uintptr_t)a->C) & 15) == 0 && (a->ldc % 8) == 0;
    // ... and on the 128x128 tile (2x2 waves; batch-1 inference, M = 608): one head per column tile, any M (edge rows are skipped in the read-back)
    const bool fused1 = (tile == 1 || tile == 101) && p.split_k <= 1 && rope_plain && (p.N % 128) == 0 && (a->rope_cols % 128) == 0 &&
                        (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin | (uintptr_t)a->C) & 15) == 0 && (a->ldc % 8) == 0;
    // ... and on the 4-wave 256x256 config (one head per wave slab, any M)
    const bool fused18 = (tile == 18 || tile == 118) && p.split_k <= 1 && rope_plain && (p.N % 128) == 0 && (a->rope_cols % 128) == 0 &&
                         (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin | (uintptr_t)a->C) & 15) == 0 && (a->ldc % 8) == 0;
    // ... and on the 4-wave 128x256 config (two heads per column tile, the RoPE column map; no K-extension)
    const bool fused22 = (tile == 22 || tile == 122) && p.split_k <= 1 && rope_plain && p.K2 == 0 && (p.N % 256) == 0 && (a->rope_cols % 128) == 0 &&
                         (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin | (uintptr_t)a->C) & 15) == 0 && (a->ldc % 8) == 0;
    const bool fused = ((tile == 16 || tile == 116) && p.split_k <= 1) || fused17 || fused1 || fused18 || fused22;
    if (fused) {
      p.rope_cos = (const bf16_bits*)a->rope_cos; p.rope_sin = (const bf16_bits*)a->rope_sin;
    } else {
      ovla_gemm_args plain = *a;
      plain.rope_cos = plain.rope_sin = nullptr;
      if (int rc = ovla_gemm_bf16(&plain, stream_)) return rc;
      ovla_rope_args r = {};
      r.qk = a->C; r.ld = a->ldc; r.rows = a->M; r.S = a->rope_S; r.n_heads = a->rope_cols / 128; r.head_dim = 128;
      r.cos_table = a->rope_cos; r.sin_table = a->rope_sin; r.inverse = 0;
      return ovla_rope(&r, stream_);
    }
  }
  switch (tile) {
    case 1: return launch_cfg<128, 128, 2, 2>(p, stream, wsb, hybrid);
    case 2: return launch_cfg<64, 128, 1, 4>(p, stream, wsb, hybrid);
    case 3: return launch_cfg<256, 128, 4, 2>(p, stream);
    case 5: return launch_cfg<128, 32, 4, 1>(p, stream, wsb, hybrid);
    case 6:
      hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3((unsigned)cdiv(p.M, 32)), dim3(256), 0, stream, p);
      OVLA_CHECK_LAUNCH("ovla_gemm_bf16(skinny)");
      return OVLA_OK;
    case 10: return launch_pipe<256, 256, 2, 4, 4>(p, stream);
    case 11: return launch_pipe<256, 128, 2, 4, 5>(p, stream);
    case 12: return launch_pipe<256, 128, 4, 2, 5>(p, stream);
    case 13: return launch_pipe<128, 256, 2, 4, 5>(p, stream);
    case 14: return launch_pipe<128, 128, 2, 2, 4>(p, stream);
    case 15: return launch_pipe<256, 256, 2, 4, 3>(p, stream);
    case 20: return launch_pipe<256, 256, 2, 4, 4, 1>(p, stream);   // round-3 tuned ring (deferred row + spread refill), 4 and 3 stages
    case 21: return launch_pipe<256, 256, 2, 4, 3, 1>(p, stream);
    case 16: return launch_cfg<256, 256, 4, 2>(p, stream, wsb, hybrid);   // 4x2 waves (64x128 wave tiles): 1-3 % behind 2x4; one head per wave slab (fused RoPE)
    case 116: return launch_cfg<256, 256, 4, 2>(p, stream, wsb, true);
    case 17: return launch_cfg<256, 256, 2, 4>(p, stream, wsb, hybrid);   // 2x4 waves (128x64 wave tiles)
    case 117: return launch_cfg<256, 256, 2, 4>(p, stream, wsb, true);
    case 18: case 118: {   // 4-wave 256x256, register-staged operands, hand-scheduled K loop
      const bool hy = hybrid || tile == 118;
      if (p.act == OVLA_ACT_SWIGLU) {   // the SwiGLU pair map on the 256x256 tile (the fine-tune step's gate | up projection: LoRA rank 32, C_pre = the projection output)
        if (p.K2 == 0) return launch_w4<0, 0, 2, false, true>(p, stream, wsb, hy);
        if (p.K2 == 32) return launch_w4<1, 0, 2, false, true>(p, stream, wsb, hy);
        ovla_set_error("ovla_gemm_bf16: act = OVLA_ACT_SWIGLU takes a K-extension of 0 or 32 columns, not %d", p.K2);
        return OVLA_EINVAL;
      }
      switch (p.K2) {
        case 0: return launch_w4<0>(p, stream, wsb, hy);
        case 32: return launch_w4<1>(p, stream, wsb, hy);
        case 64: return launch_w4<2>(p, stream, wsb, hy);
        case 96: return launch_w4<3>(p, stream, wsb, hy);
        default: ovla_set_error("ovla_gemm_bf16: the 4-wave 256x256 config takes a K-extension of 0, 32, 64 or 96 columns, not %d", p.K2); return OVLA_EINVAL;
      }
    }
    case 22: case 122: {   // 128x256 tile on 1 x 4 waves of 128x64, the same hand-scheduled loop (batch-1 shapes: M = 608 = 4.75 row tiles)
      const bool hy = hybrid || tile == 122;
      if (p.act == OVLA_ACT_SWIGLU) {   // the SwiGLU pair map (no K-extension on this tile)
        if (p.K2 != 0) { ovla_set_error("ovla_gemm_bf16: on the 128x256 config act = OVLA_ACT_SWIGLU takes no K-extension"); return OVLA_EINVAL; }
        return launch_w4<0, 0, 4, false, true>(p, stream, wsb, hy);
      }
      if (p.rope_cos) {   // the RoPE column map (only without a K-extension: the merged / adapter-free decoder of the batch-1 chunk)
        if (p.K2 != 0) { ovla_set_error("ovla_gemm_bf16: the 128x256 config fuses RoPE only without a K-extension"); return OVLA_EINVAL; }
        return launch_w4<0, 0, 4, true>(p, stream, wsb, hy);
      }
      switch (p.K2) {
        case 0: return launch_w4<0, 0, 4>(p, stream, wsb, hy);
        case 32: return launch_w4<1, 0, 4>(p, stream, wsb, hy);
        default: ovla_set_error("ovla_gemm_bf16: the 4-wave 128x256 config takes a K-extension of 0 or 32 columns, not %d", p.K2); return OVLA_EINVAL;
      }
    }
#ifdef OVLA_GEMM_ABLATE
    case 218: return launch_w4<0, 1>(p, stream, wsb, false);
    case 318: return launch_w4<0, 2>(p, stream, wsb, false);
    case 418: return launch_w4<0, 3>(p, stream, wsb, false);
    case 518: return launch_w4<0, 4>(p, stream, wsb, false);    // no lgkmcnt waits in the rows (wrong results, timing only)
    case 618: return launch_w4<0, 8>(p, stream, wsb, false);    // no vmcnt waits in the rows
    case 718: return launch_w4<0, 12>(p, stream, wsb, false);   // neither
#endif
    case 101: return launch_cfg<128, 128, 2, 2>(p, stream, wsb, true);
    case 102: return launch_cfg<64, 128, 1, 4>(p, stream, wsb, true);
    case 105: return launch_cfg<128, 32, 4, 1>(p, stream, wsb, true);
    default: ovla_set_error("ovla_gemm_bf16: unknown tile id %d", tile); return OVLA_EINVAL;
  }
}
