// attention.hip -- flash-style attention forward / backward for the short (S <= ~1.2k) OpenVLA-OFT context on gfx950.
//
// Replaces F.scaled_dot_product_attention in timm Attention (ViT: unmasked, head_dim 64 / 72) and in the
// transformers-fork LlamaAttention (head_dim 128; bidirectional + right key padding, or causal).  No KV paging: the
// whole context streams through LDS in 64-key tiles.
//
// Lane mapping (all three kernels): scores are computed TRANSPOSED, S^T = K . Q^T with mfma_f32_16x16x32_bf16, so the
// query (forward, dQ) or key (dK/dV) index sits on the lane (lane & 15) and the 4 accumulator registers x 4 lane
// groups hold the other index.  Consequences:
//   * softmax statistics, the LSE and the O / dQ / dK / dV rescale are lane-local (two xor-shuffles per row reduce);
//   * P (and dS) leave the accumulators already shaped as the NEXT MFMA's B operand: the contraction index of the
//     second product is simply taken in the permuted order {16t+4g+j} the accumulator presents it in, and the other
//     operand (V, K, dO or Q, read TRANSPOSED from the row-major LDS tile by ds_read_b64_tr_b16) follows the same
//     permutation -- no LDS round trip, no cross-lane traffic for P.
// K/V (or Q/dO) tiles are register-staged (global_load_dwordx4 issued one tile ahead, written to LDS after the barrier)
// into rows of head_dim_padded + 16 elements: that stride makes both the ds_read_b128 row reads and the transposed
// reads bank-conflict free for head_dim 128.
// The forward for contexts > 64 tokens is attn_fwd32_kernel further down: the same scheme at 32 query rows per wave on
// mfma_f32_32x32x16_bf16 (its own LDS strides).  Every kernel on a 1-D grid maps its linear block id so that the blocks of
// one (batch, head) run on one XCD (xcd_contiguous).  Measurements and what bounds these kernels: profiles/r02_attn_pmc.md.
#include "common.h"
#include <type_traits>

namespace {

constexpr int BQ = 64;   // rows per workgroup (4 waves x 16)
constexpr int BKV = 64;  // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnParams {
  const bf16_bits *Q, *K, *V, *O, *dO;
  bf16_bits *Oout, *dQ, *dK, *dV;
  int64_t q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride;
  float* lse;
  const float* lse_in;
  float* delta;
  const int32_t* kv_len;
  int B, H, S, hd, causal;
  float scale;
  const bf16_bits *rope_cos, *rope_sin;   // optional: inverse RoPE fused into the dQ / dK epilogues
  int xcd_map;                             // 1-D grids: remap the linear block id so that one (batch, head)'s blocks share an XCD
};

OVLA_DEV bf16x4_bits lds_tr16(const bf16_bits* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_bits*)p);
}

// Transposed 16(col) x 8(contraction slot) fragment: slots jj<4 -> rows row0+4g+jj, jj>=4 -> rows row0+16+4g+(jj-4).
OVLA_DEV bf16x8_bits tr_frag(const bf16_bits* tile, int row0, int col0, int stride, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const bf16_bits* a0 = tile + (row0 + 4 * g + (i >> 2)) * stride + col0 + 4 * (i & 3);
  const bf16x4_bits lo = lds_tr16(a0);
  const bf16x4_bits hi = lds_tr16(a0 + 16 * stride);
  return bf16x8_bits{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

OVLA_DEV bf16x8_bits row_frag(const bf16_bits* tile, int row, int col0, int stride) {
  return *reinterpret_cast<const bf16x8_bits*>(tile + row * stride + col0);
}

OVLA_DEV bf16x8_bits zero8() { return bf16x8_bits{0, 0, 0, 0, 0, 0, 0, 0}; }

// global row fragment: 8 elements at column col of row `row` (clamped), zero beyond hd
OVLA_DEV bf16x8_bits gload8(const bf16_bits* base, int64_t stride, int row, int col, int hd) {
  if (col >= hd) return zero8();
  return *reinterpret_cast<const bf16x8_bits*>(base + (int64_t)row * stride + col);
}

// branch-free form (a branch around a load makes the compiler drain vmcnt at the next block boundary): head_dim == DP needs no column test;
// the padded SigLIP case (72 of 96) loads a clamped column and zeroes the fragment afterwards
template <int DP>
OVLA_DEV bf16x8_bits gload8_nb(const bf16_bits* base, int64_t stride, int row, int col, int hd) {
  if constexpr (DP != 96) return *reinterpret_cast<const bf16x8_bits*>(base + (int64_t)row * stride + col);
  const bool in = col < hd;
  const bf16x8_bits v = *reinterpret_cast<const bf16x8_bits*>(base + (int64_t)row * stride + (in ? col : 0));
  return in ? v : zero8();
}

template <int DP, int NTHREADS = 256>
struct TileStage {
  static constexpr int CH = DP / 8;              // 16-byte chunks per row
  static constexpr int TOTAL = BKV * CH;
  static constexpr int PER_THREAD = (TOTAL + NTHREADS - 1) / NTHREADS;
  static constexpr int STRIDE = DP + 16;
  bf16x8_bits r[PER_THREAD];
  OVLA_DEV void load(const bf16_bits* base, int64_t stride, int row0, int row_last, int hd, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + NTHREADS * i;
      const int rr = id / CH, ch = id % CH;
      int gr = row0 + rr;
      gr = gr < row_last ? gr : row_last;
      r[i] = (TOTAL % NTHREADS == 0 || id < TOTAL) ? gload8(base, stride, gr, ch * 8, hd) : zero8();
    }
  }
  OVLA_DEV void store(bf16_bits* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + NTHREADS * i;
      const int rr = id / CH, ch = id % CH;
      if (TOTAL % NTHREADS == 0 || id < TOTAL) *reinterpret_cast<bf16x8_bits*>(tile + rr * STRIDE + ch * 8) = r[i];
    }
  }
};

OVLA_DEV short f2bf_s(float f) { return (short)f2bf(f); }

// Workgroups are handed to the 8 XCDs round-robin by linear id.  The blocks of one (batch, head) read the same K / V (or Q / dO) rows, so
// they should meet in ONE XCD's L2: linear id n runs on XCD n % 8 and becomes work item (n % 8) * (total / 8) + n / 8, which gives every
// XCD a contiguous run of work items (the total % 8 trailing ids keep their own index).
OVLA_DEV int xcd_contiguous(int n, int total) {
  const int per = total >> 3;
  return n < 8 * per ? (n & 7) * per + (n >> 3) : n;
}

// ---------------------------------------------------------------------------------------------------------------
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p) {
  constexpr int KS = DP / 32;   // contraction steps over head_dim
  constexpr int DT = DP / 16;   // output d tiles
  constexpr int STRIDE = DP + 16;
  constexpr int NTH = 64 * NW;
  constexpr int BQW = 16 * NW;  // query rows per workgroup (NW = 8: 128 rows share every K/V tile)
  // K/V tiles are double buffered in LDS: tile t+1 is written while tile t is still being read -> ONE barrier per tile
  __shared__ __attribute__((aligned(16))) bf16_bits Ks[2][BKV * STRIDE];
  __shared__ __attribute__((aligned(16))) bf16_bits Vs[2][BKV * STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQW;
  const int qrow = q0 + wave * 16 + (lane & 15);
  const int qrow_c = qrow < p.S ? qrow : p.S - 1;
  const int kvlen = p.kv_len ? p.kv_len[b] : p.S;
  int kv_end = kvlen < p.S ? kvlen : p.S;
  if (p.causal) kv_end = kv_end < (q0 + BQW) ? kv_end : (q0 + BQW);
  const int ntiles = (kv_end + BKV - 1) / BKV;

  const bf16_bits* Qb = p.Q + (int64_t)b * p.S * p.q_stride + (int64_t)h * p.hd;
  const bf16_bits* Kb = p.K + (int64_t)b * p.S * p.k_stride + (int64_t)h * p.hd;
  const bf16_bits* Vb = p.V + (int64_t)b * p.S * p.v_stride + (int64_t)h * p.hd;

  bf16x8_bits qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = gload8(Qb, p.q_stride, qrow_c, 32 * s + 8 * g, p.hd);

  f32x4 accO[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) accO[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = p.scale * LOG2E;

  // Register-staged prefetch: tile t+1 sits in registers while tile t is consumed; it is written to the other LDS
  // buffer after tile t's MFMAs and tile t+2's loads are issued right behind.  (Measured: the loop is bound by this
  // global -> register -> LDS latency chain -- with all math removed it keeps 60 % of its time -- so occupancy matters
  // more than anything else here: a second register set (2-tile distance) cost 64 VGPRs and ran 1.4x SLOWER.)
  TileStage<DP, NTH> kst, vst;
  if (ntiles > 0) {
    kst.load(Kb, p.k_stride, 0, p.S - 1, p.hd, tid);
    vst.load(Vb, p.v_stride, 0, p.S - 1, p.hd, tid);
    kst.store(Ks[0], tid);
    vst.store(Vs[0], tid);
    if (ntiles > 1) {
      kst.load(Kb, p.k_stride, BKV, p.S - 1, p.hd, tid);
      vst.load(Vb, p.v_stride, BKV, p.S - 1, p.hd, tid);
    }
  }
  __syncthreads();

  auto tile = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const bf16_bits* Kt = Ks[t & 1];
    const bf16_bits* Vt = Vs[t & 1];
    // S^T = K . Q^T : accS[nt][j] = S[q = lane&15][key = 16 nt + 4 g + j]
    f32x4 accS[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      accS[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        accS[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kt, nt * 16 + (lane & 15), 32 * s + 8 * g, STRIDE), qf[s],
                                                           accS[nt], 0, 0, 0);
    }
    const int kbase = t * BKV;
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float sv = accS[nt][j] * sl2;
        if constexpr (MASK) {
          const int key = kbase + nt * 16 + 4 * g + j;
          const bool ok = key < kvlen && (!p.causal || key <= qrow);
          sv = ok ? sv : -INFINITY;
        }
        accS[nt][j] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = __builtin_amdgcn_exp2f(accS[nt][j] - m_safe);
        accS[nt][j] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j) accO[d][j] *= alpha;
    // O^T += V^T . P^T, contraction slots (g, jj): jj<4 -> key 32 s2 + 4g + jj, jj>=4 -> key 32 s2 + 16 + 4g + jj-4
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8_bits pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pf[j] = f2bf_s(accS[2 * s2][j]);
        pf[4 + j] = f2bf_s(accS[2 * s2 + 1][j]);
      }
#pragma unroll
      for (int d = 0; d < DT; ++d)
        accO[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vt, 32 * s2, d * 16, STRIDE, lane), pf, accO[d], 0, 0, 0);
    }
    if (t + 1 < ntiles) {   // tile t+1 (in registers since the previous iteration) -> the other LDS buffer; fetch tile t+2
      kst.store(Ks[(t + 1) & 1], tid);
      vst.store(Vs[(t + 1) & 1], tid);
      if (t + 2 < ntiles) {
        kst.load(Kb, p.k_stride, (t + 2) * BKV, p.S - 1, p.hd, tid);
        vst.load(Vb, p.v_stride, (t + 2) * BKV, p.S - 1, p.hd, tid);
      }
    }
    __syncthreads();
  };
  // interior tiles (every key valid for every row of this workgroup) run a body with NO mask code at all; only the last
  // tile(s) -- key padding tail, causal diagonal -- take the masked body
  int n_free = kvlen / BKV;
  if (p.causal) n_free = n_free < (q0 / BKV) ? n_free : (q0 / BKV);   // tiles whose last key precedes the first query row
  if (n_free > ntiles) n_free = ntiles;
  for (int t = 0; t < n_free; ++t) tile(t, std::false_type{});
  for (int t = n_free; t < ntiles; ++t) tile(t, std::true_type{});
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (qrow < p.S) {
    bf16_bits* Ob = p.Oout + ((int64_t)b * p.S + qrow) * p.o_stride + (int64_t)h * p.hd;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int col = d * 16 + 4 * g;
      if (col < p.hd) {
        bf16x4_bits o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf_s(accO[d][j] * inv_l);
        *reinterpret_cast<bf16x4_bits*>(Ob + col) = o;
      }
    }
    if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.S + qrow] = m_run * LN2 + __logf(l_run);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward, 32 query rows per wave on mfma_f32_32x32x16_bf16 (round 2).  Same transposed-score idea as above at twice the rows per
// wave: every K / V fragment read from LDS now feeds 32 query rows, so the LDS array is busy half as long per FLOP (at 16 rows per
// wave it was as busy as the matrix pipe).  S^T = K . Q^T leaves accS[kb][r] = S[q = lane & 31][key = 32 kb + crow(r, hi)] with
// crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi, hi = lane >> 5: a lane owns 16 of a key block's 32 scores of ITS query row, so the row
// statistics need one cross-half exchange.  P feeds the second product straight from the accumulators: MFMA step t of key block kb
// takes registers 8t .. 8t+7, i.e. contraction slot (h, j) = key 32 kb + 16 t + 8 (j >> 2) + 4 h + (j & 3), and V^T is read
// transposed from the row-major tile in that same order (two ds_read_b64_tr_b16 per fragment).  K rows are padded to DP + 8
// elements (conflict-free ds_read_b128 for the 32-row operand), V rows to a stride of 64 mod 128 bytes (conflict-free transposed
// reads); both from the bank rules in MI355X_MICROARCH.md, checked by tools/lds_bank_sim.py.
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DP> struct Fwd32Cfg {
  static constexpr int KSTR = DP + 8;
  static constexpr int VSTR = (DP == 96) ? DP : DP + 32;
};

template <int DP, int STRIDE, int NTHREADS>
struct TileStageS {
  static constexpr int CH = DP / 8;
  static constexpr int TOTAL = BKV * CH;
  static_assert(TOTAL % NTHREADS == 0, "tile chunks must divide over the workgroup");
  static constexpr int PER_THREAD = TOTAL / NTHREADS;
  bf16x8_bits r[PER_THREAD];
  OVLA_DEV void load(const bf16_bits* base, int64_t stride, int row0, int row_last, int hd, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + NTHREADS * i;
      const int rr = id / CH, ch = id % CH;
      int gr = row0 + rr;
      gr = gr < row_last ? gr : row_last;
      r[i] = gload8_nb<DP>(base, stride, gr, ch * 8, hd);
    }
  }
  OVLA_DEV void store(bf16_bits* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int id = tid + NTHREADS * i;
      const int rr = id / CH, ch = id % CH;
      *reinterpret_cast<bf16x8_bits*>(tile + rr * STRIDE + ch * 8) = r[i];
    }
  }
};

// V^T fragment for one 32x32x16 step: lane (d = lane & 31, h) gets V[row0 + 4h + jj][col0 + d] (jj < 4) and V[row0 + 8 + 4h + jj][col0 + d]
OVLA_DEV bf16x8_bits tr_frag32(const bf16_bits* tile, int row0, int col0, int stride, int lane) {
  const int h = lane >> 5, g4 = (lane >> 4) & 1, i = lane & 15;
  const bf16_bits* a0 = tile + (row0 + 4 * h + (i >> 2)) * stride + col0 + 16 * g4 + 4 * (i & 3);
  const bf16x4_bits lo = lds_tr16(a0);
  const bf16x4_bits hi = lds_tr16(a0 + 8 * stride);
  return bf16x8_bits{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int DP>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(const AttnParams p) {
  constexpr int KS = DP / 16;   // 32x32x16 contraction steps over head_dim
  constexpr int DB = DP / 32;   // 32-wide output blocks
  constexpr int KSTR = Fwd32Cfg<DP>::KSTR, VSTR = Fwd32Cfg<DP>::VSTR;
  constexpr int BQW = 128;      // 4 waves x 32 query rows
  __shared__ __attribute__((aligned(16))) bf16_bits Ks[2][BKV * KSTR];
  __shared__ __attribute__((aligned(16))) bf16_bits Vs[2][BKV * VSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int nqb = (p.S + BQW - 1) / BQW;
  const int wi = p.xcd_map ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bh = wi / nqb, b = bh / p.H, h = bh - b * p.H, q0 = (wi - bh * nqb) * BQW;
  const int qrow = q0 + wave * 32 + (lane & 31);
  const int qrow_c = qrow < p.S ? qrow : p.S - 1;
  const int kvlen = p.kv_len ? p.kv_len[b] : p.S;
  int kv_end = kvlen < p.S ? kvlen : p.S;
  if (p.causal) kv_end = kv_end < (q0 + BQW) ? kv_end : (q0 + BQW);
  const int ntiles = (kv_end + BKV - 1) / BKV;

  const bf16_bits* Qb = p.Q + (int64_t)b * p.S * p.q_stride + (int64_t)h * p.hd;
  const bf16_bits* Kb = p.K + (int64_t)b * p.S * p.k_stride + (int64_t)h * p.hd;
  const bf16_bits* Vb = p.V + (int64_t)b * p.S * p.v_stride + (int64_t)h * p.hd;

  bf16x8_bits qf[KS];   // B operand of S^T = K . Q^T: B[k = 8 hi + j][col = q] = Q[q][16 s + 8 hi + j]
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = gload8_nb<DP>(Qb, p.q_stride, qrow_c, 16 * s + 8 * hi, p.hd);

  f32x16 accO[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) accO[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // m_run: running max of the RAW scores (the positive scale is applied inside the exp2 fma)
  const float sl2 = p.scale * LOG2E;

  TileStageS<DP, KSTR, 256> kst;
  TileStageS<DP, VSTR, 256> vst;
  if (ntiles > 0) {
    kst.load(Kb, p.k_stride, 0, p.S - 1, p.hd, tid);
    vst.load(Vb, p.v_stride, 0, p.S - 1, p.hd, tid);
    kst.store(Ks[0], tid);
    vst.store(Vs[0], tid);
  }
  __syncthreads();

  // Tile t + 1 is fetched into registers at the top of iteration t and written to the other LDS buffer at its bottom: the loads are in flight
  // under the whole tile's math and are waited for inside the iteration that issued them.  (Measured the same as the 16-row kernels' order,
  // fetch of tile t + 2 at the bottom of iteration t; this one needs no second prologue fetch.)
  auto tile = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const bf16_bits* Kt = Ks[t & 1];
    const bf16_bits* Vt = Vs[t & 1];
    // (rows beyond the context clamp to the last row: harmless re-reads, masked in the last tile)
    kst.load(Kb, p.k_stride, (t + 1) * BKV, p.S - 1, p.hd, tid);
    vst.load(Vb, p.v_stride, (t + 1) * BKV, p.S - 1, p.hd, tid);
    f32x16 accS[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accS[kb][r] = 0.f;
    {
      // K fragments are read HALF a key block (KS / 2 fragments) ahead of the MFMAs that consume them: two register sets, the reads of
      // the next half issued before the MFMAs of the current one (left to itself the compiler reads two fragments, waits, issues two MFMAs)
      constexpr int HS = KS / 2;
      const bf16_bits* krow = Kt + (lane & 31) * KSTR + 8 * hi;
      bf16x8_bits ka[HS], kb_[HS];
#pragma unroll
      for (int s = 0; s < HS; ++s) ka[s] = *reinterpret_cast<const bf16x8_bits*>(krow + 16 * s);
#pragma unroll
      for (int s = 0; s < HS; ++s) kb_[s] = *reinterpret_cast<const bf16x8_bits*>(krow + 16 * (HS + s));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < HS; ++s) accS[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s], qf[s], accS[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < HS; ++s) ka[s] = *reinterpret_cast<const bf16x8_bits*>(krow + 32 * KSTR + 16 * s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < HS; ++s) accS[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_[s], qf[HS + s], accS[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < HS; ++s) kb_[s] = *reinterpret_cast<const bf16x8_bits*>(krow + 32 * KSTR + 16 * (HS + s));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < HS; ++s) accS[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s], qf[s], accS[1], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < HS; ++s) accS[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_[s], qf[HS + s], accS[1], 0, 0, 0);
    }
    const int kbase = t * BKV;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (MASK) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = key < kvlen && (!p.causal || key <= qrow);
          accS[kb][r] = ok ? accS[kb][r] : -INFINITY;
        }
        mx = fmaxf(mx, accS[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * sl2);
    const float moff = -m_safe * sl2;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(accS[kb][r], sl2, moff));
        accS[kb][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        bf16x8_bits pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = f2bf_s(accS[kb][8 * tt + j]);
#pragma unroll
        for (int d = 0; d < DB; ++d)
          accO[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag32(Vt, kb * 32 + 16 * tt, d * 32, VSTR, lane), pf, accO[d], 0, 0, 0);
      }
    kst.store(Ks[(t + 1) & 1], tid);
    vst.store(Vs[(t + 1) & 1], tid);
    __syncthreads();
  };
  int n_free = kvlen / BKV;
  if (p.causal) n_free = n_free < (q0 / BKV) ? n_free : (q0 / BKV);
  if (n_free > ntiles) n_free = ntiles;
  for (int t = 0; t < n_free; ++t) tile(t, std::false_type{});
  for (int t = n_free; t < ntiles; ++t) tile(t, std::true_type{});
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (qrow < p.S) {
    bf16_bits* Ob = p.Oout + ((int64_t)b * p.S + qrow) * p.o_stride + (int64_t)h * p.hd;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int col = d * 32 + 8 * rq + 4 * hi;
        if (col < p.hd) {
          bf16x4_bits o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf_s(accO[d][4 * rq + j] * inv_l);
          *reinterpret_cast<bf16x4_bits*>(Ob + col) = o;
        }
      }
    if (hi == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.S + qrow] = m_run * sl2 * LN2 + __logf(l_run);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dQ: one workgroup per 64-query block, loops over K/V tiles.
// Inverse RoPE (HF rotate_half convention) on one row's gradient held as DT = hd/16 accumulator tiles: lane (row = lane & 15,
// g = lane >> 4) owns columns 16 d + 4 g + j; the rotation partner of column c is c +- hd/2 = tile d +- DT/2, same lane.  Values are
// first rounded to bf16 (what the separate pass would read back), then rotated with rope_kernel's arithmetic (inverse = 1).
template <int DT>
OVLA_DEV void inverse_rope_store(const f32x4 (&acc)[DT], float scale, bf16_bits* dst, int hd, int pos, int g, const bf16_bits* cosT, const bf16_bits* sinT) {
  constexpr int HT = DT / 2;
  const int half = hd >> 1;
#pragma unroll
  for (int d = 0; d < HT; ++d) {
    const int col = d * 16 + 4 * g;
    const bf16x4_bits cs = *reinterpret_cast<const bf16x4_bits*>(cosT + (int64_t)pos * half + col);
    const bf16x4_bits sn = *reinterpret_cast<const bf16x4_bits*>(sinT + (int64_t)pos * half + col);
    bf16x4_bits olo, ohi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bfround(acc[d][j] * scale), b = bfround(acc[d + HT][j] * scale);
      const float cc = bf2f((bf16_bits)cs[j]), s = -bf2f((bf16_bits)sn[j]);
      olo[j] = (short)f2bf(bfround(a * cc) + bfround(-b * s));
      ohi[j] = (short)f2bf(bfround(b * cc) + bfround(a * s));
    }
    *reinterpret_cast<bf16x4_bits*>(dst + col) = olo;
    *reinterpret_cast<bf16x4_bits*>(dst + half + col) = ohi;
  }
}

template <int DP, int NW, int OCC = 2, bool BATCH = (DP == 128)>
__global__ __launch_bounds__(64 * NW, OCC) void attn_bwd_dq_kernel(const AttnParams p) {
  constexpr int KS = DP / 32, DT = DP / 16, STRIDE = DP + 16;
  constexpr int NTH = 64 * NW;
  constexpr int BQW = 16 * NW;   // query rows per workgroup (NW = 8: 128 rows share every K / V tile)
  // K / V tiles double buffered in LDS (as in the forward): ONE barrier per key tile
  __shared__ __attribute__((aligned(16))) bf16_bits Ks[2][BKV * STRIDE];
  __shared__ __attribute__((aligned(16))) bf16_bits Vs[2][BKV * STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQW;
  if (p.xcd_map) {   // 1-D grid: the query blocks of one (batch, head) run on one XCD and share its L2 copy of K / V
    const int nqb = (p.S + BQW - 1) / BQW;
    const int wi = xcd_contiguous(blockIdx.x, gridDim.x), bh = wi / nqb;
    b = bh / p.H; h = bh - b * p.H; q0 = (wi - bh * nqb) * BQW;
  }
  const int qrow = q0 + wave * 16 + (lane & 15);
  const int qrow_c = qrow < p.S ? qrow : p.S - 1;
  const int kvlen = p.kv_len ? p.kv_len[b] : p.S;
  int kv_end = kvlen < p.S ? kvlen : p.S;
  if (p.causal) kv_end = kv_end < (q0 + BQW) ? kv_end : (q0 + BQW);
  const int ntiles = (kv_end + BKV - 1) / BKV;

  const bf16_bits* Qb = p.Q + (int64_t)b * p.S * p.q_stride + (int64_t)h * p.hd;
  const bf16_bits* Kb = p.K + (int64_t)b * p.S * p.k_stride + (int64_t)h * p.hd;
  const bf16_bits* Vb = p.V + (int64_t)b * p.S * p.v_stride + (int64_t)h * p.hd;
  const bf16_bits* dOb = p.dO + (int64_t)b * p.S * p.do_stride + (int64_t)h * p.hd;

  bf16x8_bits qf[KS], dof[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qf[s] = gload8(Qb, p.q_stride, qrow_c, 32 * s + 8 * g, p.hd);
    dof[s] = gload8(dOb, p.do_stride, qrow_c, 32 * s + 8 * g, p.hd);
  }
  const int64_t stat = ((int64_t)b * p.H + h) * p.S + qrow_c;
  const float Lq = p.lse_in[stat] * LOG2E;
  // delta = rowsum(dO * O), computed here from the dO fragments the lane already holds (its 8-column chunks of the row; the four
  // lane groups of a row cover every column) and written out for the dK / dV kernel, which runs after this one on the same stream
  float Dq = 0.f;
  {
    const bf16_bits* Ob = p.O + (int64_t)b * p.S * p.o_stride + (int64_t)h * p.hd;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8_bits of = gload8(Ob, p.o_stride, qrow_c, 32 * s + 8 * g, p.hd);
#pragma unroll
      for (int j = 0; j < 8; ++j) Dq += bf2f((bf16_bits)of[j]) * bf2f((bf16_bits)dof[s][j]);
    }
    Dq += __shfl_xor(Dq, 16, 64);
    Dq += __shfl_xor(Dq, 32, 64);
    if (g == 0 && qrow < p.S) p.delta[stat] = Dq;
  }
  const float sl2 = p.scale * LOG2E;

  f32x4 accQ[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) accQ[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileStage<DP, NTH> kst, vst;
  if (ntiles > 0) {
    kst.load(Kb, p.k_stride, 0, p.S - 1, p.hd, tid);
    vst.load(Vb, p.v_stride, 0, p.S - 1, p.hd, tid);
    kst.store(Ks[0], tid);
    vst.store(Vs[0], tid);
    if (ntiles > 1) {
      kst.load(Kb, p.k_stride, BKV, p.S - 1, p.hd, tid);
      vst.load(Vb, p.v_stride, BKV, p.S - 1, p.hd, tid);
    }
  }
  __syncthreads();
  auto tile = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const bf16_bits* Kt = Ks[t & 1];
    const bf16_bits* Vt = Vs[t & 1];
    const int kbase = t * BKV;
    // the two 32-key halves go through S / dP -> dS -> dQ one after the other (halves the live S / dP tiles)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f32x4 accS[2], accP[2];
      if constexpr (BATCH) {
        // all K / V row fragments of this half are read first, then the 4 KS MFMAs run back to back (left alone the compiler reads one or two
        // fragments, waits for them, issues one or two 16-cycle MFMAs: an LDS round trip every other MFMA)
        bf16x8_bits kfr[2][KS], vfr[2][KS];
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            kfr[nn][s] = row_frag(Kt, (2 * s2 + nn) * 16 + (lane & 15), 32 * s + 8 * g, STRIDE);
            vfr[nn][s] = row_frag(Vt, (2 * s2 + nn) * 16 + (lane & 15), 32 * s + 8 * g, STRIDE);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          accS[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
          accP[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) {
            accS[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[nn][s], qf[s], accS[nn], 0, 0, 0);
            accP[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[nn][s], dof[s], accP[nn], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const int nt = 2 * s2 + nn;
        accS[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        accP[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          accS[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kt, nt * 16 + (lane & 15), 32 * s + 8 * g, STRIDE), qf[s], accS[nn], 0, 0, 0);
          accP[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vt, nt * 16 + (lane & 15), 32 * s + 8 * g, STRIDE), dof[s], accP[nn], 0, 0, 0);
        }
      }
      }
      bf16x8_bits dsf;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pv = __builtin_amdgcn_exp2f(accS[nn][j] * sl2 - Lq);
          if constexpr (MASK) {
            const int key = kbase + (2 * s2 + nn) * 16 + 4 * g + j;
            const bool ok = key < kvlen && (!p.causal || key <= qrow);
            pv = ok ? pv : 0.f;
          }
          dsf[4 * nn + j] = f2bf_s(pv * (accP[nn][j] - Dq));   // dS (unscaled)
        }
      if constexpr (BATCH) {
        bf16x8_bits ktr[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) ktr[d] = tr_frag(Kt, 32 * s2, d * 16, STRIDE, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DT; ++d) accQ[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr[d], dsf, accQ[d], 0, 0, 0);
      } else {
#pragma unroll
      for (int d = 0; d < DT; ++d)
        accQ[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Kt, 32 * s2, d * 16, STRIDE, lane), dsf, accQ[d], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) {   // tile t+1 (in registers since the previous iteration) -> the other LDS buffer; fetch tile t+2
      kst.store(Ks[(t + 1) & 1], tid);
      vst.store(Vs[(t + 1) & 1], tid);
      if (t + 2 < ntiles) {
        kst.load(Kb, p.k_stride, (t + 2) * BKV, p.S - 1, p.hd, tid);
        vst.load(Vb, p.v_stride, (t + 2) * BKV, p.S - 1, p.hd, tid);
      }
    }
    __syncthreads();
  };
  int n_free = kvlen / BKV;     // tiles whose every key is valid for every row of this workgroup run the body without mask code
  if (p.causal) n_free = n_free < (q0 / BKV) ? n_free : (q0 / BKV);
  if (n_free > ntiles) n_free = ntiles;
  for (int t = 0; t < n_free; ++t) tile(t, std::false_type{});
  for (int t = n_free; t < ntiles; ++t) tile(t, std::true_type{});
  if (qrow < p.S) {
    bf16_bits* dQb = p.dQ + ((int64_t)b * p.S + qrow) * p.dq_stride + (int64_t)h * p.hd;
    if (p.rope_cos) {
      inverse_rope_store<DT>(accQ, p.scale, dQb, p.hd, qrow, g, p.rope_cos, p.rope_sin);
    } else {
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const int col = d * 16 + 4 * g;
        if (col < p.hd) {
          bf16x4_bits o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf_s(accQ[d][j] * p.scale);
          *reinterpret_cast<bf16x4_bits*>(dQb + col) = o;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// dK / dV: one workgroup per 64-key block (wave = 16 keys, key on the lane), loops over Q / dO tiles.
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_kernel(const AttnParams p) {
  constexpr int KS = DP / 32, DT = DP / 16, STRIDE = DP + 16;
  constexpr int NTH = 64 * NW;
  constexpr int BKW = 16 * NW;   // keys per workgroup (NW = 8: 128 keys share every Q / dO tile)
  // Q / dO tiles (+ their row statistics) are double buffered: tile t+1 is written while tile t is being read -> ONE barrier per tile
  __shared__ __attribute__((aligned(16))) bf16_bits Qs[2][BQ * STRIDE];
  __shared__ __attribute__((aligned(16))) bf16_bits dOs[2][BQ * STRIDE];
  __shared__ float Ls[2][BQ], Ds[2][BQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * BKW;
  if (p.xcd_map) {   // 1-D grid: the key blocks of one (batch, head) run on one XCD and share its L2 copy of Q / dO
    const int nkb = (p.S + BKW - 1) / BKW;
    const int wi = xcd_contiguous(blockIdx.x, gridDim.x), bh = wi / nkb;
    b = bh / p.H; h = bh - b * p.H; k0 = (wi - bh * nkb) * BKW;
  }
  const int krow = k0 + wave * 16 + (lane & 15);
  const int krow_c = krow < p.S ? krow : p.S - 1;
  const int kvlen = p.kv_len ? p.kv_len[b] : p.S;

  const bf16_bits* Qb = p.Q + (int64_t)b * p.S * p.q_stride + (int64_t)h * p.hd;
  const bf16_bits* Kb = p.K + (int64_t)b * p.S * p.k_stride + (int64_t)h * p.hd;
  const bf16_bits* Vb = p.V + (int64_t)b * p.S * p.v_stride + (int64_t)h * p.hd;
  const bf16_bits* dOb = p.dO + (int64_t)b * p.S * p.do_stride + (int64_t)h * p.hd;
  const float* lse_b = p.lse_in + ((int64_t)b * p.H + h) * p.S;
  const float* del_b = p.delta + ((int64_t)b * p.H + h) * p.S;

  bf16x8_bits kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = gload8(Kb, p.k_stride, krow_c, 32 * s + 8 * g, p.hd);
    vf[s] = gload8(Vb, p.v_stride, krow_c, 32 * s + 8 * g, p.hd);
  }
  const float sl2 = p.scale * LOG2E;
  f32x4 accK[DT], accV[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    accK[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    accV[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool key_ok = krow < kvlen && krow < p.S;
  const int nq = (p.S + BQ - 1) / BQ;
  const int qt0 = p.causal ? (k0 / BQ) : 0;

  TileStage<DP, NTH> qst, dst;
  float lreg = 0.f, dreg = 0.f;
  auto load_tile = [&](int qt) {
    qst.load(Qb, p.q_stride, qt * BQ, p.S - 1, p.hd, tid);
    dst.load(dOb, p.do_stride, qt * BQ, p.S - 1, p.hd, tid);
    if (tid < BQ) {
      int r = qt * BQ + tid;
      r = r < p.S ? r : p.S - 1;
      lreg = lse_b[r] * LOG2E;
      dreg = del_b[r];
    }
  };
  auto store_tile = [&](int buf) {
    qst.store(Qs[buf], tid);
    dst.store(dOs[buf], tid);
    if (tid < BQ) {
      Ls[buf][tid] = lreg;
      Ds[buf][tid] = dreg;
    }
  };
  if (qt0 < nq) {
    load_tile(qt0);
    store_tile(0);
    if (qt0 + 1 < nq) load_tile(qt0 + 1);
  }
  __syncthreads();

  auto tile = [&](int qt, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int buf = (qt - qt0) & 1;
    const bf16_bits* Qt = Qs[buf];
    const bf16_bits* dOt = dOs[buf];
    const float* Lt = Ls[buf];
    const float* Dt = Ds[buf];
    const int qbase = qt * BQ;
    // the two halves of the 64-row tile go through S / dP -> P / dS -> dV / dK one after the other (halves the live S / dP tiles)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // S[q][key], dP[q][key] with key = lane&15, q = 16 mt + 4 g + j
      f32x4 accS[2], accP[2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const int mt = 2 * s2 + mm;
        accS[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
        accP[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          accS[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qt, mt * 16 + (lane & 15), 32 * s + 8 * g, STRIDE), kf[s], accS[mm], 0, 0, 0);
          accP[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(dOt, mt * 16 + (lane & 15), 32 * s + 8 * g, STRIDE), vf[s], accP[mm], 0, 0, 0);
        }
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ql = (2 * s2 + mm) * 16 + 4 * g + j;
          float pv = __builtin_amdgcn_exp2f(accS[mm][j] * sl2 - Lt[ql]);
          if constexpr (MASK) {
            const int q = qbase + ql;
            const bool ok = key_ok && q < p.S && (!p.causal || krow <= q);
            pv = ok ? pv : 0.f;
          }
          accS[mm][j] = pv;                              // P
          accP[mm][j] = pv * (accP[mm][j] - Dt[ql]);     // dS (unscaled)
        }
      bf16x8_bits pf, dsf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pf[j] = f2bf_s(accS[0][j]);
        pf[4 + j] = f2bf_s(accS[1][j]);
        dsf[j] = f2bf_s(accP[0][j]);
        dsf[4 + j] = f2bf_s(accP[1][j]);
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        accV[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dOt, 32 * s2, d * 16, STRIDE, lane), pf, accV[d], 0, 0, 0);
        accK[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qt, 32 * s2, d * 16, STRIDE, lane), dsf, accK[d], 0, 0, 0);
      }
    }
    if (qt + 1 < nq) {   // tile qt+1 (in registers since the previous iteration) -> the other buffer; fetch tile qt+2
      store_tile(buf ^ 1);
      if (qt + 2 < nq) load_tile(qt + 2);
    }
    __syncthreads();
  };
  // Interior query tiles need no mask code when every key of this workgroup is valid (no key padding inside the block, no causal
  // diagonal in the tile) and the tile has 64 real rows; everything else takes the masked body.
  const bool keys_full = k0 + BKW <= kvlen && k0 + BKW <= p.S;
  for (int qt = qt0; qt < nq; ++qt) {
    const bool free_tile = keys_full && (qt + 1) * BQ <= p.S && (!p.causal || qt * BQ >= k0 + BKW - 1);
    if (free_tile) tile(qt, std::false_type{});
    else tile(qt, std::true_type{});
  }
  if (krow < p.S) {
    bf16_bits* dKb = p.dK + ((int64_t)b * p.S + krow) * p.dk_stride + (int64_t)h * p.hd;
    bf16_bits* dVb = p.dV + ((int64_t)b * p.S + krow) * p.dv_stride + (int64_t)h * p.hd;
    if (p.rope_cos) inverse_rope_store<DT>(accK, p.scale, dKb, p.hd, krow, g, p.rope_cos, p.rope_sin);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int col = d * 16 + 4 * g;
      if (col < p.hd) {
        bf16x4_bits ok_, ov_;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ok_[j] = f2bf_s(accK[d][j] * p.scale);
          ov_[j] = f2bf_s(accV[d][j]);
        }
        if (!p.rope_cos) *reinterpret_cast<bf16x4_bits*>(dKb + col) = ok_;
        *reinterpret_cast<bf16x4_bits*>(dVb + col) = ov_;
      }
    }
  }
}

int check_common(int B, int H, int S, int hd, const char* who) {
  if (!(B > 0 && H > 0 && S > 0)) {
    ovla_set_error("%s: empty problem B=%d H=%d S=%d", who, B, H, S);
    return OVLA_EINVAL;
  }
  if (!(hd == 64 || hd == 72 || hd == 128)) {
    ovla_set_error("%s: head_dim %d not supported (64, 72, 128)", who, hd);
    return OVLA_EINVAL;
  }
  return OVLA_OK;
}

inline bool stride_ok(int64_t s) { return (s % 8) == 0; }

}  // namespace

extern "C" int ovla_attn_fwd(const ovla_attn_fwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->Q && a->K && a->V && a->O, "ovla_attn_fwd: null pointer");
  if (int rc = check_common(a->B, a->H, a->S, a->head_dim, "ovla_attn_fwd")) return rc;
  OVLA_REQUIRE(stride_ok(a->q_stride) && stride_ok(a->k_stride) && stride_ok(a->v_stride) && (a->o_stride % 4) == 0,
               "ovla_attn_fwd: row strides must be multiples of 8 elements");
  OVLA_REQUIRE(aligned16(a->Q) && aligned16(a->K) && aligned16(a->V) && (((uintptr_t)a->O) & 7) == 0, "ovla_attn_fwd: alignment");
  AttnParams p = {};
  p.Q = (const bf16_bits*)a->Q; p.K = (const bf16_bits*)a->K; p.V = (const bf16_bits*)a->V; p.Oout = (bf16_bits*)a->O;
  p.q_stride = a->q_stride; p.k_stride = a->k_stride; p.v_stride = a->v_stride; p.o_stride = a->o_stride;
  p.lse = a->lse; p.kv_len = a->kv_len; p.B = a->B; p.H = a->H; p.S = a->S; p.hd = a->head_dim; p.causal = a->causal; p.scale = a->scale;
  static const bool fwd32 = []() { const char* e = getenv("OVLA_ATTN_FWD32"); return !(e && e[0] == '0'); }();   // A/B switch
  static const bool xcd_map = []() { const char* e = getenv("OVLA_ATTN_XCD"); return !(e && e[0] == '0'); }();   // A/B switch
  p.xcd_map = xcd_map ? 1 : 0;
  if (fwd32 && a->S > 64) {   // 4 waves x 32 query rows on the 32x32x16 MFMA
    const dim3 grid((unsigned)(cdiv(a->S, 128) * a->H * a->B));
    switch (a->head_dim) {
      case 64: hipLaunchKernelGGL((attn_fwd32_kernel<64>), grid, dim3(256), 0, stream, p); break;
      case 72: hipLaunchKernelGGL((attn_fwd32_kernel<96>), grid, dim3(256), 0, stream, p); break;
      default: hipLaunchKernelGGL((attn_fwd32_kernel<128>), grid, dim3(256), 0, stream, p); break;
    }
  } else if (a->S > 64) {   // 8 waves x 16 query rows = 128 rows per workgroup share every K/V tile
    const dim3 grid(cdiv(a->S, 128), a->H, a->B);
    switch (a->head_dim) {
      case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, 8>), grid, dim3(512), 0, stream, p); break;
      case 72: hipLaunchKernelGGL((attn_fwd_kernel<96, 8>), grid, dim3(512), 0, stream, p); break;
      default: hipLaunchKernelGGL((attn_fwd_kernel<128, 8>), grid, dim3(512), 0, stream, p); break;
    }
  } else {
    const dim3 grid(cdiv(a->S, BQ), a->H, a->B);
    switch (a->head_dim) {
      case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, 4>), grid, dim3(256), 0, stream, p); break;
      case 72: hipLaunchKernelGGL((attn_fwd_kernel<96, 4>), grid, dim3(256), 0, stream, p); break;
      default: hipLaunchKernelGGL((attn_fwd_kernel<128, 4>), grid, dim3(256), 0, stream, p); break;
    }
  }
  OVLA_CHECK_LAUNCH("ovla_attn_fwd");
  return OVLA_OK;
}

extern "C" int ovla_attn_bwd(const ovla_attn_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(a && a->Q && a->K && a->V && a->O && a->dO && a->lse && a->delta && a->dQ && a->dK && a->dV, "ovla_attn_bwd: null pointer");
  if (int rc = check_common(a->B, a->H, a->S, a->head_dim, "ovla_attn_bwd")) return rc;
  OVLA_REQUIRE(stride_ok(a->q_stride) && stride_ok(a->k_stride) && stride_ok(a->v_stride) && stride_ok(a->o_stride) && stride_ok(a->do_stride) &&
                   (a->dq_stride % 4) == 0 && (a->dk_stride % 4) == 0 && (a->dv_stride % 4) == 0,
               "ovla_attn_bwd: row strides must be multiples of 8 elements");
  OVLA_REQUIRE(aligned16(a->Q) && aligned16(a->K) && aligned16(a->V) && aligned16(a->O) && aligned16(a->dO), "ovla_attn_bwd: alignment");
  if (a->rope_cos || a->rope_sin)
    OVLA_REQUIRE(a->rope_cos && a->rope_sin && (a->head_dim == 128 || a->head_dim == 64) && (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin) & 7) == 0,
                 "ovla_attn_bwd: fused inverse RoPE needs both tables (8-byte aligned) and head_dim 64 or 128");
  AttnParams p = {};
  p.Q = (const bf16_bits*)a->Q; p.K = (const bf16_bits*)a->K; p.V = (const bf16_bits*)a->V; p.O = (const bf16_bits*)a->O;
  p.dO = (const bf16_bits*)a->dO; p.dQ = (bf16_bits*)a->dQ; p.dK = (bf16_bits*)a->dK; p.dV = (bf16_bits*)a->dV;
  p.q_stride = a->q_stride; p.k_stride = a->k_stride; p.v_stride = a->v_stride; p.o_stride = a->o_stride; p.do_stride = a->do_stride;
  p.dq_stride = a->dq_stride; p.dk_stride = a->dk_stride; p.dv_stride = a->dv_stride;
  p.lse_in = a->lse; p.delta = a->delta; p.kv_len = a->kv_len;
  p.B = a->B; p.H = a->H; p.S = a->S; p.hd = a->head_dim; p.causal = a->causal; p.scale = a->scale;
  p.rope_cos = (const bf16_bits*)a->rope_cos; p.rope_sin = (const bf16_bits*)a->rope_sin;
  static const bool xcd_map = []() { const char* e = getenv("OVLA_ATTN_XCD"); return !(e && e[0] == '0'); }();   // A/B switch
  p.xcd_map = xcd_map ? 1 : 0;
  const dim3 grid = xcd_map ? dim3((unsigned)(cdiv(a->S, BQ) * a->H * a->B)) : dim3(cdiv(a->S, BQ), a->H, a->B);
  const dim3 grid_kv4 = grid;
  const dim3 grid_kv8 = xcd_map ? dim3((unsigned)(cdiv(a->S, 128) * a->H * a->B)) : dim3(cdiv(a->S, 128), a->H, a->B);
  switch (a->head_dim) {
    case 64:
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 4>), grid, dim3(256), 0, stream, p);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, 4>), grid_kv4, dim3(256), 0, stream, p);
      break;
    case 72:
      if (a->S % 128 == 0 || a->S > 512) {   // SigLIP: 256 tokens = two exact 128-row blocks
        hipLaunchKernelGGL((attn_bwd_dq_kernel<96, 8>), grid_kv8, dim3(512), 0, stream, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<96, 8>), grid_kv8, dim3(512), 0, stream, p);
      } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<96, 4>), grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<96, 4>), grid_kv4, dim3(256), 0, stream, p);
      }
      break;
    default:   // the Llama shape: 8 waves = 128 keys share every Q / dO tile
      {
        static const bool dq_batch = []() { const char* e = getenv("OVLA_ATTN_DQ_BATCH"); return !(e && e[0] == '0'); }();   // A/B switch
        if (dq_batch) hipLaunchKernelGGL((attn_bwd_dq_kernel<128, 8>), grid_kv8, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<128, 8, 2, false>), grid_kv8, dim3(512), 0, stream, p);
      }
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 8>), grid_kv8, dim3(512), 0, stream, p);
      break;
  }
  OVLA_CHECK_LAUNCH("ovla_attn_bwd");
  return OVLA_OK;
}
