// selftest.hip -- dumps the raw lane layouts the kernels rely on (MFMA operand/accumulator maps, the transposing LDS
// read, LDS-DMA placement) so tests/ can check the assumptions on real gfx950 hardware with exact-integer data.
#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;

// out layout: [section][lane][16] floats
//  section 0: mfma_f32_16x16x32_bf16 with A[r][k] = (k == r), B[k][c] = 16 k + c (k < 16)  -> D[r][c] = 16 r + c; regs 0..3
//  section 1: ds_read_b64_tr_b16 of tile[row][col] = 16 row + col, group g reads rows 4g..4g+3; regs 0..3
//  section 2: mfma_f32_32x32x16_bf16 with A[r][k] = (k == (r & 15)), B[k][c] = 32 k + c -> D[r][c] = 32 (r&15) + c; regs 0..15
//  section 3: global_load_lds_dwordx4: LDS word w of the 1 KiB slab after each lane DMA'd its 16 bytes src[lane*8..]; regs 0..7 = lds[lane*8 + j]
__global__ __launch_bounds__(64) void selftest_kernel(float* out, const bf16_bits* src) {
  __shared__ __attribute__((aligned(16))) bf16_bits lds[2048];
  const int lane = threadIdx.x;
  // ---- section 0
  {
    bf16x8_bits a, b;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      a[j] = (short)f2bf(k == r ? 1.f : 0.f);
      b[j] = (short)f2bf(k < 16 ? (float)(16 * k + r) : 0.f);  // B[k][c = lane&15]
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[(0 * 64 + lane) * 16 + j] = c[j];
  }
  // ---- section 1
  {
    for (int i = lane; i < 16 * 16; i += 64) lds[i] = f2bf((float)i);  // tile[row][col] = 16 row + col (exact in bf16 up to 256)
    __syncthreads();
    const int g = lane >> 4, i = lane & 15;
    const bf16_bits* addr = lds + (4 * g + (i >> 2)) * 16 + 4 * (i & 3);
    const bf16x4_bits v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_bits*)addr);
    for (int j = 0; j < 4; ++j) out[(1 * 64 + lane) * 16 + j] = bf2f((bf16_bits)v[j]);
    __syncthreads();
  }
  // ---- section 2
  {
    bf16x8_bits a, b;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * hh + j;
      a[j] = (short)f2bf(k == (r & 15) ? 1.f : 0.f);
      b[j] = (short)f2bf((float)(k * 8 + (r & 7)));  // B[k][c]: small exact values 8k + (c & 7)
    }
    f32x16 c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 16; ++j) out[(2 * 64 + lane) * 16 + j] = c[j];
  }
  // ---- section 3
  {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 8; ++j) out[(3 * 64 + lane) * 16 + j] = bf2f(lds[lane * 8 + j]);
  }
}
}  // namespace

// out: fp32 [4*64*16] device; src: bf16 [512] device with src[i] = i (exact)
extern "C" int ovla_selftest_layouts(float* out, const void* src, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OVLA_REQUIRE(out && src, "ovla_selftest_layouts: null pointer");
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, out, (const bf16_bits*)src);
  OVLA_CHECK_LAUNCH("ovla_selftest_layouts");
  return OVLA_OK;
}
