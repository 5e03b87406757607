"""oracle/jpeg_oracle.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden generators, never by the product path).

CPU restatement of the JPEG encode -> decode ROUND TRIP inside the reference's `resize_image_for_policy`
(experiments/robot/openvla_utils.py:532-533: `tf.image.encode_jpeg(img)` then `tf.io.decode_image(...)`), i.e. of what libjpeg-turbo
does to the pixels for TensorFlow's default arguments:

  encode_jpeg defaults   quality 95, chroma_downsampling=True (4:2:0), baseline, no optimisation  (jpeg_set_defaults + jpeg_set_quality(95, TRUE),
                         JDCT_DEFAULT = the accurate integer DCT "islow")
  decode_image defaults  the DecodeImage op's default `UncompressFlags`: dct_method JDCT_DEFAULT (islow), fancy_upscaling = true

Entropy coding (Huffman) is lossless, so the round trip is:  RGB -> YCbCr (jccolor.c)  ->  2x2 chroma box filter with alternating bias,
edges padded by replication (jcsample.c / jcprepct.c)  ->  8x8 forward DCT "islow" on samples - 128 (jfdctint.c)  ->  quantise with the Annex-K
tables scaled for quality 95 (jcparam.c, jcdctmgr.c)  ->  dequantise  ->  inverse DCT "islow" + range limit (jidctint.c)  ->  "fancy" triangle
upsampling of the chroma planes (jdsample.c h2v2_fancy_upsample)  ->  YCbCr -> RGB (jdcolor.c).  Everything is integer arithmetic: bit-exact.

The library's C sources are not under /root/reference (libjpeg-turbo is a TensorFlow dependency): this file restates their published
algorithms.  PINNING: tests/golden/g12_jpeg_roundtrip.npz holds round trips produced by libjpeg-turbo itself (through Pillow, which links it,
in the build container: tests/golden/make_golden_jpeg.py); tests/test_oracle_pins.py requires this restatement to reproduce them bit for bit.
Against TensorFlow itself (absent): PARITY UNPINNED.  In particular the DECODE leg's IDCT is an open point: this file models it as islow
(JDCT_DEFAULT), but TensorFlow's DecodeImage / DecodeJpeg ops may select the fast integer IDCT (JDCT_IFAST, jidctfst.c) -- a reviewer's
recollection of decode_image_op.cc says they do; neither reading can be checked here (no TensorFlow, turbojpeg, simplejpeg, cv2 or imageio in
the image, and Pillow exposes islow only).  If TF decodes with ifast, some pixels of the round trip differ from this model by 1-2 LSB before the
lanczos resize.  What G12 pins is therefore the libjpeg-turbo islow codec (every stage TF and this model share: colour conversion, chroma
subsampling, FDCT, quantisation tables, fancy upsampling), NOT bit-identity with the reference's TensorFlow round trip."""
from __future__ import annotations

import numpy as np

# jcparam.c: Annex K tables, natural (row-major) order
STD_LUMA_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                       18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99],
                      dtype=np.int64)
STD_CHROMA_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                         99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99],
                        dtype=np.int64)


def quant_tables(quality: int = 95):
    """jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE) -> (luma [64], chroma [64]) int64, natural order."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for base in (STD_LUMA_Q, STD_CHROMA_Q):
        t = (base * scale + 50) // 100
        out.append(np.clip(t, 1, 255))
    return out[0], out[1]


def _fix(x: float) -> int:
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb: np.ndarray):
    """jccolor.c rgb_ycc_convert (SCALEBITS 16): uint8 [H, W, 3] -> three int64 planes in 0..255."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (_fix(0.29900) * r + _fix(0.58700) * g + _fix(0.11400) * b + half) >> 16
    cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.50000) * b + off + half - 1) >> 16
    cr = (_fix(0.50000) * r - _fix(0.41869) * g - _fix(0.08131) * b + off + half - 1) >> 16
    return y, cb, cr


def _pad_luma(p: np.ndarray, H16: int, W16: int) -> np.ndarray:
    """Replicates the last column / row up to the MCU grid (expand_right_edge / expand_bottom_edge)."""
    H, W = p.shape
    ri = np.minimum(np.arange(H16), H - 1)
    ci = np.minimum(np.arange(W16), W - 1)
    return p[ri][:, ci]


def downsample_h2v2(p: np.ndarray, H16: int, W16: int) -> np.ndarray:
    """jcsample.c h2v2_downsample on the edge-expanded plane: (a + b + c + d + bias) >> 2 with bias 1, 2, 1, 2, ... along each output row.
    Padding order as in the library: the INPUT is widened by replicating its last column and made even-height by replicating its last row;
    the remaining rows of the last iMCU row replicate the last DOWNSAMPLED row."""
    H, W = p.shape
    ci = np.minimum(np.arange(W16), W - 1)
    He = H + (H & 1)
    ri = np.minimum(np.arange(He), H - 1)
    q = p[ri][:, ci]
    s = q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2]
    bias = np.where(np.arange(W16 // 2) % 2 == 0, 1, 2)[None, :]
    d = (s + bias) >> 2                                      # [He / 2, W16 / 2]
    rr = np.minimum(np.arange(H16 // 2), d.shape[0] - 1)
    return d[rr]


F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865, F_0_899976223, F_1_175875602 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501321110, F_1_847759065, F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 12299, 15137, 16069, 16819, 20995, 25172
CONST_BITS, PASS1_BITS = 13, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, first_pass: bool):
    """One pass of jfdctint.c over the LAST axis of d [..., 8] (int64)."""
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    tmp0, tmp7, tmp1, tmp6, tmp2, tmp5, tmp3, tmp4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    sh = CONST_BITS - PASS1_BITS if first_pass else CONST_BITS + PASS1_BITS
    if first_pass:
        o0, o4 = (tmp10 + tmp11) << PASS1_BITS, (tmp10 - tmp11) << PASS1_BITS
    else:
        o0, o4 = _descale(tmp10 + tmp11, PASS1_BITS), _descale(tmp10 - tmp11, PASS1_BITS)
    z1 = (tmp12 + tmp13) * F_0_541196100
    o2 = _descale(z1 + tmp13 * F_0_765366865, sh)
    o6 = _descale(z1 + tmp12 * (-F_1_847759065), sh)
    z1, z2, z3, z4 = tmp4 + tmp7, tmp5 + tmp6, tmp4 + tmp6, tmp5 + tmp7
    z5 = (z3 + z4) * F_1_175875602
    t4, t5, t6, t7 = tmp4 * F_0_298631336, tmp5 * F_2_053119869, tmp6 * F_3_072711026, tmp7 * F_1_501321110
    z1, z2, z3, z4 = z1 * (-F_0_899976223), z2 * (-F_2_562915447), z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    o7, o5, o3, o1 = _descale(t4 + z1 + z3, sh), _descale(t5 + z2 + z4, sh), _descale(t6 + z2 + z3, sh), _descale(t7 + z1 + z4, sh)
    return np.stack([o0, o1, o2, o3, o4, o5, o6, o7], axis=-1)


def fdct_islow(blocks: np.ndarray) -> np.ndarray:
    """jfdctint.c jpeg_fdct_islow on [..., 8, 8] int64 blocks of (sample - 128): rows first, then columns; output scaled up by 8."""
    r = _fdct_1d(blocks, True)
    return np.swapaxes(_fdct_1d(np.swapaxes(r, -1, -2), False), -1, -2)


def quantize(coef: np.ndarray, qtbl: np.ndarray) -> np.ndarray:
    """jcdctmgr.c: divisor = qval << 3; round-half-away-from-zero of coef / divisor (its reciprocal form is exact for these ranges)."""
    div = (qtbl.reshape(8, 8) << 3)
    a = np.abs(coef)
    q = (a + (div >> 1)) // div
    return np.where(coef < 0, -q, q)


def _idct_1d(x, first_pass: bool):
    """One pass of jidctint.c over the LAST axis of x [..., 8]."""
    i0, i1, i2, i3, i4, i5, i6, i7 = (x[..., i] for i in range(8))
    z2, z3 = i2, i6
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    tmp0, tmp1 = (i0 + i4) << CONST_BITS, (i0 - i4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0, t1, t2, t3 = t0 * F_0_298631336, t1 * F_2_053119869, t2 * F_3_072711026, t3 * F_1_501321110
    z1, z2, z3, z4 = z1 * (-F_0_899976223), z2 * (-F_2_562915447), z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    sh = CONST_BITS - PASS1_BITS if first_pass else CONST_BITS + PASS1_BITS + 3
    outs = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([_descale(o, sh) for o in outs], axis=-1)


def range_limit_idct(x: np.ndarray) -> np.ndarray:
    """sample_range_limit + CENTERJSAMPLE indexed with (x & 1023): x + 128 clamped to 0..255 for x in [-512, 511]."""
    m = x & 1023
    return np.where(m < 128, m + 128, np.where(m < 512, 255, np.where(m < 896, 0, m - 896)))


def idct_islow(coef: np.ndarray, qtbl: np.ndarray) -> np.ndarray:
    """jidctint.c jpeg_idct_islow: dequantise, columns first, then rows, descale by 2^18, range-limit -> samples 0..255 [..., 8, 8]."""
    deq = coef * qtbl.reshape(8, 8)
    ws = np.swapaxes(_idct_1d(np.swapaxes(deq, -1, -2), True), -1, -2)
    return range_limit_idct(_idct_1d(ws, False))


def _to_blocks(p: np.ndarray) -> np.ndarray:
    H, W = p.shape
    return p.reshape(H // 8, 8, W // 8, 8).swapaxes(1, 2)


def _from_blocks(b: np.ndarray) -> np.ndarray:
    nby, nbx = b.shape[:2]
    return b.swapaxes(1, 2).reshape(nby * 8, nbx * 8)


def codec_plane(p: np.ndarray, qtbl: np.ndarray) -> np.ndarray:
    """One padded component plane (0..255, dimensions multiples of 8) through FDCT -> quantise -> dequantise -> IDCT."""
    blocks = _to_blocks(p.astype(np.int64)) - 128
    return _from_blocks(idct_islow(quantize(fdct_islow(blocks), qtbl), qtbl))


def fancy_upsample_h2v2(c: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample on the REAL downsampled plane c [ceil(H/2), ceil(W/2)]: vertical 3:1 blend with the nearer
    neighbouring row (replicated at the top / bottom), then horizontal 3:1 blend with rounding offsets 8 / 7 and the first / last column special
    cases; output cropped to [out_h, out_w]."""
    h, w = c.shape
    c = c.astype(np.int64)
    up, dn = c[np.maximum(np.arange(h) - 1, 0)], c[np.minimum(np.arange(h) + 1, h - 1)]
    rows = np.empty((2 * h, w), dtype=np.int64)
    rows[0::2], rows[1::2] = 3 * c + up, 3 * c + dn          # "colsum" rows: output row 2r uses the row above, 2r + 1 the row below
    last = rows[:, np.maximum(np.arange(w) - 1, 0)]
    nxt = rows[:, np.minimum(np.arange(w) + 1, w - 1)]
    out = np.empty((2 * h, 2 * w), dtype=np.int64)
    out[:, 0::2] = (rows * 3 + last + 8) >> 4
    out[:, 1::2] = (rows * 3 + nxt + 7) >> 4
    out[:, 0] = (rows[:, 0] * 4 + 8) >> 4                      # first column: no left neighbour
    out[:, 2 * w - 1] = (rows[:, w - 1] * 4 + 7) >> 4          # last column: no right neighbour
    return out[:out_h, :out_w]


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert (table form, SCALEBITS 16) -> uint8 [H, W, 3]."""
    half = 1 << 15
    xb, xr = cb - 128, cr - 128
    r = y + ((_fix(1.40200) * xr + half) >> 16)
    g = y + ((-_fix(0.34414) * xb + half - _fix(0.71414) * xr) >> 16)
    b = y + ((_fix(1.77200) * xb + half) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def jpeg_roundtrip(rgb: np.ndarray, quality: int = 95) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [H, W, 3] after a baseline 4:2:0 JPEG encode + decode at `quality` (libjpeg-turbo, accurate integer DCT,
    fancy upsampling): what `tf.image.encode_jpeg` + `tf.io.decode_image` do to a frame with their default arguments."""
    assert rgb.dtype == np.uint8 and rgb.ndim == 3 and rgb.shape[2] == 3
    H, W = rgb.shape[:2]
    H16, W16 = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    ql, qc = quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb)
    y2 = codec_plane(_pad_luma(y, H16, W16), ql)[:H, :W]
    hc, wc = (H + 1) // 2, (W + 1) // 2
    cb2 = codec_plane(downsample_h2v2(cb, H16, W16), qc)[:hc, :wc]
    cr2 = codec_plane(downsample_h2v2(cr, H16, W16), qc)[:hc, :wc]
    return ycc_to_rgb(y2, fancy_upsample_h2v2(cb2, H, W), fancy_upsample_h2v2(cr2, H, W))
