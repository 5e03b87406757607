"""Op-level parity of every HIP kernel (called through the C-ABI) against a plain PyTorch fp32 reference of the same op.

Tolerances: outputs are bf16 (8 significant bits); a correctly rounded result differs from the fp32 reference by at most
2^-8 relative, and accumulation-order effects add about one more ulp.  `close()` therefore checks
max|out - ref| <= tol * max|ref| with tol stated per test (default 1.5e-2) plus a much tighter mean-error bound
that catches layout bugs hidden under a loose max bound.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def close(out, ref, tol=1.5e-2, mean_tol=2.5e-3, what=""):
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    scale = ref.abs().max().item() + 1e-12
    err = (out - ref).abs()
    mx, mean = err.max().item() / scale, err.mean().item() / scale
    assert mx <= tol and mean <= mean_tol, f"{what}: max rel err {mx:.3e} (tol {tol}), mean rel err {mean:.3e} (tol {mean_tol})"
    return mx


def rnd(*shape, dev, scale=1.0, dtype=BF):
    return (torch.randn(*shape, device=dev, dtype=torch.float32) * scale).to(dtype)


# ----------------------------------------------------------------------------------------------------------------------
def test_hardware_layout_assumptions(ops, dev):
    o = ops.selftest_layouts(dev).cpu()
    lane = torch.arange(64)
    # section 0: mfma 16x16x32: D[r][c] = 16 r + c at col = lane & 15, row = 4 (lane >> 4) + reg
    for reg in range(4):
        exp = 16 * (4 * (lane >> 4) + reg) + (lane & 15)
        assert torch.equal(o[0, :, reg], exp.float()), f"mfma16 C/D layout reg {reg}"
    # section 1: tr16_b64: lane i of group g gets tile[4g + r][i], r = reg
    for reg in range(4):
        exp = 16 * (4 * (lane >> 4) + reg) + (lane & 15)
        assert torch.equal(o[1, :, reg], exp.float()), f"ds_read_tr16_b64 layout reg {reg}"
    # section 2: mfma 32x32x16: D[r][c] = B[k = r & 15][c] = 8 (r & 15) + (c & 7), col = lane & 31,
    #            row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    for reg in range(16):
        row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        exp = 8 * (row & 15) + ((lane & 31) & 7)
        assert torch.equal(o[2, :, reg], exp.float()), f"mfma32 C/D layout reg {reg}"
    # section 3: LDS-DMA writes lane-linearly: lds[lane*8 + j] = src[lane*8 + j]
    src = torch.arange(512, dtype=torch.float32).to(BF).float()  # what the kernel was fed (bf16-rounded integers)
    for j in range(8):
        assert torch.equal(o[3, :, j], src[lane * 8 + j]), "global_load_lds placement"


# ----------------------------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(128, 128, 64), (200, 136, 72), (1000, 512, 1024), (64, 256, 2048), (37, 8, 8), (300, 384, 40), (515, 1152, 1152)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 10, 11, 12, 13, 14, 15, 16, 17, 20, 21])
def test_gemm_plain(ops, dev, M, N, K, tile):
    torch.manual_seed(M * 7 + N * 3 + K + tile)
    a, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev)
    out = ops.gemm(a, b, tile=tile)
    ref = (a.float() @ b.float().T).to(BF)
    close(out, ref, what=f"gemm {M}x{N}x{K} tile {tile}")


@pytest.mark.parametrize("tile", [0, 10])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_epilogue(ops, dev, act, tile):
    torch.manual_seed(act)
    M, N, K = 333, 264, 520
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    bias, cs, res = rnd(N, dev=dev), rnd(N, dev=dev), rnd(M, N, dev=dev)
    pre = torch.empty((M, N), dtype=BF, device=dev)
    out = ops.gemm(a, b, bias=bias, act=act, colscale=cs, residual=res, c_pre=pre, tile=tile)
    z = (a.float() @ b.float().T + bias.float()).to(BF)
    close(pre, z, what="c_pre")
    zf = z.float()
    h = [zf, torch.nn.functional.gelu(zf), torch.relu(zf), torch.nn.functional.silu(zf), torch.nn.functional.gelu(zf, approximate="tanh")][act]
    ref = ((h.to(BF).float() * cs.float()).to(BF).float() + res.float()).to(BF)
    close(out, ref, what=f"epilogue act {act}")


def test_gemm_film_epilogue(ops, dev):
    torch.manual_seed(5)
    Bn, rows, N, K = 3, 50, 128, 64
    a, b = rnd(Bn * rows, K, dev=dev), rnd(N, K, dev=dev, scale=0.2)
    res, gamma, beta = rnd(Bn * rows, N, dev=dev), rnd(Bn, N, dev=dev, scale=0.3), rnd(Bn, N, dev=dev)
    pre = torch.empty((Bn * rows, N), dtype=BF, device=dev)
    out = ops.gemm(a, b, residual=res, film=(gamma, beta, rows), c_pre=pre)
    x = ((a.float() @ b.float().T).to(BF).float() + res.float()).to(BF).float().view(Bn, rows, N)
    ref = ((x * (1 + gamma.float()).to(BF).float()[:, None]).to(BF).float() + beta.float()[:, None]).to(BF).view(Bn * rows, N)
    close(out, ref, what="film epilogue")
    close(pre, x.view(Bn * rows, N).to(BF), what="pre-FiLM save")
    # backward of the modulation
    dy = rnd(Bn * rows, N, dev=dev)
    dy0 = dy.clone()
    dg = torch.zeros((Bn, N), dtype=torch.float32, device=dev)
    db = torch.zeros((Bn, N), dtype=torch.float32, device=dev)
    ops.film_bwd(dy, pre, gamma, dg, db, Bn, rows)
    d3, p3 = dy0.float().view(Bn, rows, N), pre.float().view(Bn, rows, N)
    close(dg, (d3 * p3).sum(1), tol=2e-3, mean_tol=3e-4, what="film dgamma")
    close(db, d3.sum(1), tol=2e-3, mean_tol=3e-4, what="film dbeta")
    close(dy, (d3 * (1 + gamma.float()).to(BF).float()[:, None]).reshape(Bn * rows, N), what="film dx")


@pytest.mark.parametrize("tile", [0, 10, 11, 14, 20, 21])
@pytest.mark.parametrize("groups,K2", [(1, 32), (3, 32), (2, 64), (1, 16)])
def test_gemm_lora_extension(ops, dev, groups, K2, tile):
    torch.manual_seed(groups * 10 + K2)
    M, Ng, K = 300, 256, 192
    N = Ng * groups
    a, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.1)
    t, lb = rnd(M, groups * K2, dev=dev), rnd(N, K2, dev=dev, scale=0.2)
    out = ops.gemm(a, b, a2=t, b2=lb, k2_group_n=Ng if groups > 1 else 0, tile=tile)
    ref = a.float() @ b.float().T
    for g in range(groups):
        ref[:, g * Ng:(g + 1) * Ng] += t[:, g * K2:(g + 1) * K2].float() @ lb[g * Ng:(g + 1) * Ng].float().T
    close(out, ref.to(BF), what=f"lora k-extension groups={groups} K2={K2}")


W4_SHAPES = [(128, 128, 64), (1000, 512, 1024), (300, 520, 192), (515, 1152, 1152), (256, 256, 448), (4864, 4096, 1024), (2500, 2304, 2048)]


@pytest.mark.parametrize("M,N,K", W4_SHAPES)
@pytest.mark.parametrize("tile", [18, 118, 22, 122])
@pytest.mark.parametrize("variant", ["plain", "lora", "lora64", "lora96", "lora3", "bias_res", "gelu_pre", "split2"])
def test_gemm_w4_config(ops, dev, M, N, K, tile, variant):
    """The 4-wave 256x256 configuration (tiles 18 / 118 = with the hybrid remainder schedule; hand-scheduled inline-asm K loop, LoRA K-extension as a
    prologue, read-back epilogues) against torch fp32 on the same bf16 operands: odd and even K-tile counts, edge tiles, every epilogue path, split-K."""
    torch.manual_seed(M + N + K + len(variant))
    if tile in (22, 122) and variant in ("lora64", "lora96"):
        pytest.skip("the 128x256 configuration takes a K-extension of 0 or 32 columns")
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    ref = a.float() @ b.float().T
    kw = {}
    if variant in ("lora", "bias_res", "gelu_pre"):
        t, lb = rnd(M, 32, dev=dev), rnd(N, 32, dev=dev, scale=0.2)
        kw.update(a2=t, b2=lb)
        ref = ref + t.float() @ lb.float().T
    if variant in ("lora64", "lora96"):   # the data-gradient GEMMs of the grouped LoRA linears: one 64- / 96-column K-extension, no column groups
        k2 = int(variant[4:])
        t, lb = rnd(M, k2, dev=dev), rnd(N, k2, dev=dev, scale=0.2)
        kw.update(a2=t, b2=lb)
        ref = ref + t.float() @ lb.float().T
    if variant == "lora3":
        if N % 768 != 0:
            pytest.skip("grouped K-extension needs 256-column groups")
        G, Ng = 3, N // 3
        t, lb = rnd(M, 32 * G, dev=dev), rnd(N, 32, dev=dev, scale=0.2)
        kw.update(a2=t, b2=lb, k2_group_n=Ng)
        for g in range(G):
            ref[:, g * Ng:(g + 1) * Ng] += t[:, g * 32:(g + 1) * 32].float() @ lb[g * Ng:(g + 1) * Ng].float().T
    ref = ref.to(BF).float()
    if variant == "bias_res":
        bias, res = rnd(N, dev=dev), rnd(M, N, dev=dev)
        kw.update(bias=bias, residual=res)
        ref = ((a.float() @ b.float().T + kw["a2"].float() @ kw["b2"].float().T + bias.float()).to(BF).float() + res.float()).to(BF).float()
    if variant == "gelu_pre":
        bias, pre = rnd(N, dev=dev), torch.empty(M, N, dtype=BF, device=dev)
        kw.update(bias=bias, act=1, c_pre=pre)
        z = (a.float() @ b.float().T + kw["a2"].float() @ kw["b2"].float().T + bias.float()).to(BF)
        ref = torch.nn.functional.gelu(z.float()).to(BF).float()
    if variant == "split2":
        kw.update(split_k=2)
    out = ops.gemm(a, b, tile=tile, **kw)
    close(out, ref.to(BF), what=f"w4 {variant} {M}x{N}x{K} tile {tile}")
    if variant == "gelu_pre":
        close(kw["c_pre"], z, what="w4 pre-activation")


def test_gemm_w4_refuses_what_it_cannot_do(ops, dev):
    a, b = rnd(256, 72, dev=dev), rnd(256, 72, dev=dev)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, tile=18)                                             # K not a multiple of 64
    a, b = rnd(256, 128, dev=dev), rnd(256, 128, dev=dev)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, a2=rnd(256, 16, dev=dev), b2=rnd(256, 16, dev=dev), tile=18)   # K-extension other than 32


@pytest.mark.parametrize("tile", [0, 10, 5])
@pytest.mark.parametrize("split_k", [2, 8])
def test_gemm_split_k(ops, dev, split_k, tile):
    torch.manual_seed(split_k)
    M, N, K = 64, 384, 4096
    a, b, bias = rnd(M, K, dev=dev, scale=0.3), rnd(N, K, dev=dev, scale=0.1), rnd(N, dev=dev)
    out = ops.gemm(a, b, bias=bias, act=2, split_k=split_k, tile=tile)
    ref = torch.relu((a.float() @ b.float().T + bias.float()).to(BF).float()).to(BF)
    close(out, ref, what=f"split_k {split_k}")


@pytest.mark.parametrize("M,N,K,tile", [(4864, 4096, 1024, 117), (4864, 4096, 1024, 0), (1300, 1152, 512, 101), (2500, 2304, 2048, 117), (2500, 2304, 2048, 116), (300, 520, 256, 101),
                                        (522, 1024, 4096, 102), (522, 3072, 1024, 105), (608, 4096, 4096, 105), (522, 1024, 4096, 0), (512, 4304, 1152, 0),
                                        (608, 4096, 11008, 0), (1000, 2304, 1088, 102)])
def test_gemm_hybrid_schedule(ops, dev, M, N, K, tile):
    """Full rounds data-parallel + remainder tiles split along K (partial slabs + reduce kernel with the epilogue)."""
    torch.manual_seed(M + N + K)
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    bias, res = rnd(N, dev=dev), rnd(M, N, dev=dev)
    t, lb = rnd(M, 32, dev=dev), rnd(N, 32, dev=dev, scale=0.2)
    out = ops.gemm(a, b, bias=bias, act=1, residual=res, a2=t, b2=lb, tile=tile)
    z = (a.float() @ b.float().T + t.float() @ lb.float().T + bias.float()).to(BF).float()
    ref = (torch.nn.functional.gelu(z).to(BF).float() + res.float()).to(BF)
    close(out, ref, what=f"hybrid {M}x{N}x{K} tile {tile}")


def test_gemm_auto_skinny(ops, dev):
    """tile=0 picks split-K for skinny outputs (LoRA t / dt) and small M (action head)."""
    torch.manual_seed(12)
    for M, N, K in [(4864, 96, 4096), (4864, 32, 11008), (64, 4096, 28672), (1000, 64, 1152), (4176, 32, 1024), (4100, 32, 3072), (522, 32, 1160),
                    (1030, 32, 72)]:   # the last four take the single-launch skinny kernel (N = 32, K <= 3072)
        a, b = rnd(M, K, dev=dev, scale=0.3), rnd(N, K, dev=dev, scale=0.1)
        close(ops.gemm(a, b, alpha=0.5), (0.5 * (a.float() @ b.float().T)).to(BF), what=f"auto skinny {M}x{N}x{K}")


def test_gemm_strided_views(ops, dev):
    torch.manual_seed(11)
    big = rnd(100, 512, dev=dev)
    a = big[:, 128:320]  # lda 512, K = 192
    b = rnd(64, 192, dev=dev)
    outbuf = torch.zeros((100, 256), dtype=BF, device=dev)
    ops.gemm(a, b, out=outbuf[:, 64:128])
    close(outbuf[:, 64:128], (a.float() @ b.float().T).to(BF), what="strided gemm")
    assert outbuf[:, :64].abs().max() == 0 and outbuf[:, 128:].abs().max() == 0, "wrote outside the output view"


def test_gemm_rejects_bad_arguments(ops, dev):
    a, b = rnd(16, 12, dev=dev), rnd(16, 12, dev=dev)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.gemm(a, b)


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,P,Q", [(100, 32, 200), (1000, 32, 4096), (777, 264, 32), (64, 136, 264), (4864, 32, 512), (300, 8, 16)])
def test_gemm_tn(ops, dev, M, P, Q):
    torch.manual_seed(M + P + Q)
    x, y = rnd(M, P, dev=dev), rnd(M, Q, dev=dev)
    ref = 0.5 * (x.float().T @ y.float())
    acc = torch.ones((P, Q), dtype=torch.float32, device=dev)
    ops.gemm_tn(x, y, out=acc, alpha=0.5, accumulate=True)
    close(acc - 1.0, ref, tol=2e-3, mean_tol=2e-4, what=f"gemm_tn atomic {M},{P},{Q}")
    st = ops.gemm_tn(x, y, alpha=0.5, accumulate=False, out_dtype=torch.float32)
    close(st, ref, tol=2e-3, mean_tol=2e-4, what="gemm_tn store f32")
    sb = ops.gemm_tn(x, y, alpha=0.5, accumulate=False, out_dtype=BF)
    close(sb, ref.to(BF), what="gemm_tn store bf16")


def test_gemm_tn_strided(ops, dev):
    torch.manual_seed(3)
    t = rnd(500, 96, dev=dev)
    dy = rnd(500, 384, dev=dev)
    out = ops.gemm_tn(dy[:, 128:256], t[:, 32:64], accumulate=False)
    close(out, dy[:, 128:256].float().T @ t[:, 32:64].float(), tol=2e-3, mean_tol=2e-4, what="gemm_tn strided")


def test_gemm_tn_grouped_and_block_diagonal(ops, dev):
    """The LoRA backward of a fused q|k|v linear: one block-diagonal skinny GEMM for dt and one grouped TN launch."""
    torch.manual_seed(21)
    M, G, gn, r, inn = 1000, 3, 256, 32, 384
    dy, BT = rnd(M, G * gn, dev=dev), rnd(G * r, gn, dev=dev, scale=0.2)
    dt = ops.gemm(dy, BT, alpha=0.5, a_group_n=r)
    ref = torch.cat([0.5 * dy[:, g * gn:(g + 1) * gn].float() @ BT[g * r:(g + 1) * r].float().T for g in range(G)], 1)
    close(dt, ref.to(BF), what="block-diagonal dt")
    t, x = rnd(M, G * r, dev=dev), rnd(M, inn, dev=dev)
    gB = torch.ones((G * gn, r), dtype=torch.float32, device=dev)
    gA = torch.ones((G * r, inn), dtype=torch.float32, device=dev)
    probs = [(dy[:, g * gn:(g + 1) * gn], t[:, g * r:(g + 1) * r], gB[g * gn:(g + 1) * gn]) for g in range(G)] + [(dt, x, gA)]
    ops.gemm_tn_grouped(probs)
    for g in range(G):
        close(gB[g * gn:(g + 1) * gn] - 1, dy[:, g * gn:(g + 1) * gn].float().T @ t[:, g * r:(g + 1) * r].float(), tol=2e-3, mean_tol=2e-4, what=f"grouped dB{g}")
    close(gA - 1, dt.float().T @ x.float(), tol=2e-3, mean_tol=2e-4, what="grouped dA")
    # batched transposes
    srcs = [rnd(96, 300, dev=dev), rnd(520, 32, dev=dev), rnd(64, 64, dev=dev)]
    dsts = [torch.zeros((s_.shape[1], s_.shape[0]), dtype=BF, device=dev) for s_ in srcs]
    tab = ops.transpose_table(list(zip(srcs, dsts)), dev)
    ops.transpose_batched(tab)
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, s_.T.contiguous())


def test_colsum(ops, dev):
    x = rnd(1000, 264, dev=dev)
    out = torch.zeros(264, dtype=torch.float32, device=dev)
    ops.colsum(x, out)
    close(out, x.float().sum(0), tol=1e-4, mean_tol=1e-5, what="colsum")


# ----------------------------------------------------------------------------------------------------------------------
def ref_attention(q, k, v, kv_len, causal, scale):
    """q,k,v fp32 [B,H,S,hd] (requires_grad ok)."""
    B, H, S, hd = q.shape
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.zeros((B, 1, S, S), dtype=torch.bool, device=q.device)
    if kv_len is not None:
        mask |= (torch.arange(S, device=q.device)[None, None, None, :] >= kv_len[:, None, None, None])
    if causal:
        mask |= torch.triu(torch.ones(S, S, dtype=torch.bool, device=q.device), 1)[None, None]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return p @ v, torch.logsumexp(s, dim=-1)


ATTN_CASES = [
    # B, H, S, hd, kv_len, causal
    (2, 2, 64, 128, None, False),
    (2, 3, 261, 64, None, False),
    (1, 4, 256, 72, None, False),
    (3, 2, 300, 128, [300, 250, 17], False),
    (2, 2, 200, 128, None, True),
    (2, 2, 333, 128, [333, 100], True),
    (1, 2, 50, 72, None, False),
    # round 2: the 32-row forward and the XCD-contiguous block order (block counts that are not multiples of 8, one-block grids,
    # key padding that ends on / inside / before the first tile, the ALOHA context, causal diagonals inside the 128-row blocks)
    (1, 3, 65, 128, None, False),
    (3, 5, 130, 64, [130, 64, 1], False),
    (2, 3, 129, 72, [129, 77], True),
    (1, 2, 1159, 128, [1100], False),
    (2, 2, 200, 128, [128, 192], False),
    (1, 3, 608, 128, None, True),
    (5, 1, 96, 64, None, True),
]


@pytest.mark.parametrize("B,H,S,hd,kvl,causal", ATTN_CASES)
def test_attention_fwd_bwd(ops, dev, B, H, S, hd, kvl, causal):
    torch.manual_seed(B * 100 + S + hd)
    # fused qkv layout [B*S, 3*H*hd], exactly what the QKV GEMM produces
    qkv = rnd(B * S, 3 * H * hd, dev=dev)
    q, k, v = qkv[:, : H * hd], qkv[:, H * hd: 2 * H * hd], qkv[:, 2 * H * hd:]
    kv_len = None if kvl is None else torch.tensor(kvl, dtype=torch.int32, device=dev)
    scale = hd ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv_len, causal=causal)

    def heads(t):
        return t.float().reshape(B, S, H, hd).permute(0, 2, 1, 3).contiguous().requires_grad_(True)

    qf, kf, vf = heads(q), heads(k), heads(v)
    ro, rl = ref_attention(qf, kf, vf, None if kv_len is None else kv_len.long(), causal, scale)
    ref = ro.permute(0, 2, 1, 3).reshape(B * S, H * hd)
    close(out, ref, tol=2e-2, mean_tol=3e-3, what="attn fwd")
    close(lse, rl, tol=1e-2, mean_tol=2e-3, what="attn lse")

    do = rnd(B * S, H * hd, dev=dev)
    dq, dk, dv = ops.attn_bwd(q, k, v, out, do, lse, B, S, H, hd, kv_len=kv_len, causal=causal)
    ro.backward(do.float().reshape(B, S, H, hd).permute(0, 2, 1, 3))

    def unheads(t):
        return t.permute(0, 2, 1, 3).reshape(B * S, H * hd)

    close(dv, unheads(vf.grad), tol=3e-2, mean_tol=4e-3, what="attn dV")
    close(dk, unheads(kf.grad), tol=3e-2, mean_tol=4e-3, what="attn dK")
    close(dq, unheads(qf.grad), tol=3e-2, mean_tol=4e-3, what="attn dQ")


def test_attention_spiked_scores(ops, dev):
    """Forces large running-max jumps between KV tiles (online-softmax rescale path)."""
    torch.manual_seed(0)
    B, H, S, hd = 1, 1, 256, 128
    q, k, v = rnd(S, hd, dev=dev), rnd(S, hd, dev=dev), rnd(S, hd, dev=dev)
    k[200] = q[5] * 4.0  # one key in the 4th tile dominates query 5
    k[70] = q[9] * 3.0
    out, lse = ops.attn_fwd(q, k, v, B, S, H, hd)
    ro, rl = ref_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None], None, False, hd ** -0.5)
    close(out, ro[0, 0], tol=2e-2, mean_tol=3e-3, what="spiked attn")
    close(lse, rl, tol=1e-2, what="spiked lse")


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,dim", [(37, 4096), (64, 28672), (261, 1024), (100, 1152), (5, 64)])
@pytest.mark.parametrize("rms", [True, False])
def test_norm_fwd_bwd(ops, dev, rows, dim, rms):
    torch.manual_seed(rows + dim)
    x = rnd(rows, dim, dev=dev, scale=2.0) + 0.5
    w, b = rnd(dim, dev=dev) * 0.1 + 1.0, rnd(dim, dev=dev, scale=0.1)
    w, b = w.to(BF), b.to(BF)
    eps = 1e-5 if rms else 1e-6
    y, mean, rstd = ops.norm_fwd(x, w, None if rms else b, eps=eps, rms=rms)
    xf = x.float().requires_grad_(True)
    wf, bf_ = w.float().requires_grad_(True), b.float().requires_grad_(True)
    if rms:
        ref = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))
    else:
        ref = torch.nn.functional.layer_norm(xf, (dim,), wf, bf_, eps)
    close(y, ref, what="norm fwd")
    dy = rnd(rows, dim, dev=dev)
    dw = torch.zeros(dim, dtype=torch.float32, device=dev)
    db = torch.zeros(dim, dtype=torch.float32, device=dev)
    dx = ops.norm_bwd(x, dy, w, mean, rstd, rms=rms, dweight=dw, dbias=None if rms else db)
    ref.backward(dy.float())
    close(dx, xf.grad, what="norm dx")
    close(dw, wf.grad, tol=2e-3, mean_tol=3e-4, what="norm dw")
    if not rms:
        close(db, bf_.grad, tol=2e-3, mean_tol=3e-4, what="norm db")
    base = rnd(rows, dim, dev=dev)
    acc = base.clone()
    ops.norm_bwd(x, dy, w, mean, rstd, rms=rms, dx=acc, dx_accum=True)
    close(acc, base.float() + xf.grad, what="norm dx accumulate")


def test_rope_matches_hf_formula(ops, dev):
    torch.manual_seed(1)
    B, S, H, hd = 2, 77, 3, 128
    qkv = rnd(B * S, 3 * H * hd, dev=dev)
    orig = qkv.clone()
    cos, sin = ops.rope_table(S, hd, 10000.0, dev)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))
    ang = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * inv_freq[None]
    close(cos, ang.cos(), tol=8e-3, what="cos table")
    close(sin, ang.sin(), tol=8e-3, what="sin table")
    ops.rope_(qkv, S, 2 * H, hd, cos, sin)
    c = torch.cat([ang.cos(), ang.cos()], -1).to(BF).float().repeat(B, 1)[:, None, :]  # [B*S, 1, hd]
    s = torch.cat([ang.sin(), ang.sin()], -1).to(BF).float().repeat(B, 1)[:, None, :]
    x = orig[:, : 2 * H * hd].float().view(B * S, 2 * H, hd)
    rot = torch.cat([-x[..., hd // 2:], x[..., : hd // 2]], -1)
    ref = ((x * c).to(BF).float() + (rot * s).to(BF).float()).to(BF).view(B * S, 2 * H * hd)
    close(qkv[:, : 2 * H * hd], ref, what="rope")
    assert torch.equal(qkv[:, 2 * H * hd:], orig[:, 2 * H * hd:]), "rope touched V"
    ops.rope_(qkv, S, 2 * H, hd, cos, sin, inverse=True)
    close(qkv[:, : 2 * H * hd], orig[:, : 2 * H * hd], tol=3e-2, mean_tol=5e-3, what="rope inverse")


def test_swiglu_and_act_bwd(ops, dev):
    torch.manual_seed(2)
    rows, F = 123, 264
    gu = rnd(rows, 2 * F, dev=dev, scale=1.5)
    h = ops.swiglu_fwd(gu)
    gf = gu.float().requires_grad_(True)
    ref = torch.nn.functional.silu(gf[:, :F]) * gf[:, F:]
    close(h, ref, what="swiglu fwd")
    dh = rnd(rows, F, dev=dev)
    dgu = ops.swiglu_bwd(gu, dh)
    ref.backward(dh.float())
    close(dgu, gf.grad, what="swiglu bwd")
    for act, fn in [(1, torch.nn.functional.gelu), (2, torch.relu), (3, torch.nn.functional.silu),
                    (4, lambda t: torch.nn.functional.gelu(t, approximate="tanh"))]:
        z = rnd(rows, F, dev=dev, scale=1.5)
        zf = z.float().requires_grad_(True)
        fn(zf).backward(dh.float())
        close(ops.act_bwd(z, dh, act), zf.grad, what=f"act_bwd {act}")


# ----------------------------------------------------------------------------------------------------------------------
def test_vision_front_end_glue(ops, dev):
    torch.manual_seed(4)
    B, H, W, patch = 2, 56, 56, 14
    px = rnd(B, 12, H, W, dev=dev)
    kp = 592
    cols = ops.im2col(px, 3, patch, kp)
    ref = torch.nn.functional.unfold(px[:, 3:6].float(), kernel_size=patch, stride=patch).transpose(1, 2).reshape(B * 16, 588)
    assert torch.equal(cols[:, :588].float(), ref) and cols[:, 588:].abs().max() == 0, "im2col"
    dim, npatch, npre = 64, 16, 5
    patches, pos, prefix = rnd(B * npatch, dim, dev=dev), rnd(npatch, dim, dev=dev), rnd(npre, dim, dev=dev)
    tok = ops.vit_embed(patches, pos, prefix, B, npatch, dim).view(B, npatch + npre, dim)
    assert torch.equal(tok[:, :npre], prefix[None].expand(B, -1, -1))
    close(tok[:, npre:], (patches.view(B, npatch, dim).float() + pos.float()[None]).to(BF), what="vit_embed")
    tok2 = ops.vit_embed(patches, pos, None, B, npatch, dim).view(B, npatch, dim)
    close(tok2, (patches.view(B, npatch, dim).float() + pos.float()[None]).to(BF), what="vit_embed no prefix")
    feat = torch.zeros((B, 2 * npatch, 96), dtype=BF, device=dev)
    ops.copy_rows(tok, feat, B, npatch, dim, src_batch_stride=(npatch + npre) * dim, src_row0=npre, src_ld=dim,
                  dst_batch_stride=2 * npatch * 96, dst_row0=npatch, dst_ld=96, dst_col0=32)
    assert torch.equal(feat[:, npatch:, 32:96], tok[:, npre:]) and feat[:, :npatch].abs().max() == 0 and feat[:, :, :32].abs().max() == 0
    ops.copy_rows(tok, feat, B, npatch, dim, src_batch_stride=(npatch + npre) * dim, src_row0=npre, src_ld=dim,
                  dst_batch_stride=2 * npatch * 96, dst_row0=npatch, dst_ld=96, dst_col0=32, accumulate=True)
    close(feat[:, npatch:, 32:96], 2 * tok[:, npre:].float(), what="copy_rows accumulate")
    L = 9
    x = rnd(B, L, dim, dev=dev)
    mask = torch.tensor([[1, 1, 0, 1, 1, 0, 0, 1, 1], [1, 0, 0, 0, 0, 0, 0, 0, 1]], dtype=torch.uint8, device=dev)
    mm = ops.masked_mean(x, mask, B, L, dim)
    ref = torch.stack([x[i][mask[i].bool()].float().mean(0) for i in range(B)])
    close(mm, ref, what="masked_mean")


def test_assemble_and_gather(ops, dev):
    torch.manual_seed(6)
    B, Tp, A, D, P, V = 3, 6, 14, 64, 10, 32064
    L = Tp + A + 1 + 2  # two pad columns
    ids = torch.randint(3, 31000, (B, L), device=dev)
    labels = torch.full((B, L), -100, dtype=torch.long, device=dev)
    for b_, tp in enumerate([Tp, Tp + 2, Tp - 1]):  # ragged prompts, right padded
        ids[b_, tp: tp + A] = torch.randint(31744, 32000, (A,), device=dev)
        ids[b_, tp + A] = 2
        ids[b_, tp + A + 1:] = 32000
        labels[b_, tp: tp + A + 1] = ids[b_, tp: tp + A + 1]
    table, patches, noisy = rnd(V, D, dev=dev), rnd(B, P, D, dev=dev), rnd(B, A, D, dev=dev)
    # reference restatement of modeling_prismatic.py:571-629
    cum = (labels != -100).cumsum(1)
    amask = (cum >= 1) & (labels > 31743)
    emb = table[ids]
    for use_noisy in (False, True):
        e = emb.clone()
        if use_noisy:
            for b_ in range(B):
                e[b_, amask[b_]] = noisy[b_]
        else:
            e = e * (~amask)[..., None]
        ref = torch.cat([e[:, :1], patches, e[:, 1:]], 1)
        out, apos = ops.assemble_multimodal(ids, labels, table, patches, A=A, noisy=noisy if use_noisy else None)
        assert torch.equal(out, ref), f"assemble (noisy={use_noisy})"
        for b_ in range(B):
            assert torch.equal(apos[b_].long(), b_ * (P + L) + P + torch.where(amask[b_])[0] - 1), "predicting rows of the action slots"
    src = rnd(50, D, dev=dev)
    idx = torch.tensor([3, 49, 0, 7], dtype=torch.int32, device=dev)
    g = ops.gather_rows(src, idx, D)
    assert torch.equal(g, src[idx.long()])
    dst = torch.zeros((50, D), dtype=BF, device=dev)
    ops.gather_rows(g, idx, D, dst=dst, scatter_add=True)
    assert torch.equal(dst[idx.long()], g) and dst.abs().sum() == g.abs().sum()


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("adim,mse", [(7, False), (14, False), (7, True)])
def test_head_tail(ops, dev, adim, mse):
    torch.manual_seed(adim)
    rows, dim = 64, 4096
    x, W, b = rnd(rows, dim, dev=dev), rnd(adim, dim, dev=dev, scale=0.02), rnd(adim, dev=dev, scale=0.1)
    tgt = (torch.rand(rows, adim, device=dev) * 2 - 1).to(BF)
    loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    pred = ops.head_out_fwd(x, W, b, tgt, loss_sum, mse=mse)
    xf, Wf, bf_ = x.float().requires_grad_(True), W.float().requires_grad_(True), b.float().requires_grad_(True)
    rp = xf @ Wf.T + bf_
    close(pred, rp, what="head pred")
    rloss = torch.nn.functional.mse_loss(rp, tgt.float()) if mse else torch.nn.functional.l1_loss(rp, tgt.float())
    close(loss_sum / (rows * adim), rloss.detach().reshape(1), tol=1e-2, mean_tol=1e-2, what="head loss")
    dW = torch.zeros((adim, dim), dtype=torch.float32, device=dev)
    db = torch.zeros(adim, dtype=torch.float32, device=dev)
    dx = ops.head_out_bwd(x, W, pred, tgt, 1.0 / (rows * adim), dW, db, mse=mse)
    # reference gradient evaluated at the bf16 prediction the kernel used
    d = (pred.float() - tgt.float()).to(BF).float()
    gp = (2 * d if mse else torch.sign(d)) / (rows * adim)
    gp = gp.to(BF).float()
    close(dx, gp @ W.float(), what="head dx")
    close(dW, gp.T @ x.float(), tol=2e-3, mean_tol=3e-4, what="head dW")
    close(db, gp.sum(0), tol=2e-3, mean_tol=3e-4, what="head db")


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_adamw_matches_torch(ops, dev, dtype):
    """Three steps of the fused AdamW against torch.optim.AdamW on CPU (same dtype semantics as the reference, which
    keeps LoRA/head parameters AND optimizer state in bf16: vla-scripts/finetune.py:952)."""
    torch.manual_seed(9)
    n = 10007
    p0 = (torch.randn(n) * 0.05).to(dtype)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=5e-4)
    p = p0.clone().to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g32 = torch.randn(n) * (10.0 ** torch.randint(-4, 1, (n,)).float())
        ref_p.grad = g32.to(dtype)
        opt.step()
        ops.adamw(p, m, v, g32.to(dev), step=step, lr=5e-4)
        if dtype == BF:
            mism = (p.cpu().view(torch.int16) != ref_p.data.view(torch.int16)).float().mean().item()
            assert mism <= 2e-3, f"step {step}: {mism:.2%} of bf16 parameters differ from torch"
            close(p.cpu(), ref_p.data, tol=8e-3, mean_tol=1e-5, what="adamw bf16 param")
        else:
            close(p.cpu(), ref_p.data, tol=1e-6, mean_tol=1e-7, what="adamw f32 param")
    st = opt.state[ref_p]
    close(m.cpu(), st["exp_avg"], tol=8e-3 if dtype == BF else 1e-6, mean_tol=1e-4, what="exp_avg")
    close(v.cpu(), st["exp_avg_sq"], tol=8e-3 if dtype == BF else 1e-6, mean_tol=1e-4, what="exp_avg_sq")


def test_transpose_and_casts(ops, dev):
    x = rnd(100, 264, dev=dev)
    assert torch.equal(ops.transpose(x), x.T.contiguous())
    f = torch.randn(1000, device=dev)
    assert torch.equal(ops.cvt_f32_to_bf16(f, scale=0.5), (f * 0.5).to(BF))
    assert torch.equal(ops.cvt_bf16_to_f32(x, scale=2.0), x.float() * 2.0)
    a, b = rnd(64, 128, dev=dev), rnd(64, 128, dev=dev)
    assert torch.equal(ops.add(a, b), (a.float() + b.float()).to(BF))
    s = rnd(128, dev=dev)
    assert torch.equal(ops.colscale(a, s), (a.float() * s.float()).to(BF))


@pytest.mark.parametrize("crop", [True, False])
@pytest.mark.parametrize("hw", [(224, 224), (256, 320)])
def test_image_prep_bit_exact_vs_oracle(ops, dev, crop, hw):
    """ovla_image_prep (center crop by TF's crop_and_resize rule -> uint8 -> DINOv2 / SigLIP normalisation, one launch)
    against the numpy restatement in the oracle: bit-exact (every fp32 op is individually rounded on both sides)."""
    import numpy as np
    from oracle import vla_oracle as vo

    H, W = hw
    if not crop and (H, W) != (224, 224):
        pytest.skip("without the crop the input must already be 224 x 224")
    rng = np.random.default_rng(H * 7 + W + int(crop))
    imgs = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    imgs[0, :8] = 255; imgs[1, :, :8] = 0                      # saturated borders
    got = ops.image_prep(torch.from_numpy(imgs).to(dev), crop=crop)
    assert got.shape == (1, 18, 224, 224) and got.dtype == BF
    ref = []
    for im in imgs:
        q = vo.crop_and_resize_center(im) if crop else im
        ref.append(vo.image_transform(q, (vo.IMAGENET_MEAN, vo.SIGLIP_MEAN), (vo.IMAGENET_STD, vo.SIGLIP_STD)))
    ref = torch.cat(ref, 0)[None].to(BF)
    assert torch.equal(got.cpu(), ref), f"{(got.cpu() != ref).sum().item()} of {ref.numel()} values differ"


def _resolved_tile(ops, M, N, K, tile, K2=32):
    """tile = 0 resolves per call (the 4-wave 256x256 config takes the plain / bias / residual epilogues only): a bit-for-bit comparison of a fused
    epilogue with its unfused sequence pins both GEMMs to the configuration the planner names."""
    return {17: 117, 1: 101, 2: 102, 5: 105}[ops.gemm_plan(M, N, K, K2)[0]] if tile == 0 and M > 64 and N > 128 else tile


@pytest.mark.parametrize("M,N,K,tile", [(4864, 1024, 4096, 0), (1000, 512, 256, 1), (300, 264, 1088, 101), (522, 1024, 4096, 0)])
@pytest.mark.parametrize("act", [1, 2, 3])
def test_gemm_backward_epilogue_act(ops, dev, M, N, K, tile, act):
    """dact_mode 1: C = (A B^T + A2 B2^T) * act'(z) must equal the separate act_bwd kernel applied to the plain GEMM output, bit for bit."""
    torch.manual_seed(M + act)
    a, b = rnd(M, K, dev=dev, scale=0.3), rnd(N, K, dev=dev, scale=0.1)
    t, lb = rnd(M, 32, dev=dev), rnd(N, 32, dev=dev, scale=0.2)
    z = rnd(M, N, dev=dev)
    tile = _resolved_tile(ops, M, N, K, tile)
    dh = ops.gemm(a, b, a2=t, b2=lb, tile=tile)
    ref = ops.act_bwd(z, dh, act)
    got = ops.gemm(a, b, a2=t, b2=lb, tile=tile, dact=("act", z, act))
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,F,K,tile", [(4864, 1024, 512, 0), (608, 2752, 1024, 0), (200, 256, 320, 1), (2500, 1280, 2048, 117)])
def test_gemm_backward_epilogue_swiglu(ops, dev, M, F, K, tile):
    """dact_mode 2: the down-projection's data-gradient GEMM writes d(gate|up) [M, 2F] directly == swiglu_bwd(gu, dh) of the plain output."""
    torch.manual_seed(M + F)
    a, b = rnd(M, K, dev=dev, scale=0.3), rnd(F, K, dev=dev, scale=0.1)
    t, lb = rnd(M, 32, dev=dev), rnd(F, 32, dev=dev, scale=0.2)
    gu = rnd(M, 2 * F, dev=dev)
    tile = _resolved_tile(ops, M, F, K, tile)
    dh = ops.gemm(a, b, a2=t, b2=lb, tile=tile)
    ref = ops.swiglu_bwd(gu, dh)
    got = ops.gemm(a, b, a2=t, b2=lb, tile=tile, dact=("swiglu", gu))
    assert got.shape == (M, 2 * F) and torch.equal(got, ref)


@pytest.mark.parametrize("hd,S,causal", [(128, 608, False), (128, 200, True), (64, 130, False)])
def test_attn_bwd_fused_inverse_rope(ops, dev, hd, S, causal):
    """ovla_attn_bwd with rope tables == ovla_attn_bwd followed by ovla_rope(inverse) on dq | dk, bit for bit."""
    torch.manual_seed(hd + S)
    B, H = 2, 3
    D = H * hd
    qkv = rnd(B * S, 3 * D, dev=dev)
    do = rnd(B * S, D, dev=dev, scale=0.1)
    kv_len = torch.tensor([S, S - 37], dtype=torch.int32, device=dev)
    cos, sin = ops.rope_table(S + 5, hd, 10000.0, dev)      # tables may be longer than the sequence
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv_len, causal=causal)
    d_ref = torch.empty_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, B, S, H, hd, kv_len=kv_len, causal=causal, dq=d_ref[:, :D], dk=d_ref[:, D:2 * D], dv=d_ref[:, 2 * D:])
    ops.rope_(d_ref, S, 2 * H, hd, cos, sin, inverse=True)
    d_fused = torch.empty_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, B, S, H, hd, kv_len=kv_len, causal=causal, dq=d_fused[:, :D], dk=d_fused[:, D:2 * D], dv=d_fused[:, 2 * D:],
                 rope=(cos, sin))
    assert torch.equal(d_fused, d_ref)


@pytest.mark.parametrize("M,S,tile", [(4864, 608, 0), (4864, 608, 116), (4864, 608, 117), (4864, 608, 17), (512, 128, 17), (608, 608, 0), (608, 608, 17),
                                      (1216, 304, 16), (700, 100, 1), (520, 130, 2), (608, 608, 1), (608, 608, 101), (1000, 250, 101), (4864, 608, 101),
                                      (4864, 608, 118), (4864, 608, 18), (700, 100, 18), (1216, 304, 118)])
def test_gemm_rope_epilogue(ops, dev, M, S, tile):
    """RoPE in the q|k|v projection's epilogue (fused in the 256x256 configs: 2x4 waves = the default layout, in its unrolled read-back with
    the partner-column wave map, and 4x2 waves; since round 3 also in the 128x128 config -- one head per column tile, any M: the batch-1 chunk's
    M = 608; all incl. the hybrid-remainder reduce; other schedules append one ovla_rope launch) == plain GEMM followed by the separate RoPE pass, bit for bit.  2 q | 2 k | 2 v heads of 128: q | k rotated, v untouched."""
    torch.manual_seed(M + tile)
    hd, K = 128, 512
    N = 3 * 2 * hd                      # 2 q heads | 2 k heads | 2 v heads
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    t, lb = rnd(M, 96, dev=dev), rnd(N, 32, dev=dev, scale=0.2)
    cos, sin = ops.rope_table(S + 3, hd, 10000.0, dev)
    kw = dict(a2=t, b2=lb, k2_group_n=N // 3, tile=tile)
    ref = ops.gemm(a, b, **kw)
    ops.rope_(ref, S, 4, hd, cos, sin)
    got = ops.gemm(a, b, rope=(cos, sin, S, 4 * hd), **kw)
    assert torch.equal(got, ref), f"{(got != ref).sum().item()} of {ref.numel()} differ"


@pytest.mark.parametrize("M,F,K,fold", [(608, 1408, 512, False), (608, 1408, 1024, True), (300, 256, 4096, True), (1000, 640, 192, False), (128, 128, 64, False)])
def test_gemm_swiglu_pair_epilogue(ops, dev, M, F, K, fold):
    """act = OVLA_ACT_SWIGLU on the 4-wave 128x256 configuration: the stacked [gate; up] projection with silu(gate) * up in the read-back (its column map puts
    a gate value and its up partner into the same slab row) == the projection followed by ovla_swiglu_fwd, bit for bit -- with and without the RMSNorm-fold
    row scale, edge row tiles included."""
    torch.manual_seed(M + F + K)
    x, w = rnd(M, K, dev=dev, scale=0.5), rnd(2 * F, K, dev=dev, scale=0.1)
    kw = {}
    if fold:
        part = (x.float().view(M, K // 64, 64).pow(2).sum(-1)).contiguous()
        kw["rowscale"] = (part, 1e-5, torch.empty(M, device=dev))
    gu = ops.gemm(x, w, tile=22, **kw)
    ref = ops.swiglu_fwd(gu)
    got = ops.gemm(x, w, tile=22, act=ops.ACT_SWIGLU, **kw)
    assert got.shape == (M, F) and torch.equal(got, ref), f"{(got != ref).sum().item()} of {ref.numel()} differ"
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, tile=17, act=ops.ACT_SWIGLU)


@pytest.mark.parametrize("M,F,K,tile,lora", [(4864, 1024, 512, 0, True), (2500, 1280, 2048, 118, True), (300, 256, 320 + 64, 18, True), (1000, 640, 192, 18, False),
                                            (4864, 2816, 1024, 0, True), (608, 1408, 512, 122, False)])
def test_gemm_swiglu_pair_training_shape(ops, dev, M, F, K, tile, lora):
    """act = OVLA_ACT_SWIGLU on the 256x256 4-wave configuration, the fine-tune step's gate|up projection: grouped LoRA K-extension (group of the gate rows /
    of the up rows), C_pre = the [M, 2 F] projection output kept for the backward, in-kernel read-back AND the hybrid-remainder reduce -- bit for bit the
    projection followed by ovla_swiglu_fwd."""
    torch.manual_seed(M + F + K)
    x, w = rnd(M, K, dev=dev, scale=0.5), rnd(2 * F, K, dev=dev, scale=0.1)
    kw = {}
    if lora:
        kw.update(a2=rnd(M, 64, dev=dev), b2=rnd(2 * F, 32, dev=dev, scale=0.2), k2_group_n=F)
    ptile = {0: 118}.get(tile, tile)
    gu = ops.gemm(x, w, tile=ptile, **kw)
    ref = ops.swiglu_fwd(gu)
    pre = torch.full((M, 2 * F), float("nan"), dtype=BF, device=dev)
    got = ops.gemm(x, w, tile=tile, act=ops.ACT_SWIGLU, c_pre=pre, **kw)
    assert got.shape == (M, F) and torch.equal(got, ref), f"h: {(got != ref).sum().item()} of {ref.numel()} differ"
    assert torch.equal(pre, gu), f"gate|up: {(pre != gu).sum().item()} of {gu.numel()} differ"
    got2 = ops.gemm(x, w, tile=tile, act=ops.ACT_SWIGLU, **kw)
    assert torch.equal(got2, ref)


@pytest.mark.parametrize("M,S,tile", [(608, 608, 22), (608, 608, 122), (700, 100, 22), (1216, 304, 122), (4864, 608, 122)])
def test_gemm_rope_epilogue_128x256(ops, dev, M, S, tile):
    """RoPE in the epilogue of the 4-wave 128x256 configuration (its column map for RoPE launches; no K-extension: the merged decoder of the batch-1 chunk)
    == plain GEMM of the same configuration followed by the separate RoPE pass, bit for bit.  3 q | 3 k | 2 v heads of 128: q | k rotated, v untouched."""
    torch.manual_seed(M + tile)
    hd, K = 128, 512
    N = 8 * hd
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    cos, sin = ops.rope_table(S + 3, hd, 10000.0, dev)
    ref = ops.gemm(a, b, tile=tile)
    ops.rope_(ref, S, 6, hd, cos, sin)
    got = ops.gemm(a, b, rope=(cos, sin, S, 6 * hd), tile=tile)
    assert torch.equal(got, ref), f"{(got != ref).sum().item()} of {ref.numel()} differ"


@pytest.mark.parametrize("rows,vocab,ld", [(171, 32064, 32064), (5, 1000, 1008), (64, 257, 264)])
def test_token_ce(ops, dev, rows, vocab, ld):
    """ovla_token_ce == torch cross entropy on the fp32-upcast bf16 logits (per-row loss, argmax, gradient), also in place."""
    torch.manual_seed(rows + vocab)
    logits = torch.full((rows, ld), 7.0).to(BF)
    logits[:, :vocab] = (torch.randn(rows, vocab) * 3).to(BF)
    logits[0, 5] = logits[0, 9] = 30.0                                       # a tie for the maximum: the lowest index wins
    tgt = torch.randint(0, vocab, (rows,))
    tgt[1] = int(logits[1, :vocab].float().argmax())
    lf = logits[:, :vocab].float().requires_grad_(True)
    ref_rows = torch.nn.functional.cross_entropy(lf, tgt, reduction="none")
    scale = 0.37 / rows
    (ref_rows.sum() * scale).backward()
    dl = logits.to(dev)
    loss_rows, amax, d = ops.token_ce(dl, tgt.to(dev), vocab=vocab, grad_scale=scale, inplace_grad=False)
    assert torch.allclose(loss_rows.cpu(), ref_rows.detach(), rtol=2e-5, atol=2e-5)
    assert amax[0].item() == 5 and torch.equal(amax.cpu().long()[1:], lf.detach()[1:].argmax(1))
    want = lf.grad.to(BF)
    err = (d[:, :vocab].float().cpu() - want.float()).abs().max().item()
    assert err <= 2 ** -8 * want.float().abs().max().item() + 1e-12, err
    l2, a2, d2 = ops.token_ce(dl, tgt.to(dev), vocab=vocab, grad_scale=scale, inplace_grad=True)     # gradient overwrites the logits
    assert d2.data_ptr() == dl.data_ptr() and torch.equal(d2[:, :vocab], d[:, :vocab]) and torch.equal(l2, loss_rows) and torch.equal(a2, amax)
    if ld > vocab:
        assert torch.equal(dl[:, vocab:].cpu(), logits[:, vocab:])                                 # columns past the vocabulary are untouched


@pytest.mark.parametrize("D,rows,adim,mse,train", [(256, 24, 7, False, True), (256, 8, 7, True, False), (4096, 64, 7, False, True), (4096, 8, 7, False, False),
                                                   (1024, 56, 14, True, True)])
def test_fused_head_tail_is_bit_identical_to_the_unfused_sequence(dev, D, rows, adim, mse, train):
    """ovla_head_tail_fwd (two MLPResNet blocks -> LayerNorm 2 -> fc2 -> L1 / MSE loss in ONE launch, grid-wide barriers between the stages)
    against the unfused kernel sequence the engine ran before (ovla_norm_fwd, ovla_gemm_bf16 split_k = 2 + reduce epilogue, ovla_head_out_fwd):
    predictions, loss and EVERY tensor saved for the backward are bit-identical; and against plain torch fp32 within bf16 tolerance."""
    import dataclasses
    import importlib
    import os

    load = importlib.import_module
    engine_mod, config_mod = load("openvla-oft_amd.engine"), load("openvla-oft_amd.config")
    cfg = dataclasses.replace(config_mod.OPENVLA_7B, llm_dim=D, action_dim=adim, chunk=rows)        # one "sample" of `rows` chunk steps
    g = torch.Generator().manual_seed(D + rows)
    sd = {}
    hp = "action_head.model."
    for nm, dim in (("layer_norm1", D * adim), ("layer_norm2", D), ("mlp_resnet_blocks.0.ffn.0", D), ("mlp_resnet_blocks.1.ffn.0", D)):
        sd[hp + nm + ".weight"], sd[hp + nm + ".bias"] = 1 + 0.1 * torch.randn(dim, generator=g), 0.05 * torch.randn(dim, generator=g)
    for nm, o, i in (("fc1", D, D * adim), ("mlp_resnet_blocks.0.ffn.1", D, D), ("mlp_resnet_blocks.1.ffn.1", D, D), ("fc2", adim, D)):
        sd[hp + nm + ".weight"], sd[hp + nm + ".bias"] = torch.randn(o, i, generator=g) * (1.0 / i ** 0.5), 0.02 * torch.randn(o, generator=g)
    get = lambda n: sd[n].to(dev, BF)  # noqa: E731
    head = engine_mod.build_component(engine_mod.ActionHead, dev, get, hp, cfg=cfg)
    ah = (torch.randn(rows * adim, D, generator=g) * 0.7).to(dev, BF)
    tgt = (torch.rand(rows, adim, generator=g) * 2 - 1).to(dev, BF)
    outs = {}
    for fused in (True, False):
        engine_mod._FUSE_HEAD = fused
        try:
            pred, loss_sum, saved = head.fwd(ah, target=tgt, mse=mse, train=train)
        finally:
            engine_mod._FUSE_HEAD = {"0": False, "1": True}.get(os.environ.get("OVLA_FUSE_HEAD", "auto"), None)
        torch.cuda.synchronize()
        outs[fused] = (pred.clone(), loss_sum.clone(), saved)
    assert int(head._sync[1].item()) == 0, "a grid barrier of the fused kernel timed out"
    (pf, lf, sf), (pu, lu, su) = outs[True], outs[False]
    assert torch.equal(pf, pu), f"pred: {(pf != pu).sum().item()} of {pf.numel()} differ"
    assert abs(lf.item() - lu.item()) <= 1e-6 * max(1.0, abs(lu.item())), "loss: same addends, atomic order only"
    if train:
        names = ["x0", "m0", "r0", "z1", "s1", "blocks", "x_last", "m2", "r2", "h2", "pred"]
        for nm, a, b in zip(names, sf, su):
            if nm == "blocks":
                for bi, (ba, bb) in enumerate(zip(a, b)):
                    for k, (ta, tb) in enumerate(zip(ba[:4] + ba[4], bb[:4] + bb[4])):
                        assert torch.equal(ta, tb), f"block {bi} saved tensor {k}: {(ta != tb).sum().item()} differ"
            elif torch.is_tensor(a):
                assert torch.equal(a, b), f"{nm}: {(a != b).sum().item()} of {a.numel()} differ"
        # and the backward runs off the fused kernel's saved tensors
        head.store.zero_grad()
        d1 = head.bwd(sf).clone()
        g1 = {k: v.clone() for k, v in head.store.flat_grad.items()}
        head.store.zero_grad()
        d2 = head.bwd(su)
        assert torch.equal(d1, d2) and all(torch.allclose(g1[k], head.store.flat_grad[k], rtol=1e-5, atol=1e-7) for k in g1)
    # plain torch fp32 reference of the same head
    x = ah.float().cpu().view(rows, adim * D)
    F = torch.nn.functional
    w = lambda n: sd[hp + n].to(BF).float()  # noqa: E731
    h = F.relu(F.linear(F.layer_norm(x, (adim * D,), w("layer_norm1.weight"), w("layer_norm1.bias"), 1e-5), w("fc1.weight"), w("fc1.bias")))
    for b in range(2):
        q = f"mlp_resnet_blocks.{b}.ffn."
        h = h + F.relu(F.linear(F.layer_norm(h, (D,), w(q + "0.weight"), w(q + "0.bias"), 1e-5), w(q + "1.weight"), w(q + "1.bias")))
    ref = F.linear(F.layer_norm(h, (D,), w("layer_norm2.weight"), w("layer_norm2.bias"), 1e-5), w("fc2.weight"), w("fc2.bias"))
    err = (pf.float().cpu() - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,gn,G", [(300, 264, 1), (1216, 512, 3), (4176, 1152, 1), (1000, 4304, 2)])
def test_lora_bwd_one_pass(ops, dev, M, gn, G):
    """ovla_lora_bwd: dt = bf16(s * dy_g . B_g) and dB_g += dy_g^T t_g from ONE staging of dy (peft LoRA backward, finetune.py:862-871), against
    torch fp32 on the bf16-exact operands: ragged row counts, a column count that is no multiple of the 256-column chunk, fused groups, an
    existing dB that is accumulated into; dt bit-identical between two runs (its chunk partials are added in a fixed order)."""
    r, s = 32, 0.5
    dy, t = rnd(M, G * gn, dev=dev), rnd(M, G * r, dev=dev)
    Bt = rnd(G * r, gn, dev=dev, scale=0.3)
    dB0 = torch.randn(G * gn, r, device=dev)
    dB = dB0.clone()
    dt = ops.lora_bwd(dy, Bt, t, dB, gn=gn, G=G, scale=s)
    for g in range(G):
        dyg, tg, Btg = dy[:, g * gn:(g + 1) * gn].float(), t[:, g * r:(g + 1) * r].float(), Bt[g * r:(g + 1) * r].float()
        close(dt[:, g * r:(g + 1) * r], s * (dyg @ Btg.T), what=f"dt group {g}")
        ref = dyg.T @ tg
        err = ((dB[g * gn:(g + 1) * gn] - dB0[g * gn:(g + 1) * gn]) - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, f"dB group {g}: rel err {err:.3e} (fp32 accumulation of exact bf16 products)"
    dB2 = dB0.clone()
    dt2 = ops.lora_bwd(dy, Bt, t, dB2, gn=gn, G=G, scale=s)
    assert torch.equal(dt, dt2), "dt must be bit-reproducible (it feeds the data-gradient chain)"
    # and it equals what the two-kernel path computes, up to the bf16 rounding of differently ordered fp32 sums
    dt_old = ops.gemm(dy, Bt, alpha=s, a_group_n=r if G > 1 else 0) if (G == 1 or gn % 128 == 0) else None
    if dt_old is not None:
        assert (dt.float() - dt_old.float()).abs().max().item() <= 2.0 ** -7 * dt_old.float().abs().max().item()


@pytest.mark.parametrize("M,N,K,tile,rope", [(608, 1024, 512, 1, False), (608, 768, 1024, 1, True), (1000, 1024, 512, 101, False), (300, 640, 512, 101, True),
                                             (608, 1024, 512, 22, False), (608, 768, 1024, 22, True), (1000, 1024, 512, 122, False), (300, 768, 512, 122, True), (608, 1280, 4096, 22, True)])
def test_gemm_rmsnorm_fold(ops, dev, M, N, K, tile, rope):
    """RMSNorm folded around the 128x128 GEMM and the 4-wave 128x256 one (tiles 22 / 122) (ovla_gemm_args.rowsq_out / rowscale_part; LlamaStack.fold_norms): the producer's epilogue writes the
    sums of squares of its bf16 output's 64-column groups; the consumer scales its accumulator by rstd[m] = rsqrt(sum / K + eps) before the
    (RoPE) epilogue -- against torch fp32 on the same bf16 operands, interior and edge row tiles, in-kernel epilogue and hybrid-remainder reduce,
    and bit-reproducible between runs (slots are summed in a fixed order)."""
    torch.manual_seed(M + N + tile)
    eps = 1e-5
    # producer: x = a . b^T + residual, with per-64-column sums of squares of the stored x
    a, b, res = rnd(M, 512, dev=dev, scale=0.5), rnd(K, 512, dev=dev, scale=0.1), rnd(M, K, dev=dev)
    part = torch.full((M, K // 64), float("nan"), device=dev)
    x = ops.gemm(a, b, residual=res, tile=tile, rowsq_out=part)
    ref_x = ops.gemm(a, b, residual=res, tile=tile)
    assert torch.equal(x, ref_x), "the sums of squares ride along: the output itself is unchanged"
    ref_part = x.float().view(M, K // 64, 64).pow(2).sum(-1)
    assert torch.isfinite(part).all() and ((part - ref_part).abs() <= 1e-5 * ref_part.abs().max()).all()
    assert torch.allclose(ops.row_sumsq(x), ref_part, rtol=1e-5, atol=1e-5 * ref_part.abs().max().item())
    # consumer: y = rope(rstd[m] * (x . Wn^T))
    wn = rnd(N, K, dev=dev, scale=0.05)
    rbuf = torch.empty(M, device=dev)
    S = 152 if rope else 0
    kw = {}
    if rope:
        cos, sin = ops.rope_table(S, 128, 10000.0, dev)
        kw["rope"] = (cos, sin, S, N - 256)          # leading heads rotated, the last two head widths are "v"
    y = ops.gemm(x, wn, tile=tile, rowscale=(part, eps, rbuf), **kw)
    rstd = torch.rsqrt(part.sum(-1) / K + eps)
    assert torch.allclose(rbuf, rstd, rtol=2e-6, atol=0)
    ref = ((x.float() @ wn.float().T) * rstd[:, None]).to(BF)
    if rope:
        ops.rope_(ref, S, (N - 256) // 128, 128, cos, sin)
    close(y, ref, tol=1.2e-2, what="folded RMSNorm GEMM")
    y2 = ops.gemm(x, wn, tile=tile, rowscale=(part, eps, torch.empty(M, device=dev)), **kw)
    assert torch.equal(y, y2)
    # any other tile configuration must refuse the fold instead of ignoring it
    with pytest.raises(RuntimeError):
        ops.gemm(x, wn, tile=17, rowscale=(part, eps, rbuf))


@pytest.mark.parametrize("M,N,K,tile", [(4864, 4096, 1024, 117), (608, 4096, 2048, 101), (608, 1024, 4096, 102), (1000, 768, 512, 101), (4864, 1024, 512, 0)])
def test_hybrid_reduce_in_launch_equals_the_reduce_kernel(ops, dev, M, N, K, tile):
    """The hybrid schedule's K-split remainder tiles reduced INSIDE the GEMM launch (the last-arriving K part of a tile sums the slabs in slab
    order and runs the epilogue; ovla_gemm_args.hybrid_counters) against the separate gemm_hybrid_reduce launch: bit-identical outputs incl. bias /
    residual epilogues and the LoRA K-extension, repeated launches (the counters must be all zero again after every launch)."""
    torch.manual_seed(M + N)
    a, b = rnd(M, K, dev=dev, scale=0.5), rnd(N, K, dev=dev, scale=0.1)
    t, lb, res, bias = rnd(M, 32, dev=dev), rnd(N, 32, dev=dev, scale=0.2), rnd(M, N, dev=dev), rnd(N, dev=dev)
    kw = dict(a2=t, b2=lb, residual=res, bias=bias, tile=tile)
    prev = ops._HYB_INLAUNCH
    try:
        ops._HYB_INLAUNCH = False
        ref = ops.gemm(a, b, **kw)
        ops._HYB_INLAUNCH = True
        outs = [ops.gemm(a, b, **kw) for _ in range(3)]
        torch.cuda.synchronize()
        assert int(ops._hybrid_counters(dev).abs().sum().item()) == 0, "arrival counters must be zero between launches"
    finally:
        ops._HYB_INLAUNCH = prev
    for o in outs:
        assert torch.equal(o, ref), f"{(o != ref).sum().item()} of {ref.numel()} differ"
    close(ref, a.float() @ b.float().T + t.float() @ lb.float().T + bias.float() + res.float(), tol=2e-2, what="hybrid GEMM vs torch")
