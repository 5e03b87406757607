"""Tensor-level wrappers over the C-ABI (include/ovla.h).  PyTorch is used here only as the owner of device memory
and of the HIP stream; every arithmetic op below is a hand-written gfx950 kernel in libovla_hip.so.  Nothing in this
module falls back to torch math: a missing library or a failed launch raises.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import STRUCTS

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU, ACT_GELU_TANH, ACT_SWIGLU = 0, 1, 2, 3, 4, 5
BF16 = torch.bfloat16

# bench.py sets this to a list to time individual launches with HIP events on the launch stream:
# entries are (kernel_family, start_event, end_event, algorithmic_flops)
PROFILE = None


_WS_BYTES = 96 << 20   # persistent fp32 scratch per device for split-K / hybrid-schedule partial sums (stream-ordered reuse)
_ws_cache = {}
_ws_retired = []       # outgrown workspaces stay allocated: a captured hipGraph (engine.ChunkGraph) has their raw pointers in its kernel arguments


def _workspace(device, nbytes):
    """One scratch buffer per (device, stream): launches on one stream reuse it in stream order; the two vision towers run
    on different streams and must not share partial-sum slabs.  A buffer that has been handed out is never freed: when a call needs more
    than the current one holds, a larger one takes its place for LATER launches and the old one is retired, not released -- graph replays
    recorded earlier keep writing into memory they still own."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(max(nbytes, _WS_BYTES) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = ws
    return ws


# Arrival counters of the in-launch hybrid reduce (ovla_gemm_args.hybrid_counters): one zeroed array per (device, stream), which the kernels
# keep all-zero.  OPT-IN (OVLA_HYBRID_INLAUNCH=1): bit-identical to the separate gemm_hybrid_reduce launch but measured SLOWER on every hybrid
# shape (tools/hybrid_ab.py, profiles/r03_hybrid_inlaunch_ab.txt: 4864 x 4096 x 4096 214 vs 160 us, 608 x 4096 x 4096 67.5 vs 38.5 us) -- the
# agent-scope release each K-part workgroup needs before it signals writes back its XCD's whole L2 (the other tiles' output lines included),
# which costs far more than the 15 us launch it removes.
import os as _os

_HYB_INLAUNCH = _os.environ.get("OVLA_HYBRID_INLAUNCH", "0") == "1"
_HYB_N = 4096
_hyb_cache = {}


def _hybrid_counters(device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    c = _hyb_cache.get(key)
    if c is None:
        c = _hyb_cache[key] = torch.zeros(_HYB_N, dtype=torch.int32, device=device)
    return c


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, family, flops):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((family, e0, e1, flops))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t) -> int:
    return 0 if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype=BF16, name="tensor"):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")


def check_device(index: int = 0) -> None:
    _lib.check(_lib.lib().ovla_check_device(index), "ovla_check_device")


# ----------------------------------------------------------------------------------------------------------------------
def gemm(a, b, *, out=None, bias=None, act=ACT_NONE, residual=None, colscale=None, c_pre=None, a2=None, b2=None,
         k2_group_n=0, film=None, split_k=1, tile=0, alpha=1.0, a_group_n=0, dact=None, rope=None, rowsq_out=None, rowscale=None):
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T (+ a2[M,G*K2] @ b2[N,K2]^T)); all bf16 2-D, last dim contiguous."""
    _chk(a, name="a"); _chk(b, name="b")
    M, K = a.shape
    N = b.shape[0]
    if a_group_n:   # block-diagonal: a is [M, G*K], b is [G*a_group_n, K]
        K = b.shape[1]
        assert a.shape[1] == (N // a_group_n) * K, f"gemm(block-diagonal): {a.shape} vs {b.shape}"
    assert b.shape[1] == K, f"gemm: K mismatch {a.shape} x {b.shape}"
    assert a.stride(1) == 1 and b.stride(1) == 1
    out_cols = 2 * N if (dact is not None and dact[0] == "swiglu") else (N // 2 if act == ACT_SWIGLU else N)   # ACT_SWIGLU: b = [gate; up], out = silu(gate) * up
    if out is None:
        out = torch.empty((M, out_cols), dtype=BF16, device=a.device)
    assert out.shape == (M, out_cols) and out.stride(1) == 1
    g = STRUCTS["ovla_gemm_args"]()
    g.A, g.lda, g.B, g.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
    g.C, g.ldc = out.data_ptr(), out.stride(0)
    if a2 is not None:
        assert b2 is not None and a2.stride(1) == 1 and b2.stride(1) == 1 and b2.shape[0] == N and a2.shape[0] == M
        g.A2, g.lda2, g.B2, g.ldb2, g.K2 = a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), b2.shape[1]
        g.k2_group_n = k2_group_n
    if c_pre is not None:
        assert c_pre.shape == (M, N) and c_pre.stride(0) == (N if act == ACT_SWIGLU else out.stride(0))   # ACT_SWIGLU: c_pre = the contiguous [M, 2F] projection output
        g.C_pre = c_pre.data_ptr()
    if bias is not None:
        assert bias.numel() == N
        g.bias = bias.data_ptr()
    if colscale is not None:
        assert colscale.numel() == N
        g.colscale = colscale.data_ptr()
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
    if film is not None:
        gamma, beta, rows = film
        assert gamma.shape[-1] == N and gamma.is_contiguous() and beta.is_contiguous()
        g.film_gamma, g.film_beta, g.film_rows = gamma.data_ptr(), beta.data_ptr(), rows
    if dact is not None:   # backward epilogue: ("act", z, act_id) or ("swiglu", gu)
        src = dact[1]
        assert src.stride(1) == 1 and src.shape == (M, out_cols) and src.dtype == BF16
        g.dact_src, g.ld_dact = src.data_ptr(), src.stride(0)
        g.dact_mode, g.dact_act = (2, ACT_SILU) if dact[0] == "swiglu" else (1, dact[2])
    if rope is not None:   # (cos, sin, S, cols): RoPE on output columns [0, cols) (q | k heads of a fused q|k|v projection), head_dim 128
        cos, sin, rS, rcols = rope
        assert cos.shape[1] == 64 and cos.shape[0] >= rS and cos.is_contiguous() and sin.is_contiguous()
        g.rope_cos, g.rope_sin, g.rope_S, g.rope_cols = cos.data_ptr(), sin.data_ptr(), rS, rcols
    if rowsq_out is not None:   # RMSNorm fold, producer side: fp32 [M, N / 64] sums of squares of the output's 64-column groups
        assert rowsq_out.dtype == torch.float32 and rowsq_out.shape == (M, N // 64) and rowsq_out.is_contiguous()
        g.rowsq_out = rowsq_out.data_ptr()
    if rowscale is not None:    # consumer side: (parts fp32 [M, K / 64], eps, rstd scratch fp32 [M]): C = epilogue(rstd[m] * alpha * acc)
        parts, eps, rbuf = rowscale
        assert parts.dtype == torch.float32 and parts.shape == (M, K // 64) and parts.is_contiguous() and rbuf.dtype == torch.float32 and rbuf.numel() >= M
        g.rowscale_part, g.rowscale_slots, g.rowscale_eps, g.rowscale_r = parts.data_ptr(), K // 64, eps, rbuf.data_ptr()
    g.M, g.N, g.K, g.act, g.split_k, g.tile, g.alpha, g.a_group_n = M, N, K, act, split_k, tile, alpha, a_group_n
    ws = _workspace(a.device, max(4 * split_k * M * N if split_k > 1 else 0, _WS_BYTES))
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if _HYB_INLAUNCH:
        g.hybrid_counters, g.n_hybrid_counters = _hybrid_counters(a.device).data_ptr(), _HYB_N
    e0 = _prof_begin()
    _lib.call("ovla_gemm_bf16", g, _stream())
    if e0 is not None:
        _prof_end(e0, _gemm_family(g), 2.0 * M * N * (K + (b2.shape[1] if b2 is not None else 0)))
    return out


def _gemm_family(g):
    """Profiling only: which kernel instance `ovla_gemm_bf16` runs for these arguments ("gemm_nt_t17" = gemm_nt_kernel<256,256,2,4>, "gemm_nt_t18" = the
    4-wave gemm_nt_w4_kernel, ...), from the library's own decision code (ovla_gemm_resolved_tile) -- so bench.py can report the dominant instance by itself."""
    import ctypes

    t = ctypes.c_int32()
    _lib.check(_lib.lib().ovla_gemm_resolved_tile(ctypes.byref(g), ctypes.byref(t)), "ovla_gemm_resolved_tile")
    tile = t.value % 100 if t.value >= 100 else t.value
    return f"gemm_nt_t18k{g.K2 // 32}" if tile == 18 else f"gemm_nt_t{tile}"   # the 4-wave kernel is one instantiation per K-extension width


def gemm_tn(x, y, *, out=None, alpha=1.0, accumulate=True, out_dtype=torch.float32):
    """out[P,Q] (+)= alpha * x[M,P]^T @ y[M,Q].  accumulate=True: fp32 atomic accumulation into `out`."""
    _chk(x, name="x"); _chk(y, name="y")
    M, P = x.shape
    Q = y.shape[1]
    assert y.shape[0] == M and x.stride(1) == 1 and y.stride(1) == 1
    if out is None:
        out = torch.zeros((P, Q), dtype=out_dtype, device=x.device) if accumulate else torch.empty((P, Q), dtype=out_dtype, device=x.device)
    assert out.shape == (P, Q) and out.stride(1) == 1
    g = STRUCTS["ovla_gemm_tn_args"]()
    g.X, g.ldx, g.Y, g.ldy, g.C, g.ldc = x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0)
    g.M, g.P, g.Q, g.alpha = M, P, Q, alpha
    if accumulate:
        assert out.dtype == torch.float32
        g.out_mode = 0
    else:
        g.out_mode = 1 if out.dtype == torch.float32 else 2
    e0 = _prof_begin()
    _lib.call("ovla_gemm_tn_bf16", g, _stream())
    _prof_end(e0, "gemm_tn", 2.0 * M * P * Q)
    return out


def gemm_tn_grouped(problems):
    """problems: list of (x, y, out_fp32) with out accumulated (+=).  One launch for up to 4 TN GEMMs."""
    import ctypes

    n = len(problems)
    arr = (STRUCTS["ovla_gemm_tn_args"] * n)()
    flops = 0.0
    for i, (x, y, out) in enumerate(problems):
        _chk(x, name="x"); _chk(y, name="y")
        M, P = x.shape
        Q = y.shape[1]
        assert y.shape[0] == M and x.stride(1) == 1 and y.stride(1) == 1 and out.shape == (P, Q) and out.dtype == torch.float32
        g = arr[i]
        g.X, g.ldx, g.Y, g.ldy, g.C, g.ldc = x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0)
        g.M, g.P, g.Q, g.alpha, g.out_mode = M, P, Q, 1.0, 0
        flops += 2.0 * M * P * Q
    e0 = _prof_begin()
    _lib.check(_lib.lib().ovla_gemm_tn_grouped(arr, n, _stream()), "ovla_gemm_tn_grouped")
    _prof_end(e0, "gemm_tn", flops)


def row_sumsq(x, out=None):
    """out[m, j] = sum of squares of x[m, 64 j : 64 j + 64] (fp32 [rows, dim / 64]): the first decoder layer's RMSNorm-fold input."""
    _chk(x, name="x")
    rows, dim = x.shape
    assert x.stride(1) == 1 and dim % 64 == 0
    if out is None:
        out = torch.empty((rows, dim // 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ovla_row_sumsq(x.data_ptr(), x.stride(0), out.data_ptr(), rows, dim, _stream()), "ovla_row_sumsq")
    return out


def gemm_plan(M, N, K, K2=0, k2_group_n=0, workspace_bytes=_WS_BYTES):
    """The schedule `tile = 0` would pick (ovla_gemm_plan, host only): (tile id, full tiles, remainder tiles, remainder splits, estimated seconds)."""
    import ctypes

    t, f, r, sp, est = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
    _lib.check(_lib.lib().ovla_gemm_plan(M, N, K, K2, k2_group_n, ctypes.c_int64(workspace_bytes), ctypes.byref(t), ctypes.byref(f), ctypes.byref(r),
                                         ctypes.byref(sp), ctypes.byref(est)), "ovla_gemm_plan")
    return t.value, f.value, r.value, sp.value, est.value


def lora_bwd(dy, Bt, t, dB_grad, *, gn, G, scale, dt=None):
    """One pass over dy for a LoRA linear's backward (ovla.h: ovla_lora_bwd): returns dt [M, G*r] bf16 = scale * dy_g . B_g per group and
    accumulates dB_grad [G*gn, r] fp32 += dy_g^T . t_g.  Bt: [G*r, gn] (B_g^T stacked), t: [M, G*r] (the forward's saved scale * x . A^T)."""
    _chk(dy, name="dy"); _chk(Bt, name="Bt"); _chk(t, name="t")
    M = dy.shape[0]
    r = Bt.shape[0] // G
    assert dy.shape[1] == G * gn and Bt.shape == (G * r, gn) and t.shape == (M, G * r) and dB_grad.shape == (G * gn, r) and dB_grad.dtype == torch.float32
    assert dy.stride(1) == 1 and Bt.stride(1) == 1 and t.stride(1) == 1 and dB_grad.stride(1) == 1
    if dt is None:
        dt = torch.empty((M, G * r), dtype=BF16, device=dy.device)
    need = _lib.lib().ovla_lora_bwd_workspace_bytes(M, gn, G)
    ws = _workspace(dy.device, need)
    g = STRUCTS["ovla_lora_bwd_args"]()
    g.dy, g.ld_dy, g.Bt, g.ld_bt, g.t, g.ld_t = dy.data_ptr(), dy.stride(0), Bt.data_ptr(), Bt.stride(0), t.data_ptr(), t.stride(0)
    g.dt, g.ld_dt, g.dB, g.ld_db = dt.data_ptr(), dt.stride(0), dB_grad.data_ptr(), dB_grad.stride(0)
    g.M, g.gn, g.G, g.r, g.scale = M, gn, G, r, scale
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    e0 = _prof_begin()
    _lib.call("ovla_lora_bwd", g, _stream())
    _prof_end(e0, "lora_bwd", 4.0 * M * G * gn * r)
    return dt


def transpose_table(pairs, device):
    """Builds the device descriptor table for transpose_batched from [(src, dst)] (2-D bf16 tensors that never move)."""
    import ctypes

    T = STRUCTS["ovla_transpose_args"]
    arr = (T * len(pairs))()
    starts = [0]
    for i, (src, dst) in enumerate(pairs):
        rows, cols = src.shape
        assert dst.shape == (cols, rows) and src.stride(1) == 1 and dst.stride(1) == 1
        arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].lds, arr[i].ldd = src.data_ptr(), dst.data_ptr(), rows, cols, src.stride(0), dst.stride(0)
        starts.append(starts[-1] + ((rows + 63) // 64) * ((cols + 63) // 64))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    tile_start = torch.tensor(starts, dtype=torch.int32).to(device)
    return dict(table=raw, tile_start=tile_start, n=len(pairs), total=starts[-1], keep=pairs)


def transpose_batched(tab):
    import ctypes

    ptr = ctypes.cast(ctypes.c_void_p(tab["table"].data_ptr()), ctypes.POINTER(STRUCTS["ovla_transpose_args"]))
    _lib.check(_lib.lib().ovla_transpose_batched(ptr, tab["tile_start"].data_ptr(), tab["n"], tab["total"], _stream()), "ovla_transpose_batched")


def colsum(x, out):
    """out[n] += sum_m x[m,n]  (fp32 accumulate)."""
    _chk(x); _chk(out, torch.float32)
    g = STRUCTS["ovla_colsum_args"]()
    g.X, g.ldx, g.out, g.M, g.N = x.data_ptr(), x.stride(0), out.data_ptr(), x.shape[0], x.shape[1]
    _lib.call("ovla_colsum_bf16", g, _stream())
    return out


# ----------------------------------------------------------------------------------------------------------------------
def attn_fwd(q, k, v, B, S, H, hd, *, kv_len=None, causal=False, scale=None, out=None):
    """q/k/v: 2-D views [B*S, >=H*hd] (row stride arbitrary, heads contiguous).  Returns (out [B*S, H*hd], lse [B,H,S])."""
    for t in (q, k, v):
        _chk(t)
        assert t.stride(1) == 1 and t.shape[0] == B * S
    if out is None:
        out = torch.empty((B * S, H * hd), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    g = STRUCTS["ovla_attn_fwd_args"]()
    g.Q, g.K, g.V, g.q_stride, g.k_stride, g.v_stride = q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0)
    g.O, g.o_stride, g.lse, g.kv_len = out.data_ptr(), out.stride(0), lse.data_ptr(), _p(kv_len)
    g.B, g.H, g.S, g.head_dim, g.causal = B, H, S, hd, int(causal)
    g.scale = float(scale if scale is not None else hd ** -0.5)
    e0 = _prof_begin()
    _lib.call("ovla_attn_fwd", g, _stream())
    _prof_end(e0, "attn_fwd", 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0))
    return out, lse


def attn_bwd(q, k, v, o, do, lse, B, S, H, hd, *, kv_len=None, causal=False, scale=None, dq=None, dk=None, dv=None, rope=None):
    """`rope=(cos, sin)` (ovla_rope_table): dq / dk come out with the inverse RoPE rotation applied (gradients w.r.t. the pre-RoPE q / k)."""
    for t in (q, k, v, o, do):
        _chk(t)
        assert t.stride(1) == 1
    dev = q.device
    if dq is None:
        dq = torch.empty((B * S, H * hd), dtype=BF16, device=dev)
    if dk is None:
        dk = torch.empty((B * S, H * hd), dtype=BF16, device=dev)
    if dv is None:
        dv = torch.empty((B * S, H * hd), dtype=BF16, device=dev)
    delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    g = STRUCTS["ovla_attn_bwd_args"]()
    g.Q, g.K, g.V, g.q_stride, g.k_stride, g.v_stride = q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0)
    g.O, g.dO, g.o_stride, g.do_stride = o.data_ptr(), do.data_ptr(), o.stride(0), do.stride(0)
    g.lse, g.delta = lse.data_ptr(), delta.data_ptr()
    g.dQ, g.dK, g.dV, g.dq_stride, g.dk_stride, g.dv_stride = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dq.stride(0), dk.stride(0), dv.stride(0)
    g.kv_len = _p(kv_len)
    g.B, g.H, g.S, g.head_dim, g.causal = B, H, S, hd, int(causal)
    g.scale = float(scale if scale is not None else hd ** -0.5)
    if rope is not None:
        assert rope[0].shape[0] >= S and rope[0].shape[1] == hd // 2 and rope[0].is_contiguous() and rope[1].is_contiguous()
        g.rope_cos, g.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
    e0 = _prof_begin()
    _lib.call("ovla_attn_bwd", g, _stream())
    _prof_end(e0, "attn_bwd", 10.0 * B * H * S * S * hd * (0.5 if causal else 1.0))
    return dq, dk, dv


# ----------------------------------------------------------------------------------------------------------------------
def norm_fwd(x, weight, bias=None, *, eps, rms, out=None, save_stats=True):
    _chk(x); _chk(weight)
    rows, dim = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if (save_stats and not rms) else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    g = STRUCTS["ovla_norm_fwd_args"]()
    g.x, g.y, g.weight, g.bias, g.mean, g.rstd = x.data_ptr(), out.data_ptr(), weight.data_ptr(), _p(bias), _p(mean), _p(rstd)
    g.rows, g.dim, g.eps, g.is_rms = rows, dim, eps, int(rms)
    _lib.call("ovla_norm_fwd", g, _stream())
    return out, mean, rstd


def norm_bwd(x, dy, weight, mean, rstd, *, rms, dx=None, dx_accum=False, dweight=None, dbias=None):
    _chk(x); _chk(dy)
    rows, dim = x.shape
    assert x.is_contiguous() and dy.is_contiguous()
    if dx is None:
        assert not dx_accum
        dx = torch.empty_like(x)
    g = STRUCTS["ovla_norm_bwd_args"]()
    g.x, g.dy, g.weight, g.mean, g.rstd, g.dx = x.data_ptr(), dy.data_ptr(), weight.data_ptr(), _p(mean), rstd.data_ptr(), dx.data_ptr()
    g.dweight, g.dbias, g.rows, g.dim, g.is_rms, g.dx_accum = _p(dweight), _p(dbias), rows, dim, int(rms), int(dx_accum)
    _lib.call("ovla_norm_bwd", g, _stream())
    return dx


def rope_table(S, hd, theta, device):
    cos = torch.empty((S, hd // 2), dtype=BF16, device=device)
    sin = torch.empty((S, hd // 2), dtype=BF16, device=device)
    _lib.check(_lib.lib().ovla_rope_table(cos.data_ptr(), sin.data_ptr(), S, hd, theta, _stream()), "ovla_rope_table")
    return cos, sin


def rope_(qk, S, n_heads, hd, cos, sin, inverse=False):
    """In place on the first n_heads*hd columns of qk [rows, ld]."""
    _chk(qk)
    g = STRUCTS["ovla_rope_args"]()
    g.qk, g.ld, g.rows, g.S, g.n_heads, g.head_dim = qk.data_ptr(), qk.stride(0), qk.shape[0], S, n_heads, hd
    g.cos_table, g.sin_table, g.inverse = cos.data_ptr(), sin.data_ptr(), int(inverse)
    _lib.call("ovla_rope", g, _stream())
    return qk


def swiglu_fwd(gu):
    _chk(gu)
    rows, F2 = gu.shape
    h = torch.empty((rows, F2 // 2), dtype=BF16, device=gu.device)
    g = STRUCTS["ovla_swiglu_fwd_args"]()
    g.gu, g.h, g.rows, g.F = gu.data_ptr(), h.data_ptr(), rows, F2 // 2
    _lib.call("ovla_swiglu_fwd", g, _stream())
    return h


def swiglu_bwd(gu, dh):
    _chk(gu); _chk(dh)
    rows, F2 = gu.shape
    dgu = torch.empty_like(gu)
    g = STRUCTS["ovla_swiglu_bwd_args"]()
    g.gu, g.dh, g.dgu, g.rows, g.F = gu.data_ptr(), dh.data_ptr(), dgu.data_ptr(), rows, F2 // 2
    _lib.call("ovla_swiglu_bwd", g, _stream())
    return dgu


def act_bwd(z, dh, act):
    _chk(z); _chk(dh)
    assert z.is_contiguous() and dh.is_contiguous()
    dz = torch.empty_like(z)
    g = STRUCTS["ovla_act_bwd_args"]()
    g.z, g.dh, g.dz, g.n, g.act = z.data_ptr(), dh.data_ptr(), dz.data_ptr(), z.numel(), act
    _lib.call("ovla_act_bwd", g, _stream())
    return dz


def add(a, b=None, out=None):
    _chk(a)
    if out is None:
        out = torch.empty_like(a)
    g = STRUCTS["ovla_add_args"]()
    g.a, g.b, g.out, g.n = a.data_ptr(), _p(b), out.data_ptr(), a.numel()
    _lib.call("ovla_add_bf16", g, _stream())
    return out


def colscale(x, scale, out=None):
    _chk(x)
    if out is None:
        out = torch.empty_like(x)
    g = STRUCTS["ovla_colscale_args"]()
    g.x, g.scale, g.out, g.rows, g.dim = x.data_ptr(), scale.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1]
    _lib.call("ovla_colscale_bf16", g, _stream())
    return out


# ----------------------------------------------------------------------------------------------------------------------
def im2col(pixels, c0, patch, k_padded, n_img=1, img_cstride=6):
    _chk(pixels)
    B, C, H, W = pixels.shape
    assert pixels.is_contiguous()
    out = torch.empty((B * n_img * (H // patch) * (W // patch), k_padded), dtype=BF16, device=pixels.device)
    g = STRUCTS["ovla_im2col_args"]()
    g.pixels, g.out, g.ldo, g.B, g.C_total, g.c0, g.H, g.W, g.patch = pixels.data_ptr(), out.data_ptr(), k_padded, B, C, c0, H, W, patch
    g.n_img, g.img_cstride = n_img, img_cstride
    _lib.call("ovla_im2col", g, _stream())
    return out


def vit_embed(patches, pos, prefix, B, n_patches, dim):
    n_prefix = 0 if prefix is None else prefix.shape[0]
    tokens = torch.empty((B * (n_patches + n_prefix), dim), dtype=BF16, device=patches.device)
    g = STRUCTS["ovla_vit_embed_args"]()
    g.patches, g.pos, g.prefix, g.tokens = patches.data_ptr(), pos.data_ptr(), _p(prefix), tokens.data_ptr()
    g.B, g.n_patches, g.n_prefix, g.dim = B, n_patches, n_prefix, dim
    _lib.call("ovla_vit_embed", g, _stream())
    return tokens


def copy_rows(src, dst, B, rows, dim, *, src_batch_stride, src_row0, src_ld, dst_batch_stride, dst_row0, dst_ld, dst_col0=0,
              accumulate=False):
    g = STRUCTS["ovla_copy_rows_args"]()
    g.src, g.dst, g.B, g.rows, g.dim = src.data_ptr(), dst.data_ptr(), B, rows, dim
    g.src_batch_stride, g.src_row0, g.src_ld = src_batch_stride, src_row0, src_ld
    g.dst_batch_stride, g.dst_row0, g.dst_ld, g.dst_col0, g.accumulate = dst_batch_stride, dst_row0, dst_ld, dst_col0, int(accumulate)
    _lib.call("ovla_copy_rows", g, _stream())
    return dst


def film_bwd(dy, x_pre, gamma, dgamma, dbeta, B, rows_per_batch):
    """In place: dy <- dy * (1 + gamma[b]); accumulates dgamma / dbeta (fp32 [B, dim])."""
    g = STRUCTS["ovla_film_bwd_args"]()
    g.dy, g.x_pre, g.gamma, g.dgamma, g.dbeta = dy.data_ptr(), x_pre.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
    g.B, g.rows_per_batch, g.dim = B, rows_per_batch, dy.shape[1]
    _lib.call("ovla_film_bwd", g, _stream())
    return dy


def masked_mean(x, row_mask, B, L, dim):
    out = torch.empty((B, dim), dtype=BF16, device=x.device)
    g = STRUCTS["ovla_masked_mean_args"]()
    g.x, g.row_mask, g.out, g.B, g.L, g.dim = x.data_ptr(), row_mask.data_ptr(), out.data_ptr(), B, L, dim
    _lib.call("ovla_masked_mean", g, _stream())
    return out


def language_average(ids, labels, table, out, *, action_token_begin=31743):
    """out[:B] = mean of table[ids] over the non-action text positions (ovla.h: ovla_language_average); ids / labels int64 [B, L] on the device."""
    B, L = ids.shape
    assert ids.dtype == torch.int64 and labels.dtype == torch.int64 and ids.is_contiguous() and labels.is_contiguous() and ids.is_cuda and labels.is_cuda
    assert out.shape[0] >= B and out.shape[1] == table.shape[1] and out.is_contiguous()
    g = STRUCTS["ovla_language_average_args"]()
    g.ids, g.labels, g.embed_table, g.out = ids.data_ptr(), labels.data_ptr(), table.data_ptr(), out.data_ptr()
    g.B, g.L, g.D, g.vocab, g.action_token_begin = B, L, table.shape[1], table.shape[0], action_token_begin
    _lib.call("ovla_language_average", g, _stream())
    return out


def image_prep(images_u8, *, crop: bool, crop_scale: float = 0.9, out_size: int = 224,
               mean=(0.485, 0.456, 0.406, 0.5, 0.5, 0.5), std=(0.229, 0.224, 0.225, 0.5, 0.5, 0.5)):
    """uint8 [n_img, H, W, 3] (device) -> bf16 [1, 6 * n_img, out, out] pixel_values: center crop (TF crop_and_resize rule) +
    uint8 re-quantisation + DINOv2 / SigLIP normalisation in one launch (openvla_utils.py:542-622, processing_prismatic.py:128-145)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3 and images_u8.is_contiguous() and images_u8.is_cuda
    n, H, W, _ = images_u8.shape
    out = torch.empty((1, 6 * n, out_size, out_size), dtype=BF16, device=images_u8.device)
    g = STRUCTS["ovla_image_prep_args"]()
    g.src, g.dst, g.n_img, g.H, g.W, g.out, g.crop = images_u8.data_ptr(), out.data_ptr(), n, H, W, out_size, int(crop)
    g.crop_scale = float(crop_scale)
    for i in range(6):
        g.mean[i], g.std[i] = mean[i], std[i]
    _lib.call("ovla_image_prep", g, _stream())
    return out


def image_resize(images_u8, row_spans, col_spans):
    """uint8 [n, H, W, 3] (device) -> uint8 [n, out_h, out_w, 3]: separable span resampling (ovla.h: ovla_image_resize; TF's
    scale_and_translate kernel, rows first).  `*_spans` = (starts int32 [out], weights fp32 [out, span]) on the device
    (image_prep.lanczos3_spans)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3 and images_u8.is_contiguous() and images_u8.is_cuda
    n, H, W, _ = images_u8.shape
    (rs, rw), (cs, cw) = row_spans, col_spans
    for st, wt in ((rs, rw), (cs, cw)):
        assert st.dtype == torch.int32 and wt.dtype == torch.float32 and wt.dim() == 2 and wt.shape[0] == st.shape[0] and st.is_cuda and wt.is_cuda
        assert st.is_contiguous() and wt.is_contiguous()
    oh, ow = rs.shape[0], cs.shape[0]
    out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=images_u8.device)
    wsb = int(_lib.lib().ovla_image_resize_workspace_bytes(n, W, oh))
    ws = torch.empty(wsb, dtype=torch.uint8, device=images_u8.device)
    g = STRUCTS["ovla_image_resize_args"]()
    g.src, g.dst, g.workspace, g.workspace_bytes = images_u8.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb
    g.row_starts, g.row_weights, g.col_starts, g.col_weights = rs.data_ptr(), rw.data_ptr(), cs.data_ptr(), cw.data_ptr()
    g.n_img, g.H, g.W, g.out_h, g.out_w, g.row_span, g.col_span = n, H, W, oh, ow, rw.shape[1], cw.shape[1]
    _lib.call("ovla_image_resize", g, _stream())
    return out


def jpeg_roundtrip(images_u8, quality: int = 95):
    """uint8 [n, H, W, 3] (device) -> the same frames after a baseline 4:2:0 JPEG encode + decode at `quality` (ovla.h: ovla_jpeg_roundtrip;
    tf.image.encode_jpeg + tf.io.decode_image of experiments/robot/openvla_utils.py:532-533)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3 and images_u8.is_contiguous() and images_u8.is_cuda
    n, H, W, _ = images_u8.shape
    out = torch.empty_like(images_u8)
    wsb = int(_lib.lib().ovla_jpeg_roundtrip_workspace_bytes(n, H, W))
    ws = torch.empty(wsb, dtype=torch.uint8, device=images_u8.device)
    g = STRUCTS["ovla_jpeg_roundtrip_args"]()
    g.src, g.dst, g.workspace, g.workspace_bytes = images_u8.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb
    g.n_img, g.H, g.W, g.quality = n, H, W, int(quality)
    _lib.call("ovla_jpeg_roundtrip", g, _stream())
    return out


AUG_CROP, AUG_BRIGHTNESS, AUG_CONTRAST, AUG_SATURATION, AUG_HUE = 1, 2, 4, 8, 16
AUG_ALL = 31


def image_augment(images_u8, params, *, ops_mask: int = AUG_ALL, out_size: int = 224,
                  mean=(0.485, 0.456, 0.406, 0.5, 0.5, 0.5), std=(0.229, 0.224, 0.225, 0.5, 0.5, 0.5)):
    """Training-time frames: uint8 [n, H, W, 3] + fp32 params [n, 8] (both on the device) -> bf16 [n, 6, out, out]: dlimp's
    augment_image ops (crop-and-resize box, brightness, contrast, saturation, hue; `ops_mask` selects them), uint8 re-quantisation
    and both backbones' normalisations (ovla.h: ovla_image_augment).  ops_mask = 0 is the evaluation path."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3 and images_u8.is_contiguous() and images_u8.is_cuda
    n, H, W, _ = images_u8.shape
    assert params.dtype == torch.float32 and tuple(params.shape) == (n, 8) and params.is_contiguous() and params.device == images_u8.device
    out = torch.empty((n, 6, out_size, out_size), dtype=BF16, device=images_u8.device)
    wsb = int(_lib.lib().ovla_image_augment_workspace_bytes(n, out_size))
    ws = torch.empty(wsb, dtype=torch.uint8, device=images_u8.device)
    g = STRUCTS["ovla_image_augment_args"]()
    g.src, g.dst, g.params, g.workspace, g.workspace_bytes = images_u8.data_ptr(), out.data_ptr(), params.data_ptr(), ws.data_ptr(), wsb
    g.n_img, g.H, g.W, g.out, g.ops_mask = n, H, W, out_size, int(ops_mask)
    for i in range(6):
        g.mean[i], g.std[i] = mean[i], std[i]
    _lib.call("ovla_image_augment", g, _stream())
    return out


def assemble_multimodal(ids, labels, table, patches, *, A, noisy=None, ignore_index=-100, action_token_begin=31743, action_dim=7):
    """ids/labels int64 [B,L]; table bf16 [V,D]; patches bf16 [B,P,D] -> (out [B,P+L,D], action_rows int32 [B,A]:
    flattened row of the hidden state that predicts each action slot)."""
    B, L = ids.shape
    P, D = patches.shape[1], patches.shape[2]
    out = torch.empty((B, P + L, D), dtype=BF16, device=table.device)
    action_pos = torch.full((B, A), -1, dtype=torch.int32, device=table.device)
    g = STRUCTS["ovla_assemble_args"]()
    g.ids, g.labels, g.embed_table, g.patches, g.noisy = ids.data_ptr(), labels.data_ptr(), table.data_ptr(), patches.data_ptr(), _p(noisy)
    g.out, g.action_pos = out.data_ptr(), action_pos.data_ptr()
    g.B, g.L, g.P, g.D, g.A, g.vocab = B, L, P, D, A, table.shape[0]
    g.ignore_index, g.action_token_begin, g.action_dim = ignore_index, action_token_begin, action_dim
    _lib.call("ovla_assemble_multimodal", g, _stream())
    return out, action_pos


def gather_rows(src, index, dim, *, dst=None, scatter_add=False, n_dst_rows=None):
    """gather: dst[i] = src[index[i]];  scatter_add: dst[index[i]] += src[i]."""
    n = index.numel()
    if dst is None:
        dst = torch.empty((n, dim), dtype=BF16, device=src.device)
    g = STRUCTS["ovla_gather_rows_args"]()
    g.src, g.index, g.dst, g.n, g.dim = src.data_ptr(), index.data_ptr(), dst.data_ptr(), n, dim
    g.src_ld, g.dst_ld, g.scatter_add = src.stride(0), dst.stride(0), int(scatter_add)
    _lib.call("ovla_gather_rows", g, _stream())
    return dst


# ----------------------------------------------------------------------------------------------------------------------
def token_ce(logits, targets, *, vocab=None, grad_scale=None, inplace_grad=True):
    """Next-token cross entropy on gathered rows (ovla.h: ovla_token_ce): logits bf16 [rows, ld >= vocab], targets int64 [rows] ->
    (loss_rows fp32 [rows], argmax int32 [rows], dlogits bf16 or None).  With grad_scale the gradient overwrites `logits`
    (inplace_grad) or goes to a new tensor."""
    _chk(logits)
    rows = logits.shape[0]
    vocab = logits.shape[1] if vocab is None else vocab
    assert targets.dtype == torch.int64 and targets.numel() == rows and targets.is_contiguous() and targets.device == logits.device
    loss_rows = torch.empty(rows, dtype=torch.float32, device=logits.device)
    amax = torch.empty(rows, dtype=torch.int32, device=logits.device)
    d = None
    if grad_scale is not None:
        d = logits if inplace_grad else torch.empty_like(logits)
    g = STRUCTS["ovla_token_ce_args"]()
    g.logits, g.ld, g.targets, g.loss_rows, g.argmax = logits.data_ptr(), logits.stride(0), targets.data_ptr(), loss_rows.data_ptr(), amax.data_ptr()
    g.dlogits, g.ld_d = _p(d), (d.stride(0) if d is not None else 0)
    g.rows, g.vocab, g.grad_scale = rows, vocab, float(grad_scale or 0.0)
    _lib.call("ovla_token_ce", g, _stream())
    return loss_rows, amax, d


def head_out_fwd(x, W, b, target=None, loss_sum=None, mse=False):
    rows, dim = x.shape
    adim = W.shape[0]
    pred = torch.empty((rows, adim), dtype=BF16, device=x.device)
    g = STRUCTS["ovla_head_out_fwd_args"]()
    g.x, g.W, g.b, g.pred, g.target, g.loss_sum = x.data_ptr(), W.data_ptr(), _p(b), pred.data_ptr(), _p(target), _p(loss_sum)
    g.rows, g.dim, g.adim, g.mse = rows, dim, adim, int(mse)
    _lib.call("ovla_head_out_fwd", g, _stream())
    return pred


def head_tail_fwd(x0, blocks, ln2, out_w, out_b, *, rows_real, target=None, loss_sum=None, mse=False, train=False, sync=None, ksplit=2, eps=1e-5):
    """The fused action-head tail (ovla.h: ovla_head_tail_fwd): x0 bf16 [R, D] (R a multiple of 16, <= 64) -> two MLPResNet blocks, LayerNorm 2,
    fc2 and the loss in one launch.  blocks = [(ln_w, ln_b, W [D, D], bias)] * 2, ln2 = (w, b).  Returns a dict with pred [rows_real, adim] and
    everything the (unfused) backward reads: per block hb, zb, x_out, mean, rstd (hb / zb / stats only when train), h2, mean2, rstd2."""
    _chk(x0)
    R, D = x0.shape
    assert x0.is_contiguous() and len(blocks) == 2
    dev = x0.device
    adim = out_w.shape[0]
    new = lambda: torch.empty((R, D), dtype=BF16, device=dev)  # noqa: E731
    stat = lambda: torch.empty(R, dtype=torch.float32, device=dev)  # noqa: E731
    out = dict(hb=[new(), new()], zb=[new() if train else None for _ in range(2)], xo=[new(), new()],
               mean=[stat() if train else None for _ in range(2)], rstd=[stat() if train else None for _ in range(2)], h2=new(),
               mean2=stat() if train else None, rstd2=stat() if train else None, pred=torch.empty((rows_real, adim), dtype=BF16, device=dev))
    if sync is None:
        sync = torch.zeros(2, dtype=torch.int32, device=dev)
    g = STRUCTS["ovla_head_tail_args"]()
    g.x0 = x0.data_ptr()
    for b, (lw, lb, W, bias) in enumerate(blocks):
        assert tuple(W.shape) == (D, D) and W.is_contiguous() and lw.numel() == D and lb.numel() == D and bias.numel() == D
        g.ln_w[b], g.ln_b[b], g.W[b], g.bias[b] = lw.data_ptr(), lb.data_ptr(), W.data_ptr(), bias.data_ptr()
        g.hb[b], g.zb[b], g.xo[b] = _p(out["hb"][b]), _p(out["zb"][b]), out["xo"][b].data_ptr()
        g.mean[b], g.rstd[b] = _p(out["mean"][b]), _p(out["rstd"][b])
    g.ln2_w, g.ln2_b, g.h2, g.mean2, g.rstd2 = ln2[0].data_ptr(), ln2[1].data_ptr(), out["h2"].data_ptr(), _p(out["mean2"]), _p(out["rstd2"])
    g.W2, g.b2, g.pred, g.target, g.loss_sum = out_w.data_ptr(), _p(out_b), out["pred"].data_ptr(), _p(target), _p(loss_sum)
    g.sync, g.rows, g.rows_real, g.dim, g.adim, g.mse, g.ksplit, g.eps = sync.data_ptr(), R, rows_real, D, adim, int(mse), ksplit, eps
    _lib.call("ovla_head_tail_fwd", g, _stream())
    out["sync"] = sync
    return out


def head_out_bwd(x, W, pred, target, dloss_scale, dW, db, mse=False, dpred=None):
    """Backward of the head tail.  Either (pred, target, dloss_scale) -- fused L1/MSE gradient -- or an explicit dpred."""
    rows, dim = x.shape
    adim = W.shape[0]
    dx = torch.empty_like(x)
    g = STRUCTS["ovla_head_out_bwd_args"]()
    g.x, g.W, g.pred, g.target, g.dpred = x.data_ptr(), W.data_ptr(), _p(pred), _p(target), _p(dpred)
    g.dloss_scale, g.mse = dloss_scale, int(mse)
    g.dx, g.dW, g.db, g.rows, g.dim, g.adim = dx.data_ptr(), _p(dW), _p(db), rows, dim, adim
    _lib.call("ovla_head_out_bwd", g, _stream())
    return dx


def adamw(param, exp_avg, exp_avg_sq, grad, *, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
    g = STRUCTS["ovla_adamw_args"]()
    g.param, g.exp_avg, g.exp_avg_sq, g.grad = param.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), grad.data_ptr()
    g.n, g.is_bf16, g.step = param.numel(), int(param.dtype == BF16), step
    g.lr, g.beta1, g.beta2, g.eps, g.weight_decay, g.grad_scale = lr, beta1, beta2, eps, weight_decay, grad_scale
    assert grad.dtype == torch.float32 and exp_avg.dtype == param.dtype and exp_avg_sq.dtype == param.dtype
    _lib.call("ovla_adamw", g, _stream())


def cvt_f32_to_bf16(src, dst=None, scale=1.0):
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    g = STRUCTS["ovla_cvt_args"]()
    g.src, g.dst, g.n, g.scale = src.data_ptr(), dst.data_ptr(), src.numel(), scale
    _lib.call("ovla_cvt_f32_to_bf16", g, _stream())
    return dst


def cvt_bf16_to_f32(src, dst=None, scale=1.0):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().ovla_cvt_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), scale, _stream()), "ovla_cvt_bf16_to_f32")
    return dst


def transpose(src, dst=None):
    _chk(src)
    rows, cols = src.shape
    assert src.stride(1) == 1
    if dst is None:
        dst = torch.empty((cols, rows), dtype=BF16, device=src.device)
    g = STRUCTS["ovla_transpose_args"]()
    g.src, g.dst, g.rows, g.cols, g.lds, g.ldd = src.data_ptr(), dst.data_ptr(), rows, cols, src.stride(0), dst.stride(0)
    _lib.call("ovla_transpose_bf16", g, _stream())
    return dst


def selftest_layouts(device):
    out = torch.zeros((4, 64, 16), dtype=torch.float32, device=device)
    src = torch.arange(512, dtype=torch.float32, device=device).to(BF16)
    _lib.check(_lib.lib().ovla_selftest_layouts(out.data_ptr(), src.data_ptr(), _stream()), "ovla_selftest_layouts")
    return out
