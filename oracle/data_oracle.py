"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/ benchmarks' checker legs, never by the product path).

CPU restatement of the reference's training data path for checking
`openvla-oft_amd/prismatic/vla/datasets/rlds_free.py` (host logic) and `ovla_image_augment` (HIP kernels):

  chunk_act_obs_loop            prismatic/vla/datasets/rlds/traj_transforms.py:14-59   (pure-Python loops over the gather indices)
  normalize_loop                prismatic/vla/datasets/rlds/utils/data_utils.py:52-94  (element loops, float32)
  batch_transform_ids           prismatic/vla/datasets/datasets.py:36-97               (prompt | '' | action ids | </s>, label masking)
  augment_image                 dlimp `augment_image` as configured at prismatic/vla/datasets/datasets.py:159-174, called from
                                prismatic/vla/datasets/rlds/obs_transforms.py:18-45; ops restated from TensorFlow 2.15's CPU kernels
                                (crop_and_resize_op.cc, adjust_contrast_op.cc, adjust_saturation_op.cc, adjust_hue_op.cc) in explicit
                                float32, one rounding per operation
  pixel_values                  prismatic/extern/hf/processing_prismatic.py:128-145   (to_tensor + per-backbone normalise)

PARITY UNPINNED against TensorFlow / dlimp themselves: neither is installable in this environment (tensorflow==2.15.0 and
`dlimp @ git+https://github.com/moojink/dlimp_openvla`, pyproject.toml:52-55, are absent and there is no network); the reference
holds no fixtures for this path.  What IS pinned: the index arithmetic and normalisation against hand-computed cases
(tests/test_data_path.py), and the HIP kernels bit-for-bit against this file.
"""
import numpy as np

F = np.float32


# ---- host logic ------------------------------------------------------------------------------------------------------------------
def chunk_act_obs_loop(traj_len: int, window_size: int, future_action_window_size: int):
    eff = traj_len - future_action_window_size
    obs, act, pad = [], [], []
    for t in range(max(eff, 0)):
        o_row, p_row = [], []
        for w in range(-window_size + 1, 1):
            o_row.append(max(t + w, 0))
            p_row.append(t + w >= 0)
        a_row = [min(max(t + w, 0), traj_len - 1) for w in range(-window_size + 1, 1 + future_action_window_size)]
        obs.append(o_row); act.append(a_row); pad.append(p_row)
    return obs, act, pad


def normalize_loop(x, stats, kind: str):
    """kind in {"normal", "bounds", "bounds_q99"}; x [T, D] -> float32 [T, D]"""
    x = np.asarray(x, F)
    out = np.empty_like(x)
    D = x.shape[1]
    mask = stats.get("mask", [True] * D)
    for t in range(x.shape[0]):
        for d in range(D):
            v = x[t, d]
            if kind == "normal":
                if mask[d]:
                    v = F(F(v - F(stats["mean"][d])) / F(F(stats["std"][d]) + F(1e-8)))
            else:
                lo, hi = (stats["min"][d], stats["max"][d]) if kind == "bounds" else (stats["q01"][d], stats["q99"][d])
                if mask[d]:
                    v = F(F(F(F(2) * F(v - F(lo))) / F(F(F(hi) - F(lo)) + F(1e-8))) - F(1))
                    v = min(max(v, F(-1)), F(1))
                if stats["min"][d] == stats["max"][d]:
                    v = F(0)
            out[t, d] = v
    return out


def batch_transform_ids(prompt_ids, action_ids, predict_stop_token=True, empty_token_id=29871, stop=2, ignore=-100):
    ids = list(prompt_ids)
    if ids[-1] != empty_token_id:
        ids.append(empty_token_id)
    ids = ids + [int(a) for a in action_ids] + [stop]
    labels = list(ids)
    for i in range(len(ids) - (len(action_ids) + 1)):
        labels[i] = ignore
    if not predict_stop_token:
        labels[-1] = ignore
    return ids, labels


# ---- image ops (float32, TF CPU kernel arithmetic) ----------------------------------------------------------------------------------
def _clip01(x):
    return np.clip(x, F(0), F(1)).astype(F)


def crop_and_resize(img, box, out: int):
    """crop_and_resize_op.cc (CPU, bilinear, extrapolation_value 0): img float32 [H, W, 3] in [0,1]."""
    H, W = img.shape[:2]
    y1, x1, y2, x2 = (F(b) for b in box)

    def coords(lo, hi, n):
        scale = (F(hi - lo) * F(n - 1) / F(out - 1)).astype(F)
        c = (F(lo * F(n - 1)) + (np.arange(out, dtype=F) * scale).astype(F)).astype(F)
        ok = (c >= 0) & (c <= F(n - 1))
        cc = np.where(ok, c, F(0)).astype(F)
        lo_i, hi_i = np.floor(cc), np.ceil(cc)
        return lo_i.astype(np.int64), hi_i.astype(np.int64), (cc - lo_i).astype(F), ok

    y0, yb, wy, oky = coords(y1, y2, H)
    x0, xr, wx, okx = coords(x1, x2, W)
    wy, wx = wy[:, None, None], wx[None, :, None]
    tl, tr, bl, br = img[y0][:, x0], img[y0][:, xr], img[yb][:, x0], img[yb][:, xr]
    top = (tl + ((tr - tl).astype(F) * wx).astype(F)).astype(F)
    bot = (bl + ((br - bl).astype(F) * wx).astype(F)).astype(F)
    res = (top + ((bot - top).astype(F) * wy).astype(F)).astype(F)
    return np.where((oky[:, None] & okx[None, :])[:, :, None], res, F(0)).astype(F)


def adjust_contrast(x, factor):
    mean = (x.astype(np.float64).sum(axis=(0, 1)) / float(x.shape[0] * x.shape[1])).astype(F)   # fp64 accumulation (as the HIP kernel)
    return (((x - mean).astype(F) * F(factor)).astype(F) + mean).astype(F)


def rgb_to_hsv(r, g, b):
    vv = np.maximum(r, np.maximum(g, b))
    rng = (vv - np.minimum(r, np.minimum(g, b))).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(vv > 0, (rng / vv).astype(F), F(0)).astype(F)
        norm = (F(1) / (F(6) * rng).astype(F)).astype(F)
        h_r = (norm * (g - b).astype(F)).astype(F)
        h_g = ((norm * (b - r).astype(F)).astype(F).astype(np.float64) + 2.0 / 6.0).astype(F)
        h_b = ((norm * (r - g).astype(F)).astype(F).astype(np.float64) + 4.0 / 6.0).astype(F)
    h = np.where(r == vv, h_r, np.where(g == vv, h_g, h_b)).astype(F)
    h = np.where(rng <= 0, F(0), h).astype(F)
    h = np.where(h < 0, (h + F(1)).astype(F), h).astype(F)
    return h, s, vv


def hsv_to_rgb(h, s, v):
    c = (s * v).astype(F)
    m = (v - c).astype(F)
    dh = (h * F(6)).astype(F)
    cat = dh.astype(np.int32)
    fm = np.where(dh >= 6, dh - F(6), np.where(dh >= 4, dh - F(4), np.where(dh >= 2, dh - F(2), dh))).astype(F)
    x = (c * (F(1) - np.abs((fm - F(1)).astype(F))).astype(F)).astype(F)
    z = np.zeros_like(c)
    rr = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4, cat == 5], [c, x, z, z, x, c], z)
    gg = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4, cat == 5], [x, c, c, x, z, z], z)
    bb = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4, cat == 5], [z, z, x, c, c, x], z)
    return (rr + m).astype(F), (gg + m).astype(F), (bb + m).astype(F)


def adjust_saturation(x, factor):
    h, s, v = rgb_to_hsv(x[..., 0], x[..., 1], x[..., 2])
    s = np.minimum(F(1), np.maximum(F(0), (s * F(factor)).astype(F))).astype(F)
    return np.stack(hsv_to_rgb(h, s, v), axis=-1)


def adjust_hue(x, delta):
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    c1, c3, c0, c4 = (r < g) & (b < r), (r < g) & (b > g), ~(r < g) & (b < g), ~(r < g) & (b > r)
    c2, c5 = (r < g) & ~c1 & ~c3, ~(r < g) & ~c0 & ~c4
    conds = [c0, c1, c2, c3, c4, c5]
    v_max = np.select(conds, [r, g, g, b, b, r])
    v_mid = np.select(conds, [g, r, b, g, r, b])
    v_min = np.select(conds, [b, b, r, r, g, g])
    cat = np.select(conds, [0, 1, 2, 3, 4, 5]).astype(np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = ((v_mid - v_min).astype(F) / (v_max - v_min).astype(F)).astype(F)
    h = (cat.astype(F) + np.where((cat & 1) == 0, ratio, (F(1) - ratio).astype(F))).astype(F)
    h = np.where(v_max == v_min, F(0), h).astype(F)
    h = (h + (F(delta) * F(6))).astype(F)
    h = np.where(h < 0, (h + F(6)).astype(F), h).astype(F)
    h = np.where(h >= 6, (h - F(6)).astype(F), h).astype(F)
    c = h.astype(np.int32)
    ratio = (h - c.astype(F)).astype(F)
    ratio = np.where((c & 1) != 0, (F(1) - ratio).astype(F), ratio).astype(F)
    mid = (v_min + (ratio * (v_max - v_min).astype(F)).astype(F)).astype(F)
    sel = [c == 0, c == 1, c == 2, c == 3, c == 4]
    r2 = np.select(sel, [v_max, mid, v_min, v_min, mid], v_max)
    g2 = np.select(sel, [mid, v_max, v_max, mid, v_min], v_min)
    b2 = np.select(sel, [v_min, v_min, mid, v_max, v_max], mid)
    return np.stack([r2, g2, b2], axis=-1).astype(F)


CROP, BRIGHTNESS, CONTRAST, SATURATION, HUE = 1, 2, 4, 8, 16


def augment_image(img_u8, params, ops_mask: int = 31, out: int = 224):
    """uint8 [H, W, 3] + (y1, x1, y2, x2, brightness delta, contrast factor, saturation factor, hue delta) -> uint8 [out, out, 3]"""
    x = (img_u8.astype(F) / F(255)).astype(F)
    if ops_mask & CROP:
        x = _clip01(crop_and_resize(x, params[0:4], out))
    if ops_mask & BRIGHTNESS:
        x = _clip01((x + F(params[4])).astype(F))
    if ops_mask & CONTRAST:
        x = _clip01(adjust_contrast(x, params[5]))
    if ops_mask & SATURATION:
        x = _clip01(adjust_saturation(x, params[6]))
    if ops_mask & HUE:
        x = _clip01(adjust_hue(x, params[7]))
    return (x * F(255)).astype(F).astype(np.uint8)


def pixel_values(img_u8, mean=(0.485, 0.456, 0.406, 0.5, 0.5, 0.5), std=(0.229, 0.224, 0.225, 0.5, 0.5, 0.5)):
    """uint8 [H, W, 3] -> float32 [6, H, W] (DINOv2-normalised channels, then SigLIP-normalised ones)"""
    x = (img_u8.astype(F) / F(255)).astype(F).transpose(2, 0, 1)
    m, s = np.asarray(mean, F), np.asarray(std, F)
    a = ((x - m[:3, None, None]).astype(F) / s[:3, None, None]).astype(F)
    b = ((x - m[3:, None, None]).astype(F) / s[3:, None, None]).astype(F)
    return np.concatenate([a, b], axis=0)


# ---- lanczos3 resize (scale_and_translate_op.cc, TF 2.15) -------------------------------------------------------------------------
def lanczos3_kernel(x):
    """LanczosKernelFunc(radius 3): float32, kPI = 3.14159265359f"""
    x = abs(F(x))
    if x > F(3):
        return F(0)
    if x <= F(1e-3):
        return F(1)
    pi = F(3.14159265359)
    px = F(pi * x)
    return F(F(F(F(3) * np.sin(px, dtype=F)) * np.sin(F(px / F(3)), dtype=F)) / F(F(F(pi * pi) * x) * x))


def lanczos3_spans_loop(in_size: int, out_size: int):
    """ComputeSpansCore, element by element (scale = out / in, translate = 0, antialias = True)."""
    scale = F(F(out_size) / F(in_size))
    inv_scale = F(1.0 / np.float64(scale))
    ks = max(inv_scale, F(1))
    span = min(2 * int(np.ceil(F(3) * ks)) + 1, in_size)
    inv_ks = F(F(1) / ks)
    starts, weights = np.zeros(out_size, np.int32), np.zeros((out_size, span), F)
    for x in range(out_size):
        s = F(F(F(x) + F(0.5)) * inv_scale + F(-inv_scale * F(0)))
        if s < 0 or s > F(in_size):
            continue
        rk = F(F(3) * ks)
        a = min(max(int(np.ceil(F(F(s - rk) - F(0.5)))), 0), in_size - 1)
        b = min(max(int(np.floor(F(F(s + rk) - F(0.5)))), 0), in_size - 1) + 1
        tmp, total = [], F(0)
        for src in range(a, b):
            w = lanczos3_kernel(F(F(F(F(src) + F(0.5)) - s) * inv_ks))
            total = F(total + w)
            tmp.append(w)
        if abs(total) >= F(1000) * np.finfo(F).tiny:
            inv = F(F(1) / total)
            for j, w in enumerate(tmp):
                weights[x, j] = F(w * inv)
        starts[x] = a
    return starts, weights


def resize_lanczos3(img_u8, out_h: int, out_w: int):
    """GatherRows then GatherColumns (sequential float32 accumulation from 0 in span order), tf.round, clip, uint8."""
    H, W = img_u8.shape[:2]
    rs, rw = lanczos3_spans_loop(H, out_h)
    cs, cw = lanczos3_spans_loop(W, out_w)
    x = img_u8.astype(F)
    mid = np.zeros((out_h, W, 3), F)
    for y in range(out_h):
        for k in range(min(rs[y] + rw.shape[1], H) - rs[y]):
            mid[y] = (mid[y] + (x[rs[y] + k] * rw[y, k]).astype(F)).astype(F)
    out = np.zeros((out_h, out_w, 3), F)
    for xo in range(out_w):
        for k in range(min(cs[xo] + cw.shape[1], W) - cs[xo]):
            out[:, xo] = (out[:, xo] + (mid[:, cs[xo] + k] * cw[xo, k]).astype(F)).astype(F)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
