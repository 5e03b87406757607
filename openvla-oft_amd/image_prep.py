"""Host-side image preparation of the inference glue, without TensorFlow / torchvision.

  resize_image_for_policy  experiments/robot/openvla_utils.py:516-540  (JPEG encode/decode round trip, then lanczos3 + antialias resize: both on the device)
  center_crop_image        experiments/robot/openvla_utils.py:542-622  (tf.image.crop_and_resize of the central
                           sqrt(0.9) x sqrt(0.9) box back to 224 x 224, bilinear, on float [0,1], back to uint8)
  apply_transform          prismatic/extern/hf/processing_prismatic.py:128-145 (per-backbone to_tensor + normalise, channel
                           stack: DINOv2 ImageNet mean/std first, SigLIP 0.5/0.5 second -- timm data configs)
Parity status: TensorFlow is not installable here, so the crop follows the published arithmetic of TF 2.15's CropAndResize
kernel in explicit float32 and is PARITY UNPINNED against TF itself.  `ops.image_prep` (ovla_image_prep) is the same arithmetic
as one HIP launch, bit for bit.
"""
import numpy as np
import torch

OPENVLA_IMAGE_SIZE = 224
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
SIGLIP_MEAN, SIGLIP_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)


def check_image_format(image) -> None:
    ok = isinstance(image, np.ndarray) and len(image.shape) == 3 and image.shape[-1] == 3 and image.dtype == np.uint8
    assert ok, "Incorrect image format detected! Make sure that the input image is a numpy array with shape (H, W, 3) and dtype np.uint8!"


def center_crop_image(image_u8: np.ndarray, crop_scale: float = 0.9, out_size: int = OPENVLA_IMAGE_SIZE) -> np.ndarray:
    """Host mirror of ovla_image_prep's crop (used when the images are not on the device): experiments/robot/openvla_utils.py:542-622 restated in explicit float32, operation by operation, after the published
    algorithm of tensorflow==2.15.0 (pyproject.toml:52; absent here -> PARITY UNPINNED):
      convert_image_dtype(u8 -> f32):   x * f32(1/255)
      box (crop_and_resize, :565-579):  side = clip(sqrt(f32(crop_scale)), 0, 1); y1 = (1 - side) / 2; y2 = y1 + side       (f32)
      CropAndResize kernel:             scale = (y2 - y1) * (H - 1) / (out - 1);  in_y = y1 * (H - 1) + i * scale
                                        top = floor(in_y), bottom = ceil(in_y), lerp = in_y - top
                                        v = tl + (tr - tl) * x_lerp  (top and bottom rows), out = top + (bottom - top) * y_lerp
      clip_by_value(0, 1); convert_image_dtype(f32 -> u8, saturate): trunc(x * 255.5)."""
    f = np.float32
    img = image_u8.astype(f) * f(1.0 / 255.0)
    H, W = img.shape[:2]
    side = np.clip(np.sqrt(f(crop_scale)), f(0), f(1)).astype(f)
    o1 = ((f(1) - side) / f(2)).astype(f)
    o2 = (o1 + side).astype(f)

    def coords(n):
        scale = ((o2 - o1) * f(n - 1) / f(out_size - 1)).astype(f)
        c = (o1 * f(n - 1) + np.arange(out_size, dtype=f) * scale).astype(f)
        lo, hi = np.floor(c), np.ceil(c)
        return lo.astype(np.int64), hi.astype(np.int64), (c - lo).astype(f)

    y0, y1, wy = coords(H)
    x0, x1, wx = coords(W)
    wy, wx = wy[:, None, None], wx[None, :, None]
    tl, tr, bl, br = img[y0][:, x0], img[y0][:, x1], img[y1][:, x0], img[y1][:, x1]
    top = (tl + ((tr - tl) * wx).astype(f)).astype(f)
    bot = (bl + ((br - bl) * wx).astype(f)).astype(f)
    out = (top + ((bot - top) * wy).astype(f)).astype(f)
    return (np.clip(out, f(0), f(1)) * f(255.5)).astype(np.uint8)


_SPAN_CACHE = {}


def lanczos3_spans(in_size: int, out_size: int):
    """The per-output-pixel sample spans of tf.image.resize(method="lanczos3", antialias=True) along one axis, after TF 2.15's
    scale_and_translate_op.cc `ComputeSpansCore` (float32 throughout): scale = out / in, kernel widened by max(1 / scale, 1) when
    downsampling, span = [ceil(s - r k - 0.5), floor(s + r k - 0.5)] clamped to the image, weights = lanczos3((i + 0.5 - s) / k)
    normalised to sum 1.  -> (starts int32 [out], weights float32 [out, span_size]); PARITY UNPINNED against TensorFlow."""
    key = (in_size, out_size)
    if key in _SPAN_CACHE:
        return _SPAN_CACHE[key]
    f = np.float32
    radius = f(3.0)
    scale = f(f(out_size) / f(in_size))
    inv_scale = f(1.0 / np.float64(scale))
    kernel_scale = max(inv_scale, f(1.0))
    span_size = min(2 * int(np.ceil(radius * kernel_scale)) + 1, in_size)
    inv_ks = f(f(1.0) / kernel_scale)
    x = np.arange(out_size, dtype=f)
    sample = ((x + f(0.5)) * inv_scale + f(-inv_scale * f(0.0))).astype(f)
    rk = f(radius * kernel_scale)
    start = np.clip(np.ceil((sample - rk).astype(f) - f(0.5)).astype(np.int64), 0, in_size - 1)
    end = np.clip(np.floor((sample + rk).astype(f) - f(0.5)).astype(np.int64), 0, in_size - 1) + 1
    k = np.arange(span_size)[None, :]
    src = start[:, None] + k
    pos = np.abs((((src.astype(f) + f(0.5)) - sample[:, None]).astype(f) * inv_ks).astype(f))
    pi = f(3.14159265359)
    with np.errstate(divide="ignore", invalid="ignore"):
        px = (pi * pos).astype(f)
        num = ((radius * np.sin(px).astype(f)).astype(f) * np.sin((px / radius).astype(f)).astype(f)).astype(f)
        den = (((pi * pi).astype(f) * pos).astype(f) * pos).astype(f)
        w = (num / den).astype(f)
    w = np.where(pos <= f(1e-3), f(1.0), w)
    w = np.where(pos > radius, f(0.0), w)
    w = np.where(src < end[:, None], w, f(0.0)).astype(f)
    total = np.zeros(out_size, f)
    for j in range(span_size):                       # TF accumulates the weight sum in span order
        total = (total + w[:, j]).astype(f)
    ok = np.abs(total) >= f(1000.0) * np.finfo(f).tiny
    w = np.where(ok[:, None], (w * (f(1.0) / np.where(ok, total, f(1.0))).astype(f)[:, None]).astype(f), f(0.0)).astype(f)
    outside = (sample < 0) | (sample > f(in_size))
    w[outside] = 0
    start = np.where(outside, 0, start)
    _SPAN_CACHE[key] = (start.astype(np.int32), np.ascontiguousarray(w))
    return _SPAN_CACHE[key]


def resize_image_for_policy(img: np.ndarray, resize_size, device=None, jpeg: bool = True) -> np.ndarray:
    """experiments/robot/openvla_utils.py:516-540: tf.image.encode_jpeg -> tf.io.decode_image (ovla_jpeg_roundtrip: libjpeg-turbo's baseline
    4:2:0 quality-95 codec without the entropy coder, bit-identical to libjpeg-turbo's accurate-DCT (islow) path, fixture G12; PARITY UNPINNED
    against TensorFlow itself, whose decode may use the fast IDCT: oracle/jpeg_oracle.py header), then tf.image.resize(lanczos3, antialias=True) ->
    round -> clip -> uint8 (ovla_image_resize), all on the device.  img uint8 [H, W, 3] -> uint8 [h, w, 3].  `jpeg=False` skips the codec (the
    training collator's resize of stored frames, rlds/obs_transforms.py:83, has none)."""
    import importlib

    ops = importlib.import_module(__package__ + ".ops")      # the HIP library: required
    assert isinstance(resize_size, (int, tuple))
    oh, ow = (resize_size, resize_size) if isinstance(resize_size, int) else resize_size
    check_image_format(img)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    spans = [tuple(torch.from_numpy(a).to(dev) for a in lanczos3_spans(n_in, n_out)) for n_in, n_out in ((img.shape[0], oh), (img.shape[1], ow))]
    frames = torch.from_numpy(np.ascontiguousarray(img))[None].to(dev)
    if jpeg:
        frames = ops.jpeg_roundtrip(frames)
    out = ops.image_resize(frames, spans[0], spans[1])
    return out[0].cpu().numpy()


def prepare_images_for_vla(images, cfg):
    """experiments/robot/openvla_utils.py:678-708: frames that are not 224 x 224 go through resize_image_for_policy (JPEG round trip +
    lanczos3 antialias resize), then the center crop if configured."""
    out = []
    for image in images:
        check_image_format(image)
        if image.shape != (OPENVLA_IMAGE_SIZE, OPENVLA_IMAGE_SIZE, 3):
            image = resize_image_for_policy(image, OPENVLA_IMAGE_SIZE)
        out.append(center_crop_image(image) if cfg.center_crop else image)
    return out


def apply_transform(image_u8: np.ndarray) -> torch.Tensor:
    """uint8 HWC (already 224 x 224: resize and center-crop are identities) -> float32 (6, H, W)."""
    x = torch.from_numpy(image_u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    outs = []
    for mean, std in ((IMAGENET_MEAN, IMAGENET_STD), (SIGLIP_MEAN, SIGLIP_STD)):
        outs.append((x - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None])
    return torch.cat(outs, dim=0)
