"""Host-side image preparation of the inference glue, without TensorFlow / torchvision.

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


def prepare_images_for_vla(images, cfg):
    """experiments/robot/openvla_utils.py:678-708.  Inputs that are not 224x224 would need the reference's
    JPEG round-trip + lanczos3 resize (TF); that branch is not available offline and raises."""
    out = []
    for image in images:
        check_image_format(image)
        if image.shape != (OPENVLA_IMAGE_SIZE, OPENVLA_IMAGE_SIZE, 3):
            raise NotImplementedError("resize_image_for_policy (TF jpeg + lanczos3) is not available in this port; pass 224x224 images")
        out.append(center_crop_image(image) if cfg.center_crop else image)
    return out


def apply_transform(image_u8: np.ndarray) -> torch.Tensor:
    """uint8 HWC (already 224 x 224: resize and center-crop are identities) -> float32 (6, H, W)."""
    x = torch.from_numpy(image_u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    outs = []
    for mean, std in ((IMAGENET_MEAN, IMAGENET_STD), (SIGLIP_MEAN, SIGLIP_STD)):
        outs.append((x - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None])
    return torch.cat(outs, dim=0)
