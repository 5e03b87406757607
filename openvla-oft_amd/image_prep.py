"""Host-side image preparation of the inference glue, without TensorFlow / torchvision.

  center_crop_image        experiments/robot/openvla_utils.py:542-622  (tf.image.crop_and_resize of the central
                           sqrt(0.9) x sqrt(0.9) box back to 224 x 224, bilinear, on float [0,1], back to uint8)
  apply_transform          prismatic/extern/hf/processing_prismatic.py:128-145 (per-backbone to_tensor + normalise, channel
                           stack: DINOv2 ImageNet mean/std first, SigLIP 0.5/0.5 second -- timm data configs)
Parity status: TensorFlow is not installable here, so the crop follows TF's documented sampling rule
(y = y1 (H-1) + i (y2-y1)(H-1)/(out-1), bilinear) and is PARITY UNPINNED against TF itself.
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
    img = image_u8.astype(np.float32) / 255.0
    H, W = img.shape[:2]
    side = float(np.clip(np.sqrt(crop_scale), 0, 1))
    off = (1 - side) / 2
    ys = off * (H - 1) + np.arange(out_size, dtype=np.float32) * (side * (H - 1) / (out_size - 1))
    xs = off * (W - 1) + np.arange(out_size, dtype=np.float32) * (side * (W - 1) / (out_size - 1))
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    wy, wx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    top = img[y0][:, x0] * (1 - wx) + img[y0][:, x1] * wx
    bot = img[y1][:, x0] * (1 - wx) + img[y1][:, x1] * wx
    return (np.clip(top * (1 - wy) + bot * wy, 0, 1) * 255.5).astype(np.uint8)


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
