"""Generates tests/golden/g12_jpeg_roundtrip.npz: JPEG encode -> decode round trips produced by libjpeg-turbo ITSELF (through Pillow, which
links it) in the build container, for the settings TensorFlow's encode_jpeg / decode_image defaults select (baseline, quality 95, 4:2:0,
accurate integer DCT, fancy upsampling).  Only data is committed (inputs + the library's outputs); oracle/jpeg_oracle.py and the HIP kernel
(ovla_jpeg_roundtrip) must reproduce the outputs bit for bit.

  python tests/golden/make_golden_jpeg.py
"""
import io
from pathlib import Path

import numpy as np
from PIL import Image, features


def libjpeg_roundtrip(img: np.ndarray, quality: int = 95) -> np.ndarray:
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=quality, subsampling=2, optimize=False, progressive=False)   # subsampling 2 = 4:2:0
    buf.seek(0)
    return np.array(Image.open(buf).convert("RGB"))


def scene(rng, h, w):
    """A camera-like frame: smooth shading + edges + sensor noise."""
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 17.0 + yy / 29.0), 127 + 90 * np.cos(xx / 11.0 - yy / 41.0), 60 + 0.6 * yy + 0.2 * xx], -1)
    base[h // 3: h // 2, w // 4: w // 2] = (230, 40, 60)
    return np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)


def main():
    assert features.check_feature("libjpeg_turbo"), "the fixtures must come from libjpeg-turbo (the codec TensorFlow links)"
    rng = np.random.default_rng(12)
    cases = {"scene_96x128": scene(rng, 96, 128), "scene_odd_75x101": scene(rng, 75, 101), "noise_48x64": rng.integers(0, 256, (48, 64, 3), dtype=np.uint8),
             "noise_odd_37x53": rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), "tiny_9x17": scene(rng, 9, 17),
             "saturated_32x32": (rng.integers(0, 2, (32, 32, 3)) * 255).astype(np.uint8), "flat_16x16": np.full((16, 16, 3), 200, np.uint8)}
    out = {}
    for k, img in cases.items():
        out[k + "__in"] = img
        out[k + "__q95"] = libjpeg_roundtrip(img, 95)
    out["scene_96x128__q50"] = libjpeg_roundtrip(cases["scene_96x128"], 50)
    out["scene_96x128__q100"] = libjpeg_roundtrip(cases["scene_96x128"], 100)
    np.savez_compressed(Path(__file__).resolve().parent / "g12_jpeg_roundtrip.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
