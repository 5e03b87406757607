"""Extracts the reference's only data fixture, experiments/robot/libero/sample_libero_spatial_observation.pkl, WITHOUT unpickling it: the
file is walked opcode by opcode with `pickletools.genops` (a parser: nothing in the file is executed, no object is constructed), and the
numpy buffers it carries (`numpy.core.numeric._frombuffer(bytes, dtype, shape, order)` records: full_image, wrist_image uint8 224 x 224 x 3,
state float64 [8]) plus the task description string are written to tests/golden/g13_libero_observation.npz.  Data only.

  python tests/golden/extract_libero_observation.py [/root/reference/experiments/robot/libero/sample_libero_spatial_observation.pkl]
"""
import pickletools
import sys
from pathlib import Path

import numpy as np

BYTES_OPS = {"BINBYTES", "SHORT_BINBYTES", "BINBYTES8", "BYTEARRAY8"}
STR_OPS = {"BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8"}
INT_OPS = {"BININT", "BININT1", "BININT2", "LONG1"}
KEYS = ("full_image", "wrist_image", "state", "task_description")
DTYPES = {"u1": np.uint8, "f8": np.float64, "f4": np.float32, "i8": np.int64}


def extract(path):
    data = Path(path).read_bytes()
    ops = [(op.name, arg) for op, arg, _ in pickletools.genops(data)]
    key_at = [i for i, (name, arg) in enumerate(ops) if name in STR_OPS and arg in KEYS]
    out, dtype = {}, None
    for n, i in enumerate(key_at):
        key = ops[i][1]
        seg = ops[i + 1: key_at[n + 1] if n + 1 < len(key_at) else len(ops)]     # everything that belongs to this dict value
        if key == "task_description":
            out[key] = next(arg for name, arg in seg if name in STR_OPS)
            continue
        raw = bytes(next(arg for name, arg in seg if name in BYTES_OPS))
        for name, arg in seg:       # the dtype code is spelled out the first time a dtype occurs and taken from the pickle memo afterwards
            if name in STR_OPS and arg in DTYPES:
                dtype = DTYPES[arg]
        shape = None
        for j, (name, arg) in enumerate(seg):   # the shape is the run of small integers closed by TUPLE1 / TUPLE3 (the dtype state tuple closes with TUPLE)
            if name in ("TUPLE1", "TUPLE2", "TUPLE3"):
                k, dims = j - 1, []
                while k >= 0 and seg[k][0] in INT_OPS:
                    dims.insert(0, int(seg[k][1]))
                    k -= 1
                if dims and all(d > 0 for d in dims):
                    shape = dims
        assert dtype is not None and shape is not None and len(raw) == int(np.prod(shape)) * np.dtype(dtype).itemsize, (key, dtype, shape, len(raw))
        out[key] = np.frombuffer(raw, dtype=dtype).reshape(shape).copy()
    return out


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/experiments/robot/libero/sample_libero_spatial_observation.pkl"
    obs = extract(src)
    assert obs["full_image"].shape == (224, 224, 3) and obs["full_image"].dtype == np.uint8 and obs["wrist_image"].shape == (224, 224, 3)
    assert obs["state"].shape == (8,) and obs["state"].dtype == np.float64 and isinstance(obs["task_description"], str)
    np.savez_compressed(Path(__file__).resolve().parent / "g13_libero_observation.npz", full_image=obs["full_image"], wrist_image=obs["wrist_image"],
                        state=obs["state"], task_description=np.array(obs["task_description"]))
    print({k: (v.shape, str(v.dtype)) if hasattr(v, "shape") else v for k, v in obs.items()})
    print("state", obs["state"], "| image means", obs["full_image"].mean(), obs["wrist_image"].mean())


if __name__ == "__main__":
    main()
