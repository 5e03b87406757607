"""Writes a small LIBERO-shaped episode store (layout of openvla-oft_amd/prismatic/vla/datasets/rlds_free.py) for trying the fine-tune
driver without the real demonstrations:  python tools/make_synthetic_episodes.py datasets/rlds [n_episodes]"""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
synth = importlib.import_module("openvla-oft_amd.synthetic")
root = sys.argv[1] if len(sys.argv) > 1 else "datasets/rlds"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
synth.write_synthetic_episodes(root, "libero_spatial_no_noops", n)
print(f"wrote {n} episodes under {root}/libero_spatial_no_noops")
