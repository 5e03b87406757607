"""G20: the reference's OWN RLDSBatchTransform (prismatic/vla/datasets/datasets.py:26-97) and PurePromptBuilder
(prismatic/models/backbones/llm/prompting/base_prompter.py:28-73), executed in the build container on synthetic RLDS frames.

datasets.py imports the TF / dlimp pipeline at module level (`prismatic.vla.datasets.rlds`, `.rlds.oxe`) and `prismatic.models.backbones.vision`;
those are registered as NAMES-ONLY modules (nothing RLDSBatchTransform.__call__ touches lives in them).  The tokenizer is tests/duck_tokenizer.py (no
tokenizer files exist offline), the image transform the identity on the uint8 array.  What this pins: prompt construction, the decode -> re-tokenise
round trip of the action string, the IGNORE_INDEX masking arithmetic (`labels[: -(action_chunk_len + 1)]`), predict_stop_token, which observation keys
become wrist images, the proprio passthrough incl. the ur5e branch.

    python tests/golden/make_golden_batch_transform.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

import transformers  # noqa: F401

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.duck_tokenizer import DuckTokenizer  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def names_only(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    for pk in ["prismatic", "prismatic.vla", "prismatic.models", "prismatic.models.backbones", "prismatic.models.backbones.llm", "prismatic.util", "prismatic.vla.datasets"]:
        names_only(pk)
    names_only("prismatic.models.backbones.vision", ImageTransform=object)
    names_only("prismatic.vla.datasets.rlds", make_interleaved_dataset=None, make_single_dataset=None)
    names_only("prismatic.vla.datasets.rlds.oxe", OXE_NAMED_MIXTURES={}, get_oxe_dataset_kwargs_and_weights=None)
    _load("prismatic.vla.constants", "prismatic/vla/constants.py")
    _load("prismatic.util.data_utils", "prismatic/util/data_utils.py")
    at_mod = _load("prismatic.vla.action_tokenizer", "prismatic/vla/action_tokenizer.py")
    bp = _load("prismatic.models.backbones.llm.prompting.base_prompter", "prismatic/models/backbones/llm/prompting/base_prompter.py")
    names_only("prismatic.models.backbones.llm.prompting", PromptBuilder=bp.PromptBuilder, PurePromptBuilder=bp.PurePromptBuilder)
    ds = _load("prismatic.vla.datasets.datasets", "prismatic/vla/datasets/datasets.py")

    tok = DuckTokenizer()
    rng = np.random.default_rng(20)
    g20 = {}
    cases = [("libero", b"libero_spatial_no_noops", "Pick up the black bowl and place it on the plate", True, True, True),
             ("nostop", b"libero_object_no_noops", "open the middle drawer of the cabinet", True, True, False),
             ("primary_only", b"bridge_orig", "PUT the carrot  on the plate", False, False, True),
             ("ur5e", b"ur5e_pick_place", "pick the box", True, True, True)]
    for tag, name, lang, wrist, prop, stop in cases:
        obs = {"image_primary": rng.integers(0, 256, (1, 12, 12, 3), dtype=np.uint8), "image_wrist": rng.integers(0, 256, (1, 12, 12, 3), dtype=np.uint8),
               "proprio": rng.uniform(-1, 1, (1, 8)).astype(np.float32)}
        if tag == "ur5e":
            obs = {"image_camera_front_image": obs["image_primary"], "image_camera_gripper_image": obs["image_wrist"], "joint_positions": obs["proprio"]}
        frame = {"dataset_name": name, "action": rng.uniform(-1.2, 1.2, (8, 7)).astype(np.float32), "observation": obs,
                 "task": {"language_instruction": lang.encode()}}
        bt = ds.RLDSBatchTransform(at_mod.ActionTokenizer(tok), tok, image_transform=lambda im: torch.from_numpy(np.asarray(im)), prompt_builder_fn=bp.PurePromptBuilder,
                                   predict_stop_token=stop, use_wrist_image=wrist, use_proprio=prop)
        out = bt(frame)
        g20[f"{tag}.dataset_name"] = np.frombuffer(name, dtype=np.uint8)
        g20[f"{tag}.language"] = np.frombuffer(lang.encode(), dtype=np.uint8)
        g20[f"{tag}.flags"] = np.array([wrist, prop, stop])
        g20[f"{tag}.action"] = frame["action"]
        for k, v in obs.items():
            g20[f"{tag}.obs.{k}"] = v
        g20[f"{tag}.input_ids"], g20[f"{tag}.labels"] = out["input_ids"].numpy(), out["labels"].numpy()
        g20[f"{tag}.pixel_values"] = out["pixel_values"].numpy()
        g20[f"{tag}.actions"] = np.asarray(out["actions"])
        if wrist:
            g20[f"{tag}.pixel_values_wrist"] = out["pixel_values_wrist"].numpy()
        if "proprio" in out:
            g20[f"{tag}.proprio"] = np.asarray(out["proprio"])
        pb = bp.PurePromptBuilder("openvla")
        pb.add_turn("human", f"What action should the robot take to {lang.lower()}?")
        g20[f"{tag}.prompt_ids"] = np.array(tok(pb.get_prompt()).input_ids)
    np.savez_compressed(OUT / "g20_ref_batch_transform.npz", **g20)
    print("wrote", (OUT / "g20_ref_batch_transform.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
