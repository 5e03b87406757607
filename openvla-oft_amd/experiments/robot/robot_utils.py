"""Evaluation-loop helpers the LIBERO / ALOHA run scripts call (mirror of experiments/robot/robot_utils.py: set_seed_everywhere
:37, get_model :55, get_image_resize_size :79, get_action :99-146, normalize_gripper_action :149, invert_gripper_action :181).

`get_action` is the function the simulator loops call once per open-loop chunk (run_libero_eval.py:294); it forwards to
`get_vla_action`, i.e. to the HIP engine -- there is no other model family and no CPU path behind it."""
from __future__ import annotations

import os
import random
import time
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from .openvla_utils import get_vla, get_vla_action

ACTION_DIM = 7
DATE = time.strftime("%Y_%m_%d")
DATE_TIME = time.strftime("%Y_%m_%d-%H_%M_%S")
DEVICE = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
MODEL_IMAGE_SIZES = {"openvla": 224}


def _require_openvla(cfg: Any) -> None:
    family = getattr(cfg, "model_family", None)
    if family not in MODEL_IMAGE_SIZES:
        raise ValueError(f"Unsupported model family: {family}")


def set_seed_everywhere(seed: int) -> None:
    """Seeds python / numpy / torch (host and device generators), as the eval scripts do before building the environment."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def get_model(cfg: Any, wrap_diffusion_policy_for_droid: bool = False):
    _require_openvla(cfg)
    model = get_vla(cfg)
    print(f"Loaded model: {type(model)}")
    return model


def get_image_resize_size(cfg: Any) -> Union[int, tuple]:
    """Side length (int = square) the policy's frames are resized to before `get_action`."""
    _require_openvla(cfg)
    return MODEL_IMAGE_SIZES[cfg.model_family]


def get_action(cfg: Any, model, obs: Dict[str, Any], task_label: str, processor: Optional[Any] = None, action_head=None,
               proprio_projector=None, noisy_action_projector=None, use_film: bool = False) -> Union[List[np.ndarray], np.ndarray]:
    """One query of the policy: observation dict + instruction -> list of `num_open_loop_steps` un-normalised actions."""
    _require_openvla(cfg)
    with torch.no_grad():
        return get_vla_action(cfg=cfg, vla=model, processor=processor, obs=obs, task_label=task_label, action_head=action_head,
                              proprio_projector=proprio_projector, noisy_action_projector=noisy_action_projector, use_film=use_film)


def normalize_gripper_action(action: np.ndarray, binarize: bool = True) -> np.ndarray:
    """Last dimension [0, 1] -> [-1, +1] (the dataset wrapper leaves the gripper un-normalised); optionally snapped to its sign."""
    out = np.array(action, copy=True)
    g = 2.0 * out[..., -1] - 1.0
    out[..., -1] = np.sign(g) if binarize else g
    return out


def invert_gripper_action(action: np.ndarray) -> np.ndarray:
    """Flips the gripper sign: the RLDS loader uses 1 = open, the simulators expect -1 = open."""
    out = np.array(action, copy=True)
    out[..., -1] = -out[..., -1]
    return out
