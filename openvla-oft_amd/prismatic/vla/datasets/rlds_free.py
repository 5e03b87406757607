"""RLDS-free data path: the reference's TFDS -> dlimp -> RLDSBatchTransform -> collator pipeline
restated without TensorFlow, on an episode store of memory-mapped `.npy` files, with the per-frame image work on the device.

What the reference does, and where it is mirrored here:

  make_dataset_from_rlds.restructure   rlds/dataset.py:132-204      -> `restructure`            (camera / state key extraction)
  libero_dataset_transform             rlds/oxe/transforms.py:833-847 -> `libero_dataset_transform` (gripper: clip to [0,1], invert)
  make_oxe_dataset_kwargs masks        rlds/oxe/materialize.py:36-46  -> `action_masks`           (normalise all but the gripper)
  get_dataset_statistics               rlds/utils/data_utils.py:176-258 -> `get_dataset_statistics` (mean/std/max/min/q01/q99, sha256-named cache)
  normalize_action_and_proprio         rlds/utils/data_utils.py:52-94 -> `normalize_action_and_proprio`
  chunk_act_obs                        rlds/traj_transforms.py:14-59  -> `chunk_indices`           (window 1 + NUM_ACTIONS_CHUNK-1 future actions,
                                                                                                   last action repeated past the end)
  skip_unlabeled                       rlds/dataset.py:302-306        -> episodes without an instruction are dropped
  RLDSDataset config                   datasets.py:100-176            -> `EpisodeDataset`          (same constructor arguments)
  dlimp augment_image (un-vendored dependency `dlimp @ git+https://github.com/moojink/dlimp_openvla`, pyproject.toml:55, no pinned revision; algorithm restated from its published
  source: float [0,1], ops in `augment_order`, clip after each, `cast(x * 255, uint8)`)  -> `sample_augment_params` + ops.image_augment
  RLDSBatchTransform.__call__          datasets.py:36-97              -> `RLDSBatchTransform`
  PaddedCollatorForActionPrediction    prismatic/util/data_utils.py:95-156 -> `DeviceCollator` (same batch dict; pixel_values made on the GPU)

Differences by design (MI355X-first, documented in DESIGN.md):
  * storage is one directory per episode holding `image.npy`, `wrist_image.npy`, `state.npy`, `action.npy` (np.load(mmap_mode="r"),
    never pickled) and `language_instruction.txt`: frames are random-access, so an epoch is a true permutation of all (episode, step)
    pairs instead of a 256k-frame shuffle buffer over a sequential TFRecord stream;
  * ranks take disjoint strides of that permutation (the reference lets every rank draw its own shuffled stream, finetune.py:1022);
  * crop-and-resize, colour jitter, uint8 re-quantisation and both backbones' normalisations run as two HIP launches per batch
    (`ovla_image_augment`) on uint8 frames copied to HBM as they are -- the host never touches a float pixel; frames stored at
    another resolution are first resized there too (`ovla_image_resize`: dlimp resize_image's lanczos3 + antialias);
  * random numbers come from numpy's PCG64 (TF's stateless Philox stream cannot be matched without TensorFlow): the augmentation
    DISTRIBUTIONS are the reference's, the draws are not.
Parity: TensorFlow and dlimp are not installable here, so everything that is TF arithmetic is PARITY UNPINNED against TF itself and
checked against `oracle/data_oracle.py` (independent loop restatement) and hand-computed cases instead (tests/test_data_path.py).
"""
from __future__ import annotations

import hashlib
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import constants as C
from ..action_tokenizer import ActionTokenizer

IGNORE_INDEX = C.IGNORE_INDEX

# rlds/oxe/configs.py:662-724 (the datasets this fine-tune path is used with)
OXE_DATASET_CONFIGS: Dict[str, Dict[str, Any]] = {
    **{name: {"image_obs_keys": {"primary": "image", "secondary": None, "wrist": "wrist_image"},
              "state_obs_keys": ["EEF_state", "gripper_state"], "action_encoding": "EEF_POS", "standardize": "libero"}
       for name in ("libero_spatial_no_noops", "libero_object_no_noops", "libero_goal_no_noops", "libero_10_no_noops", "libero_4_task_suites_no_noops")},
    **{name: {"image_obs_keys": {"primary": "image", "secondary": None, "left_wrist": "left_wrist_image", "right_wrist": "right_wrist_image"},
              "state_obs_keys": ["state"], "action_encoding": "JOINT_POS_BIMANUAL", "standardize": "aloha"}
       for name in ("aloha1_fold_shorts_20_demos", "aloha1_fold_shirt_30_demos", "aloha1_scoop_X_into_bowl_45_demos", "aloha1_put_X_into_pot_300_demos")},
}

# datasets.py:159-174 (image_aug=True)
DEFAULT_AUGMENT_KWARGS = dict(random_resized_crop=dict(scale=[0.9, 0.9], ratio=[1.0, 1.0]), random_brightness=[0.2], random_contrast=[0.8, 1.2],
                              random_saturation=[0.8, 1.2], random_hue=[0.05],
                              augment_order=["random_resized_crop", "random_brightness", "random_contrast", "random_saturation", "random_hue"])


# ------------------------------------------------------------------------------------------------------------------------------
# episode store
# ------------------------------------------------------------------------------------------------------------------------------
def write_episode(root: Path, dataset_name: str, index: int, *, arrays: Dict[str, np.ndarray], language_instruction: str) -> Path:
    """One episode = one directory of plain `.npy` arrays with a common leading (time) dimension + the instruction as text."""
    d = Path(root) / dataset_name / f"episode_{index:06d}"
    d.mkdir(parents=True, exist_ok=True)
    T = None
    for key, arr in arrays.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype == object:
            raise TypeError(f"{key}: object arrays are not stored (files are read with allow_pickle=False)")
        T = arr.shape[0] if T is None else T
        if arr.shape[0] != T:
            raise ValueError(f"{key}: leading dimension {arr.shape[0]} != trajectory length {T}")
        np.save(d / f"{key}.npy", arr, allow_pickle=False)
    (d / "language_instruction.txt").write_text(language_instruction)
    return d


class Episode:
    def __init__(self, path: Path):
        self.path = Path(path)
        self.language_instruction = (self.path / "language_instruction.txt").read_text() if (self.path / "language_instruction.txt").exists() else ""
        self._arrays: Dict[str, np.ndarray] = {}

    def keys(self) -> List[str]:
        return sorted(p.stem for p in self.path.glob("*.npy"))

    def __getitem__(self, key: str) -> np.ndarray:
        if key not in self._arrays:
            self._arrays[key] = np.load(self.path / f"{key}.npy", mmap_mode="r", allow_pickle=False)
        return self._arrays[key]

    def __len__(self) -> int:
        return int(self["action"].shape[0])


def list_episodes(root: Path, dataset_name: str, split: str = "train") -> List[Episode]:
    """split "train": `<root>/<name>/episode_*`; "val": `<root>/<name>/val/episode_*` (rlds/dataset.py:238 reads the builder's
    `train` / `val` split; a dataset without one fails there too)."""
    d = Path(root) / dataset_name
    if split == "val":
        if not (d / "val").is_dir():
            raise ValueError(f"Unknown split 'val': {d} has no val/ directory of episodes")
        d = d / "val"
    if not d.is_dir():
        raise FileNotFoundError(f"no episode store at {d} (expected {d}/episode_000000/action.npy ...; see tools/make_synthetic_episodes.py)")
    return [Episode(p) for p in sorted(d.glob("episode_*")) if (p / "action.npy").exists()]


# ------------------------------------------------------------------------------------------------------------------------------
# trajectory-level transforms (numpy restatements; float32 like the reference's tf.float32 tensors)
# ------------------------------------------------------------------------------------------------------------------------------
def invert_gripper_actions(actions: np.ndarray) -> np.ndarray:
    """rlds/utils/data_utils.py:127-128"""
    return (np.float32(1) - actions).astype(np.float32)


def binarize_gripper_actions(actions: np.ndarray) -> np.ndarray:
    """rlds/utils/data_utils.py:97-124: intermediate values take the state reached after them (reverse scan)."""
    actions = np.asarray(actions, np.float32)
    open_mask, closed_mask = actions > 0.95, actions < 0.05
    out = np.empty_like(actions)
    carry = actions[-1]
    for i in range(actions.shape[0] - 1, -1, -1):
        if open_mask[i] or closed_mask[i]:
            carry = np.float32(open_mask[i])
        out[i] = carry
    return out


def rel2abs_gripper_actions(actions: np.ndarray) -> np.ndarray:
    """rlds/utils/data_utils.py:131-153: +1 closing / -1 opening relative commands -> absolute 0 = closed, 1 = open."""
    actions = np.asarray(actions, np.float32)
    th = np.where(actions < -0.1, 1, np.where(actions > 0.1, -1, 0)).astype(np.int32)
    nz = np.flatnonzero(th != 0)
    start = -th[nz[0]] if nz.size else 0     # tf.argmax of an all-False mask is 0 and thresholded[0] == 0 there
    carry = 1 if start == 0 else int(start)
    out = np.empty(actions.shape[0], np.float32)
    for i in range(actions.shape[0]):
        if th[i] != 0:
            carry = int(th[i])
        out[i] = carry
    return (out / np.float32(2) + np.float32(0.5)).astype(np.float32)


def libero_dataset_transform(traj: Dict[str, Any]) -> Dict[str, Any]:
    """rlds/oxe/transforms.py:833-847: gripper -1 (open) .. +1 (close) -> clip to [0, 1] -> flip (+1 = open, 0 = close)."""
    action = np.asarray(traj["action"], np.float32)
    gripper = invert_gripper_actions(np.clip(action[:, -1:], np.float32(0), np.float32(1)))
    traj["action"] = np.concatenate([action[:, :6], gripper], axis=1)
    state = traj["observation"]["state"]
    traj["observation"]["EEF_state"] = state[:, :6]
    traj["observation"]["gripper_state"] = state[:, -2:]
    return traj


def aloha_dataset_transform(traj: Dict[str, Any]) -> Dict[str, Any]:
    """rlds/oxe/transforms.py:850-852"""
    return traj


STANDARDIZATION_TRANSFORMS: Dict[str, Callable] = {"libero": libero_dataset_transform, "aloha": aloha_dataset_transform}


def action_masks(action_encoding: str) -> Tuple[List[bool], List[bool]]:
    """rlds/oxe/materialize.py:36-46 -> (absolute_action_mask, action_normalization_mask)"""
    if action_encoding == "EEF_POS":
        return [False] * 6 + [True], [True] * 6 + [False]
    if action_encoding == "EEF_R6":
        return [False] * 9 + [True], [True] * 9 + [False]
    if action_encoding == "JOINT_POS_BIMANUAL":
        return [True] * 14, [True] * 14
    raise ValueError(f"unknown action encoding {action_encoding}")


def restructure(ep: Episode, cfg: Dict[str, Any], load_camera_views: Sequence[str]) -> Dict[str, Any]:
    """rlds/dataset.py:132-204 after the dataset's standardize_fn: `image_<view>` arrays (lazily memory-mapped), `proprio` = the
    configured state keys concatenated as float32, float32 actions, the instruction."""
    traj = {"observation": {k: ep[k] for k in ep.keys() if k != "action"}, "action": np.asarray(ep["action"], np.float32),
            "language_instruction": ep.language_instruction}
    traj = STANDARDIZATION_TRANSFORMS[cfg["standardize"]](traj)
    obs = traj["observation"]
    new_obs: Dict[str, Any] = {}
    for new, old in cfg["image_obs_keys"].items():
        if new in load_camera_views:
            new_obs[f"image_{new}"] = None if old is None else obs[old]     # None = padding view
    missing = set(load_camera_views) - set(cfg["image_obs_keys"])
    if missing:
        raise ValueError(f"Cannot load dataset; missing camera views `{missing}`")
    new_obs["proprio"] = np.concatenate([np.asarray(obs[k], np.float32) for k in cfg["state_obs_keys"]], axis=1)
    return {"observation": new_obs, "action": traj["action"], "language_instruction": traj["language_instruction"]}


def get_dataset_statistics(trajs: Sequence[Dict[str, Any]], hash_dependencies: Tuple[str, ...], save_dir: Optional[Path] = None) -> Dict[str, Any]:
    """rlds/utils/data_utils.py:176-258 (same JSON schema and cache-file naming; the arithmetic is numpy's in both)."""
    unique_hash = hashlib.sha256("".join(hash_dependencies).encode("utf-8"), usedforsecurity=False).hexdigest()
    path = None if save_dir is None else Path(save_dir) / f"dataset_statistics_{unique_hash}.json"
    if path is not None and path.exists():
        return json.loads(path.read_text())
    actions = np.concatenate([t["action"] for t in trajs])
    proprios = np.concatenate([t["observation"]["proprio"] if "proprio" in t["observation"] else np.zeros_like(t["action"]) for t in trajs])

    def stats(x):
        return {"mean": x.mean(0).tolist(), "std": x.std(0).tolist(), "max": x.max(0).tolist(), "min": x.min(0).tolist(),
                "q01": np.quantile(x, 0.01, axis=0).tolist(), "q99": np.quantile(x, 0.99, axis=0).tolist()}

    metadata = {"action": stats(actions), "proprio": stats(proprios), "num_transitions": int(actions.shape[0]), "num_trajectories": len(trajs)}
    if path is not None:
        try:
            path.write_text(json.dumps(metadata))
        except OSError:
            pass
    return metadata


def normalize_action_and_proprio(traj: Dict[str, Any], metadata: Dict[str, Any], normalization_type) -> Dict[str, Any]:
    """rlds/utils/data_utils.py:52-94, float32 arithmetic in the reference's operation order."""
    f = np.float32
    for key in ("action", "proprio"):
        x = np.asarray(traj["action"] if key == "action" else traj["observation"]["proprio"], f)
        md = {k: np.asarray(v) for k, v in metadata[key].items()}
        if normalization_type == C.NormalizationType.NORMAL:
            mask = md.get("mask", np.ones_like(md["mean"], dtype=bool)).astype(bool)
            x = np.where(mask, (x - md["mean"].astype(f)) / (md["std"].astype(f) + f(1e-8)), x).astype(f)
        elif normalization_type in (C.NormalizationType.BOUNDS, C.NormalizationType.BOUNDS_Q99):
            lo, hi = (md["min"], md["max"]) if normalization_type == C.NormalizationType.BOUNDS else (md["q01"], md["q99"])
            lo, hi = lo.astype(f), hi.astype(f)
            mask = md.get("mask", np.ones_like(md["min"], dtype=bool)).astype(bool)
            x = np.where(mask, np.clip(f(2) * (x - lo) / (hi - lo + f(1e-8)) - f(1), f(-1), f(1)), x).astype(f)
            x = np.where(md["min"] == md["max"], f(0), x).astype(f)     # unused dimensions -> 0
        else:
            raise ValueError(f"Unknown Normalization Type {normalization_type}")
        if key == "action":
            traj["action"] = x
        else:
            traj["observation"]["proprio"] = x
    return traj


def chunk_indices(traj_len: int, window_size: int, future_action_window_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """rlds/traj_transforms.py:14-59 -> (observation indices [T', W], action indices [T', W + F], pad_mask [T', W]) with
    T' = traj_len - F: observations before the start repeat step 0 (pad_mask False), actions past the end repeat the last one."""
    eff = traj_len - future_action_window_size
    if eff <= 0:
        z = np.zeros((0, window_size), np.int64)
        return z, np.zeros((0, window_size + future_action_window_size), np.int64), z.astype(bool)
    base = np.arange(eff)[:, None]
    obs_idx = np.arange(-window_size + 1, 1)[None, :] + base
    act_idx = np.arange(-window_size + 1, 1 + future_action_window_size)[None, :] + base
    return np.maximum(obs_idx, 0), np.minimum(np.maximum(act_idx, 0), traj_len - 1), obs_idx >= 0


# ------------------------------------------------------------------------------------------------------------------------------
# augmentation parameters (host) -- the pixels are processed by ops.image_augment on the device
# ------------------------------------------------------------------------------------------------------------------------------
AUG_PARAM_COLS = 8   # y1, x1, y2, x2 (normalised crop box), brightness delta, contrast factor, saturation factor, hue delta


AUG_BITS = {"random_resized_crop": 1, "random_brightness": 2, "random_contrast": 4, "random_saturation": 8, "random_hue": 16}


def augment_ops_mask(augment_kwargs: Optional[Dict[str, Any]] = None) -> int:
    """ovla_image_augment's `ops_mask` for an augment_kwargs dict (the ops named in its `augment_order`)."""
    kw = DEFAULT_AUGMENT_KWARGS if augment_kwargs is None else augment_kwargs
    return sum(AUG_BITS[o] for o in kw["augment_order"])


def identity_augment_params(n: int) -> np.ndarray:
    p = np.zeros((n, AUG_PARAM_COLS), np.float32)
    p[:, 2:4] = 1.0
    p[:, 5:7] = 1.0
    return p


def sample_augment_params(rng: np.random.Generator, n: int, augment_kwargs: Optional[Dict[str, Any]] = None) -> np.ndarray:
    """Per-image draws of dlimp's augment_image ops (same distributions, float32 arithmetic):
      random_resized_crop(scale, ratio): area ~ U[scale], log-ratio ~ U[log ratio]; h = clip(sqrt(area / r), 0, 1), w = clip(sqrt(area * r), 0, 1);
                                         offsets ~ U[0, 1 - h], U[0, 1 - w]; box = (y, x, y + h, x + w) for crop_and_resize
      random_brightness(max_delta): delta ~ U[-max_delta, max_delta)      random_contrast(lower, upper): factor ~ U[lower, upper)
      random_saturation(lower, upper): factor ~ U[lower, upper)            random_hue(max_delta): delta ~ U[-max_delta, max_delta)
    Ops missing from `augment_order` keep their identity parameter.  The device kernel applies them in the reference's default order."""
    kw = DEFAULT_AUGMENT_KWARGS if augment_kwargs is None else augment_kwargs
    order = list(kw["augment_order"])
    canonical = [o for o in DEFAULT_AUGMENT_KWARGS["augment_order"] if o in order]
    if order != canonical:
        raise NotImplementedError(f"augment_order {order}: the device kernel applies the ops in the order {canonical}")
    f = np.float32
    p = identity_augment_params(n)
    u = lambda lo, hi: rng.uniform(lo, hi, size=n).astype(f)  # noqa: E731
    if "random_resized_crop" in order:
        rc = kw["random_resized_crop"]
        area = u(rc["scale"][0], rc["scale"][1])
        ratio = np.exp(u(np.log(f(rc["ratio"][0])), np.log(f(rc["ratio"][1])))).astype(f)
        h = np.clip(np.sqrt(area / ratio), 0, 1).astype(f)
        w = np.clip(np.sqrt(area * ratio), 0, 1).astype(f)
        y = (rng.uniform(0, 1, size=n).astype(f) * (f(1) - h)).astype(f)
        x = (rng.uniform(0, 1, size=n).astype(f) * (f(1) - w)).astype(f)
        p[:, 0], p[:, 1], p[:, 2], p[:, 3] = y, x, (y + h).astype(f), (x + w).astype(f)
    if "random_brightness" in order:
        p[:, 4] = u(-kw["random_brightness"][0], kw["random_brightness"][0])
    if "random_contrast" in order:
        p[:, 5] = u(kw["random_contrast"][0], kw["random_contrast"][1])
    if "random_saturation" in order:
        p[:, 6] = u(kw["random_saturation"][0], kw["random_saturation"][1])
    if "random_hue" in order:
        p[:, 7] = u(-kw["random_hue"][0], kw["random_hue"][0])
    return p


# ------------------------------------------------------------------------------------------------------------------------------
# per-sample transform + collator
# ------------------------------------------------------------------------------------------------------------------------------
def build_prompt(lang: str) -> str:
    """PurePromptBuilder("openvla") with one human turn (prompting/base_prompter.py:28-73): the text BEFORE the action tokens."""
    message = f"What action should the robot take to {lang}?".replace("<image>", "").strip()
    return f"In: {message}\nOut: "


@dataclass
class RLDSBatchTransform:
    """datasets.py:27-97.  `base_tokenizer(text) -> list of ids incl. BOS` (e.g. PrismaticProcessor.tokenizer).  The reference decodes
    the action ids to text and re-tokenises prompt + action text + '</s>'; for the Llama-2 tokenizer that yields
    [prompt ids ..., 29871 (the '' after 'Out: '), action ids ..., 2] (modeling_prismatic.py:974 relies on the same layout), which is
    built here directly from ids.  Images stay uint8: `image_transform` (both backbones' normalisation) runs on the device later."""
    action_tokenizer: ActionTokenizer
    base_tokenizer: Callable[[str], Sequence[int]]
    image_transform: Any = None
    prompt_builder_fn: Any = None
    predict_stop_token: bool = True
    use_wrist_image: bool = False
    use_proprio: bool = False
    empty_token_id: int = 29871

    def __call__(self, rlds_batch: Dict[str, Any]) -> Dict[str, Any]:
        name = rlds_batch["dataset_name"]
        name_b = name if isinstance(name, bytes) else str(name).encode()
        obs = rlds_batch["observation"]
        img = obs["image_camera_front_image"][0] if b"ur5e" in name_b else obs["image_primary"][0]
        lang = rlds_batch["task"]["language_instruction"]
        lang = (lang.decode() if isinstance(lang, bytes) else lang).lower()
        actions = np.asarray(rlds_batch["action"], np.float32)
        act_ids = np.asarray(self.action_tokenizer.token_ids(actions.reshape(-1)), np.int64)
        prompt_ids = list(self.base_tokenizer(build_prompt(lang).rstrip()))
        if prompt_ids[-1] != self.empty_token_id:
            prompt_ids.append(self.empty_token_id)
        input_ids = np.concatenate([np.asarray(prompt_ids, np.int64), act_ids, [C.STOP_INDEX]])
        labels = input_ids.copy()
        labels[: -(act_ids.size + 1)] = IGNORE_INDEX
        if not self.predict_stop_token:
            labels[-1] = IGNORE_INDEX
        out = dict(image=np.asarray(img), input_ids=torch.from_numpy(input_ids), labels=torch.from_numpy(labels), dataset_name=name, actions=actions)
        if self.use_wrist_image:
            out["image_wrist"] = [np.asarray(obs[k][0]) for k in obs.keys() if k.startswith("image_") and ("wrist" in k or "gripper" in k)]
        if self.use_proprio and "proprio" in obs:
            out["proprio"] = obs["proprio"]
        elif self.use_proprio and b"ur5e" in name_b:
            out["proprio"] = obs["joint_positions"]
        return out


class EpisodeDataset:
    """Drop-in for RLDSDataset (datasets.py:100-196): same constructor arguments, `dataset_statistics`, `__len__`, `__iter__` over
    `batch_transform` outputs; infinite in train mode (the RLDS loader repeats implicitly, finetune.py:968-971).  `train=False` reads
    the `val/` sub-directory of the store (the builder's "val" split, rlds/dataset.py:238), one finite pass; statistics always cover
    both splits.  `shuffle=False` / `repeat=False` give a sequential / single pass over the chosen split."""

    def __init__(self, data_root_dir: Path, data_mix: str, batch_transform: RLDSBatchTransform, resize_resolution: Tuple[int, int] = (224, 224),
                 shuffle_buffer_size: int = 256_000, train: bool = True, image_aug: bool = False, *, seed: int = 0, rank: int = 0, world_size: int = 1,
                 shuffle: bool = True, repeat: Optional[bool] = None):
        if data_mix not in OXE_DATASET_CONFIGS:
            raise KeyError(f"dataset `{data_mix}` is not configured (known: {sorted(OXE_DATASET_CONFIGS)})")
        self.cfg = OXE_DATASET_CONFIGS[data_mix]
        self.data_root_dir, self.data_mix, self.batch_transform = Path(data_root_dir), data_mix, batch_transform
        self.resize_resolution, self.train, self.image_aug = tuple(resize_resolution), train, image_aug
        self.seed, self.rank, self.world_size = seed, rank, world_size
        self.shuffle, self.repeat = shuffle, (train if repeat is None else repeat)    # the validation loader is one finite pass
        if "aloha" in data_mix:
            self.load_camera_views = ("primary", "left_wrist", "right_wrist")
        else:
            self.load_camera_views = ("primary", "wrist")
        train_eps = list_episodes(self.data_root_dir, data_mix)
        val_eps = list_episodes(self.data_root_dir, data_mix, "val") if (not train or (self.data_root_dir / data_mix / "val").is_dir()) else []
        all_trajs = [restructure(ep, self.cfg, self.load_camera_views) for ep in train_eps + val_eps]      # statistics over split="all" (rlds/dataset.py:213)
        trajs = all_trajs[: len(train_eps)] if train else all_trajs[len(train_eps):]
        stats = get_dataset_statistics(all_trajs, (data_mix, str(self.cfg["state_obs_keys"]), self.cfg["standardize"], str(len(all_trajs))),
                                       save_dir=self.data_root_dir / data_mix)
        absolute_mask, norm_mask = action_masks(self.cfg["action_encoding"])
        stats = {**stats, "action": {**stats["action"], "mask": norm_mask}}
        self.dataset_statistics = {data_mix: stats}
        self.future = C.NUM_ACTIONS_CHUNK - 1
        self.trajs, index = [], []
        for t in trajs:
            if not t["language_instruction"]:           # skip_unlabeled=True
                continue
            t = normalize_action_and_proprio(t, stats, C.ACTION_PROPRIO_NORMALIZATION_TYPE)
            obs_idx, act_idx, _ = chunk_indices(t["action"].shape[0], 1, self.future)
            t["act_idx"] = act_idx
            index.extend((len(self.trajs), int(s)) for s in obs_idx[:, 0])
            self.trajs.append(t)
        self.index = np.asarray(index, np.int64).reshape(-1, 2)
        self.dataset_length = int(self.index.shape[0])

    def __len__(self) -> int:
        return self.dataset_length

    def frame(self, traj_i: int, step: int) -> Dict[str, Any]:
        """One flattened frame in the layout `dataset.as_numpy_iterator()` hands to RLDSBatchTransform (window_size = 1)."""
        t = self.trajs[traj_i]
        obs = {k: (np.zeros((1, *self.resize_resolution, 3), np.uint8) if v is None else v[step:step + 1])   # stored size; the collator resizes
               for k, v in t["observation"].items() if k.startswith("image_")}
        obs["proprio"] = t["observation"]["proprio"][step:step + 1]
        return {"observation": obs, "task": {"language_instruction": t["language_instruction"].encode()}, "action": t["action"][t["act_idx"][step]],
                "dataset_name": self.data_mix.encode()}

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        epoch = 0
        while True:
            order = np.random.default_rng([self.seed, epoch]).permutation(self.dataset_length) if self.shuffle else np.arange(self.dataset_length)
            for j in order[self.rank::self.world_size]:
                yield self.batch_transform(self.frame(int(self.index[j, 0]), int(self.index[j, 1])))
            if not self.repeat:
                return
            epoch += 1


@dataclass
class DeviceCollator:
    """PaddedCollatorForActionPrediction (prismatic/util/data_utils.py:95-156) with the image work on the GPU: the uint8 frames of the
    batch are copied to HBM once and `ops.image_augment` writes `pixel_values` [B, 6 * n_images, 224, 224] bf16 (primary image first,
    then the wrist images, per image DINOv2-normalised channels then SigLIP-normalised ones -- the reference's channel order)."""
    model_max_length: int
    pad_token_id: int
    device: Any = "cuda:0"
    image_aug: bool = False
    augment_kwargs: Optional[Dict[str, Any]] = None
    seed: int = 0
    image_size: int = 224
    padding_side: str = "right"
    _rng: Any = field(default=None, repr=False)

    def __call__(self, instances: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
        from .... import ops      # the HIP library: required (no host fallback for the pixel path)

        assert self.padding_side == "right", f"Invalid Tokenizer `{self.padding_side = }`"
        if self._rng is None:
            self._rng = np.random.default_rng(self.seed)
        pad = torch.nn.utils.rnn.pad_sequence
        input_ids = pad([i["input_ids"] for i in instances], batch_first=True, padding_value=self.pad_token_id)[:, : self.model_max_length]
        labels = pad([i["labels"] for i in instances], batch_first=True, padding_value=IGNORE_INDEX)[:, : self.model_max_length]
        frames = [[i["image"], *i.get("image_wrist", [])] for i in instances]
        n_img = len(frames[0])
        flat = np.ascontiguousarray(np.stack([f for fr in frames for f in fr]))
        params = sample_augment_params(self._rng, flat.shape[0], self.augment_kwargs) if self.image_aug else identity_augment_params(flat.shape[0])
        mask = augment_ops_mask(self.augment_kwargs) if self.image_aug else 0
        dev_frames = torch.from_numpy(flat).to(self.device, non_blocking=True)
        if tuple(dev_frames.shape[1:3]) != (self.image_size, self.image_size):
            # decode_and_resize (rlds/obs_transforms.py:48-99 -> dlimp resize_image: lanczos3, antialias, round, uint8) before the augmentation
            from .... import image_prep

            spans = [tuple(torch.from_numpy(a).to(self.device) for a in image_prep.lanczos3_spans(n, self.image_size)) for n in dev_frames.shape[1:3]]
            dev_frames = ops.image_resize(dev_frames, spans[0], spans[1])
        pv = ops.image_augment(dev_frames, torch.from_numpy(params).to(self.device, non_blocking=True), ops_mask=mask, out_size=self.image_size)
        out = dict(pixel_values=pv.view(len(instances), 6 * n_img, pv.shape[-2], pv.shape[-1]), input_ids=input_ids,
                   attention_mask=input_ids.ne(self.pad_token_id), labels=labels,
                   actions=torch.stack([torch.from_numpy(np.copy(i["actions"])) for i in instances]))
        out["proprio"] = torch.Tensor(np.squeeze(np.stack([i["proprio"] for i in instances]))) if "proprio" in instances[0] else None
        if "dataset_name" in instances[0]:
            out["dataset_names"] = [i["dataset_name"] for i in instances]
        return out


def batches(dataset: EpisodeDataset, collator: DeviceCollator, batch_size: int) -> Iterator[Dict[str, Any]]:
    """DataLoader(train_dataset, batch_size, collate_fn=collator, num_workers=0) (finetune.py:1010-1016)."""
    buf: List[Dict[str, Any]] = []
    for sample in dataset:
        buf.append(sample)
        if len(buf) == batch_size:
            yield collator(buf)
            buf = []
    if buf:
        yield collator(buf)
