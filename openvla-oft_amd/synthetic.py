"""Synthetic LIBERO/ALOHA-shaped batches in the layout the reference's data path hands to run_forward_pass.

Mirrors (no TensorFlow / RLDS here -- SURVEY.md section 8d):
  * RLDSBatchTransform      prismatic/vla/datasets/datasets.py:36-97   (prompt ids | action ids | stop; labels = -100 on the prompt)
  * PaddedCollatorForActionPrediction  prismatic/util/data_utils.py:95-156 (right padding with 32000 / -100, wrist images
    concatenated on the channel dim, actions stacked, proprio squeezed)
  * ActionTokenizer.__call__           prismatic/vla/action_tokenizer.py:38-47 (ids = 32000 - digitize(clip(a), linspace(-1,1,256)))
"""
from __future__ import annotations

import numpy as np
import torch

IGNORE_INDEX = -100
PAD_TOKEN_ID = 32000
STOP_INDEX = 2
EMPTY_TOKEN = 29871  # '' after "Out:" (modeling_prismatic.py:974)


def action_token_ids(actions: np.ndarray, vocab_size: int = 32000, n_bins: int = 256) -> np.ndarray:
    bins = np.linspace(-1, 1, n_bins)
    return vocab_size - np.digitize(np.clip(actions, -1.0, 1.0), bins)


def collate(instances, pad_token_id: int = PAD_TOKEN_ID, model_max_length: int = 2048):
    """prismatic/util/data_utils.py:102-156"""
    ids = [torch.as_tensor(i["input_ids"]) for i in instances]
    labs = [torch.as_tensor(i["labels"]) for i in instances]
    input_ids = torch.nn.utils.rnn.pad_sequence(ids, batch_first=True, padding_value=pad_token_id)[:, :model_max_length]
    labels = torch.nn.utils.rnn.pad_sequence(labs, batch_first=True, padding_value=IGNORE_INDEX)[:, :model_max_length]
    pv = torch.stack([torch.as_tensor(i["pixel_values"]) for i in instances])
    if "pixel_values_wrist" in instances[0]:
        pv = torch.cat((pv, torch.stack([torch.as_tensor(i["pixel_values_wrist"]) for i in instances])), dim=1)
    out = dict(pixel_values=pv, input_ids=input_ids, attention_mask=input_ids.ne(pad_token_id), labels=labels,
               actions=torch.stack([torch.from_numpy(np.copy(i["actions"])) for i in instances]))
    out["proprio"] = torch.Tensor(np.squeeze(np.stack([i["proprio"] for i in instances]))) if "proprio" in instances[0] else None
    return out


def make_instance(rng: np.random.Generator, *, prompt_len: int, chunk: int, action_dim: int, proprio_dim: int, num_images: int,
                  image_size: int = 224):
    """One RLDSBatchTransform-shaped sample: [BOS, prompt..., '' ] + action ids + stop."""
    actions = rng.uniform(-1, 1, size=(chunk, action_dim)).astype(np.float32)
    prompt = np.concatenate([[1], rng.integers(3, 31743, size=prompt_len - 2), [EMPTY_TOKEN]]).astype(np.int64)
    act_ids = action_token_ids(actions.reshape(-1)).astype(np.int64)
    input_ids = np.concatenate([prompt, act_ids, [STOP_INDEX]])
    labels = input_ids.copy()
    labels[: -(act_ids.size + 1)] = IGNORE_INDEX
    inst = dict(input_ids=input_ids, labels=labels, actions=actions,
                pixel_values=rng.standard_normal((6, image_size, image_size), dtype=np.float32),
                proprio=rng.uniform(-1, 1, size=(1, proprio_dim)).astype(np.float32))
    if num_images > 1:
        inst["pixel_values_wrist"] = rng.standard_normal((6 * (num_images - 1), image_size, image_size), dtype=np.float32)
    return inst


def make_batch(batch_size: int = 8, *, seed: int = 0, prompt_lens=None, chunk: int = 8, action_dim: int = 7, proprio_dim: int = 8,
               num_images: int = 2, image_size: int = 224):
    """Tp = 38 with two of eight rows at Tp = 34 (right padded) to exercise the mask."""
    rng = np.random.default_rng(seed)
    if prompt_lens is None:
        prompt_lens = [38] * batch_size
        for i in (1, 5):
            if i < batch_size and batch_size > 2:
                prompt_lens[i] = 34
    inst = [make_instance(rng, prompt_len=tp, chunk=chunk, action_dim=action_dim, proprio_dim=proprio_dim, num_images=num_images,
                          image_size=image_size) for tp in prompt_lens]
    return collate(inst)


def write_synthetic_episodes(root, dataset_name: str = "libero_spatial_no_noops", n_episodes: int = 4, *, seed: int = 0, min_len: int = 40,
                             max_len: int = 60, image_size: int = 224, unlabeled_every: int = 0):
    """A small LIBERO-shaped episode store (the layout of prismatic/vla/datasets/rlds_free.py): smooth random frames, 8-d state, 7-d
    raw actions with the gripper in {-1, +1} as the LIBERO RLDS builder stores it.  Stands in for the real demonstrations, which
    cannot be fetched offline."""
    import importlib
    rlds_free = importlib.import_module(__package__ + ".prismatic.vla.datasets.rlds_free")
    rng = np.random.default_rng(seed)
    tasks = ["pick up the black bowl and place it on the plate", "put the cream cheese in the bowl", "turn on the stove", "open the top drawer"]
    for e in range(n_episodes):
        T = int(rng.integers(min_len, max_len + 1))
        base = rng.integers(0, 256, size=(2, image_size // 8, image_size // 8, 3)).astype(np.float32)
        frames = []
        for cam in range(2):
            up = np.kron(base[cam], np.ones((8, 8, 1), np.float32))
            drift = rng.normal(0, 6, size=(T, 1, 1, 3)).astype(np.float32).cumsum(axis=0)
            frames.append(np.clip(up[None] + drift + rng.normal(0, 3, size=(T, image_size, image_size, 3)), 0, 255).astype(np.uint8))
        action = rng.normal(0, 0.3, size=(T, 7)).astype(np.float32)
        action[:, -1] = np.where(np.sin(np.arange(T) / 7.0 + e) > 0, 1.0, -1.0)
        state = rng.normal(0, 0.5, size=(T, 8)).astype(np.float32).cumsum(axis=0) * 0.05
        lang = "" if (unlabeled_every and (e + 1) % unlabeled_every == 0) else tasks[e % len(tasks)]
        rlds_free.write_episode(root, dataset_name, e, arrays={"image": frames[0], "wrist_image": frames[1], "state": state, "action": action},
                                language_instruction=lang)
    return root
