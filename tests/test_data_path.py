"""the RLDS-free training data path (prismatic/vla/datasets/rlds_free.py + ovla_image_augment).

CPU tests: host logic against hand-computed cases and the loop restatements in oracle/data_oracle.py.
GPU tests: the HIP augmentation kernels bit-for-bit against the oracle's float32 restatement, through the C-ABI; the collator and
the fine-tune driver end to end on an episode store.
TensorFlow / dlimp are absent: PARITY UNPINNED against them (see oracle/data_oracle.py); what is pinned here is our own restatement."""
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import data_oracle as do  # noqa: E402

load = importlib.import_module
D = load("openvla-oft_amd.prismatic.vla.datasets")
R = load("openvla-oft_amd.prismatic.vla.datasets.rlds_free")
C = load("openvla-oft_amd.prismatic.vla.constants")
AT = load("openvla-oft_amd.prismatic.vla.action_tokenizer")
synth = load("openvla-oft_amd.synthetic")
BF = torch.bfloat16


class Tok:
    vocab_size = 32000

    def __call__(self, text):
        return [1] + [3 + (sum(map(ord, w)) * 7919) % 30000 for w in text.split()]


# ---------------------------------------------------------------------------------------------------------------------------- CPU
def test_chunk_indices_known_answers_and_loop_restatement():
    """rlds/traj_transforms.py:14-59"""
    o, a, p = R.chunk_indices(5, 1, 2)
    assert o.tolist() == [[0], [1], [2]] and a.tolist() == [[0, 1, 2], [1, 2, 3], [2, 3, 4]] and p.all()
    o, a, p = R.chunk_indices(3, 2, 1)      # a window of 2 observations: the first one is padding at t = 0
    assert o.tolist() == [[0, 0], [0, 1]] and p.tolist() == [[False, True], [True, True]] and a.tolist() == [[0, 0, 1], [0, 1, 2]]
    for T, W, F in [(1, 1, 0), (9, 1, 7), (8, 1, 7), (7, 1, 7), (3, 1, 7), (20, 3, 4), (40, 1, 24)]:
        o, a, p = R.chunk_indices(T, W, F)
        lo, la, lp = do.chunk_act_obs_loop(T, W, F)
        assert o.tolist() == lo and a.tolist() == la and p.tolist() == lp, (T, W, F)
        assert o.shape == (max(T - F, 0), W) and a.shape == (max(T - F, 0), W + F)


def test_normalize_action_and_proprio():
    """rlds/utils/data_utils.py:52-94: q01 -> -1, q99 -> (almost) +1, clipping, masked dimensions untouched, min == max -> 0."""
    rng = np.random.default_rng(0)
    act = rng.normal(0, 1, (50, 7)).astype(np.float32)
    act[:, 4] = 0.25                                        # an unused dimension
    pro = rng.normal(0, 2, (50, 8)).astype(np.float32)
    trajs = [{"action": act, "observation": {"proprio": pro}}]
    md = R.get_dataset_statistics(trajs, ("x",))
    md["action"]["mask"] = [True] * 6 + [False]
    for kind, name in ((C.NormalizationType.BOUNDS_Q99, "bounds_q99"), (C.NormalizationType.BOUNDS, "bounds"), (C.NormalizationType.NORMAL, "normal")):
        t = R.normalize_action_and_proprio({"action": act.copy(), "observation": {"proprio": pro.copy()}}, md, kind)
        assert np.array_equal(t["action"], do.normalize_loop(act, md["action"], name)), kind
        assert np.array_equal(t["observation"]["proprio"], do.normalize_loop(pro, md["proprio"], name)), kind
        assert np.array_equal(t["action"][:, 6], act[:, 6])                 # the gripper is not normalised
        if name != "normal":
            assert np.all(t["action"][:, 4] == 0) and t["action"][:, :4].min() == -1 and t["action"][:, :4].max() == 1
    one = {"action": np.array([[md["action"]["q01"][0]] + [0.0] * 6, [md["action"]["q99"][0]] + [0.0] * 6], np.float32), "observation": {"proprio": pro[:2]}}
    t = R.normalize_action_and_proprio(one, md, C.NormalizationType.BOUNDS_Q99)
    assert t["action"][0, 0] == -1 and abs(t["action"][1, 0] - 1) < 1e-6


def test_gripper_and_libero_transforms():
    """rlds/oxe/transforms.py:833-847, rlds/utils/data_utils.py:97-153"""
    a = np.zeros((4, 7), np.float32)
    a[:, -1] = [-1.0, 1.0, 0.5, -0.2]
    st = np.arange(32, dtype=np.float32).reshape(4, 8)
    t = R.libero_dataset_transform({"action": a, "observation": {"state": st}})
    assert t["action"][:, -1].tolist() == [1.0, 0.0, 0.5, 1.0]
    assert np.array_equal(t["observation"]["EEF_state"], st[:, :6]) and np.array_equal(t["observation"]["gripper_state"], st[:, -2:])
    assert R.binarize_gripper_actions(np.array([1.0, 0.5, 0.0, 0.3, 0.97, 0.4], np.float32)).tolist() == [1, 0, 0, 1, 1, 0.4000000059604645]
    assert R.rel2abs_gripper_actions(np.array([0, 0, 1, 0, -1, 0], np.float32)).tolist() == [1, 1, 0, 0, 1, 1]
    assert R.rel2abs_gripper_actions(np.zeros(3, np.float32)).tolist() == [1, 1, 1]
    assert R.action_masks("EEF_POS") == ([False] * 6 + [True], [True] * 6 + [False]) and R.action_masks("JOINT_POS_BIMANUAL") == ([True] * 14, [True] * 14)


def test_dataset_statistics_schema_and_cache(tmp_path):
    """rlds/utils/data_utils.py:176-258: keys, values, the sha256-named cache file (second call reads it)."""
    rng = np.random.default_rng(1)
    trajs = [{"action": rng.normal(size=(n, 7)).astype(np.float32), "observation": {"proprio": rng.normal(size=(n, 8)).astype(np.float32)}} for n in (5, 9)]
    md = R.get_dataset_statistics(trajs, ("a", "b"), save_dir=tmp_path)
    allact = np.concatenate([t["action"] for t in trajs])
    assert set(md) == {"action", "proprio", "num_transitions", "num_trajectories"} and md["num_transitions"] == 14 and md["num_trajectories"] == 2
    assert set(md["action"]) == {"mean", "std", "max", "min", "q01", "q99"}
    assert md["action"]["q99"] == np.quantile(allact, 0.99, axis=0).tolist() and md["action"]["std"] == allact.std(0).tolist()
    import hashlib
    f = tmp_path / f"dataset_statistics_{hashlib.sha256(b'ab').hexdigest()}.json"
    assert f.exists() and json.loads(f.read_text()) == md
    f.write_text(json.dumps({**md, "num_trajectories": 99}))
    assert R.get_dataset_statistics([], ("a", "b"), save_dir=tmp_path)["num_trajectories"] == 99


def test_batch_transform_ids_and_labels():
    """datasets.py:36-97"""
    tok = Tok()
    at = AT.ActionTokenizer(tok)
    rng = np.random.default_rng(2)
    actions = rng.uniform(-1.2, 1.2, (8, 7)).astype(np.float32)
    frame = {"observation": {"image_primary": np.zeros((1, 8, 8, 3), np.uint8), "image_wrist": np.ones((1, 8, 8, 3), np.uint8), "proprio": np.ones((1, 8), np.float32)},
             "task": {"language_instruction": b"Pick Up The Bowl"}, "action": actions, "dataset_name": b"libero_spatial_no_noops"}
    for stop in (True, False):
        out = R.RLDSBatchTransform(at, tok, use_wrist_image=True, use_proprio=True, predict_stop_token=stop)(frame)
        prompt = tok("In: What action should the robot take to pick up the bowl?\nOut:")
        ids, labels = do.batch_transform_ids(prompt, at.token_ids(actions.reshape(-1)), predict_stop_token=stop)
        assert out["input_ids"].tolist() == ids and out["labels"].tolist() == labels
        assert (out["labels"] != -100).sum().item() == 56 + int(stop) and out["input_ids"][-1].item() == 2
        assert len(out["image_wrist"]) == 1 and out["image_wrist"][0].max() == 1 and out["proprio"].shape == (1, 8)
    assert R.build_prompt("open the drawer") == "In: What action should the robot take to open the drawer?\nOut: "


def test_episode_dataset_frames_and_rank_partition(tmp_path):
    synth.write_synthetic_episodes(tmp_path, n_episodes=4, unlabeled_every=4, min_len=20, max_len=30, image_size=32)
    tok = Tok()
    bt = R.RLDSBatchTransform(AT.ActionTokenizer(tok), tok, use_wrist_image=True, use_proprio=True)
    ds = D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, resize_resolution=(32, 32), shuffle=False, repeat=False)
    eps = R.list_episodes(tmp_path, "libero_spatial_no_noops")
    assert len(ds) == sum(len(e) - 7 for e in eps[:3])                      # the unlabeled episode is skipped, T - 7 frames each
    stats = ds.dataset_statistics["libero_spatial_no_noops"]
    assert stats["num_trajectories"] == 4 and stats["action"]["mask"] == [True] * 6 + [False]
    # frame (episode 1, step 5): the chunk is the normalised actions 5..12 of that episode, the images are frame 5
    raw = R.libero_dataset_transform({"action": np.asarray(eps[1]["action"]), "observation": {"state": np.asarray(eps[1]["state"])}})["action"]
    want = do.normalize_loop(raw, stats["action"], "bounds_q99")[5:13]
    fr = ds.frame(1, 5)
    assert np.array_equal(fr["action"], want) and np.array_equal(fr["observation"]["image_primary"][0], eps[1]["image"][5])
    assert np.array_equal(fr["observation"]["image_wrist"][0], eps[1]["wrist_image"][5]) and fr["observation"]["proprio"].shape == (1, 8)
    samples = list(ds)
    assert len(samples) == len(ds) and samples[0]["actions"].shape == (8, 7)
    # training order: every epoch is a permutation; ranks take disjoint strides of it
    seen = []
    for rank in range(2):
        it = iter(D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, resize_resolution=(32, 32), train=True, seed=3, rank=rank, world_size=2))
        seen.append([tuple(next(it)["actions"].reshape(-1)[:3]) for _ in range(len(ds) // 2)])
    assert len(set(seen[0]) & set(seen[1])) == 0 and len(set(seen[0]) | set(seen[1])) == 2 * (len(ds) // 2)
    with pytest.raises(KeyError):
        D.RLDSDataset(tmp_path, "not_a_dataset", bt)
    # the validation split is the store's val/ directory (the builder's "val" split); statistics cover both
    with pytest.raises(ValueError, match="Unknown split 'val'"):
        D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, resize_resolution=(32, 32), train=False)
    synth.write_synthetic_episodes(tmp_path / "v", n_episodes=1, seed=9, min_len=20, max_len=20, image_size=32)
    (tmp_path / "v" / "libero_spatial_no_noops").rename(tmp_path / "libero_spatial_no_noops" / "val")
    val = D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, resize_resolution=(32, 32), train=False)
    assert len(val) == 13 and len(list(val)) == 13 and val.dataset_statistics["libero_spatial_no_noops"]["num_trajectories"] == 5
    assert len(D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, resize_resolution=(32, 32))) == len(ds)


def test_augment_parameter_distributions():
    """datasets.py:159-174 kwargs through dlimp's op definitions: 90 %-area square crops inside the image, jitter ranges."""
    p = R.sample_augment_params(np.random.default_rng(0), 4000)
    h, w = p[:, 2] - p[:, 0], p[:, 3] - p[:, 1]
    assert np.allclose(h * w, 0.9, atol=1e-5) and np.allclose(h, w, atol=1e-6) and p[:, :2].min() >= 0 and p[:, 2:4].max() <= 1 + 1e-6
    assert p[:, 0].max() > 0.04 and abs(p[:, 0].mean() - (1 - np.sqrt(0.9)) / 2) < 2e-3
    for col, lo, hi in ((4, -0.2, 0.2), (5, 0.8, 1.2), (6, 0.8, 1.2), (7, -0.05, 0.05)):
        assert lo <= p[:, col].min() < lo + 0.01 and hi - 0.01 < p[:, col].max() <= hi
    assert R.augment_ops_mask() == 31 and R.augment_ops_mask(dict(augment_order=["random_resized_crop", "random_hue"])) == 17
    ident = R.identity_augment_params(2)
    assert ident.tolist() == [[0, 0, 1, 1, 0, 1, 1, 0]] * 2
    with pytest.raises(NotImplementedError):
        R.sample_augment_params(np.random.default_rng(0), 1, dict(augment_order=["random_hue", "random_brightness"], random_hue=[0.1], random_brightness=[0.1]))


def test_oracle_image_ops_known_answers():
    """Hand-computable cases that pin what the oracle (and, through it, the HIP kernel) means by each op."""
    F = np.float32
    img = np.random.default_rng(3).integers(0, 256, (16, 16, 3), dtype=np.uint8)
    ident = np.array([0, 0, 1, 1, 0, 1, 1, 0], F)
    assert np.array_equal(do.augment_image(img, ident, do.CROP, out=16), img)                     # the full box at the same size is the identity
    assert np.array_equal(do.augment_image(img, ident, 0, out=16), img)
    red = np.zeros((4, 4, 3), F); red[..., 0] = 1
    assert np.array_equal(do.adjust_hue(red, 1 / 3)[0, 0].round(5), [0, 1, 0]) and np.array_equal(do.adjust_hue(red, -1 / 3)[0, 0].round(5), [0, 0, 1])
    x = np.array([[[0.8, 0.4, 0.2]]], F)
    assert np.allclose(do.adjust_saturation(x, 0.0), 0.8) and np.allclose(do.adjust_saturation(x, 1.0), x, atol=1e-6)
    assert np.allclose(do.adjust_hue(x, 0.0), x, atol=1e-6)
    h, s, v = do.rgb_to_hsv(x[..., 0], x[..., 1], x[..., 2])
    assert np.allclose([h[0, 0], s[0, 0], v[0, 0]], [(1 / 3) / 6, 0.75, 0.8], atol=1e-6)            # hue 20 degrees
    y = np.stack([np.full((2, 2), 0.2, F), np.full((2, 2), 0.5, F), np.array([[0.0, 1.0], [0.0, 1.0]], F)], -1)
    c0 = do.adjust_contrast(y, 0.0)
    assert np.allclose(c0[..., 0], 0.2) and np.allclose(c0[..., 2], 0.5)                             # factor 0 -> the channel means
    assert np.allclose(do.adjust_contrast(y, 2.0)[..., 2], [[-0.5, 1.5], [-0.5, 1.5]])
    half = do.crop_and_resize(np.arange(25, dtype=F).reshape(5, 5, 1).repeat(3, 2) / F(25), (0, 0, 0.5, 0.5), 3)
    assert np.allclose(half[..., 0] * 25, [[0, 1, 2], [5, 6, 7], [10, 11, 12]])                      # box (0, 0, .5, .5) of a 5x5 ramp at 3x3: rows/cols 0, 1, 2
    bright = do.augment_image(np.full((2, 2, 3), 100, np.uint8), np.array([0, 0, 1, 1, 0.2, 1, 1, 0], F), do.BRIGHTNESS, out=2)
    assert np.all(bright == int(F(F(100) / F(255) + F(0.2)) * F(255)))
    pv = do.pixel_values(np.full((2, 2, 3), 255, np.uint8))
    assert np.allclose(pv[:3, 0, 0], (1 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225]), atol=1e-6) and np.allclose(pv[3:], 1.0)


def test_lanczos3_spans():
    """scale_and_translate_op.cc ComputeSpansCore: vectorised host code == element loop; weights of a span sum to 1; same-size is
    the identity after rounding; kernel values at the known points."""
    ip = load("openvla-oft_amd.image_prep")
    for a, b in [(256, 224), (224, 224), (128, 224), (480, 224), (640, 224), (7, 5), (3, 9)]:
        s, w = ip.lanczos3_spans(a, b)
        s2, w2 = do.lanczos3_spans_loop(a, b)
        assert np.array_equal(s, s2) and np.array_equal(w, w2) and np.allclose(w.sum(1), 1, atol=1e-6), (a, b)
        assert w.shape[1] == min(2 * int(np.ceil(3 * max(a / b, 1))) + 1, a) and s.min() >= 0 and (s + 1 <= a).all()
    assert do.lanczos3_kernel(0) == 1 and do.lanczos3_kernel(3.5) == 0 and abs(do.lanczos3_kernel(1)) < 1e-7 and abs(do.lanczos3_kernel(2)) < 1e-7
    assert abs(do.lanczos3_kernel(0.5) - 3 * 1 * 0.5 / (np.pi ** 2 * 0.25)) < 1e-6
    img = np.random.default_rng(0).integers(0, 256, (24, 24, 3), dtype=np.uint8)
    assert np.array_equal(do.resize_lanczos3(img, 24, 24), img)
    assert np.all(do.resize_lanczos3(np.full((32, 40, 3), 77, np.uint8), 16, 24) == 77)


# ---------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ops():
    return load("openvla-oft_amd.ops")


def _frames(rng, n, H, W):
    base = rng.integers(0, 256, (n, H // 8 + 1, W // 8 + 1, 3)).astype(np.float32)
    up = np.stack([np.kron(b, np.ones((8, 8, 1), np.float32))[:H, :W] for b in base])
    img = np.clip(up + rng.normal(0, 12, up.shape), 0, 255).astype(np.uint8)
    img[0, : H // 3] = 0                      # black, white and grey regions: zero chroma / zero range branches
    img[0, H // 3: H // 2] = 255
    img[-1, :, : W // 4] = 128
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [31, 0, 1, 2, 4, 8, 16, 3, 28, 13])
@pytest.mark.parametrize("H,W,out", [(224, 224, 224), (200, 256, 224), (64, 64, 64)])
def test_image_augment_matches_oracle_bit_for_bit(ops, dev, mask, H, W, out):
    if not (mask & 1) and (H != out or W != out):
        pytest.skip("without crop-and-resize the frames must already be out x out")
    rng = np.random.default_rng(mask * 131 + H)
    n = 5
    img = _frames(rng, n, H, W)
    params = R.sample_augment_params(rng, n)
    params[1] = R.identity_augment_params(1)[0]                      # an identity row among the random ones
    params[2, 4:] = [0.2, 1.2, 1.2, 0.05]                            # the range ends
    params[3, 4:] = [-0.2, 0.8, 0.8, -0.05]
    got = ops.image_augment(torch.from_numpy(img).to(dev), torch.from_numpy(params).to(dev), ops_mask=mask, out_size=out)
    want = np.stack([do.pixel_values(do.augment_image(img[i], params[i], mask, out)) for i in range(n)])
    want = torch.from_numpy(want).to(BF)
    assert got.shape == (n, 6, out, out) and got.dtype == BF
    neq = (got.cpu().view(torch.int16) != want.view(torch.int16)).sum().item()
    assert neq == 0, f"{neq} of {want.numel()} bf16 values differ (mask {mask})"


@pytest.mark.gpu
def test_image_augment_eval_path_equals_image_prep(ops, dev):
    """ops_mask = 0 == ovla_image_prep without the crop (the inference-side kernel), bit for bit."""
    img = torch.from_numpy(_frames(np.random.default_rng(5), 3, 224, 224)).to(dev)
    a = ops.image_augment(img, torch.from_numpy(R.identity_augment_params(3)).to(dev), ops_mask=0)
    b = ops.image_prep(img, crop=False)
    assert torch.equal(a.view(1, 18, 224, 224), b)


@pytest.mark.gpu
def test_image_augment_rejects_bad_arguments(ops, dev):
    _lib = load("openvla-oft_amd._lib")
    img = torch.zeros((2, 100, 100, 3), dtype=torch.uint8, device=dev)
    prm = torch.from_numpy(R.identity_augment_params(2)).to(dev)
    with pytest.raises(_lib.OvlaError, match="without crop-and-resize"):
        ops.image_augment(img, prm, ops_mask=2)
    with pytest.raises(_lib.OvlaError, match="unknown bits"):
        ops.image_augment(img, prm, ops_mask=64)


@pytest.mark.gpu
def test_device_collator_batch_layout(tmp_path, ops, dev):
    """PaddedCollatorForActionPrediction's batch dict (prismatic/util/data_utils.py:95-156) with pixel_values made on the device:
    primary image channels first, then the wrist image's; same parameters drawn from the same seed reproduce the oracle."""
    synth.write_synthetic_episodes(tmp_path, n_episodes=2, min_len=12, max_len=14)
    tok = Tok()
    bt = R.RLDSBatchTransform(AT.ActionTokenizer(tok), tok, use_wrist_image=True, use_proprio=True)
    ds = D.RLDSDataset(tmp_path, "libero_spatial_no_noops", bt, shuffle=False, repeat=False)
    inst = [s for _, s in zip(range(3), ds)]
    inst[1]["input_ids"], inst[1]["labels"] = inst[1]["input_ids"][2:], inst[1]["labels"][2:]          # a shorter row -> right padding
    batch = R.DeviceCollator(2048, 32000, device=dev, image_aug=True, seed=11)(inst)
    assert batch["pixel_values"].shape == (3, 12, 224, 224) and batch["pixel_values"].dtype == BF and batch["pixel_values"].is_cuda
    assert batch["input_ids"].shape == batch["labels"].shape == batch["attention_mask"].shape and batch["input_ids"][1, -2:].tolist() == [32000, 32000]
    assert batch["labels"][1, -2:].tolist() == [-100, -100] and not batch["attention_mask"][1, -1] and batch["attention_mask"][0].all()
    assert batch["actions"].shape == (3, 8, 7) and batch["proprio"].shape == (3, 8) and batch["dataset_names"] == [b"libero_spatial_no_noops"] * 3
    params = R.sample_augment_params(np.random.default_rng(11), 6)
    for i in range(3):
        for j, frame in enumerate([inst[i]["image"], inst[i]["image_wrist"][0]]):
            want = torch.from_numpy(do.pixel_values(do.augment_image(frame, params[2 * i + j], 31, 224))).to(BF)
            assert torch.equal(batch["pixel_values"][i, 6 * j: 6 * j + 6].cpu(), want), (i, j)
    ev = R.DeviceCollator(2048, 32000, device=dev, image_aug=False)(inst)
    assert torch.equal(ev["pixel_values"][0, :6].cpu(), torch.from_numpy(do.pixel_values(inst[0]["image"])).to(BF))
    # frames stored at another resolution are resized on the device first (dlimp resize_image), then augmented
    small = R.DeviceCollator(2048, 32000, device=dev, image_aug=True, seed=11, image_size=96)(inst)
    want = do.pixel_values(do.augment_image(do.resize_lanczos3(inst[0]["image"], 96, 96), params[0], 31, 96))
    assert small["pixel_values"].shape == (3, 12, 96, 96) and torch.equal(small["pixel_values"][0, :6].cpu(), torch.from_numpy(want).to(BF))


@pytest.mark.gpu
def test_finetune_reads_an_episode_store(tmp_path, dev):
    """finetune(cfg) with cfg.data_root_dir pointing at an episode store: the reference's dataset + collator wiring
    (finetune.py:981-1016) end to end on the reduced-size model; dataset_statistics.json lands next to the checkpoint."""
    from oracle import vla_oracle as vo

    ft, config_mod = load("openvla-oft_amd.vla_scripts.finetune"), load("openvla-oft_amd.config")
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=0).items()}
    mc = config_mod.VLAConfig.from_any(ocfg)
    side = mc.dino.image_size
    synth.write_synthetic_episodes(tmp_path / "data", n_episodes=3, min_len=12, max_len=16, image_size=side)
    synth.write_synthetic_episodes(tmp_path / "v", n_episodes=1, seed=9, min_len=12, max_len=12, image_size=side)
    (tmp_path / "v" / "libero_spatial_no_noops").rename(tmp_path / "data" / "libero_spatial_no_noops" / "val")
    cfg = ft.FinetuneConfig(run_root_dir=tmp_path / "runs", data_root_dir=tmp_path / "data", dataset_name="libero_spatial_no_noops", batch_size=2,
                            num_images_in_input=2, use_proprio=True, max_steps=3, save_freq=2, wandb_log_freq=1, image_aug=True, use_val_set=True, val_freq=2)
    lines = []
    hist = ft.finetune(cfg, model_config=mc, state_dict=sd, log=lines.append, tokenizer=Tok())
    assert len(hist["loss_value"]) == 3 and all(np.isfinite(hist["loss_value"]))
    # run_validation (finetune.py:678-760) at step 2: one pass over the val/ split (5 frames -> 3 batches), loss without gradients
    assert [s for s, _ in hist["val"]] == [2] and hist["val"][0][1]["val/num_batches"] == 3 and np.isfinite(hist["val"][0][1]["val/loss"])
    assert any("val/loss_value" in str(l) for l in lines)
    assert any("episode store" in str(l) for l in lines)
    ck = list((tmp_path / "runs").glob("*--2_chkpt"))[0]
    stats = json.loads((ck / "dataset_statistics.json").read_text())
    assert stats["libero_spatial_no_noops"]["action"]["mask"] == [True] * 6 + [False] and stats["libero_spatial_no_noops"]["num_trajectories"] == 4


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,oh,ow", [(256, 256, 224, 224), (128, 160, 224, 224), (480, 640, 224, 224), (224, 224, 224, 224), (50, 30, 20, 45)])
def test_image_resize_matches_oracle_bit_for_bit(ops, dev, H, W, oh, ow):
    """ovla_image_resize (lanczos3 + antialias, rows then columns, tf.round, uint8) == oracle/data_oracle.py resize_lanczos3."""
    ip = load("openvla-oft_amd.image_prep")
    img = _frames(np.random.default_rng(H + W), 2, H, W)
    spans = [tuple(torch.from_numpy(a).to(dev) for a in ip.lanczos3_spans(n_in, n_out)) for n_in, n_out in ((H, oh), (W, ow))]
    got = ops.image_resize(torch.from_numpy(img).to(dev), spans[0], spans[1]).cpu().numpy()
    for i in range(2):
        want = do.resize_lanczos3(img[i], oh, ow)
        assert np.array_equal(got[i], want), f"{(got[i] != want).sum()} of {want.size} bytes differ"
    if (H, W) == (oh, ow):
        assert np.array_equal(got, img)


@pytest.mark.gpu
def test_prepare_images_accepts_simulator_frames(ops, dev):
    """experiments/robot/openvla_utils.py:678-708 with 256 x 256 LIBERO simulator frames: JPEG round trip -> lanczos3 resize -> center crop
    (resize_image_for_policy :516-540, then crop_and_resize); host API and the all-device path (device_pixel_values) agree bit for bit, and
    both equal the oracle chain (libjpeg-turbo-pinned codec, TF-restated resize and crop)."""
    from oracle import jpeg_oracle as jo

    ip, utils = load("openvla-oft_amd.image_prep"), load("openvla-oft_amd.experiments.robot.openvla_utils")
    frames = list(_frames(np.random.default_rng(9), 2, 256, 256))
    cfg = type("Cfg", (), {"center_crop": True})()
    host = ip.prepare_images_for_vla(frames, cfg)
    assert all(h.shape == (224, 224, 3) and h.dtype == np.uint8 for h in host)
    assert np.array_equal(host[0], ip.center_crop_image(do.resize_lanczos3(jo.jpeg_roundtrip(frames[0]), 224, 224)))
    pv = utils.device_pixel_values(frames, cfg)
    want = torch.cat([ip.apply_transform(h) for h in host])[None].to(BF)
    assert torch.equal(pv.cpu(), want)
    assert np.array_equal(ip.resize_image_for_policy(frames[1], (112, 200)), do.resize_lanczos3(jo.jpeg_roundtrip(frames[1]), 112, 200))
    assert np.array_equal(ip.resize_image_for_policy(frames[1], (112, 200), jpeg=False), do.resize_lanczos3(frames[1], 112, 200))


@pytest.mark.gpu
def test_jpeg_roundtrip_kernel_is_bit_identical_to_libjpeg_turbo(dev):
    """ovla_jpeg_roundtrip (two launches: per-MCU colour conversion / chroma box filter / FDCT / quantise / dequantise / IDCT, then fancy
    upsampling + colour conversion) against libjpeg-turbo's own outputs (G12) and, at the sizes the policies see (256 x 256 LIBERO frames,
    480 x 640 ALOHA frames, a batch of 3), against the oracle that is pinned to them."""
    import importlib

    from oracle import jpeg_oracle as jo

    ops = importlib.import_module("openvla-oft_amd.ops")
    g = np.load(Path(__file__).resolve().parent / "golden" / "g12_jpeg_roundtrip.npz")
    for name in sorted(k[:-4] for k in g.files if k.endswith("__in")):
        img = torch.from_numpy(g[name + "__in"])[None].to(dev)
        for q in (95, 50, 100):
            if f"{name}__q{q}" in g.files:
                got = ops.jpeg_roundtrip(img, quality=q)[0].cpu().numpy()
                assert np.array_equal(got, g[f"{name}__q{q}"]), f"{name} q{q}: {int((got != g[f'{name}__q{q}']).sum())} bytes differ from libjpeg-turbo"
    rng = np.random.default_rng(5)
    for shape in ((3, 256, 256, 3), (2, 480, 640, 3), (1, 250, 243, 3)):
        smooth = np.clip(rng.normal(128, 50, (shape[0], shape[1] // 8 + 1, shape[2] // 8 + 1, 3)).repeat(8, 1).repeat(8, 2)[:, : shape[1], : shape[2]]
                         + rng.normal(0, 5, shape), 0, 255).astype(np.uint8)
        got = ops.jpeg_roundtrip(torch.from_numpy(smooth).to(dev)).cpu().numpy()
        want = np.stack([jo.jpeg_roundtrip(f) for f in smooth])
        assert np.array_equal(got, want), f"{shape}: {int((got != want).sum())} bytes differ"
        err = np.abs(got.astype(int) - smooth.astype(int))
        assert 0 < err.max() and err.mean() < 12, "a lossy codec at quality 95: close to, not equal to, the input (4:2:0 smears the block edges)"
    # resize_image_for_policy = round trip + lanczos3 antialias resize (the eval loops' 256 -> 224 path)
    ip = importlib.import_module("openvla-oft_amd.image_prep")
    frame = smooth[0]
    a, b = ip.resize_image_for_policy(frame, 224), ip.resize_image_for_policy(frame, 224, jpeg=False)
    assert a.shape == (224, 224, 3) and a.dtype == np.uint8 and not np.array_equal(a, b)
    with pytest.raises(Exception, match="quality"):
        ops.jpeg_roundtrip(torch.from_numpy(frame)[None].to(dev), quality=0)


@pytest.mark.parametrize("tag", ["libero", "nostop", "primary_only", "ur5e"])
def test_g20_batch_transform_matches_the_reference_class(tag):
    """The mirror's RLDSBatchTransform against the REFERENCE's own class executed in the build container (prismatic/vla/datasets/datasets.py:26-97 with its
    PurePromptBuilder, base_prompter.py:28-73; tests/golden/make_golden_batch_transform.py, shared stand-in tokenizer tests/duck_tokenizer.py): input ids,
    labels (IGNORE_INDEX masking arithmetic, predict_stop_token), which observation keys become wrist images, actions and proprio passthrough, the ur5e
    key names -- and the oracle's restatement of the id / label layout (oracle/data_oracle.py: batch_transform_ids)."""
    from tests.duck_tokenizer import DuckTokenizer, mirror_tokenizer

    g = dict(np.load(Path(__file__).resolve().parent / "golden" / "g20_ref_batch_transform.npz", allow_pickle=False))
    wrist, prop, stop = (bool(x) for x in g[f"{tag}.flags"])
    name, lang = bytes(g[f"{tag}.dataset_name"]), bytes(g[f"{tag}.language"])
    obs = {k[len(tag) + 5:]: v for k, v in g.items() if k.startswith(f"{tag}.obs.")}
    frame = {"dataset_name": name, "action": g[f"{tag}.action"], "observation": obs, "task": {"language_instruction": lang}}
    at = AT.ActionTokenizer(DuckTokenizer())
    out = R.RLDSBatchTransform(at, mirror_tokenizer, use_wrist_image=wrist, use_proprio=prop, predict_stop_token=stop)(frame)
    assert np.array_equal(out["input_ids"].numpy(), g[f"{tag}.input_ids"]) and np.array_equal(out["labels"].numpy(), g[f"{tag}.labels"])
    assert np.array_equal(np.asarray(out["actions"]), g[f"{tag}.actions"])
    assert np.array_equal(out["image"], g[f"{tag}.pixel_values"]), "primary frame (identity image transform on both sides)"
    if wrist:
        assert np.array_equal(np.concatenate(out["image_wrist"], 0), g[f"{tag}.pixel_values_wrist"])
    if prop:
        assert np.array_equal(np.asarray(out["proprio"]), g[f"{tag}.proprio"])
    # the prompt the mirror builds is the reference prompt builder's, token for token (get_prompt() strips the trailing space; the mirror tokenises the stripped text)
    assert mirror_tokenizer(R.build_prompt(lang.decode().lower()).rstrip()) == g[f"{tag}.prompt_ids"].tolist()
    ids, labels = do.batch_transform_ids(mirror_tokenizer(R.build_prompt(lang.decode().lower()).rstrip()), at.token_ids(g[f"{tag}.action"].reshape(-1)), predict_stop_token=stop)
    assert ids == g[f"{tag}.input_ids"].tolist() and labels == g[f"{tag}.labels"].tolist()
