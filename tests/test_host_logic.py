"""CPU tests of the host-side logic of the product package (no GPU, no compute kernels)."""
import importlib
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

load = importlib.import_module
G = Path(__file__).resolve().parent / "golden"


def test_constants_follow_reference_platform_rule(monkeypatch):
    C = load("openvla-oft_amd.prismatic.vla.constants")
    assert (C.IGNORE_INDEX, C.ACTION_TOKEN_BEGIN_IDX, C.STOP_INDEX) == (-100, 31743, 2)
    C.set_platform("aloha")
    assert (C.NUM_ACTIONS_CHUNK, C.ACTION_DIM, C.PROPRIO_DIM, C.ACTION_PROPRIO_NORMALIZATION_TYPE.value) == (25, 14, 14, "bounds")
    monkeypatch.setattr(sys, "argv", ["run_bridge_eval.py"])
    assert C.detect_robot_platform() == "BRIDGE"
    monkeypatch.setattr(sys, "argv", ["train.py"])
    assert C.detect_robot_platform() == "LIBERO"
    C.set_platform("libero")
    assert (C.NUM_ACTIONS_CHUNK, C.ACTION_DIM, C.PROPRIO_DIM) == (8, 7, 8)


def test_masks_and_tokenizer_match_reference_fixtures():
    tu = load("openvla-oft_amd.prismatic.training.train_utils")
    at = load("openvla-oft_amd.prismatic.vla.action_tokenizer")
    g = dict(np.load(G / "g1_masks.npz"))
    lab = torch.from_numpy(g["labels"])
    assert np.array_equal(tu.get_current_action_mask(lab).numpy().astype(bool), g["current"])
    assert np.array_equal(tu.get_next_actions_mask(lab).numpy().astype(bool), g["next"])
    g4 = dict(np.load(G / "g4_action_tokenizer.npz"))
    tok = at.ActionTokenizer(type("T", (), {"vocab_size": 32000, "decode": lambda self, x: x, "batch_decode": lambda self, x: x})())
    assert np.array_equal(tok.token_ids(g4["actions"]), g4["token_ids"])
    assert np.array_equal(tok.decode_token_ids_to_actions(g4["all_ids"]), g4["decoded"])
    assert tok.action_token_begin_idx == int(g4["begin_idx"])
    synth = load("openvla-oft_amd.synthetic")
    assert np.array_equal(synth.action_token_ids(g4["actions"]), g4["token_ids"])


def test_synthetic_batch_layout():
    synth = load("openvla-oft_amd.synthetic")
    b = synth.make_batch(8, seed=0)
    assert b["pixel_values"].shape == (8, 12, 224, 224) and b["input_ids"].shape == (8, 95) and b["actions"].shape == (8, 8, 7)
    assert b["proprio"].shape == (8, 8) and b["attention_mask"].dtype == torch.bool
    lens = b["attention_mask"].sum(1).tolist()
    assert lens == [95, 91, 95, 95, 95, 91, 95, 95]                           # two ragged rows, right padded
    assert (b["input_ids"][1, 91:] == 32000).all() and (b["labels"][1, 91:] == -100).all()
    lab = b["labels"][0]
    assert (lab[:38] == -100).all() and (lab[38:94] > 31743).all() and lab[94] == 2
    assert (b["input_ids"][:, 0] == 1).all() and b["input_ids"][0, 37] == 29871


def test_learning_rate_schedule_and_run_id():
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    cfg = ft.FinetuneConfig(learning_rate=5e-4, num_steps_before_decay=100, lr_warmup_steps=0)
    assert ft.learning_rate_at(cfg, 0) == 5e-4 and ft.learning_rate_at(cfg, 99) == 5e-4 and ft.learning_rate_at(cfg, 100) == pytest.approx(5e-5)
    # cross-check against torch's MultiStepLR
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=5e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.1)
    for step in range(105):
        assert opt.param_groups[0]["lr"] == pytest.approx(ft.learning_rate_at(cfg, step))
        opt.step(); sch.step()
    w = ft.FinetuneConfig(learning_rate=1.0, lr_warmup_steps=10)
    assert ft.learning_rate_at(w, 0) == pytest.approx(0.19) and ft.learning_rate_at(w, 9) == pytest.approx(1.0)
    rid = ft.get_run_id(ft.FinetuneConfig(vla_path="openvla/openvla-7b", dataset_name="libero_spatial_no_noops"))
    assert rid == "openvla-7b+libero_spatial_no_noops+b8+lr-0.0005+lora-r32+dropout-0.0--image_aug"
    assert ft.remove_ddp_in_checkpoint({"module.fc1.weight": 1, "fc2.bias": 2}) == {"fc1.weight": 1, "fc2.bias": 2}


def test_image_prep():
    ip = load("openvla-oft_amd.image_prep")
    from oracle import vla_oracle as vo

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    crop = ip.center_crop_image(img)
    assert crop.shape == (224, 224, 3) and crop.dtype == np.uint8
    assert np.array_equal(crop, vo.crop_and_resize_center(img))
    flat = np.full((224, 224, 3), 77, dtype=np.uint8)
    assert np.array_equal(ip.center_crop_image(flat), flat)                      # constant images are fixed points
    ident = ip.center_crop_image(img, crop_scale=1.0)
    assert np.abs(ident.astype(int) - img.astype(int)).max() <= 1
    t = ip.apply_transform(img)
    assert t.shape == (6, 224, 224) and torch.allclose(t, vo.image_transform(img, [vo.IMAGENET_MEAN, vo.SIGLIP_MEAN], [vo.IMAGENET_STD, vo.SIGLIP_STD]))
    with pytest.raises(AssertionError):
        ip.check_image_format(img.astype(np.float32))


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (only tests, smoke() and bench.py's cpu_baseline leg may)."""
    pkg = Path(__file__).resolve().parent.parent / "openvla-oft_amd"
    offenders = []
    for f in pkg.rglob("*.py"):
        if f.name == "smoke.py":
            continue
        txt = f.read_text()
        if "import oracle" in txt or "from oracle" in txt:
            offenders.append(str(f))
    assert not offenders, offenders


def test_peft_adapter_directory_roundtrip(tmp_path):
    """lora_adapter/ in peft's on-disk format (finetune.py:617-619 writes it with save_pretrained; merge_lora_weights_and_save.py
    reads it back with PeftModel.from_pretrained).  peft itself is absent: parity of the layout is unpinned, restated from
    its published format (adapter_config.json + adapter_model.safetensors, `base_model.model.` key prefix)."""
    import json

    from safetensors.torch import load_file, save_file

    w = importlib.import_module("openvla-oft_amd.weights")
    t = {"language_model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.randn(4, 16).to(torch.bfloat16),
         "language_model.model.layers.0.self_attn.q_proj.lora_B.weight": torch.randn(16, 4).to(torch.bfloat16),
         "action_head.model.fc1.weight": torch.randn(3, 3)}
    d = w.save_lora_adapter(tmp_path / "lora_adapter", t, r=4, lora_alpha=2)
    raw = load_file(str(d / "adapter_model.safetensors"))
    assert set(raw) == {"base_model.model." + k for k in t if ".lora_" in k}, "only adapter tensors, peft key prefix"
    cfg = json.loads((d / "adapter_config.json").read_text())
    assert cfg["peft_type"] == "LORA" and cfg["r"] == 4 and cfg["lora_alpha"] == 2 and cfg["target_modules"] == "all-linear"
    back, cfg2 = w.load_lora_adapter(d)
    assert cfg2 == cfg and set(back) == {k for k in t if ".lora_" in k}
    assert all(torch.equal(back[k], t[k]) for k in back)
    # adapters saved with an explicit adapter name and without the prefix are accepted too
    save_file({"x.lora_A.default.weight": torch.zeros(2, 2), "base_model.model.x.lora_B.default.weight": torch.zeros(2, 2)},
              str(d / "adapter_model.safetensors"))
    back, _ = w.load_lora_adapter(d)
    assert set(back) == {"x.lora_A.weight", "x.lora_B.weight"}
    (d / "adapter_config.json").write_text(json.dumps({**cfg, "use_dora": True}))
    with pytest.raises(ValueError, match="DoRA"):
        w.load_lora_adapter(d)


def test_save_sharded_layout(tmp_path):
    m = importlib.import_module("openvla-oft_amd.vla_scripts.merge_lora_weights_and_save")
    import json

    from safetensors.torch import load_file

    sd = {f"w{i}": torch.full((256,), float(i)) for i in range(5)}     # 1 KiB each
    files = m.save_sharded(sd, tmp_path / "a", max_shard_bytes=2048)
    assert [f.name for f in files] == ["model-00001-of-00003.safetensors", "model-00002-of-00003.safetensors", "model-00003-of-00003.safetensors"]
    idx = json.loads((tmp_path / "a" / "model.safetensors.index.json").read_text())
    assert idx["metadata"]["total_size"] == 5 * 1024 and idx["weight_map"]["w4"] == "model-00003-of-00003.safetensors"
    got = {}
    for f in files:
        got.update(load_file(str(f)))
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    assert [f.name for f in m.save_sharded(sd, tmp_path / "b")] == ["model.safetensors"]


def test_gemm_schedule_decisions():
    """ovla_gemm_plan: the host-side cost model behind `tile = 0` (no launch, runs without a GPU: the CU count falls back to 256).
    Pins the decisions the measurements in tools/gemm_sweep.py / gemm_m608.py / gemm_small_m.py justified."""
    import ctypes

    lib = importlib.import_module("openvla-oft_amd._lib").lib()

    def plan(M, N, K, K2=0, group=0):
        t, f, r, s = (ctypes.c_int32() for _ in range(4))
        e = ctypes.c_double()
        assert lib.ovla_gemm_plan(M, N, K, K2, group, 96 << 20, ctypes.byref(t), ctypes.byref(f), ctypes.byref(r), ctypes.byref(s), ctypes.byref(e)) == 0
        return t.value, f.value, r.value, s.value, e.value

    dims = {17: (256, 256), 1: (128, 128), 2: (64, 128), 5: (128, 32)}
    cases = {  # (M, N, K, K2, k2_group_n) -> expected tile
        (4864, 12288, 4096, 32, 4096): 17, (4864, 22016, 4096, 32, 11008): 17, (4864, 4096, 11008, 32, 0): 17, (4864, 4096, 4096, 32, 0): 17,   # Llama, B = 8
        (4176, 3072, 1024, 32, 0): 17, (4176, 1024, 1024, 32, 0): 2,                                                                             # ViT, B = 8
        (608, 12288, 4096, 0, 0): 1, (608, 4096, 4096, 0, 0): 1, (608, 4096, 11008, 0, 0): 1,                                                  # Llama, batch-1 inference
        (522, 3072, 1024, 0, 0): 5, (512, 1152, 1152, 0, 0): 5, (522, 1024, 4096, 0, 0): 2,                                                     # ViT, batch-1 inference
    }
    for (M, N, K, K2, g), want in cases.items():
        tile, full, rem, sp, est = plan(M, N, K, K2, g)
        bm, bn = dims[tile]
        tiles = -(-M // bm) * -(-N // bn)
        assert tile == want, f"{(M, N, K)}: tile {tile}, expected {want}"
        assert (full + rem == tiles) if rem else (full == tiles and sp == 1)
        assert 1 <= sp <= 8 and 0 < est < 5e-3
        if g:
            assert g % bn == 0, "a LoRA group never straddles an N tile"
    # the o-proj shape: 304 tiles of 256x256 = one full round of the 256 CUs + 48 tiles split along K over the idle CUs
    assert plan(4864, 4096, 4096, 32)[1:4] == (256, 48, 5)
    # more work never gets cheaper
    assert plan(4864, 8192, 4096)[4] > plan(4864, 4096, 4096)[4] and plan(4864, 4096, 8192)[4] > plan(4864, 4096, 4096)[4]


def test_deploy_wire_codec_roundtrip():
    """json_numpy-style wire format of the /act endpoint (deploy.py:80-95), incl. the "encoded" double encoding."""
    import json

    d = importlib.import_module("openvla-oft_amd.vla_scripts.deploy")
    obs = {"full_image": np.arange(24, dtype=np.uint8).reshape(2, 4, 3), "state": np.linspace(-1, 1, 8), "instruction": "pick up the bowl"}
    wire = json.loads(json.dumps(d._encode(obs)))
    back, double = d.decode_payload(wire)
    assert not double and back["instruction"] == obs["instruction"]
    assert back["full_image"].dtype == np.uint8 and np.array_equal(back["full_image"], obs["full_image"]) and np.array_equal(back["state"], obs["state"])
    back2, double2 = d.decode_payload({"encoded": json.dumps(d._encode(obs))})
    assert double2 and np.array_equal(back2["full_image"], obs["full_image"])
    with pytest.raises(AssertionError, match="Only uses encoded payload"):
        d.decode_payload({"encoded": "{}", "x": 1})


def test_vision_backbone_checkpoint_key_layout_roundtrip():
    """`vision_backbone--{step}_checkpoint.pt` in the reference's layout (FiLMedPrismaticVisionBackbone over peft-wrapped towers:
    finetune.py:640-655, film_vit_wrapper.py:49-54,192, peft's base_layer / lora_X.default naming) <-> the engine's names."""
    w = importlib.import_module("openvla-oft_amd.weights")
    eng = {f"vision_backbone.featurizer.blocks.3.attn.qkv.{t}": i for i, t in enumerate(("weight", "bias", "lora_A.weight", "lora_B.weight"))}
    eng.update({"vision_backbone.featurizer.blocks.3.scale.weight": 5, "vision_backbone.featurizer.blocks.3.shift.bias": 6,
                "vision_backbone.featurizer.blocks.3.norm1.weight": 7, "vision_backbone.featurizer.blocks.3.ls1.scale_factor": 8,
                "vision_backbone.featurizer.patch_embed.proj.weight": 9, "vision_backbone.featurizer.pos_embed": 10,
                "vision_backbone.fused_featurizer.blocks.0.mlp.fc1.weight": 11, "projector.fc1.weight": 12})
    ref = w.vision_backbone_keys_to_reference(eng)
    assert "projector.fc1.weight" not in ref
    assert ref["vision_backbone.featurizer.blocks.3.block.attn.qkv.base_layer.weight"] == 0
    assert ref["vision_backbone.featurizer.blocks.3.block.attn.qkv.lora_A.default.weight"] == 2
    assert ref["vision_backbone.featurizer.blocks.3.scale.weight"] == 5 and ref["vision_backbone.featurizer.blocks.3.block.ls1.scale_factor"] == 8
    assert ref["vision_backbone.featurizer.patch_embed.proj.weight"] == 9
    assert ref["vision_backbone.fused_featurizer.blocks.0.block.mlp.fc1.weight"] == 11, "a Linear without adapters keeps its plain name"
    back = w.vision_backbone_keys_from_reference({"module." + k: v for k, v in ref.items()})
    assert back == {k: v for k, v in eng.items() if k.startswith("vision_backbone.")}
    assert w.vision_backbone_keys_from_reference({"featurizer.blocks.3.scale.weight": 1}) == {"vision_backbone.featurizer.blocks.3.scale.weight": 1}


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset: the process starts N children through torch.distributed.run on 127.0.0.1 BEFORE any
    torch.cuda call and returns their status; under a launcher (WORLD_SIZE set) it does not."""
    import subprocess
    import sys

    bench = importlib.import_module("bench")
    calls = []

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)

    def no_cuda(*a, **k):
        raise AssertionError("the launcher parent must not touch the GPU")

    import torch

    monkeypatch.setattr(torch.cuda, "set_device", no_cuda)
    monkeypatch.setattr(torch.cuda, "is_available", no_cuda)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "WORLD_SIZE" not in env
    # a mismatch under a launcher is an error message, not a silent single-rank run
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code) and len(calls) == 1
