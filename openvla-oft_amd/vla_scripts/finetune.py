"""LoRA fine-tune driver (mirror of vla-scripts/finetune.py: FinetuneConfig :79-131, run_forward_pass :280-451,
save_training_checkpoint :584-675, finetune :763-1154) on the HIP engine.

  torchrun --standalone --nnodes 1 --nproc-per-node N -m ...   (one process per GPU, RCCL through torch.distributed)

What differs from the reference, deliberately:
  * the optimisation step runs the engine's fused path: explicit forward/backward kernels, gradients accumulated in flat
    fp32 buffers, one fused AdamW launch per dtype (same arithmetic as torch.optim.AdamW on bf16 / fp32 parameters), one
    large RCCL all-reduce per bucket instead of four DDP wrappers;
  * the frozen lm_head + cross-entropy the reference computes and discards in L1 mode is not executed;
  * the data source is pluggable: `dataset` is any iterable of collated batches (prismatic/util/data_utils.py:102-156
    layout).  The TF/RLDS pipeline is out of scope; without one, seeded synthetic
    LIBERO-shaped batches are used (`synthetic.make_batch`).
`run_forward_pass` keeps the reference signature and works through torch autograd for glue that wants it.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterable, Optional, Tuple

import torch

from .. import synthetic
from ..config import OPENVLA_7B, VLAConfig
from ..dp import GradReducer
from ..engine import VLAEngine
from ..prismatic.training.train_utils import get_current_action_mask, get_next_actions_mask
from ..prismatic.vla import constants as C
from ..weights import load_lora_adapter, make_getter, random_state_dict, save_lora_adapter, vision_backbone_keys_from_reference


@dataclass
class FinetuneConfig:
    # fmt: off
    vla_path: str = "openvla/openvla-7b"
    data_root_dir: Path = Path("datasets/rlds")
    dataset_name: str = "aloha_scoop_x_into_bowl"
    run_root_dir: Path = Path("runs")
    shuffle_buffer_size: int = 100_000
    use_l1_regression: bool = True
    use_diffusion: bool = False
    num_diffusion_steps: int = 50
    use_film: bool = False
    num_images_in_input: int = 1
    use_proprio: bool = False
    batch_size: int = 8
    learning_rate: float = 5e-4
    lr_warmup_steps: int = 0
    num_steps_before_decay: int = 100_000
    grad_accumulation_steps: int = 1
    max_steps: int = 200_000
    use_val_set: bool = False
    val_freq: int = 10_000
    val_time_limit: int = 180
    save_freq: int = 10_000
    save_latest_checkpoint_only: bool = False
    resume: bool = False
    resume_step: Optional[int] = None
    image_aug: bool = True
    diffusion_sample_freq: int = 50
    use_lora: bool = True
    lora_rank: int = 32
    lora_dropout: float = 0.0
    merge_lora_during_training: bool = True
    wandb_entity: str = "your-wandb-entity"
    wandb_project: str = "your-wandb-project"
    run_id_note: Optional[str] = None
    run_id_override: Optional[str] = None
    wandb_log_freq: int = 10
    # fmt: on


def remove_ddp_in_checkpoint(state_dict) -> dict:
    """finetune.py:134-156"""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def get_run_id(cfg) -> str:
    """finetune.py:159-190"""
    if cfg.run_id_override is not None:
        return cfg.run_id_override
    if cfg.resume:
        run_id = cfg.vla_path.split("/")[-1]
        if "chkpt" in run_id.split("--")[-1]:
            run_id = "--".join(run_id.split("--")[:-1])
        return run_id
    run_id = f"{cfg.vla_path.split('/')[-1]}+{cfg.dataset_name}+b{cfg.batch_size * cfg.grad_accumulation_steps}+lr-{cfg.learning_rate}"
    if cfg.use_lora:
        run_id += f"+lora-r{cfg.lora_rank}+dropout-{cfg.lora_dropout}"
    if cfg.image_aug:
        run_id += "--image_aug"
    if cfg.run_id_note is not None:
        run_id += f"--{cfg.run_id_note}"
    return run_id


def learning_rate_at(cfg, gradient_step_idx: int) -> float:
    """MultiStepLR(milestones=[num_steps_before_decay], gamma=0.1) + optional linear warm-up 10% -> 100%
    (finetune.py:958-962, 1094-1098).  The step index is the number of optimizer steps already taken."""
    lr = cfg.learning_rate * (0.1 if gradient_step_idx >= cfg.num_steps_before_decay else 1.0)
    if cfg.lr_warmup_steps > 0:
        progress = min((gradient_step_idx + 1) / cfg.lr_warmup_steps, 1.0)
        lr = cfg.learning_rate * (0.1 + 0.9 * progress)
    return lr


def run_diffusion_sampling(vla, action_head, noisy_action_projector, proprio_projector, batch, batch_size, num_patches, actions_shape, device_id,
                           current_action_mask, next_actions_mask, use_proprio, use_film) -> torch.Tensor:
    """finetune.py:454-540: reverse diffusion for the whole batch -- start from N(0, 1) noise, and for every DDIM timestep predict the
    noise from the action rows of the VLM's last hidden state and step x_t -> x_{t-1}.  The reference re-runs the vision backbone in every
    step; its output does not depend on t, so here the projected patches of the first step are reused (same numbers, one tower pass).
    METRICS-ONLY deviation from the reference's arithmetic (this function feeds the logged diffusion L1 numbers, never a gradient): the start
    noise is drawn on the host and `scheduler.step` runs there in fp32 on the bf16-rounded sample, rounding to bf16 once per step, where the
    reference keeps `curr_noisy_actions` / `noise_pred` in bf16 on the device throughout (finetune.py:490-540); `predict_action`'s diffusion
    branch (modeling.py) is the deployment path and rounds where the reference does.  The per-step `.cpu()` is one host sync per DDIM step."""
    head, eng = action_head.module, vla.module.engine
    cfg = eng.cfg
    A = cfg.num_action_tokens
    cur = torch.randn((batch_size, C.NUM_ACTIONS_CHUNK, C.ACTION_DIM)).to(torch.bfloat16)
    head.noise_scheduler.set_timesteps(head.num_diffusion_steps)
    cached = None
    for t in head.noise_scheduler.timesteps:
        temb = head.time_encoder(torch.full((batch_size,), float(t))).to(torch.bfloat16).unsqueeze(1)          # (B, 1, llm_dim)
        out = eng.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"],
                          proprio=batch["proprio"] if use_proprio else None, train=False, noisy_actions=cur, timestep_emb=temb,
                          proprio_projector=proprio_projector.module.comp if use_proprio else None,
                          noisy_action_projector=noisy_action_projector.module.comp, cached_patches=cached, sel="actions")
        cached = out["patches"]
        ah, _ = eng.action_hidden(out)
        noise_pred = head.predict_noise(ah.view(batch_size, A, cfg.llm_dim)).reshape(cur.shape).float().cpu()
        cur = head.noise_scheduler.step(noise_pred, t, cur.float()).prev_sample.to(torch.bfloat16)
    return cur.reshape(actions_shape).to(device_id)


def run_forward_pass(vla, action_head, noisy_action_projector, proprio_projector, batch, action_tokenizer, device_id, use_l1_regression,
                     use_diffusion, use_proprio, use_film, num_patches, compute_diffusion_l1=False, num_diffusion_steps=None
                     ) -> Tuple[torch.Tensor, Dict[str, float]]:
    """finetune.py:280-451 with the reference's signature and all three objectives, through the autograd bridge of modeling.py
    (`loss.backward()` then drives the engine's explicit backward):
      discrete (neither head flag)   loss = output.loss (next-token cross entropy), token accuracies and decoded L1 of the current /
                                     next actions from output.logits[:, num_patches:-1].argmax(2)                       (:357-378)
      use_l1_regression              L1Loss(gt, head.predict_action(action rows))                                       (:396-400)
      use_diffusion                  MSE(noise_pred, noise); with compute_diffusion_l1 additionally a full DDIM sampling
                                     (run_diffusion_sampling) for the current / next L1 metrics                         (:402-430)"""
    metrics: Dict[str, float] = {}
    gt = batch["actions"].to(device_id).to(torch.bfloat16)
    noisy = None
    if use_diffusion:                                                                   # :327-333
        noisy = action_head.module.sample_noisy_actions(gt)
    output = vla(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"], labels=batch["labels"],
                 output_hidden_states=True, proprio=batch["proprio"] if use_proprio else None,
                 proprio_projector=proprio_projector if use_proprio else None, use_film=use_film,
                 noisy_actions=noisy["noisy_actions"] if use_diffusion else None,
                 noisy_action_projector=noisy_action_projector if use_diffusion else None,
                 diffusion_timestep_embeddings=noisy["diffusion_timestep_embeddings"] if use_diffusion else None)
    ids = batch["labels"][:, 1:].to(device_id)
    cur, nxt = get_current_action_mask(ids), get_next_actions_mask(ids)
    if not (use_l1_regression or use_diffusion):                                        # :357-378
        from ..prismatic.training.train_utils import compute_actions_l1_loss, compute_token_accuracy

        loss = output.loss
        predicted = output.logits[:, num_patches:-1].argmax(dim=2)
        metrics["loss_value"] = loss.item()
        for name, m in (("curr_action", cur), ("next_actions", nxt)):
            metrics[f"{name}_accuracy"] = compute_token_accuracy(predicted, ids, mask=m).item()
            metrics[f"{name}_l1_loss"] = compute_actions_l1_loss(action_tokenizer, predicted, ids, mask=m).item()
        return loss, metrics
    last = output.hidden_states[-1]
    text_hidden = last[:, num_patches:-1]
    B = batch["input_ids"].shape[0]
    ah = text_hidden[cur | nxt].reshape(B, C.NUM_ACTIONS_CHUNK * C.ACTION_DIM, -1).to(torch.bfloat16)
    pred = None
    if use_l1_regression:
        pred = action_head.module.predict_action(ah)
        loss = torch.nn.L1Loss()(gt, pred)
    if use_diffusion:                                                                   # :402-430
        noise_pred = action_head.module.predict_noise(ah).reshape(noisy["noise"].shape)
        loss = torch.nn.functional.mse_loss(noise_pred, noisy["noise"].to(noise_pred.device), reduction="mean")
        if compute_diffusion_l1:
            with torch.no_grad():
                pred = run_diffusion_sampling(vla, action_head, noisy_action_projector, proprio_projector, batch, B, num_patches, gt.shape, device_id,
                                              cur, nxt, use_proprio, use_film)
    metrics["loss_value"] = loss.item()
    if action_head is not None and hasattr(action_head.module, "comp"):
        action_head.module.comp.check_fused_tail()
    if pred is not None:                                                                # :437-448 (should_log_l1_loss)
        metrics["curr_action_l1_loss"] = torch.nn.L1Loss()(gt[:, 0], pred[:, 0]).item()
        metrics["next_actions_l1_loss"] = torch.nn.L1Loss()(gt[:, 1:], pred[:, 1:]).item()
    return loss, metrics


_COMPONENT_PREFIXES = {"proprio_projector": "proprio_projector.", "noisy_action_projector": "noisy_action_projector.", "action_head": "action_head.",
                       "vision_backbone": "vision_backbone."}   # FiLM scale/shift Linears (finetune.py:640-645 saves the wrapped backbone)


def save_training_checkpoint(run_dir: Path, log_step: int, engine: VLAEngine, dataset_statistics: Optional[dict], rank: int,
                             latest_only: bool = False, save_optimizer: bool = True) -> Path:
    """finetune.py:584-675: `{component}--{step}_checkpoint.pt` (+ `lora_adapter/` in peft's format, `dataset_statistics.json`).
    The LoRA merge of the reference (:663-675) is the separate merge_lora_weights_and_save step.  Beyond the reference:
    `optimizer_state--{step}_checkpoint.safetensors` (AdamW moments + step counters), so a resumed run continues the same
    trajectory instead of restarting the moments from zero."""
    ckpt = run_dir if latest_only else Path(str(run_dir) + f"--{log_step}_chkpt")
    suffix = "latest_checkpoint" if latest_only else f"{log_step}_checkpoint"
    if rank == 0:
        (ckpt / "lora_adapter").mkdir(parents=True, exist_ok=True)
        if dataset_statistics is not None:
            (ckpt / "dataset_statistics.json").write_text(json.dumps(dataset_statistics))
        exp = {k: v.detach().to("cpu") for k, v in engine.export_trainable("data").items()}
        save_lora_adapter(ckpt / "lora_adapter", exp, r=engine.cfg.lora_rank, lora_alpha=engine.cfg.lora_alpha)   # peft on-disk format
        for comp, prefix in _COMPONENT_PREFIXES.items():
            if comp == "vision_backbone":
                continue
            sd = {k[len(prefix):]: v.contiguous() for k, v in exp.items() if k.startswith(prefix) and ".lora_" not in k}
            if sd:
                torch.save(sd, ckpt / f"{comp}--{suffix}.pt")
        if engine.use_film:
            # finetune.py:640-655: the whole FiLM-wrapped backbone (frozen towers + their adapters + scale / shift), in the reference's key layout,
            # so that get_vla(cfg.use_film) -- here or in the reference -- finds what _apply_film_to_vla loads (openvla_utils.py:311-349)
            torch.save({k: v.detach().to("cpu").contiguous() for k, v in engine.vision_backbone_state_dict().items()}, ckpt / f"vision_backbone--{suffix}.pt")
        if save_optimizer:
            from safetensors.torch import save_file

            save_file({k: v.detach().to("cpu").contiguous() for k, v in engine.optimizer_state_dict().items()},
                      str(ckpt / f"optimizer_state--{suffix}.safetensors"))
    return ckpt


def load_training_checkpoint(ckpt: Path, log_step: Optional[int], engine: VLAEngine, load_optimizer: bool = True) -> dict:
    """Resume (finetune.py:134-156 `load_checkpoint(module_name, path, step)` per component + the lora adapter the reference
    re-attaches through its merged checkpoint): reads `{component}--{step}_checkpoint.pt` (`latest` when step is None), the
    peft adapter directory and, if present, the optimizer state.  Only `weights_only=True` / safetensors loaders are used."""
    ckpt = Path(ckpt)
    suffix = "latest_checkpoint" if log_step is None else f"{log_step}_checkpoint"
    sd: Dict[str, torch.Tensor] = {}
    adapter_dir = ckpt / "lora_adapter"
    if adapter_dir.is_dir():
        adapter, _ = load_lora_adapter(adapter_dir)
        sd.update(adapter)
    for comp, prefix in _COMPONENT_PREFIXES.items():
        f = ckpt / f"{comp}--{suffix}.pt"
        if f.is_file():
            part = remove_ddp_in_checkpoint(torch.load(str(f), weights_only=True, map_location="cpu"))
            if comp == "vision_backbone":    # reference key layout (or the FiLM-only files of earlier rounds); only the trainable tensors are taken
                part = vision_backbone_keys_from_reference(part)
                sd.update({k: v for k, v in part.items() if ".lora_" in k or ".scale." in k or ".shift." in k})
            else:
                sd.update({prefix + k: v for k, v in part.items()})
    missing = engine.load_trainable(sd, strict=False)
    opt = ckpt / f"optimizer_state--{suffix}.safetensors"
    loaded_opt = False
    if load_optimizer and opt.is_file():
        from safetensors.torch import load_file

        engine.load_optimizer_state_dict(load_file(str(opt)))
        loaded_opt = True
    return {"missing": missing, "optimizer": loaded_opt, "tensors": len(sd)}


def run_validation(engine: VLAEngine, cfg: FinetuneConfig, val_batches: Iterable[dict], log_step: int, log, *, discrete: bool, make_diffusion=None,
                   metric_fn=None) -> Dict[str, float]:
    """finetune.py:678-760: the training loss (and, for the discrete objective, the token metrics) on the validation loader without
    gradients, cut short after `cfg.val_time_limit` seconds, averaged over the batches seen, logged with a `val/` prefix."""
    import time

    start, rows = time.time(), []
    for batch in val_batches:
        if not cfg.use_proprio:
            batch = {**batch, "proprio": None}
        loss_sum, count, pred = engine.eval_step(batch, diffusion=make_diffusion(batch) if make_diffusion else None, discrete=discrete)
        m = {"loss_value": loss_sum.item() / count}
        if metric_fn is not None:
            m.update(metric_fn(batch, pred))
        m["loss"] = m["loss_value"]
        rows.append(m)
        if time.time() - start > cfg.val_time_limit:
            break
    if not rows:
        return {}
    avg = {f"val/{k}": sum(r[k] for r in rows) / len(rows) for k in rows[0]}
    avg["val/num_batches"] = len(rows)
    log(f"step {log_step}: " + " ".join(f"{k} {v:.5f}" for k, v in avg.items()))
    return avg


def finetune(cfg: FinetuneConfig, *, model_config: VLAConfig = OPENVLA_7B, state_dict: Optional[Dict[str, torch.Tensor]] = None,
             dataset: Optional[Iterable[dict]] = None, dataset_statistics: Optional[dict] = None, log=print, tokenizer=None,
             val_dataset=None) -> Dict[str, list]:
    """finetune.py:763-1154.  Returns the logged metric history.  Data source, in this order: `dataset` (any iterable of collated
    batches); an episode store at `cfg.data_root_dir / cfg.dataset_name` (prismatic/vla/datasets/rlds_free.py: the reference's
    RLDSDataset + RLDSBatchTransform + collator, finetune.py:981-1016, with the frames augmented and normalised on the device --
    needs `tokenizer(text) -> ids` or tokenizer files under `cfg.vla_path`); otherwise seeded synthetic batches of the recipe's shape."""
    assert cfg.use_lora, "Only LoRA fine-tuning is supported. Please set --use_lora=True!"
    assert not (cfg.use_l1_regression and cfg.use_diffusion), "Cannot do both L1 regression and diffusion. Please pick one of them!"
    discrete = not (cfg.use_l1_regression or cfg.use_diffusion)    # next-token cross entropy on the action tokens (finetune.py:357-378)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    run_dir = cfg.run_root_dir / get_run_id(cfg)
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
    model_config = VLAConfig(**{**model_config.__dict__, "num_images": cfg.num_images_in_input, "lora_rank": cfg.lora_rank,
                                "lora_alpha": min(cfg.lora_rank, 16), "action_dim": C.ACTION_DIM, "chunk": C.NUM_ACTIONS_CHUNK,
                                "proprio_dim": C.PROPRIO_DIM, "norm_type": C.ACTION_PROPRIO_NORMALIZATION_TYPE.value})
    if state_dict is None:
        log(f"[finetune] no checkpoint at `{cfg.vla_path}` is loadable offline: using seeded random weights of the architecture")
        state_dict = random_state_dict(model_config, dev, seed=0, lm_head=discrete, film=cfg.use_film, diffusion=cfg.use_diffusion)
    get, has = make_getter(state_dict, dev)
    engine = VLAEngine(model_config, get, dev, lora=True, use_proprio=cfg.use_proprio, head="none" if discrete else ("diffusion" if cfg.use_diffusion else "l1"),
                       use_film=cfg.use_film, has=has)
    if cfg.use_diffusion:   # DiffusionActionHead.sample_noisy_actions (action_heads.py:167-197), host side
        from ..diffusion import DDIMScheduler, SinusoidalPositionalEncoding

        sched, time_enc = DDIMScheduler(cfg.num_diffusion_steps), SinusoidalPositionalEncoding(model_config.llm_dim)
        noise_gen = torch.Generator().manual_seed(7 + rank)
    log(f"# total trainable params: {engine.num_trainable()}")
    if cfg.resume:   # finetune.py:193-209: the trainable components come from the checkpoint directory `vla_path` at `resume_step`
        assert cfg.resume_step is not None, "resume=True needs resume_step"
        info = load_training_checkpoint(Path(cfg.vla_path), cfg.resume_step, engine)
        log(f"[finetune] resumed step {cfg.resume_step} from {cfg.vla_path}: {info['tensors']} tensors, optimizer state: {info['optimizer']}, "
            f"{len(info['missing'])} trainable tensors kept at their initial values")
    reducer = GradReducer(engine.stores, world) if world > 1 else None
    engine.attach_reducer(reducer, overlap=cfg.grad_accumulation_steps == 1 and os.environ.get("OVLA_DP_OVERLAP", "1") != "0")   # overlap only when every backward ends a step
    if dataset is None and (Path(cfg.data_root_dir) / cfg.dataset_name).is_dir():
        from ..prismatic.vla import datasets as D
        from ..prismatic.vla.action_tokenizer import ActionTokenizer

        if tokenizer is None:
            from transformers import AutoTokenizer   # tokenizer files inside the checkpoint directory

            tok = AutoTokenizer.from_pretrained(cfg.vla_path)
            tokenizer = lambda text: tok(text, add_special_tokens=True).input_ids  # noqa: E731
        vocab = type("Vocab", (), {"vocab_size": model_config.vocab - model_config.pad_to_multiple_of})()   # 32000 (constants.py / modeling_prismatic.py:732)
        side = model_config.dino.image_size
        train_dataset = D.EpisodeDataset(cfg.data_root_dir, cfg.dataset_name,
                                         D.RLDSBatchTransform(ActionTokenizer(vocab), tokenizer, use_wrist_image=cfg.num_images_in_input > 1,
                                                              use_proprio=cfg.use_proprio),
                                         resize_resolution=(side, side), shuffle_buffer_size=cfg.shuffle_buffer_size, image_aug=cfg.image_aug,
                                         seed=7, rank=rank, world_size=world)
        if dataset_statistics is None:
            dataset_statistics = train_dataset.dataset_statistics    # saved next to every checkpoint (finetune.py:1003-1005)
        collator = D.DeviceCollator(2048, model_config.pad_token_id, device=dev, image_aug=cfg.image_aug, seed=1000 + rank, image_size=side)
        dataset = D.batches(train_dataset, collator, cfg.batch_size)
        if cfg.use_val_set and val_dataset is None:    # finetune.py:990-1000, 1017-1026: the store's val/ split, augmented like the training frames
            val_ds = D.EpisodeDataset(cfg.data_root_dir, cfg.dataset_name, train_dataset.batch_transform, resize_resolution=(side, side),
                                      shuffle_buffer_size=cfg.shuffle_buffer_size // 10, image_aug=cfg.image_aug, train=False, seed=11, rank=rank,
                                      world_size=world)
            val_collator = D.DeviceCollator(2048, model_config.pad_token_id, device=dev, image_aug=cfg.image_aug, seed=2000 + rank, image_size=side)
            val_dataset = lambda: D.batches(val_ds, val_collator, cfg.batch_size)  # noqa: E731
        log(f"[finetune] episode store {cfg.data_root_dir / cfg.dataset_name}: {len(train_dataset)} frames, image_aug={cfg.image_aug}")
    if dataset is None:
        def synthetic_stream():
            step = 0
            while True:
                yield synthetic.make_batch(cfg.batch_size, seed=1000 * (step + 1) + rank, num_images=cfg.num_images_in_input,
                                           chunk=C.NUM_ACTIONS_CHUNK, action_dim=C.ACTION_DIM, proprio_dim=C.PROPRIO_DIM,
                                           image_size=model_config.dino.image_size)
                step += 1
        dataset = synthetic_stream()
    if cfg.use_val_set and val_dataset is None:
        raise ValueError("use_val_set=True needs a validation source: an episode store with a val/ split, or `val_dataset` (a callable returning an iterable of batches)")
    history = {"loss_value": [], "learning_rate": [], "val": []}
    if discrete:
        from ..prismatic.training.train_utils import compute_actions_l1_loss, compute_token_accuracy
        from ..prismatic.vla.action_tokenizer import ActionTokenizer

        metric_tok = ActionTokenizer(type("Vocab", (), {"vocab_size": model_config.vocab - model_config.pad_to_multiple_of})())
        history.update({k: [] for k in ("curr_action_accuracy", "curr_action_l1_loss", "next_actions_accuracy", "next_actions_l1_loss")})
    def make_diffusion(batch):   # DiffusionActionHead.sample_noisy_actions (action_heads.py:167-197)
        gt = batch["actions"].to("cpu", torch.float32)
        noise = torch.randn(gt.shape, generator=noise_gen).to(torch.bfloat16).float()
        tsteps = torch.randint(0, cfg.num_diffusion_steps, (gt.shape[0],), generator=noise_gen)
        return dict(noise=noise, noisy_actions=sched.add_noise(gt, noise, tsteps).to(torch.bfloat16), timestep_emb=time_enc(tsteps.float()).to(torch.bfloat16))

    def token_metrics(batch, pred_ids):   # finetune.py:358-377
        gt_ids = batch["labels"][:, 1:].to("cpu")
        out = {}
        for name, m in (("curr_action", get_current_action_mask(gt_ids)), ("next_actions", get_next_actions_mask(gt_ids))):
            out[f"{name}_accuracy"] = compute_token_accuracy(pred_ids, gt_ids, m).item()
            out[f"{name}_l1_loss"] = compute_actions_l1_loss(metric_tok, pred_ids, gt_ids, m).item()
        return out

    engine.zero_grad()
    for batch_idx, batch in enumerate(dataset):
        if not cfg.use_proprio:
            batch = {**batch, "proprio": None}
        diffusion = make_diffusion(batch) if cfg.use_diffusion else None
        if discrete:
            loss_sum, count, pred_ids = engine.train_step_discrete(batch, loss_scale=1.0 / cfg.grad_accumulation_steps)
        else:
            loss_sum, count, _ = engine.train_step_fwd_bwd(batch, loss_scale=1.0 / cfg.grad_accumulation_steps, diffusion=diffusion)
        gradient_step_idx = batch_idx // cfg.grad_accumulation_steps
        log_step = gradient_step_idx if not cfg.resume else cfg.resume_step + gradient_step_idx
        if (batch_idx + 1) % cfg.grad_accumulation_steps == 0:
            if reducer is not None:
                reducer.all_reduce()
            lr = learning_rate_at(cfg, gradient_step_idx)
            engine.adamw_step(lr, grad_scale=1.0 / world)
            engine.refresh_derived()
            engine.zero_grad()
            if log_step % cfg.wandb_log_freq == 0:
                history["loss_value"].append(loss_sum.item() / count)
                if engine.head is not None:
                    engine.head.check_fused_tail()      # (the host is synchronised here anyway: raises if a fused-tail grid barrier timed out)
                history["learning_rate"].append(lr)
                if discrete:
                    for k, v in token_metrics(batch, pred_ids).items():
                        history[k].append(v)
                if rank == 0:
                    log(f"step {log_step}: loss {history['loss_value'][-1]:.5f} lr {lr:.2e}")
            if cfg.use_val_set and log_step > 0 and log_step % cfg.val_freq == 0:   # finetune.py:1128-1144
                history["val"].append((log_step, run_validation(engine, cfg, val_dataset(), log_step, log if rank == 0 else (lambda *_: None), discrete=discrete,
                                                                make_diffusion=make_diffusion if cfg.use_diffusion else None,
                                                                metric_fn=token_metrics if discrete else None)))
            if gradient_step_idx > 0 and log_step % cfg.save_freq == 0:
                save_training_checkpoint(run_dir, log_step, engine, dataset_statistics, rank, cfg.save_latest_checkpoint_only)
                if world > 1:
                    dist.barrier()
            if log_step + 1 >= cfg.max_steps:
                break
    return history
