"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 (reduced-size model, same structure as
OpenVLA-7B: dual ViT -> projector -> Llama with LoRA -> L1 head; forward + backward + AdamW), checked against the CPU
oracle on the same seeded inputs."""
import importlib

import torch


def run(device: torch.device) -> None:
    from oracle import vla_oracle as vo  # checker only

    pkg = __package__
    engine_mod, weights_mod, synth, config_mod = (importlib.import_module(f"{pkg}.{m}") for m in ("engine", "weights", "synthetic", "config"))
    BF = torch.bfloat16
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=0).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(sd, device)
    eng = engine_mod.VLAEngine(cfg, get, device, lora=True, use_proprio=True, head="l1", has=has)
    batch = synth.make_batch(2, seed=3, prompt_lens=[9, 7], image_size=56)
    for k in ("pixel_values", "proprio", "actions"):
        batch[k] = batch[k].to(BF).float()
    with torch.no_grad():
        loss_ref, pred_ref, _ = vo.Oracle(ocfg, sd, mode="bf16").train_forward(batch)
    eng.zero_grad()
    loss_sum, count, pred = eng.train_step_fwd_bwd(batch)
    eng.adamw_step(lr=5e-4)
    eng.refresh_derived()
    torch.cuda.synchronize()
    loss = loss_sum.item() / count
    err = (pred.float().cpu().view(2, 8, 7) - pred_ref).abs().max().item()
    gnorm = sum(g.float().norm().item() ** 2 for g in eng.export_trainable("grad").values()) ** 0.5
    print(f"[smoke] loss hip {loss:.5f} oracle(bf16-emu) {loss_ref.item():.5f}; pred Linf err {err:.3e}; grad norm {gnorm:.4f}")
    assert abs(loss - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item())), "smoke: loss mismatch vs oracle"
    assert err < 5e-2, "smoke: action prediction mismatch vs oracle"
    assert gnorm > 0 and gnorm == gnorm, "smoke: gradients missing"
