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
        loss_ref, pred_ref, _ = vo.Oracle(ocfg, sd, mode="bf16").train_forward(batch)     # emulation of the reference's bf16 rounding points
        _, pred32, _ = vo.Oracle(ocfg, sd, mode="fp32").train_forward(batch)               # exact arithmetic on the same weights
    eng.zero_grad()
    loss_sum, count, pred = eng.train_step_fwd_bwd(batch)
    eng.adamw_step(lr=5e-4)
    eng.refresh_derived()
    torch.cuda.synchronize()
    loss = loss_sum.item() / count
    got = pred.float().cpu().view(2, 8, 7)
    err, err32, emu32 = (got - pred_ref).abs().max().item(), (got - pred32).abs().max().item(), (pred_ref - pred32).abs().max().item()
    # the head's output is a bf16 tensor: two evaluations agree bit for bit or differ by whole ulps of the output value
    # (2^-6 = 1.56e-2 for |a| in [2, 4), the range of this seeded model's largest predictions)
    ulp = 2.0 ** (torch.floor(torch.log2(pred32.abs().max())).item() - 7)
    gnorm = sum(g.float().norm().item() ** 2 for g in eng.export_trainable("grad").values()) ** 0.5
    print(f"[smoke] loss hip {loss:.5f} oracle(bf16-emu) {loss_ref.item():.5f}; pred Linf: hip-emu {err:.3e} ({err / ulp:.1f} output ulp), hip-fp32 {err32:.3e}, "
          f"emu-fp32 {emu32:.3e}; grad norm {gnorm:.4f}")
    assert abs(loss - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item())), "smoke: loss mismatch vs oracle"
    assert err32 <= 1.5 * emu32 + ulp, "smoke: the HIP path must be as close to exact arithmetic as the emulated bf16 reference path (+1 output ulp)"
    assert err <= 3 * ulp, "smoke: action prediction more than 3 bf16 ulp of the output away from the emulated reference path"
    assert gnorm > 0 and gnorm == gnorm, "smoke: gradients missing"
