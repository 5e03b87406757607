#!/usr/bin/env python
"""bench.py -- OpenVLA-OFT LoRA fine-tune step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

Workload (config.workload): BASELINE.json configs[2] -- LoRA (rank 32) fine-tune of OpenVLA-7B on synthetic
LIBERO-Spatial batches, bf16, batch 8 per GPU, 2 x 224x224 images + proprio, L1-regression head, S = 608.
One step = zero_grad + forward + backward + gradient all-reduce (N > 1) + fused AdamW + derived-weight refresh, with
inputs resident in HBM.  Weights are seeded random tensors of the real architecture (no checkpoint exists offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"


def flops_per_sample(cfg, S, I):
    """Executed algorithmic FLOPs of one train-step sample (SURVEY.md 8d formulas; the discarded last ViT block and the
    lm_head/CE the reference computes and throws away in L1 mode are NOT executed and NOT counted)."""
    D, F, L = cfg.llm_dim, cfg.llm_ff, cfg.llm_layers
    r = cfg.lora_rank
    llm_lin = S * 2 * L * (4 * D * D + 3 * D * F)
    llm_attn = 4 * L * S * S * D
    lora_llm = S * 2 * L * r * (2 * D * 4 + 2 * (D + F) + (D + F))  # q,k,v,o: (D+D) each; gate,up: (D+F) each; down: (F+D)
    vit = 0
    for vc in (cfg.dino, cfg.siglip):
        T = vc.n_patches + vc.n_prefix
        blocks = vc.depth - 1
        vit += I * (2 * T * blocks * (4 * vc.dim * vc.dim + 2 * vc.dim * vc.mlp_hidden) + 4 * blocks * T * T * vc.dim
                    + 2 * vc.n_patches * vc.patch_k * vc.dim)
    vd = cfg.vision_dim
    proj = 2 * I * cfg.dino.n_patches * (vd * 4 * vd + 4 * vd * D + D * D)
    head = 2 * cfg.chunk * (cfg.action_dim * D * D + 2 * D * D + D * cfg.action_dim)
    fwd = llm_lin + llm_attn + vit + proj + head
    # frozen base: backward = data gradients only (= forward FLOPs) + attention backward (2.5x forward attention);
    # LoRA adds fwd + dgrad + wgrad of the skinny GEMMs; the head trains in full (fwd + dgrad + wgrad)
    train = 2 * (llm_lin + vit + proj) + 3.5 * llm_attn + 3 * lora_llm + 3 * head
    return fwd, train


def cpu_baseline(cfg, S, train_flops_per_sample):
    """Times the CPU oracle (oracle/vla_oracle.py, kind "port") on a bounded slice of the same workload: one sample
    (batch 1, S = 608) through ONE full-width Llama decoder layer + final norm + L1 head, forward + backward, fp32, all
    host cores.  Its measured FLOP rate is converted to samples/s of the full step by the FLOP ratio."""
    from oracle import vla_oracle as vo

    # the threads this process may actually run on (a 1-GPU box grants a CPU share, not the host's whole core count:
    # one thread per *granted* core; oversubscribing 256 threads onto 16 cores ran 10x slower)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:   # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    if quota:
        cores = min(cores, quota)
    cores = min(cores, int(os.environ.get("OVLA_CPU_BASELINE_THREADS", "64")))
    torch.set_num_threads(cores)
    ocfg = vo.OracleConfig(llm_dim=cfg.llm_dim, llm_layers=1, llm_heads=cfg.llm_heads, llm_ff=cfg.llm_ff, vocab=cfg.vocab)
    g = torch.Generator().manual_seed(0)
    D, F, r = cfg.llm_dim, cfg.llm_ff, cfg.lora_rank
    sd = {}

    def lin(name, o, i, lora=True, bias=False):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * 0.02
        if bias:
            sd[name + ".bias"] = torch.zeros(o)
        if lora:
            sd[name + ".lora_A.weight"] = (torch.randn(r, i, generator=g) / r).requires_grad_(True)
            sd[name + ".lora_B.weight"] = (torch.randn(o, r, generator=g) * 0.01).requires_grad_(True)

    p = "language_model.model.layers.0."
    for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
        lin(p + "self_attn." + n, D, D)
    lin(p + "mlp.gate_proj", F, D); lin(p + "mlp.up_proj", F, D); lin(p + "mlp.down_proj", D, F)
    sd[p + "input_layernorm.weight"] = torch.ones(D); sd[p + "post_attention_layernorm.weight"] = torch.ones(D)
    sd["language_model.model.norm.weight"] = torch.ones(D)
    hp = "action_head.model."
    for nm, dim in (("layer_norm1", D * cfg.action_dim), ("layer_norm2", D), ("mlp_resnet_blocks.0.ffn.0", D), ("mlp_resnet_blocks.1.ffn.0", D)):
        sd[hp + nm + ".weight"] = torch.ones(dim, requires_grad=True); sd[hp + nm + ".bias"] = torch.zeros(dim, requires_grad=True)
    for nm, o, i in (("fc1", D, D * cfg.action_dim), ("mlp_resnet_blocks.0.ffn.1", D, D), ("mlp_resnet_blocks.1.ffn.1", D, D), ("fc2", cfg.action_dim, D)):
        sd[hp + nm + ".weight"] = (torch.randn(o, i, generator=g) * 0.02).requires_grad_(True)
        sd[hp + nm + ".bias"] = torch.zeros(o, requires_grad=True)
    o = vo.Oracle(ocfg, sd, mode="fp32")
    x = torch.randn(1, S, D, generator=g)
    A = cfg.action_dim * cfg.chunk
    tgt = torch.rand(1, cfg.chunk, cfg.action_dim, generator=g) * 2 - 1

    def one():
        h = o.llm(x, torch.ones(1, S, dtype=torch.bool))
        pred = o.l1_head(h[:, S - 1 - A: S - 1])
        (tgt - pred).abs().mean().backward()

    one()  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 12.0:
        one()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    lin_f = S * 2 * (4 * D * D + 3 * D * F)
    attn_f = 4 * S * S * D
    lora_f = S * 2 * r * (2 * D * 4 + 3 * (D + F))
    head_f = 2 * cfg.chunk * (cfg.action_dim * D * D + 2 * D * D)
    sample_flops = 2 * lin_f + 3.5 * attn_f + 3 * lora_f + 3 * head_f
    rate = sample_flops / dt
    return {"value": rate / train_flops_per_sample, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 fwd+bwd of 1 sample (S={S}) through 1 of {cfg.llm_layers} full-width Llama layers + final norm + L1 head, "
                      f"{reps} reps of {dt:.2f} s = {rate / 1e12:.3f} TFLOP/s, scaled by FLOPs to the full {train_flops_per_sample / 1e12:.1f} TFLOP/sample step"}


def eager_baseline(cfg, batch_size, steps, warmup, dev):
    """the same step executed by stock PyTorch-ROCm eager ops (hipBLASLt GEMMs, SDPA, unfused LoRA as
    separate matmuls, torch.optim.AdamW, autograd), bf16 weights and activations, including the lm_head + fp32 logits +
    cross-entropy that the reference computes and discards in L1 mode (finetune.py:338-351).  Runs the oracle's module
    code in its "native" mode on the GPU: a timing baseline, not a parity reference."""
    from oracle import vla_oracle as vo
    load = importlib.import_module
    weights_mod, synth = load("openvla-oft_amd.weights"), load("openvla-oft_amd.synthetic")
    sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=True)
    # the towers' last block exists in the reference and runs although its output is discarded
    for prefix, vc in (("vision_backbone.featurizer.", cfg.dino), ("vision_backbone.fused_featurizer.", cfg.siglip)):
        src = f"{prefix}blocks.{vc.depth - 2}."
        for k in [k for k in sd if k.startswith(src)]:
            sd[k.replace(src, f"{prefix}blocks.{vc.depth - 1}.")] = sd[k].clone()
    trainable = [k for k in sd if ".lora_" in k or k.startswith(("action_head.", "proprio_projector."))]
    for k in trainable:
        sd[k].requires_grad_(True)
    ocfg = vo.OracleConfig(**{f: getattr(cfg, f) for f in ("llm_dim", "llm_layers", "llm_heads", "llm_ff", "vocab", "rms_eps", "rope_theta",
                                                          "num_images", "lora_rank", "lora_alpha", "action_dim", "chunk", "proprio_dim")},
                           dino=vo.VitConfig(**{f: getattr(cfg.dino, f) for f in ("dim", "depth", "heads", "mlp_hidden", "n_prefix", "layerscale", "patch", "image_size")}),
                           siglip=vo.VitConfig(**{f: getattr(cfg.siglip, f) for f in ("dim", "depth", "heads", "mlp_hidden", "n_prefix", "layerscale", "patch", "image_size")}))

    class FullDepth(vo.Oracle):   # run ALL ViT blocks like timm does (the oracle skips the discarded one)
        def vit(self, img, prefix, vc, film_avg=None):
            out = super().vit(img, prefix, vc, film_avg)
            B, H, hd = img.shape[0], vc.heads, vc.dim // vc.heads
            p = f"{prefix}blocks.{vc.depth - 1}."
            x = torch.cat([out.new_zeros(B, vc.n_prefix, vc.dim), out], 1)
            h = torch.nn.functional.layer_norm(x, (vc.dim,), self.W(p + "norm1.weight"), self.W(p + "norm1.bias"), vc.eps)
            qkv = self.linear(h, p + "attn.qkv").reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
            a = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, -1, vc.dim)
            x = x + self.linear(a, p + "attn.proj")
            h = torch.nn.functional.layer_norm(x, (vc.dim,), self.W(p + "norm2.weight"), self.W(p + "norm2.bias"), vc.eps)
            self._dead = x + self.linear(self.act(self.linear(h, p + "mlp.fc1")), p + "mlp.fc2")
            return out

    o = FullDepth(ocfg, sd, mode="native")
    opt = torch.optim.AdamW([sd[k] for k in trainable], lr=5e-4)
    batch = synth.make_batch(batch_size, seed=1000)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    for k in ("pixel_values", "actions", "proprio"):
        batch[k] = batch[k].to(torch.bfloat16)

    def step():
        loss, _, _ = o.train_forward(batch)
        # the discarded language-modelling loss of PrismaticForConditionalGeneration.forward(labels=...) (modeling_prismatic.py:632-643)
        hidden, P = o._last_hidden
        logits = (hidden @ sd["language_model.lm_head.weight"].T).float()
        labels = torch.cat([batch["labels"][:, :1], torch.full((batch_size, P), -100, device=dev), batch["labels"][:, 1:]], 1)
        torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=-100)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    orig = o.multimodal_hidden

    def keep(*a, **k):
        r = orig(*a, **k)
        o._last_hidden = (r[0].detach(), r[1])
        return r

    o.multimodal_hidden = keep
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"kind": "torch-eager (stock PyTorch-ROCm ops, bf16, hipBLASLt + SDPA + torch.optim.AdamW)", "ms_per_step": 1e3 * dt,
            "samples_per_s": batch_size / dt, "loss": loss.item(), "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference recipe: 8, LIBERO.md:91-113)")
    ap.add_argument("--tiny", action="store_true", help="reduced-size model (plumbing check only; NOT a valid benchmark number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the batch-1 inference leg (configs[1])")
    ap.add_argument("--aloha", action="store_true", help="SURVEY.md 8(d) config 5 instead of the headline workload: ALOHA shapes (3 images, 25x14 "
                    "chunk, proprio 14, S=1159), FiLM + diffusion head, batch 4 -- a side measurement, not the BASELINE metric")
    ap.add_argument("--eager-baseline", action="store_true", help="time the stock PyTorch-ROCm eager step instead (BASELINE.md B1) and exit")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("OVLA_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the multi-process path on a 1-GPU box
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("openvla-oft_amd")
    load = importlib.import_module
    engine_mod, weights_mod, synth, config_mod, ops = (load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                       load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"), load("openvla-oft_amd.ops"))
    dp_mod = load("openvla-oft_amd.dp")
    if args.tiny:
        cfg = config_mod.VLAConfig(llm_dim=256, llm_layers=2, llm_heads=2, llm_ff=512,
                                   dino=config_mod.VitConfig(128, 3, 2, 256, n_prefix=5, layerscale=True),
                                   siglip=config_mod.VitConfig(144, 3, 2, 536))
    else:
        cfg = config_mod.OPENVLA_7B
    if args.aloha:
        import dataclasses

        cfg = dataclasses.replace(cfg, num_images=3, chunk=25, action_dim=14, proprio_dim=14)
        if args.batch == 8:
            args.batch = 4           # ALOHA.md:69
    if args.eager_baseline:
        res = eager_baseline(cfg, args.batch, args.steps, args.warmup, dev)
        print(json.dumps({"metric": "fine-tune samples/s (action-chunks/s) OpenVLA-7B bf16 [torch eager baseline]", "value": res["samples_per_s"],
                          "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, **res}))
        return
    t_init = time.time()
    sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False, film=args.aloha, diffusion=args.aloha)   # identical on every rank
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="diffusion" if args.aloha else "l1", use_film=args.aloha, has=has)
    del sd, get
    torch.cuda.empty_cache()
    batch = synth.make_batch(args.batch, seed=1000 + rank, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim,
                             proprio_dim=cfg.proprio_dim)
    # inputs resident in HBM before the timed region (ids/labels/mask stay host-side like the reference collator's output;
    # they are ~6 KB per step)
    batch["pixel_values"] = batch["pixel_values"].to(dev, torch.bfloat16)
    batch["actions"] = batch["actions"].to(dev, torch.bfloat16)
    batch["proprio"] = batch["proprio"].to(dev, torch.bfloat16)
    S = 1 + cfg.num_images * cfg.dino.n_patches + 1 + int(args.aloha) + (batch["input_ids"].shape[1] - 1)   # (+1: diffusion timestep token)
    reducer = dp_mod.GradReducer(eng.stores, world) if world > 1 else None
    # gradient all-reduce overlapped with the backward (buckets ship as they complete); OVLA_DP_OVERLAP=0 reduces everything after it
    eng.attach_reducer(reducer, overlap=os.environ.get("OVLA_DP_OVERLAP", "1") != "0")
    if rank == 0:
        print(f"[bench] init {time.time() - t_init:.1f}s, trainable params {eng.num_trainable() / 1e6:.1f} M, S={S}, "
              f"HBM allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)

    diffusion = None
    if args.aloha:   # finetune.py:336-350: noise, timestep and x_t are drawn per step on the host (a chunk is 350 numbers per sample)
        dmod = load("openvla-oft_amd.diffusion")
        sched, enc = dmod.DDIMScheduler(50), dmod.SinusoidalPositionalEncoding(cfg.llm_dim)
        gen = torch.Generator().manual_seed(1234 + rank)

    def step():
        nonlocal diffusion
        eng.zero_grad()
        if args.aloha:
            noise = torch.randn(batch["actions"].shape, generator=gen)
            ts = torch.randint(0, 50, (args.batch,), generator=gen)
            diffusion = dict(noise=noise, noisy_actions=sched.add_noise(batch["actions"].float().cpu(), noise, ts).to(torch.bfloat16),
                             timestep_emb=enc(ts.float()).to(torch.bfloat16))
        loss_sum, count, _ = eng.train_step_fwd_bwd(batch, diffusion=diffusion)
        if reducer is not None:
            reducer.all_reduce()
        eng.adamw_step(lr=5e-4, grad_scale=1.0 / world)
        eng.refresh_derived()
        return loss_sum

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_sum = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed
    final_loss = loss_sum.item() / (args.batch * cfg.chunk * cfg.action_dim)

    roofline = cpu = None
    # dominant kernel (gemm_nt) timed launch by launch with HIP events on the launch stream over one more step.  EVERY rank
    # runs the step (it contains the gradient collectives); only rank 0 records events.
    if rank == 0:
        ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    if rank == 0:
        fam = {}
        for family, e0, e1, fl in ops.PROFILE:
            d = fam.setdefault(family, [0, 0.0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
        ops.PROFILE = None
        # the dominant kernel = the gemm_nt instance with the most time in the step (ops._gemm_family names the instance each
        # launch runs: t17 = gemm_nt_kernel<256,256,2,4>, t1 = <128,128,2,2>, t2 = <64,128,1,4>, t5 = <128,32,4,1>; a launch's
        # events also cover its split-K / hybrid reduce kernel)
        inst = {"gemm_nt_t17": "gemm_nt_kernel<256,256,2,4>", "gemm_nt_t1": "gemm_nt_kernel<128,128,2,2>", "gemm_nt_t2": "gemm_nt_kernel<64,128,1,4>",
                "gemm_nt_t5": "gemm_nt_kernel<128,32,4,1>"}
        gemms = {k: v for k, v in fam.items() if k.startswith("gemm_nt")}
        dom = max(gemms, key=lambda k: gemms[k][1])
        n, ms, fl = gemms[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        all_n, all_ms, all_fl = (sum(v[i] for v in gemms.values()) for i in range(3))
        roofline = {"bound": "mfma", "kernel": inst.get(dom, dom) + " (bf16 NT GEMM + LoRA K-extension + fused epilogue; incl. its hybrid-remainder reduce)",
                    "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                    # HBM-side bytes per launch from PMC counters cannot be collected from inside this process; the value below is
                    # the measured, gfx950-corrected FETCH_SIZE*2 + WRITE_SIZE of the most expensive shape (gate|up forward,
                    # 4864x22016x4096; 4.34e8 algorithmic bytes), see profiles/r01_pmc_gemm_gate_up.md
                    "traffic": 1.62e9 if not args.tiny else None,
                    "launches_per_step": n, "avg_launch_us": 1e3 * ms / n, "flops_per_step": fl, "ms_per_step": ms,
                    "all_gemm_nt": {"launches": all_n, "ms": all_ms, "flops_per_step": all_fl, "tflops": all_fl / (all_ms * 1e-3) / 1e12,
                                    "by_instance": {inst.get(k, k): {"launches": v[0], "ms": v[1], "tflops": v[2] / (v[1] * 1e-3) / 1e12}
                                                    for k, v in sorted(gemms.items())}},
                    "other_kernels": {k: {"launches": v[0], "ms": v[1], "tflops": (v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else None)}
                                      for k, v in fam.items() if not k.startswith("gemm_nt")},
                    "event_timed_ms_per_step": sum(v[1] for v in fam.values())}
        fwd_f, train_f = flops_per_sample(cfg, S, cfg.num_images)
        roofline["step_tflops_per_sample"] = train_f / 1e12
        roofline["step_mfma_frac"] = (train_f * args.batch / (ms_per_step * 1e-3)) / 1e12 / PEAK_BF16_TFLOPS
        if not args.no_cpu_baseline and not args.tiny and world == 1:   # the CPU baseline is reported at N = 1 only
            cpu = cpu_baseline(cfg, S, train_f)
    # ---- BASELINE.json configs[1]: single-chunk inference, batch 1; rank 0 only, no collectives.  Runs LAST on the training
    # engine: (1) adapters applied on the fly (the state during fine-tuning evaluation), (2) adapters merged into the base
    # weights as the reference deploys them (merge_lora_weights_and_save.py), (3) the merged forward replayed from a hipGraph.
    infer = None
    if rank == 0 and not args.no_inference and not args.aloha:
        b1 = synth.make_batch(1, seed=77, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
        b1["pixel_values"] = b1["pixel_values"].to(dev, torch.bfloat16)
        b1["proprio"] = b1["proprio"].to(dev, torch.bfloat16).reshape(1, -1)

        def infer_once():
            out = eng.forward(b1["input_ids"], b1["attention_mask"], b1["pixel_values"], b1["labels"], proprio=b1["proprio"], train=False,
                              sel="actions")     # as predict_action does: the last layer's projections only on the 56 action rows
            ah, _ = eng.action_hidden(out)
            return eng.head.fwd(ah)[0]

        def time_it(fn, n=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ti = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - ti) / n, r

        ms_lora, _ = time_it(infer_once)
        eng.merge_lora()
        ms_merged, pred_e = time_it(infer_once)
        graph = engine_mod.ChunkGraph(eng, 1, b1["input_ids"].shape[1], b1["pixel_values"].shape, head=eng.head, use_proprio=True)
        graph.load(b1["input_ids"], b1["attention_mask"], b1["pixel_values"], b1["labels"], b1["proprio"])
        graph.capture()

        def graph_once():   # what a deployment does per observation: refresh the static inputs, replay, read the actions back
            graph.load(b1["input_ids"], b1["attention_mask"], b1["pixel_values"], b1["labels"], b1["proprio"])
            return graph.replay()[0]

        ms_graph, pred_g = time_it(graph_once, n=50)
        assert torch.equal(pred_g, pred_e), "hipGraph replay must reproduce the eager actions bit for bit"
        infer = {"workload": "BASELINE.json configs[1]: OpenVLA-7B L1-regression inference, 2x224x224 images + proprio, bf16, batch 1 (one 8x7 action chunk per forward)",
                 "ms_per_chunk": ms_graph, "chunks_per_s": 1e3 / ms_graph, "actions_per_s": 1e3 / ms_graph * cfg.chunk,
                 "mode": "LoRA merged (W += 0.5 B A on device) + hipGraph replay",
                 "ms_per_chunk_merged_eager": ms_merged, "ms_per_chunk_unmerged_eager": ms_lora}

    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({
            "metric": "fine-tune samples/s (action-chunks/s) OpenVLA-7B bf16", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights of the OpenVLA-7B architecture, synthetic LIBERO-shaped batches)",
            "config": {"workload": ("SURVEY 8(d) config 5 [side measurement]: ALOHA shapes, 3x224x224 images + proprio 14, 25x14 chunk, FiLM + diffusion head (MSE), LoRA r=32 step" if args.aloha else
                                    "BASELINE.json configs[2]: LoRA r=32 fine-tune step (fwd+bwd+AdamW), 2x224x224 images + proprio, L1 head") + (" [TINY MODEL - not a benchmark]" if args.tiny else ""),
                       "global_batch": world * args.batch, "seq_len": S, "parallelism": f"dp{world}", "mask_mode": cfg.mask_mode,
                       "final_loss": final_loss},
            "roofline": roofline, "cpu_baseline": cpu, "inference_batch1": infer}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
