#!/usr/bin/env python
"""bench.py -- OpenVLA-OFT LoRA fine-tune step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when WORLD_SIZE is unset, this process starts them itself as children
  (the same torch.distributed.run command on 127.0.0.1, before anything here touches the GPU), relays rank 0's JSON line and exits with their status.

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


def flops_per_sample(cfg, S, I, sel_rows=None):
    """Algorithmic FLOPs of one train-step sample (SURVEY.md 8d formulas) -> dict:
      fwd / train          EXECUTED by this path: the discarded last ViT block and the lm_head + CE the reference computes and throws
                           away in L1 mode are not run; with `sel_rows` (= A action rows per sample) the last decoder layer runs its
                           o / gate|up / down projections on those rows only (LlamaStack.fwd(sel=...)), forward and data gradient;
      fwd_ref / train_ref  REFERENCE-EQUIVALENT: what finetune.py:280-451 executes for the same step (all S rows in every layer,
                           all `depth` ViT blocks in the forward, lm_head + cross entropy in the forward)."""
    D, F, L = cfg.llm_dim, cfg.llm_ff, cfg.llm_layers
    r = cfg.lora_rank
    qkv_lin, rest_lin = 2 * 3 * D * D, 2 * (D * D + 3 * D * F)             # per token per layer: q|k|v ; o + gate|up + down
    qkv_lora, rest_lora = 2 * r * 3 * (D + D), 2 * r * ((D + D) + 2 * (D + F) + (F + D))
    last_rows = S if sel_rows is None else sel_rows
    llm_lin = S * (L - 1) * (qkv_lin + rest_lin) + S * qkv_lin + last_rows * rest_lin
    lora_llm = S * (L - 1) * (qkv_lora + rest_lora) + S * qkv_lora + last_rows * rest_lora
    llm_lin_ref, lora_llm_ref = S * L * (qkv_lin + rest_lin), S * L * (qkv_lora + rest_lora)
    llm_attn = 4 * L * S * S * D
    vit = vit_last = 0
    for vc in (cfg.dino, cfg.siglip):
        T = vc.n_patches + vc.n_prefix
        blk = 2 * T * (4 * vc.dim * vc.dim + 2 * vc.dim * vc.mlp_hidden) + 4 * T * T * vc.dim
        vit += I * ((vc.depth - 1) * blk + 2 * vc.n_patches * vc.patch_k * vc.dim)
        vit_last += I * blk
    vd = cfg.vision_dim
    proj = 2 * I * cfg.dino.n_patches * (vd * 4 * vd + 4 * vd * D + D * D)
    head = 2 * cfg.chunk * (cfg.action_dim * D * D + 2 * D * D + D * cfg.action_dim)
    lm_head = 2 * S * D * cfg.vocab
    fwd = llm_lin + llm_attn + vit + proj + head
    # frozen base: backward = data gradients only (= forward FLOPs) + attention backward (2.5x forward attention);
    # LoRA adds fwd + dgrad + wgrad of the skinny GEMMs; the head trains in full (fwd + dgrad + wgrad)
    train = 2 * (llm_lin + vit + proj) + 3.5 * llm_attn + 3 * lora_llm + 3 * head
    fwd_ref = llm_lin_ref + llm_attn + vit + vit_last + proj + head + lm_head
    train_ref = 2 * (llm_lin_ref + vit + proj) + vit_last + lm_head + 3.5 * llm_attn + 3 * lora_llm_ref + 3 * head
    return dict(fwd=fwd, train=train, fwd_ref=fwd_ref, train_ref=train_ref)


def _host_cores():
    """The threads this process may actually run on (a 1-GPU box grants a CPU share, not the host's whole core count: one thread
    per *granted* core; oversubscribing 256 threads onto 16 cores ran 10x slower)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return min(cores, int(os.environ.get("OVLA_CPU_BASELINE_THREADS", "64")))


def _oracle_cfg(cfg):
    from oracle import vla_oracle as vo

    vit_fields = ("dim", "depth", "heads", "mlp_hidden", "n_prefix", "layerscale", "patch", "image_size")
    return vo.OracleConfig(**{f: getattr(cfg, f) for f in ("llm_dim", "llm_layers", "llm_heads", "llm_ff", "vocab", "rms_eps", "rope_theta",
                                                           "num_images", "lora_rank", "lora_alpha", "action_dim", "chunk", "proprio_dim")},
                           dino=vo.VitConfig(**{f: getattr(cfg.dino, f) for f in vit_fields}),
                           siglip=vo.VitConfig(**{f: getattr(cfg.siglip, f) for f in vit_fields}))


def cpu_baseline(cfg, sd_dev, b1, fl, gpu_chunks_per_s=None):
    """SURVEY.md 8(d) "CPU baseline" = BASELINE.json configs[0] at full size: ONE action chunk (batch 1, 2 x 224 x 224 images +
    proprio, S = 608, L1 head) through the CPU oracle (oracle/vla_oracle.py, kind "port": the reference itself cannot run on the GPU
    box) in fp32 on the host cores this process is granted, full-size weights (the same seeded tensors as the GPU run, LoRA merged
    away as in deployment, copied to host memory as fp32 before the clock starts).  ~8.9 TFLOP: about ten seconds on 16 cores.
    The fine-tune metric's CPU figure is this measured FLOP rate applied to the train step's FLOPs (`train_step_estimate`)."""
    from oracle import vla_oracle as vo

    cores = _host_cores()
    torch.set_num_threads(cores)
    t_load = time.perf_counter()
    sd = {k: v.float().cpu() for k, v in sd_dev.items() if ".lora_" not in k and not k.startswith("language_model.lm_head")}
    t_load = time.perf_counter() - t_load
    o = vo.Oracle(_oracle_cfg(cfg), sd, mode="fp32")
    ids, am = b1["input_ids"][:, : -(cfg.num_action_tokens + 1)], b1["attention_mask"][:, : -(cfg.num_action_tokens + 1)]   # the prompt
    pv, prop = b1["pixel_values"].float().cpu(), b1["proprio"].float().cpu().reshape(-1).numpy()
    with torch.no_grad():
        o.vit(pv[:, :3], "vision_backbone.featurizer.", o.cfg.dino)          # warm-up: thread pool, allocator (one tower, 0.16 TFLOP)
        t0 = time.perf_counter()
        actions, _ = o.predict_action(ids, am, pv, proprio=prop, head="l1")
        dt = time.perf_counter() - t0
    assert actions.shape == (cfg.chunk, cfg.action_dim) and bool((actions == actions).all())
    rate = fl["fwd"] / dt
    out = {"value": 1.0 / dt, "unit": "action-chunks/s (one batch-1 forward = one 8x7 chunk: BASELINE.json configs[0] at full size)", "cores": cores,
           "kind": "port",
           "sample": f"oracle fp32 predict_action, 1 chunk, S={b1['input_ids'].shape[1] + 1 + cfg.num_images * cfg.dino.n_patches}, all {cfg.llm_layers} layers + both towers "
                     f"+ projector + L1 head at full size: {dt:.2f} s on {cores} threads = {rate / 1e12:.3f} TFLOP/s (weights copied to host as fp32 "
                     f"beforehand: {t_load:.1f} s, not timed)",
           "seconds_per_chunk": dt, "tflops": rate / 1e12,
           "train_step_estimate": {"value": rate / fl["train_ref"], "unit": "samples/s",
                                   "how": f"the measured CPU FLOP rate applied to the reference-equivalent {fl['train_ref'] / 1e12:.1f} TFLOP of one "
                                          "fine-tune sample (an extrapolation, not a timed CPU train step)"}}
    if gpu_chunks_per_s:
        out["gpu_over_cpu_same_workload"] = gpu_chunks_per_s * dt
    return out


def _eager_state(cfg, sd):
    """The bench's seeded weights as the stock-eager baseline needs them: the towers' last block exists in the reference and runs
    although its output is discarded (blocks[depth-1] := a copy of blocks[depth-2])."""
    sd = dict(sd)
    for prefix, vc in (("vision_backbone.featurizer.", cfg.dino), ("vision_backbone.fused_featurizer.", cfg.siglip)):
        src = f"{prefix}blocks.{vc.depth - 2}."
        for k in [k for k in sd if k.startswith(src)]:
            sd[k.replace(src, f"{prefix}blocks.{vc.depth - 1}.")] = sd[k]
    return sd


def _full_depth_oracle():
    from oracle import vla_oracle as vo

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

    return FullDepth


def eager_baseline(cfg, batch_size, steps, warmup, dev, sd=None):
    """the same step executed by stock PyTorch-ROCm eager ops (hipBLASLt GEMMs, SDPA, unfused LoRA as
    separate matmuls, torch.optim.AdamW, autograd), bf16 weights and activations, including the lm_head + fp32 logits +
    cross-entropy that the reference computes and discards in L1 mode (finetune.py:338-351).  Runs the oracle's module
    code in its "native" mode on the GPU: a timing baseline, not a parity reference."""
    load = importlib.import_module
    weights_mod, synth = load("openvla-oft_amd.weights"), load("openvla-oft_amd.synthetic")
    if sd is None:
        sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=True)
    sd = _eager_state(cfg, sd)
    trainable = [k for k in sd if ".lora_" in k or k.startswith(("action_head.", "proprio_projector."))]
    for k in trainable:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    o = _full_depth_oracle()(_oracle_cfg(cfg), sd, mode="native")
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
    torch.cuda.reset_peak_memory_stats()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"kind": "B1: torch-eager (stock PyTorch-ROCm ops, bf16, hipBLASLt + SDPA + autograd + torch.optim.AdamW; the oracle's modules in mode='native')",
            "ms_per_step": 1e3 * dt, "samples_per_s": batch_size / dt, "steps": steps, "warmup": warmup, "loss": loss.item(),
            "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}


def eager_inference_baseline(cfg, dev, sd, b1, n=20):
    """one predict_action-equivalent forward (batch 1, L1 head) by stock PyTorch-ROCm eager ops on the merged
    (adapter-free) weights the reference deploys, all ViT blocks, no lm_head (the L1 branch of predict_action does not call it:
    modeling_prismatic.py:879-927)."""
    sd = {k: v for k, v in _eager_state(cfg, sd).items() if ".lora_" not in k}
    o = _full_depth_oracle()(_oracle_cfg(cfg), sd, mode="native")
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b1.items()}
    pv, prop = b["pixel_values"].to(torch.bfloat16), b["proprio"].to(torch.bfloat16)

    def once():
        with torch.no_grad():
            hidden, P = o.multimodal_hidden(b["input_ids"], b["attention_mask"], pv, b["labels"], prop)
            A = cfg.num_action_tokens
            ah = hidden[:, -(A + 1): -1]                       # rows that predict the action slots (shift by one; stop is the last id)
            return o.l1_head(ah)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = once()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    assert tuple(r.shape) == (1, cfg.chunk, cfg.action_dim)
    return {"kind": "B3: torch-eager batch-1 chunk (stock PyTorch-ROCm ops, bf16, merged weights)", "ms_per_chunk": ms, "chunks_per_s": 1e3 / ms, "reps": n}


def measured_traffic(tiny: bool):
    """`roofline.traffic`: HBM-side bytes per launch of the dominant kernel from PMC counters.  They cannot be collected from inside
    this process (rocprofv3 --pmc passes; tools/pmc_traffic.sh + tools/pmc_traffic_parse.py write profiles/pmc_traffic.json), so the
    value is read from that file together with where it came from; it is flagged stale when gemm_nt.hip has changed since."""
    import hashlib

    f = ROOT / "profiles" / "pmc_traffic.json"
    if tiny or not f.exists():
        return None, {"traffic_source": None}
    rec = json.loads(f.read_text())
    sha = hashlib.sha256((ROOT / "openvla-oft_amd" / "csrc" / "gemm_nt.hip").read_bytes()).hexdigest()[:16]
    return rec["bytes_per_launch"], {"traffic_source": rec.get("source"), "traffic_algorithmic_bytes": rec.get("algorithmic_bytes"),
                                     "traffic_shape_mnk": rec.get("shape"), "traffic_stale": rec.get("gemm_nt_sha16") != sha}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through torch.distributed.run (the command the
    driver would type: vla-scripts/finetune.py:212-224,796 run under torchrun the same way), on 127.0.0.1 with a free port.  This parent never
    initialises the GPU and never execs; the children's stdout / stderr pass through (rank 0 prints the JSON line), their status is returned."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print(f"[bench] WORLD_SIZE unset: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference recipe: 8, LIBERO.md:91-113)")
    ap.add_argument("--tiny", action="store_true", help="reduced-size model (plumbing check only; NOT a valid benchmark number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the in-process stock PyTorch-ROCm eager legs (BASELINE.md B1 and B3)")
    ap.add_argument("--baseline-steps", type=int, default=10, help="timed steps of the B1 eager leg (after 3 warm-up steps)")
    ap.add_argument("--no-inference", action="store_true", help="skip the batch-1 inference leg (configs[1])")
    ap.add_argument("--aloha", action="store_true", help="SURVEY.md 8(d) config 5 instead of the headline workload: ALOHA shapes (3 images, 25x14 "
                    "chunk, proprio 14, S=1159), FiLM + diffusion head, batch 4 -- a side measurement, not the BASELINE metric")
    ap.add_argument("--eager-baseline", action="store_true", help="time the stock PyTorch-ROCm eager step instead (BASELINE.md B1) and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))          # no HIP call has happened in this process; the ranks are fresh children
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N`, or under "
                         f"`torch.distributed.run --nproc-per-node N`)")
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
        res.pop("steps"), res.pop("warmup")
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
    n_ranks_seen = 1
    if world > 1:    # every rank adds a one: the line below then shows how many processes really took part in the collectives
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed
    final_loss = loss_sum.item() / (args.batch * cfg.chunk * cfg.action_dim)
    eng.head.check_fused_tail()

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
        # launch runs: t17 = gemm_nt_kernel<256,256,2,4>, t18k<KEXT> = its 4-wave configuration gemm_nt_w4_kernel<KEXT, 0, 2, false, false> (KEXT = LoRA K-extension / 32), t1 = <128,128,2,2>, t2 = <64,128,1,4>, t5 = <128,32,4,1>; a launch's
        # events also cover its split-K / hybrid reduce kernel)
        inst = {"gemm_nt_t17": "gemm_nt_kernel<256,256,2,4>", "gemm_nt_t18k0": "gemm_nt_w4_kernel<0, 0, 2, false, false>", "gemm_nt_t18k1": "gemm_nt_w4_kernel<1, 0, 2, false, false>", "gemm_nt_t18k2": "gemm_nt_w4_kernel<2, 0, 2, false, false>", "gemm_nt_t18k3": "gemm_nt_w4_kernel<3, 0, 2, false, false>", "gemm_nt_t1": "gemm_nt_kernel<128,128,2,2>",
                "gemm_nt_t2": "gemm_nt_kernel<64,128,1,4>", "gemm_nt_t5": "gemm_nt_kernel<128,32,4,1>", "gemm_nt_t6": "gemm_skinny_kernel<2>"}
        gemms = {k: v for k, v in fam.items() if k.startswith("gemm_nt")}
        dom = max(gemms, key=lambda k: gemms[k][1])
        n, ms, fl = gemms[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, traffic_info = measured_traffic(args.tiny)
        all_n, all_ms, all_fl = (sum(v[i] for v in gemms.values()) for i in range(3))
        roofline = {"bound": "mfma", "kernel": inst.get(dom, dom) + " (bf16 NT GEMM + LoRA K-extension + fused epilogue; incl. its hybrid-remainder reduce)",
                    "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                    # HBM-side bytes per launch (PMC, gfx950-corrected FETCH_SIZE*2 + WRITE_SIZE) of the most expensive shape:
                    # read from profiles/pmc_traffic.json (measured_traffic)
                    "traffic": traffic, **traffic_info,
                    "launches_per_step": n, "avg_launch_us": 1e3 * ms / n, "flops_per_step": fl, "ms_per_step": ms,
                    "all_gemm_nt": {"launches": all_n, "ms": all_ms, "flops_per_step": all_fl, "tflops": all_fl / (all_ms * 1e-3) / 1e12,
                                    "by_instance": {inst.get(k, k): {"launches": v[0], "ms": v[1], "tflops": v[2] / (v[1] * 1e-3) / 1e12}
                                                    for k, v in sorted(gemms.items())}},
                    "other_kernels": {k: {"launches": v[0], "ms": v[1], "tflops": (v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else None)}
                                      for k, v in fam.items() if not k.startswith("gemm_nt")},
                    "event_timed_ms_per_step": sum(v[1] for v in fam.values())}
        sel_on = os.environ.get("OVLA_LAST_LAYER_SEL", "1") != "0" and (args.batch * cfg.num_action_tokens) % 8 == 0
        flp = flops_per_sample(cfg, S, cfg.num_images, sel_rows=cfg.num_action_tokens if sel_on else None)
        # skipped work leaves the count -- the fraction is priced on EXECUTED FLOPs; the reference-equivalent count
        # (every row in the last layer, all ViT blocks, lm_head + CE) is reported beside it
        roofline["step_tflops_per_sample"] = flp["train"] / 1e12
        roofline["step_tflops_per_sample_reference_equivalent"] = flp["train_ref"] / 1e12
        roofline["step_mfma_frac"] = (flp["train"] * args.batch / (ms_per_step * 1e-3)) / 1e12 / PEAK_BF16_TFLOPS
        roofline["step_mfma_frac_reference_equivalent"] = (flp["train_ref"] * args.batch / (ms_per_step * 1e-3)) / 1e12 / PEAK_BF16_TFLOPS
    # ---- BASELINE.json configs[1]: single-chunk inference, batch 1; rank 0 only, no collectives.  Runs LAST on the training
    # engine: (1) adapters applied on the fly (the state during fine-tuning evaluation), (2) adapters merged into the base
    # weights as the reference deploys them (merge_lora_weights_and_save.py), (3) the merged forward replayed from a hipGraph.
    infer = None
    if rank == 0 and not args.aloha:
        b1 = synth.make_batch(1, seed=77, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
        b1["pixel_values"] = b1["pixel_values"].to(dev, torch.bfloat16)
        b1["proprio"] = b1["proprio"].to(dev, torch.bfloat16).reshape(1, -1)
    if rank == 0 and not args.no_inference and not args.aloha:

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
        # roofline of the chunk ("B = 1 inference: report both"): M = S = 608 sits at the knee, so both floors are quoted.
        # Per-kernel-family times of ONE eager chunk (HIP events on the launch stream; the two vision streams overlap, so they sum to more than
        # the chunk's wall time) name the dominant kernel and its own rate.
        ops.PROFILE = []
        infer_once()
        torch.cuda.synchronize()
        ifam = {}
        for family, e0, e1, fl in ops.PROFILE:
            d = ifam.setdefault(family, [0, 0.0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
        ops.PROFILE = None
        flp1 = flops_per_sample(cfg, S, cfg.num_images, sel_rows=cfg.num_action_tokens)
        w_bytes = 2.0 * (cfg.llm_layers * (4 * cfg.llm_dim ** 2 + 3 * cfg.llm_dim * cfg.llm_ff) + cfg.vision_dim * 4 * cfg.vision_dim + 4 * cfg.vision_dim * cfg.llm_dim +
                         cfg.llm_dim ** 2 + sum((vc.depth - 1) * (4 * vc.dim ** 2 + 2 * vc.dim * vc.mlp_hidden) for vc in (cfg.dino, cfg.siglip)) +
                         cfg.action_dim * cfg.llm_dim ** 2 + 2 * cfg.llm_dim ** 2)
        igemm = {k: v for k, v in ifam.items() if k.startswith("gemm_nt")}
        idom = max(igemm, key=lambda k: igemm[k][1])
        inst1 = {"gemm_nt_t17": "gemm_nt_kernel<256,256,2,4>", "gemm_nt_t1": "gemm_nt_kernel<128,128,2,2>", "gemm_nt_t2": "gemm_nt_kernel<64,128,1,4>",
                 "gemm_nt_t5": "gemm_nt_kernel<128,32,4,1>", "gemm_nt_t22": "gemm_nt_w4_kernel<0, 0, 4, RMAP, GMAP> (the 128x256 tile on 4 waves: plain / with the RoPE column map / with the SwiGLU pair map)"}
        infer_roofline = {
            "mfma": {"flops_per_chunk": flp1["fwd"], "achieved": flp1["fwd"] / (ms_graph * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": flp1["fwd"] / (ms_graph * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "floor_ms": flp1["fwd"] / (PEAK_BF16_TFLOPS * 1e12) * 1e3},
            "hbm": {"weight_bytes_per_chunk": w_bytes, "achieved": w_bytes / (ms_graph * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": w_bytes / (ms_graph * 1e-3) / 1e9 / 8000.0, "floor_ms": w_bytes / 8e12 * 1e3},
            "dominant_kernel": {"kernel": inst1.get(idom, idom) + " (incl. its hybrid / split-K reduce)", "launches": igemm[idom][0], "ms": igemm[idom][1],
                                "achieved": igemm[idom][2] / (igemm[idom][1] * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": igemm[idom][2] / (igemm[idom][1] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS},
            "by_family_ms": {k: {"launches": v[0], "ms": v[1]} for k, v in sorted(ifam.items(), key=lambda kv: -kv[1][1])},
            "profile": "profiles/r03_infer_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/infer_profile.py)"}
        infer = {"workload": "BASELINE.json configs[1]: OpenVLA-7B L1-regression inference, 2x224x224 images + proprio, bf16, batch 1 (one 8x7 action chunk per forward)",
                 "ms_per_chunk": ms_graph, "chunks_per_s": 1e3 / ms_graph, "actions_per_s": 1e3 / ms_graph * cfg.chunk,
                 "mode": "LoRA merged (W += 0.5 B A on device) + hipGraph replay",
                 "ms_per_chunk_merged_eager": ms_merged, "ms_per_chunk_unmerged_eager": ms_lora, "roofline": infer_roofline}

    # ---- measured baselines, same process, same box, AFTER every timed region (rank 0, N = 1 only):
    #   B1  stock PyTorch-ROCm eager fine-tune step (the denominator of north_star's ">= 1.5x" target), B3 stock eager batch-1 chunk,
    #   cpu_baseline  the CPU oracle on one full-size chunk (BASELINE.json configs[0] at full size)
    baselines = None
    if rank == 0 and world == 1 and not args.tiny and not args.aloha and not (args.no_eager_baseline and args.no_cpu_baseline):
        graph = None
        torch.cuda.empty_cache()
        sd0 = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=True)      # the same seeded tensors the engine was built from
        if not args.no_eager_baseline:
            e1 = eager_baseline(cfg, args.batch, args.baseline_steps, 3, dev, sd=sd0)
            torch.cuda.empty_cache()
            e3 = eager_inference_baseline(cfg, dev, sd0, b1)
            baselines = {"B1_torch_eager_train_step": e1, "B3_torch_eager_inference_batch1": e3,
                         "speedup_vs_B1": value / e1["samples_per_s"],
                         "speedup_vs_B3": (infer["chunks_per_s"] / e3["chunks_per_s"]) if infer else None,
                         "note": "BASELINE.md publishes no number for this metric (vs_baseline stays null); B1 / B3 are BASELINE.md section 2's "
                                 "self-measured baselines, run in this process on this GPU after the timed region"}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, sd0, b1, flp, infer["chunks_per_s"] if infer else None)
        del sd0
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({
            "metric": "fine-tune samples/s (action-chunks/s) OpenVLA-7B bf16", "value": value, "unit": "samples/s", "n_gpus": world,
            "n_ranks_seen": n_ranks_seen, "dist_backend": backend if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights of the OpenVLA-7B architecture, synthetic LIBERO-shaped batches)",
            "config": {"workload": ("SURVEY 8(d) config 5 [side measurement]: ALOHA shapes, 3x224x224 images + proprio 14, 25x14 chunk, FiLM + diffusion head (MSE), LoRA r=32 step" if args.aloha else
                                    "BASELINE.json configs[2]: LoRA r=32 fine-tune step (fwd+bwd+AdamW), 2x224x224 images + proprio, L1 head") + (" [TINY MODEL - not a benchmark]" if args.tiny else ""),
                       "global_batch": world * args.batch, "seq_len": S, "parallelism": f"dp{world}", "mask_mode": cfg.mask_mode,
                       "final_loss": final_loss},
            "roofline": roofline, "cpu_baseline": cpu, "inference_batch1": infer, "measured_baselines": baselines}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
