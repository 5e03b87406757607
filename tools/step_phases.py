"""Wall time of the phases of one full 7B LoRA step (synchronised between phases): vision fwd, Llama fwd, head, Llama bwd, vision bwd, optimizer."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
ops, engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.ops"), load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                   load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"))
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False)
get, has = weights_mod.make_getter(sd, dev)
eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
del sd, get
batch = synth.make_batch(8, seed=1000, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
for k in ("pixel_values", "actions", "proprio"):
    batch[k] = batch[k].to(dev, torch.bfloat16)
def step():
    eng.zero_grad(); eng.train_step_fwd_bwd(batch); eng.adamw_step(lr=5e-4); eng.refresh_derived()
for _ in range(3): step()
torch.cuda.synchronize()
marks = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    return w
eng.vision_fwd = timed("vision_fwd (2 towers + projector)", eng.vision_fwd)
eng.vision_bwd = timed("vision_bwd", eng.vision_bwd)
eng.llm.fwd = timed("llama_fwd", eng.llm.fwd)
eng.llm.bwd = timed("llama_bwd", eng.llm.bwd)
eng.head.fwd = timed("head_fwd", eng.head.fwd)
eng.head.bwd = timed("head_bwd", eng.head.bwd)
eng.adamw_step = timed("adamw", eng.adamw_step)
eng.refresh_derived = timed("refresh_derived", eng.refresh_derived)
N = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3 / N
for k, v in marks.items(): print(f"{k:36s} {v / N:8.2f} ms")
print(f"{'sum of phases':36s} {sum(marks.values()) / N:8.2f} ms;  step (with the extra syncs) {tot:.2f} ms")
