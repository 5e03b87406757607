"""Host time to ENQUEUE one full 7B LoRA step (no synchronisation) next to the GPU time of the step: is the launch thread ahead of the GPU?"""
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
N = 6
t0 = time.perf_counter()
marks = []
for _ in range(N):
    step(); marks.append(time.perf_counter())
t_enq = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("host enqueue per step (ms):", [round((b - a) * 1e3, 1) for a, b in zip([t0] + marks[:-1], marks)])
print(f"all {N} steps enqueued after {(t_enq - t0) * 1e3:.1f} ms; GPU done after {(t1 - t0) * 1e3:.1f} ms ({(t1 - t0) * 1e3 / N:.1f} ms/step)")
# phase-level: how long does the host need for the vision forward's launches alone, and how long does the GPU take for it?
torch.cuda.synchronize()
pv = batch["pixel_values"]
t0 = time.perf_counter(); out = eng.vision_fwd(pv, True); t_h = time.perf_counter(); torch.cuda.synchronize(); t_g = time.perf_counter()
print(f"vision_fwd: host enqueue {(t_h - t0) * 1e3:.2f} ms, GPU complete {(t_g - t0) * 1e3:.2f} ms")
