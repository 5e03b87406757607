"""Vision towers at B = 8, two images: forward / backward time of each tower alone, of both back to back on one stream, and of both on two streams."""
import importlib, os, sys, time
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
pv = batch["pixel_values"].to(dev, torch.bfloat16)
I = 2
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
def tower_fb(tower, c0):
    vc = tower.vc; T = vc.n_patches + vc.n_prefix
    def f():
        tok, sv = tower.fwd(pv, c0, I, True, None)
        tower.bwd(torch.ones((8 * I * T, vc.dim), dtype=torch.bfloat16, device=dev), sv)
    return f
def tower_f(tower, c0):
    return lambda: tower.fwd(pv, c0, I, True, None)
print(f"DINOv2 alone : fwd {timed(tower_f(eng.dino, 0)):6.2f} ms, fwd+bwd {timed(tower_fb(eng.dino, 0)):6.2f} ms")
print(f"SigLIP alone : fwd {timed(tower_f(eng.siglip, 3)):6.2f} ms, fwd+bwd {timed(tower_fb(eng.siglip, 3)):6.2f} ms")
def both_seq():
    tower_fb(eng.dino, 0)(); tower_fb(eng.siglip, 3)()
side = torch.cuda.Stream()
def both_par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        tower_fb(eng.siglip, 3)()
    tower_fb(eng.dino, 0)()
    main.wait_stream(side)
print(f"both, one stream : {timed(both_seq):6.2f} ms;  both, two streams : {timed(both_par):6.2f} ms")
