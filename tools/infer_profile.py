"""Batch-1 inference (merged LoRA) for rocprofv3 --kernel-trace --stats: 30 eager forwards (graph replays hide kernel names from some tools)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
ops, engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.ops"), load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                   load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"))
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False, lora=False)
get, has = weights_mod.make_getter(sd, dev)
eng = engine_mod.VLAEngine(cfg, get, dev, lora=False, use_proprio=True, head="l1", has=has)
del sd, get
b1 = synth.make_batch(1, seed=77, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
b1["pixel_values"] = b1["pixel_values"].to(dev, torch.bfloat16); b1["proprio"] = b1["proprio"].to(dev, torch.bfloat16).reshape(1, -1)
def once():
    out = eng.forward(b1["input_ids"], b1["attention_mask"], b1["pixel_values"], b1["labels"], proprio=b1["proprio"], train=False)
    ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
    return eng.head.fwd(ah)[0]
for _ in range(30): once()
torch.cuda.synchronize()
