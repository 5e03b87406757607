"""End-to-end latency of get_vla_action at full size (random OpenVLA-7B-shaped weights): uint8 frames + state + instruction in,
8 x 7 actions out -- device image prep (ovla_image_prep), tokenisation stub, merged LoRA, hipGraph replay, un-normalisation."""
import importlib, sys, time, types
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
weights_mod, config_mod, modeling, utils = (load("openvla-oft_amd.weights"), load("openvla-oft_amd.config"), load("openvla-oft_amd.modeling"),
                                            load("openvla-oft_amd.experiments.robot.openvla_utils"))
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False, lora=True)
stats = {"libero": {"action": {"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False]}, "proprio": {"q01": [-1.0] * 8, "q99": [1.0] * 8}}}
sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
vla = modeling.OpenVLAForActionPrediction(cfg, {k: v for k, v in sd.items() if not k.startswith(("action_head.", "proprio_projector."))}, device=dev, norm_stats=stats)
head = modeling.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, 7, device=dev, state_dict=sub("action_head."))
pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev, state_dict=sub("proprio_projector."))
del sd
vla.merge_and_unload()
rng = np.random.default_rng(0)
tok = lambda text: [1] + rng.integers(3, 31000, 36).tolist() + [29871]      # 38 prompt tokens (no tokenizer files offline)
proc = utils.PrismaticProcessor(tok)
c = types.SimpleNamespace(num_images_in_input=2, use_proprio=True, center_crop=True, unnorm_key="libero", num_open_loop_steps=8)
def obs():
    return {"full_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), "wrist_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8),
            "state": rng.uniform(-1, 1, 8)}
for mode in ("eager", "graph"):
    vla.enable_graph_replay(mode == "graph")
    for _ in range(3):
        a = utils.get_vla_action(c, vla, proc, obs(), "pick up the black bowl", action_head=head, proprio_projector=pp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        a = utils.get_vla_action(c, vla, proc, obs(), "pick up the black bowl", action_head=head, proprio_projector=pp)
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"get_vla_action end to end ({mode}): {dt:.2f} ms/call = {1e3 / dt:.1f} chunks/s; actions {np.stack(a).shape}, finite {np.isfinite(np.stack(a)).all()}")
