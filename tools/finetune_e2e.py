"""End-to-end fine-tune at full size from an episode store: synthetic LIBERO-shaped episodes -> EpisodeDataset -> RLDSBatchTransform ->
DeviceCollator (ovla_image_augment) -> VLAEngine step.  Reports the wall time per step INCLUDING the data path and the loss curve."""
import importlib, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
ft, synth = load("openvla-oft_amd.vla_scripts.finetune"), load("openvla-oft_amd.synthetic")
class Tok:
    vocab_size = 32000
    def __call__(self, text): return [1] + [3 + (sum(map(ord, w)) * 7919) % 30000 for w in text.split()]
root = Path(tempfile.mkdtemp())
synth.write_synthetic_episodes(root / "data", n_episodes=6, min_len=50, max_len=70)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = ft.FinetuneConfig(run_root_dir=root / "runs", data_root_dir=root / "data", dataset_name="libero_spatial_no_noops", batch_size=8, num_images_in_input=2,
                        use_proprio=True, max_steps=steps, save_freq=10 ** 9, wandb_log_freq=10, image_aug=True, learning_rate=5e-4)
stamps = []
def log(msg):
    stamps.append((time.perf_counter(), str(msg))); print(msg, flush=True)
t0 = time.perf_counter()
hist = ft.finetune(cfg, log=log, tokenizer=Tok())
torch.cuda.synchronize()
t1 = time.perf_counter()
step_lines = [(t, m) for t, m in stamps if m.startswith("step ")]
if len(step_lines) >= 3:
    (ta, _), (tb, _) = step_lines[1], step_lines[-1]
    n = 10 * (len(step_lines) - 2)
    print(f"steady state: {(tb - ta) / n * 1e3:.1f} ms per step incl. the data path = {8 * n / (tb - ta):.1f} samples/s over {n} steps; total {t1 - t0:.1f} s")
print("loss", [round(x, 3) for x in hist["loss_value"]])
