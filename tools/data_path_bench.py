"""Throughput of the RLDS-free data path at the fine-tune recipe's shape (batch 8, two 224 x 224 cameras,
image_aug on): host part (frame fetch + RLDSBatchTransform), device collator (H2D copy of the uint8 frames + ovla_image_augment),
and the same image work by the CPU oracle (numpy float32 restatement of the TF ops) for scale."""
import importlib, sys, tempfile, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import data_oracle as do
load = importlib.import_module
synth, D, R, AT = (load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.prismatic.vla.datasets"), load("openvla-oft_amd.prismatic.vla.datasets.rlds_free"),
                   load("openvla-oft_amd.prismatic.vla.action_tokenizer"))
ops = load("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
class Tok:
    vocab_size = 32000
    def __call__(self, text): return [1] + [3 + (sum(map(ord, w)) * 7919) % 30000 for w in text.split()]
root = tempfile.mkdtemp()
synth.write_synthetic_episodes(root, n_episodes=8, min_len=60, max_len=80)
tok = Tok()
ds = D.RLDSDataset(root, "libero_spatial_no_noops", D.RLDSBatchTransform(AT.ActionTokenizer(tok), tok, use_wrist_image=True, use_proprio=True), image_aug=True)
col = D.DeviceCollator(2048, 32000, device=dev, image_aug=True, seed=0)
it = iter(ds)
B, iters = 8, 40
t0 = time.perf_counter(); insts = [[next(it) for _ in range(B)] for _ in range(iters)]; t_host = (time.perf_counter() - t0) / iters
for b in insts[:3]: col(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in insts: out = col(b)
torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / iters
frames = torch.from_numpy(np.stack([f for i in insts[0] for f in (i["image"], i["image_wrist"][0])])).to(dev)
prm = torch.from_numpy(R.sample_augment_params(np.random.default_rng(0), 16)).to(dev)
for _ in range(5): ops.image_augment(frames, prm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.image_augment(frames, prm)
e1.record(); torch.cuda.synchronize()
t_k = e0.elapsed_time(e1) / 50 * 1e-3
alg = 16 * (224 * 224 * 3 * (1 + 4 + 4) + 6 * 224 * 224 * 2)     # u8 read + fp32 intermediate write + read + bf16 write
p = R.sample_augment_params(np.random.default_rng(0), 4)
t0 = time.perf_counter()
for i in range(4): do.pixel_values(do.augment_image(insts[0][i]["image"], p[i]))
t_cpu = (time.perf_counter() - t0) / 4
print(f"host part (mmap frame fetch + ids/labels)      {t_host * 1e3:7.2f} ms / batch of {B}  -> {B / t_host:8.0f} samples/s")
print(f"device collator (H2D + ovla_image_augment)     {t_dev * 1e3:7.2f} ms / batch of {B}  -> {B / t_dev:8.0f} samples/s")
print(f"ovla_image_augment alone, 16 frames            {t_k * 1e6:7.1f} us  = {alg / t_k / 1e9:6.0f} GB/s of {alg / 1e6:.1f} MB algorithmic (HBM-bound; peak 8000)")
print(f"CPU oracle (numpy fp32), one frame             {t_cpu * 1e3:7.2f} ms  -> {1 / (2 * t_cpu):8.1f} samples/s on one core (2 frames per sample)")
print(f"whole loader                                   {(t_host + t_dev) * 1e3:7.2f} ms / batch  -> {B / (t_host + t_dev):8.0f} samples/s  (the training step consumes ~47 samples/s)")
