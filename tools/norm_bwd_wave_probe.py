"""Root-cause probe for the round-1 'wave-per-row LayerNorm backward is not reproducible beside the second vision stream' finding.

Runs the full-size fine-tune step (BASELINE.json configs[2]) several times from one state with the SigLIP-width LayerNorm backward
(dim 1152, no weight gradient) swapped for tools/probe/norm_bwd_wave_probe.hip = the removed kernel + a per-lane trace (s1 / s2
partials, hashes of every raw dword the lane loaded from x, dy, w, dx, and m1 / m2 as each lane received them from the reduction).
For every (call, row) whose dx differs between two runs it reports WHICH traced quantity differs first:
   hash of a loaded operand differs  -> the kernel READ different bytes: a memory-ordering / buffer-lifetime problem outside the kernel
   hashes equal, s1 / s2 differ      -> arithmetic inside the lane differs (cannot happen on deterministic hardware)
   partials equal, m1 / m2 differ    -> the cross-lane reduction differs
   all equal, dx differs             -> the accumulate / store path
Usage (GPU box):  python tools/norm_bwd_wave_probe.py [runs=3]"""
import ctypes, importlib, os, subprocess, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# PROBE_NOPK=1: the same source compiled WITHOUT packed-FP32 VALU instructions (-target-feature -packed-fp32-ops)
NOPK = os.environ.get("PROBE_NOPK", "0") == "1"
so = ROOT / "tools" / "probe" / ("libnormprobe_nopk.so" if NOPK else "libnormprobe.so")
if not so.exists():
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + (["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if NOPK else [])
                   + [str(so.with_name("norm_bwd_wave_probe.hip")), "-o", str(so)], check=True)
print(f"probe library: {so.name}", flush=True)
load = importlib.import_module
ops, engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.ops"), load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                   load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"))
probe = ctypes.CDLL(str(so))
probe.probe_norm_bwd.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False)
get, has = weights_mod.make_getter(sd, dev)
eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
del sd, get
batch = synth.make_batch(8, seed=1000)
orig, rec = ops.norm_bwd, []
KEEP_INPUTS = os.environ.get("PROBE_KEEP_INPUTS", "0") == "1"


def patched(x, dy, weight, mean, rstd, *, rms, dx=None, dx_accum=False, dweight=None, dbias=None):
    rows, dim = x.shape
    if not (dim == 1152 and dweight is None and dbias is None and not rms and dx is not None):
        return orig(x, dy, weight, mean, rstd, rms=rms, dx=dx, dx_accum=dx_accum, dweight=dweight, dbias=dbias)
    for t in (x, dy, dx):   # host-side operand checks before a hand-written kernel is launched
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape) == (rows, dim) and t.data_ptr() % 16 == 0
    assert weight.numel() == dim and mean.numel() == rows and rstd.numel() == rows and mean.dtype == rstd.dtype == torch.float32
    dbg = torch.empty((rows, 64, 8), dtype=torch.int32, device=x.device)
    rc = probe.probe_norm_bwd(x.data_ptr(), dy.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dbg.data_ptr(),
                              rows, dim, 0, int(dx_accum), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    # dy is not written by the kernel: it is cloned AFTER the launch (clones BEFORE it made the effect disappear: normprobe2.log)
    pre = (dy.clone(), None) if KEEP_INPUTS else None
    rec.append((dbg, dx.clone(), torch.cuda.current_stream().cuda_stream, pre, weight))
    return dx


ops.norm_bwd = patched
NAMES = ["s1", "s2", "hash(x)", "hash(dy)", "hash(w)", "hash(dx_in)", "m1", "m2"]


def campaign(label, n):
    runs = []
    for r in range(n):
        rec.clear()
        eng.zero_grad()
        eng.train_step_fwd_bwd(batch)
        torch.cuda.synchronize()
        runs.append(list(rec))
    print(f"== {label}: {len(runs[0])} probed calls per run, streams used {sorted(set(c[2] for c in runs[0]))}", flush=True)
    total = 0
    first_kind = {}
    for r in range(1, n):
        for ci, (c0, c1) in enumerate(zip(runs[0], runs[r])):
            (d0, o0), (d1, o1) = c0[:2], c1[:2]
            bad = (o0 != o1).any(dim=1).nonzero().flatten()
            if bad.numel() == 0:
                continue
            total += bad.numel()
            for row in bad[:3].tolist():
                a, b = d0[row].cpu(), d1[row].cpu()                      # [64, 8]
                diff_cols = [(NAMES[c], (a[:, c] != b[:, c]).nonzero().flatten().tolist()) for c in range(8) if (a[:, c] != b[:, c]).any()]
                uni = [len(set(t[:, c].tolist())) for t in (a, b) for c in (6, 7)]
                kind = ("loaded operand differs: " + ",".join(nm for nm, _ in diff_cols if nm.startswith("hash"))) if any(nm.startswith("hash") for nm, _ in diff_cols) else \
                       ("lane arithmetic" if any(nm in ("s1", "s2") for nm, _ in diff_cols) else ("reduction" if diff_cols else "accumulate/store"))
                first_kind[kind] = first_kind.get(kind, 0) + 1
                nel = int((o0[row] != o1[row]).sum())
                extra = ""
                if c0[3] is not None:    # were the operands, as cloned on the same stream right before the launch, identical?
                    extra = f" | post-launch clones of dy equal: {bool((c0[3][0][row] == c1[3][0][row]).all())}"
                if c0[3] is not None and diff_cols and diff_cols[0][0] == "s1" and not any(nm.startswith("hash") for nm, _ in diff_cols):
                    # per-lane s1 recomputed on the host in the kernel's own order: which run is right, and what does the wrong one look like?
                    lanes = diff_cols[0][1][:4]
                    g, wv = c0[3][0][row].float().cpu(), c0[4].float().cpu()
                    for ln in lanes:
                        terms = [(g[(ln + 64 * i) * 8 + j] * wv[(ln + 64 * i) * 8 + j]).item() for i in range(3) if ln + 64 * i < 144 for j in range(8)]
                        acc = torch.zeros((), dtype=torch.float32)
                        for t in terms:
                            acc = acc + torch.tensor(t, dtype=torch.float32)
                        va, vb = a[ln, 0:1].view(torch.float32).item(), b[ln, 0:1].view(torch.float32).item()
                        d = vb - va
                        near = min(range(len(terms)), key=lambda k: abs(abs(terms[k]) - abs(d)))
                        print(f"      lane {ln}: s1 run0 {va:.9e} run{r} {vb:.9e} host {acc.item():.9e} | run{r}-run0 {d:.3e}; closest single term g*w[{near}] = {terms[near]:.3e}; "
                              f"partial sums after each chunk: {[float(sum(terms[:8 * (k + 1)])) for k in range(len(terms) // 8)]}", flush=True)
                print(f"  run {r} call {ci} row {row}: {nel} dx elements differ | differing trace columns (name, first lanes, #lanes): "
                      f"{[(nm, l[:6], len(l)) for nm, l in diff_cols]} | distinct m1/m2 values within the wave (run0 m1, m2, run{r} m1, m2): {uni}{extra}", flush=True)
            if total > 400:
                break
    print(f"== {label}: {total} differing (call, row) pairs; classification of the inspected ones: {first_kind}", flush=True)
    return total


n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ["OVLA_VIT_STREAMS"] = "2"
t2 = campaign("two vision streams (SigLIP on the side stream)", n)
os.environ["OVLA_VIT_STREAMS"] = "1"
t1 = campaign("one stream (control)", n)
print(f"SUMMARY two-stream differing rows {t2}, one-stream differing rows {t1}")
