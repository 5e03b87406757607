"""The RCCL leg of the data-parallel path on the one GPU a test box has: a single-rank `nccl` (= RCCL on ROCm) process group runs
the same GradReducer code the N-GPU job runs -- communicator creation with `device_id`, bucketed all-reduce launched on the comm
stream from `notify` / `all_reduce`, stream joins, barrier.  (Reduction over one rank is the identity; the multi-rank arithmetic is
covered by the world-size-2 gloo tests in test_dp_gloo.py.)"""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_grad_reducer_on_a_single_rank_rccl_group(dev):
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group already exists in this test process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    try:
        dp = importlib.import_module("openvla-oft_amd.dp")

        class Store:
            def __init__(self):
                self.flat_grad = {torch.float32: torch.arange(5_000_000, dtype=torch.float32, device=dev),
                                  torch.bfloat16: torch.ones(3_000_000, dtype=torch.float32, device=dev)}

        st = Store()
        red = dp.GradReducer([st], 2)          # world 2 from the reducer's point of view, so every code path runs
        red.notify(st, torch.float32, 3_000_000)    # a frontier in the middle of the buffer ships the whole buckets below it
        red.notify(st, torch.float32, 5_000_000)
        red.all_reduce()                            # the rest + the other dtype; joins the comm stream
        red.all_reduce()                            # idempotent until reset()
        torch.cuda.synchronize()
        assert torch.equal(st.flat_grad[torch.float32], torch.arange(5_000_000, dtype=torch.float32, device=dev))
        assert torch.equal(st.flat_grad[torch.bfloat16], torch.ones(3_000_000, dtype=torch.float32, device=dev))
        red.reset()
        st.flat_grad[torch.float32].mul_(2)
        red.all_reduce()
        dist.barrier()
        torch.cuda.synchronize()
        assert st.flat_grad[torch.float32][7].item() == 14.0
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()
