"""Single-rank RCCL check of the collectives the multi-GPU path issues (torch.distributed backend "nccl" = RCCL):
init with a bound device, all-reduce of a complex128 panel viewed as f64 pairs, MAX all-reduce of a flag, barrier.
Usage: python tools/nccl_selftest.py"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.randn(50000, 64, dtype=torch.complex128, device="cuda")
ref = x.clone()
dist.all_reduce(torch.view_as_real(x), op=dist.ReduceOp.SUM)
torch.cuda.current_stream().synchronize()
flag = torch.tensor([3.0], dtype=torch.float64, device="cuda")
dist.all_reduce(flag, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl ok:", bool(torch.equal(x, ref)), float(flag.item()), torch.cuda.get_device_name(0))
dist.destroy_process_group()
