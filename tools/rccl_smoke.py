"""RCCL API smoke test on whatever GPUs are visible (1 rank per GPU; works with a single GPU): the collectives the
view-parallel step uses.  torchrun --standalone --nproc-per-node N tools/rccl_smoke.py   (or plain python for N = 1)"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
x = torch.full((1 << 20,), float(rank + 1), device="cuda")
dist.all_reduce(x); assert float(x[0]) == world * (world + 1) / 2
g = torch.empty(world * 4, 3, device="cuda"); dist.all_gather_into_tensor(g, torch.full((4, 3), float(rank), device="cuda"))
assert float(g[-1, 0]) == world - 1
m = torch.tensor([rank], device="cuda", dtype=torch.int32); dist.all_reduce(m, op=dist.ReduceOp.MAX); assert int(m) == world - 1
t = torch.tensor([1.5 + rank], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
if rank == 0:
    print("rccl smoke ok: world", world, "backend", dist.get_backend())
dist.destroy_process_group()
