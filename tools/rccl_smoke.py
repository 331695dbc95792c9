"""Diagnostic: the torch.distributed / RCCL calls bench.py and loss.py make (init with device_id, all_gather_into_tensor, all_reduce MAX,
barrier), at whatever world size the launcher provides (1 on a one-GPU box)."""
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
rank, world, lr = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
x = torch.full((4, 6), float(rank), device=dev)
out = torch.empty(world * 4, 6, device=dev)
dist.all_gather_into_tensor(out, x)
t = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert out.shape[0] == world * 4 and float(t) == world
if rank == 0:
    print("rccl ok: world", world, "gathered", out[:, 0].tolist())
dist.destroy_process_group()
