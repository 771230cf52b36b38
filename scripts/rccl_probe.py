"""One-rank RCCL probe: init the nccl backend, run the collectives bench.py uses."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.arange(12, dtype=torch.float64, device=dev).reshape(2, 2, 3)
parts = [torch.empty_like(x)]
dist.gather(x, parts, dst=0)
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
assert torch.equal(parts[0], x)
print("rccl probe ok", torch.cuda.get_device_name(0), "HSA_ENABLE_IPC_MODE_LEGACY=", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
dist.destroy_process_group()
