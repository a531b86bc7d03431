import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.getcwd())
from gffx_amd import shard
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
print(shard.allgather_hit_counts(123, 456, device=dev))
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print(t.item())
dist.barrier(); dist.destroy_process_group(); print("nccl ok")
