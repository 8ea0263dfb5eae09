"""What the c10d flight recorder says about a finished collective (world 1, RCCL)."""
import os, pickle, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
t0 = time.perf_counter()
from torch._C._distributed_c10d import _dump_nccl_trace
for i in range(40):
    d = pickle.loads(_dump_nccl_trace())
    ent = d.get("entries", [])
    print(round(time.perf_counter() - t0, 3), [(e.get("state"), e.get("retired")) for e in ent][-3:])
    if ent and all(e.get("retired") for e in ent):
        break
    time.sleep(0.05)
print(sorted(ent[-1].keys()) if ent else d.keys())
dist.destroy_process_group()
