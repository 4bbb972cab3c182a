import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from adcraft_amd import distributed as D
from adcraft_amd.engine import StepEngine
e = StepEngine(8, 16, device_id=0)          # the engine's own HIP use next to torch's context
v = D.all_reduce_sum(np.arange(10, dtype=np.float64), device="cuda")
dist.barrier(); torch.cuda.synchronize()
print("nccl world-1 all_reduce ok:", v[:4], "engine alive:", e.num_envs)
e.close(); dist.destroy_process_group()
