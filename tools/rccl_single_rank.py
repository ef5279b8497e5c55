"""RCCL bring-up of physicl_amd.dist.CounterComm with ONE rank (all a one-GPU box allows): new_group("nccl"), the
probe all-reduce and a counter all-reduce on a device tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
import numpy as np
from physicl_amd.dist import CounterComm
c = CounterComm(0, 1, "nccl", 0, _init=False)
c.world = 1
c._torch, c._dist = torch, dist
torch.cuda.set_device(0)
c._dev = torch.device("cuda", 0)
c._init_group("nccl")
print("backend after init:", c.backend, "group:", c._group)
# exercise the collective path itself on the RCCL group with one rank
c.world = 2   # force the collective branch
out = c.allreduce_sum(np.arange(160, dtype=np.int64))
print("allreduce ok:", out[:5], out.sum())
c.world = 1
dist.destroy_process_group()
