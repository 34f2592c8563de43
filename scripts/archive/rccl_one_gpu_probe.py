#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE GPU when each rank claims a host of its own (NCCL_HOSTID) and the ranks talk over the
loopback interface?  usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
scripts/rccl_one_gpu_probe.py"""
import os, sys
rank = int(os.environ["RANK"])
os.environ["NCCL_HOSTID"] = f"onegpu-rank{rank}"
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_DEBUG", "WARN")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.full((1024,), float(rank + 1), device="cuda:0", dtype=torch.float64)
dist.all_reduce(x)
torch.cuda.synchronize()
g = [torch.zeros(4, device="cuda:0", dtype=torch.float64) for _ in range(dist.get_world_size())]
dist.all_gather(g, torch.full((4,), float(rank), device="cuda:0", dtype=torch.float64))
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce -> {x[0].item()} all_gather -> {[t[0].item() for t in g]}", flush=True)
dist.barrier()
dist.destroy_process_group()
