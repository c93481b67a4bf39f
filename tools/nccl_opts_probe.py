"""RCCL at world size 1 with the process-group options bench.py uses for N > 1 (high-priority stream): does the call
pattern work on this torch / RCCL build?  (One GPU is enough to find an API error before the 8-GPU run does.)"""
import os
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
t = torch.ones(157381, device=dev)
w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
e = torch.cuda.Event(enable_timing=True); e.record()
w.wait()
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MIN); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("ok", float(t[0]), float(tt), dist.get_backend())
dist.destroy_process_group()
