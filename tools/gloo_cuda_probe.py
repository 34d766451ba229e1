"""Does the gloo backend accept CUDA tensors for the in-place all_gather_into_tensor the sharded path uses?
(two processes on ONE GPU: lets the N > 1 code path be exercised end to end on a 1-GPU box)"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
whole = torch.zeros(8, device="cuda")
mine = whole[rank * 4:(rank + 1) * 4]
mine.fill_(float(rank + 1))
dist.all_gather_into_tensor(whole, mine)
torch.cuda.synchronize()
t = torch.tensor([float(rank)], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
if rank == 0:
    print("gloo all_gather on CUDA tensors:", whole.tolist(), t.item())
dist.destroy_process_group()
