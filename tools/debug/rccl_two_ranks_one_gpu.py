import os, torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="env://")
torch.cuda.set_device(0)
x = torch.ones(1024, device="cuda:0") * (dist.get_rank() + 1)
try:
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ok", float(x[0]))
except Exception as e:
    print("rank", dist.get_rank(), "FAILED", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
