"""Development check: asynchronous all_gather_into_tensor on the RCCL backend with two alternating buffer pairs, the pattern bench.py
uses for N > 1 (one rank is enough to exercise the API on a one-GPU box)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mine=[torch.arange(1024, dtype=torch.uint8, device=dev) for _ in range(2)]
g=[torch.zeros(1024, dtype=torch.uint8, device=dev) for _ in range(2)]
pend=[None,None]
for k in range(6):
    b=k&1
    if pend[b] is not None: pend[b].wait()
    mine[b].add_(1)
    pend[b]=dist.all_gather_into_tensor(g[b], mine[b], async_op=True)
for p in pend: p.wait()
torch.cuda.synchronize()
print("ok", int(g[0][5]), int(g[1][5]))
dist.destroy_process_group()
