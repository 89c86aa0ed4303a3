import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
a = torch.zeros(4608, dtype=torch.uint8, device="cuda"); b = torch.zeros(4608, dtype=torch.uint8, device="cuda")
x = torch.randn(4096,4096, device="cuda")
for _ in range(5): dist.all_gather_into_tensor(b, a)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(100): dist.all_gather_into_tensor(b, a)
torch.cuda.synchronize(); print("allgather alone us/iter", (time.perf_counter()-t)*1e4)
t=time.perf_counter()
for _ in range(100):
    y = x @ x
    dist.all_gather_into_tensor(b, a)
torch.cuda.synchronize(); print("matmul+allgather us/iter", (time.perf_counter()-t)*1e4)
t=time.perf_counter()
for _ in range(100):
    y = x @ x
torch.cuda.synchronize(); print("matmul alone us/iter", (time.perf_counter()-t)*1e4)
dist.destroy_process_group()
