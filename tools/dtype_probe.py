import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for dt in (torch.uint16, torch.uint8, torch.int32, torch.bfloat16):
    x = torch.zeros(8, dtype=dt, device="cuda")
    try:
        dist.all_reduce(x) if dt != torch.uint16 else dist.broadcast(x, 0)
        torch.cuda.synchronize()
        print("RESULT", dt, "ok")
    except Exception as e:
        print("RESULT", dt, "fails:", str(e)[:100])
dist.destroy_process_group()
