"""Stand-in worker for tests/test_bench_launcher.py: what bench.py's ranks do around the measurement, on gloo (no GPU)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if "--fail" in sys.argv and rank == world - 1:
    sys.exit(3)
if rank == 0:
    print("RCCL version : banner noise on stdout", flush=True)
    print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "launched": os.environ.get("KAN_BENCH_LAUNCHED")}), flush=True)
dist.barrier()
dist.destroy_process_group()
