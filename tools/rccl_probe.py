"""What does this lease offer for RCCL?  (1) how many GPUs are visible, (2) a 1-rank `nccl` process group: bucketed all-reduce of the
flat gradient buffer through training.GradAllReducer (RCCL really executes), (3) optionally 2 ranks on ONE device (expected to be refused
by RCCL: duplicate GPU)."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd.training import GradAllReducer
rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
print(f"rank {rank}/{world}: visible GPUs = {torch.cuda.device_count()}", flush=True)
torch.cuda.set_device(0 if os.environ.get("ONE_DEVICE") else rank % max(1, torch.cuda.device_count()))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=rank, world_size=world)
g = torch.full((38_436_836,), float(rank + 1), device="cuda")
red = GradAllReducer(g)
for _ in range(3):
    g.fill_(float(rank + 1)); torch.cuda.synchronize(); t0 = time.perf_counter()
    if world > 1:
        red.start(); mul = red.wait()
    else:                                   # GradAllReducer skips the collective on one rank; call RCCL on its buckets directly
        work = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in red.buckets]
        for w in work: w.wait()
        mul = 1.0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
want = sum(range(1, world + 1))
print(f"rank {rank}: RCCL all-reduce of 153.7 MB in {len(red.buckets)} buckets over {world} rank(s): {dt*1e3:.2f} ms, sum ok = {bool((g == want).all())}, 1/world = {mul}", flush=True)
dist.destroy_process_group()
