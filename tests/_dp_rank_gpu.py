"""Worker for tests/test_gpu_dp.py::test_two_ranks_on_one_gpu_*: one data-parallel rank of a KAN-VGG11 step.  Both ranks use cuda:0 (a 1-GPU lease) and
talk over gloo -- RCCL refuses two ranks on one device -- so everything above one rank EXCEPT the RCCL transport runs for real: bucket arming from the
autograd hooks of conv-KAN layers, gradient sinks written by the weight-gradient kernels, the side-stream collective, the mean over ranks, two
processes importing (and build-checking) the library at once."""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
out = sys.argv[1]
dist.init_process_group("gloo", rank=rank, world_size=world)
from convkan_amd.models import vggkan                      # noqa: E402
from convkan_amd.parallel import BucketedGradReducer      # noqa: E402
from convkan_amd import ops                                # noqa: E402

torch.manual_seed(3)                                       # same replica on every rank
model = vggkan(3, 10, arch="VGG11", kan_conv="KAN", dropout_linear=0.0).cuda().train()
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(32 * world, 3, 32, 32, device="cuda", generator=g)
t = torch.randint(0, 10, (32 * world,), device="cuda", generator=g)
xs, ts = x[rank * 32:(rank + 1) * 32], t[rank * 32:(rank + 1) * 32]
red = BucketedGradReducer(model.parameters())
assert red.world == world and red.cuda and not red.avg_in_collective and len(ops.GRAD_SINKS) > 0
for _ in range(2):                                         # twice: buckets re-arm, sinks are re-offered
    model.zero_grad(set_to_none=True)
    F.cross_entropy(model(xs), ts).backward()
    red.finish()
torch.cuda.synchronize()
views = {id(p): v for b in red.buckets for p, v in zip(b.params, b.views)}
assert all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in model.parameters())      # every gradient lives in its bucket
torch.save({n: p.grad.detach().cpu() for n, p in model.named_parameters()}, f"{out}.rank{rank}")
dist.barrier()
red.remove()
dist.destroy_process_group()
