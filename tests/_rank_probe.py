"""Helper of tests/test_parallel_cpu.py: a stand-in for bench.py's rank body, launched through
pixelsplat_amd.parallel.launch_ranks (torch.distributed.run) on CPU with gloo."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelsplat_amd import parallel as P  # noqa: E402

rank, world, local = P.init_from_env("gloo")
w = torch.nn.Parameter(torch.zeros(5))
red = P.GradientReducer([w], world)
(w * torch.arange(5.0) * (rank + 1)).sum().backward()
red.finish()
P.barrier(world)
t = P.max_over_ranks(1.0 + rank, world)
if rank == 0:
    print(json.dumps(dict(n_gpus=world, grad=w.grad.tolist(), t=t, out=sys.argv[1:])), flush=True)
P.shutdown(world)
