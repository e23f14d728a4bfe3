"""The RCCL leg of the N-GPU path, executed for real on the one-GPU box: a ONE-rank communicator
(PIXELSPLAT_FORCE_COMM=1, backend "nccl" = RCCL).  A one-rank all-reduce changes no value, but the rest
is what N ranks run: `init_process_group("nccl", device_id=...)`, the buckets' asynchronous collectives
on RCCL's own stream ordered against the compute stream, `finish()`'s waits, the exposed-wait timing,
hipGraph capture and replay beside a live communicator (watchdog thread, `thread_local` capture mode),
and the communicator block of the bench line.  The two-rank schedule itself is covered on gloo
(tests/test_zz_bench_ranks_gpu.py, tests/test_parallel_cpu.py); only xGMI traffic between devices is left
to the driver's 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_REDUCER_SCRIPT = r"""
import json, torch
from pixelsplat_amd import parallel as P
rank, world, local = P.init_from_env()
assert (rank, world) == (0, 1) and P.active(world)
import torch.distributed as dist
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.GELU(), torch.nn.Linear(512, 256),
                          torch.nn.Linear(256, 8)).to(dev)            # last layer unused
x = torch.randn(4096, 256, device=dev)
used = list(net[:3].parameters())
ref = [g.clone() for g in torch.autograd.grad(net[:3](x).square().mean(), used)]
red = P.GradientReducer(list(net.parameters()), world, bucket_bytes=256 << 10, extra_payload_bytes=8 << 20)
ok = True
for step in range(3):
    for p in net.parameters():
        p.grad = None
    net[:3](x).square().mean().backward()
    red.launch_extra_payload()
    y = (x @ x.t()[:, :512]).sum()          # compute-stream work the collectives overlap with
    red.finish()
    ok &= all(torch.equal(p.grad, g) for p, g in zip(used, ref))
    ok &= all(float(p.grad.abs().max()) == 0.0 for p in net[3].parameters())
launches_hooks = red.stats["launches"]
red.remove()
for p, g in zip(used, ref):
    p.grad = g.clone()
red.reduce_now(); red.finish()
ok &= all(torch.equal(p.grad, g) for p, g in zip(used, ref))
info = P.comm_info(world, dev)
print(json.dumps(dict(ok=bool(ok), stats=red.stats, launches_hooks=launches_hooks, info=info,
                      exposed=red.exposed_ms(), y=float(y))))
P.barrier(world); P.shutdown(world)
"""


def _env():
    env = dict(os.environ, PIXELSPLAT_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PIXELSPLAT_DIST_BACKEND"):
        env.pop(k, None)
    return env


def _last_json(out, strict=True):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    if strict:      # bench.py: ONE line on stdout (RCCL's version banner is kept off it)
        assert len(lines) == 1, out.stdout[-2000:]
    return json.loads([ln for ln in lines if ln.startswith("{")][-1])


def test_gradient_reducer_over_rccl(gpu_device):
    out = subprocess.run([sys.executable, "-c", _REDUCER_SCRIPT], env=_env(), capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    rec = _last_json(out, strict=False)
    assert rec["ok"]                                            # reduced == local gradients, bit for bit
    st, info = rec["stats"], rec["info"]
    assert st["steps"] == 4 and st["rebucketed"] and st["buckets"] >= 3
    assert rec["launches_hooks"] > 0 and st["launches"] > rec["launches_hooks"]
    assert st["launches_before_finish"] > 0                     # buckets launched from the hooks
    assert info["rccl_nranks"] == 1 and info["rccl_version"]
    assert len(rec["exposed"]) == 4 and all(e >= 0.0 for e in rec["exposed"])


@pytest.mark.parametrize("mode", ["eager", "auto"])
def test_bench_step_beside_a_live_rccl_communicator(gpu_device, mode):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--size", "64",
           "--batch", "1", "--no-cpu-baseline", "--no-probes", "--launch", mode, "--grad-payload-mb", "16"]
    rec = _last_json(subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT))
    comm = rec["comm"]
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["step_check"]["ok"], rec["step_check"]
    assert comm["backend"] == "nccl" and comm["forced_one_rank_communicator"] and comm["rccl_nranks"] == 1
    assert comm["buckets"] >= 1 and comm["launches_total_at_end_of_timed_region"] >= 4 * comm["buckets"]
    assert comm["extra_payload_bytes_per_step"] == 16_000_000
    assert comm["exposed_ms_per_step"] >= 0.0
    if mode == "eager":
        assert rec["launch"] == "eager"
        assert comm["launches_before_finish_total"] == comm["launches_total_at_end_of_timed_region"]
    else:
        assert rec["launch"] in ("hipgraph", "eager")
        if rec["launch"] == "eager":
            assert rec["launch_fallback"]                       # a fallback must say why
