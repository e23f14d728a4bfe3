"""Rank body of tests/test_ddp_gpu.py: the connected training step (pixelsplat_amd/training_step.py) wrapped in
torch's DistributedDataParallel exactly as the reference's trainer does it -- Lightning's
`ddp_find_unused_parameters_true` strategy (/root/reference/src/main.py:94-98) is
DistributedDataParallel(find_unused_parameters=True) around the model, per-rank batches (main.py:106 seeds by rank).

    python -m torch.distributed.run --nproc-per-node N tests/_ddp_probe.py --steps K [--views 2]

Every rank: (1) one stand-alone step WITHOUT DDP -> its own parameter gradients; their mean over the ranks (an
all-reduce outside DDP) is what DDP must leave in `.grad`; (2) the same step through DDP, `--check-steps` times,
compared after each; (3) `--steps` more DDP steps (fault / leak soak), compared again at the end.  Rank 0 prints one
JSON line.  Warnings are recorded: torch's "AccumulateGrad node's stream does not match" (a gradient produced on a
side stream without handing it back to the stream its consumer runs on) must not appear."""
import argparse
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PIXELSPLAT_DETERMINISTIC", "1")      # the rasterizer's > 4-tile gradients without float atomics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check-steps", type=int, default=3)
    args = ap.parse_args()
    from pixelsplat_amd import parallel as P
    from tests.test_connected_gpu import GOLD, _build

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    rank, world, local = P.init_from_env()
    assert dist.is_initialized(), "the probe needs a process group (PIXELSPLAT_FORCE_COMM=1 for one rank)"
    dev = torch.device("cuda", local)
    g = np.load(GOLD)
    caught = []
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        net = _build(g, dev)                                   # identical weights on every rank (the golden's)
        # a parameter NO rank's step touches (the reference wraps modules with such parameters, which is why its
        # strategy is ddp_find_unused_parameters_true): the reducer must not wait for its gradient
        net.register_parameter("never_used", torch.nn.Parameter(torch.zeros(3, device=dev)))
        t = lambda k: torch.from_numpy(g[k]).to(dev)
        gen = torch.Generator().manual_seed(1000 + rank)       # per-rank batch: the golden's features, perturbed
        feats0 = t("features_in") + 0.1 * torch.randn(g["features_in"].shape, generator=gen).to(dev)
        context = {k: t("ctx_" + k) for k in ("extrinsics", "intrinsics", "near", "far")}
        target = {k: t("tgt_" + k) for k in ("extrinsics", "intrinsics", "near", "far")}
        target["image"] = t("target")
        params = dict(net.named_parameters())

        def run(module):
            feats = feats0.clone().requires_grad_(True)
            for p in params.values():
                p.grad = None
            out = module(feats, context, target, 0, True)      # deterministic depth sampling
            out.loss.backward()
            torch.cuda.synchronize()
            return float(out.loss), feats.grad.clone()

        # (1) stand-alone gradients and their mean over the ranks
        loss_alone, dfeat_alone = run(net)
        alone = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in params.items()}
        used_local = {n for n, a in alone.items() if a is not None}
        expect = {}
        for n, p in params.items():
            m = alone[n].clone() if alone[n] is not None else torch.zeros_like(p)
            dist.all_reduce(m)
            expect[n] = m / world
        rerun_loss, dfeat_again = run(net)                     # the step is reproducible (deterministic modes)
        repeat_err = max(float((p.grad - alone[n]).abs().max() / alone[n].abs().max().clamp_min(1e-30))
                         for n, p in params.items() if alone[n] is not None)

        # (2) through DDP
        ddp = DDP(net, device_ids=[local], find_unused_parameters=True, gradient_as_bucket_view=True)

        def compare():
            worst, worst_name, missing = 0.0, None, []
            for n, p in params.items():
                e = expect[n]
                scale = float(e.abs().max())
                if p.grad is None:
                    if scale != 0.0:
                        missing.append(n)
                    continue
                err = float((p.grad - e).abs().max()) / max(scale, 1e-30) if scale > 0 else float(p.grad.abs().max())
                if err > worst:
                    worst, worst_name = err, n
            return worst, worst_name, missing

        checks = []
        for _ in range(args.check_steps):
            loss_ddp, dfeat_ddp = run(ddp)
            w_, n_, miss = compare()
            checks.append(dict(worst=w_, where=n_, missing=miss, loss=loss_ddp,
                               dfeat=float((dfeat_ddp - dfeat_alone).abs().max() / dfeat_alone.abs().max())))
        # (3) soak
        for _ in range(args.steps):
            run(ddp)
        w_, n_, miss = compare()
        final = dict(worst=w_, where=n_, missing=miss)
        caught = [str(w.message)[:200] for w in wlist]
    stream_warnings = [m for m in caught if "stream does not match" in m or "AccumulateGrad" in m]
    unused = sorted(n for n in params if n not in used_local)
    rec = dict(world=world, backend=dist.get_backend(), rank=rank, loss_alone=loss_alone, repeat_err=repeat_err,
               checks=checks, final=final, soak_steps=args.steps, unused_parameters=unused,
               n_parameters=len(params), stream_warnings=stream_warnings,
               other_warnings=sorted({m for m in caught if m not in stream_warnings})[:8],
               mem_mb=round(torch.cuda.max_memory_allocated() / 2 ** 20, 1))
    allrec = [None] * world
    dist.all_gather_object(allrec, rec)
    if rank == 0:
        print(json.dumps(dict(ranks=allrec)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
