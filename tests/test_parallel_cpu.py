"""N > 1 launch contract on CPU: two gloo ranks on 127.0.0.1 (what bench.py does under
torchrun with RCCL), exercising rendezvous, barrier, max-over-ranks timing, per-rank seeds
and the whole-job throughput aggregation."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pixelsplat_amd import parallel as P
    from pixelsplat_amd.synthetic import make_cameras

    r, w, _ = P.init_from_env("gloo")
    assert (r, w) == (rank, world)
    P.barrier(w)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    t = P.max_over_ranks(elapsed, w)
    gen = torch.Generator().manual_seed(P.rank_seed(0, r))
    _, tgt = make_cameras(1, 2, 4, (64, 64), gen)
    out[rank] = (t, P.aggregate_throughput(28, 10, w, t), float(tgt.extrinsics[0, 0, 0, 3]))
    P.barrier(w)
    P.shutdown(w)


def test_two_rank_gloo_contract():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res[0][0] == res[1][0] == 2.0                    # max over ranks
    assert res[0][1] == 2 * 28 * 10 / 2.0                   # whole-job views/s
    assert res[0][2] != res[1][2]                           # ranks render different batches


def test_single_rank_is_a_noop():
    from pixelsplat_amd import parallel as P
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert P.env_rank() == (0, 1, 0)
    assert P.max_over_ranks(3.5, 1) == 3.5
    assert P.aggregate_throughput(28, 20, 1, 0.14) == 28 * 20 / 0.14
