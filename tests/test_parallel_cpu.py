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


# ---- gradient all-reduce (GradientReducer) ---------------------------------------------
def _reducer_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pixelsplat_amd import parallel as P

    P.init_from_env("gloo")
    torch.manual_seed(0)                       # same initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16),
                              torch.nn.Tanh(), torch.nn.Linear(16, 3))
    unused_here = torch.nn.Parameter(torch.ones(7))     # used by rank 0 only
    never_used = torch.nn.Parameter(torch.ones(3))
    params = list(net.parameters()) + [unused_here, never_used]
    # 400-byte buckets: several buckets, some holding more than one tensor
    red = P.GradientReducer(params, world, bucket_bytes=400)
    gen = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 6, generator=gen), torch.randn(8, 3, generator=gen)
    half = slice(rank * 4, rank * 4 + 4)
    results = []
    for step in range(2):                      # twice: the hooks re-arm after finish()
        for p in params:
            p.grad = None
        loss = (net(x_all[half]) - y_all[half]).square().mean()
        if rank == 0:
            loss = loss + (unused_here * torch.arange(7.0)).sum()
        loss.backward()
        red.finish()
        results.append([p.grad.clone() for p in params])
    out[rank] = (results, dict(red.stats))
    P.shutdown(world)


def test_gradient_reducer_two_ranks_equals_single_process():
    """Two gloo ranks, each with half of a batch: the reduced gradients equal the single-process
    gradients of the whole batch (mean of the two half-batch means), bucket by bucket; a parameter
    used on one rank only gets that rank's gradient / world on BOTH ranks
    (find_unused_parameters semantics, /root/reference/src/main.py:95), one used nowhere zeros."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_reducer_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16),
                              torch.nn.Tanh(), torch.nn.Linear(16, 3))
    gen = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 6, generator=gen), torch.randn(8, 3, generator=gen)
    (net(x_all) - y_all).square().mean().backward()
    ref = [p.grad for p in net.parameters()] + [torch.arange(7.0) / 2, torch.zeros(3)]
    for rank in range(world):
        results, stats = res[rank]
        assert stats["buckets"] >= 3 and stats["launches"] >= 2 * 3
        # after the first step the parameter no rank uses sits in a bucket of its own at the end,
        # and on rank 0 (which uses everything else) every other bucket launches from the hooks
        assert stats["rebucketed"]
        if rank == 0:
            assert stats["launches_before_finish"] >= stats["buckets"] - 1
        assert stats["bytes_per_step"] == 4 * sum(r.numel() for r in ref)
        for step_grads in results:
            for got, want in zip(step_grads, ref):
                torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-7)
    for a, b in zip(res[0][0][1], res[1][0][1]):
        assert torch.equal(a, b)                # every rank holds the same reduced gradient


def test_gradient_reducer_single_rank_is_transparent():
    from pixelsplat_amd import parallel as P
    w = torch.nn.Parameter(torch.ones(4))
    red = P.GradientReducer([w], 1)
    (w * 3).sum().backward()
    red.finish()
    red.launch_extra_payload()
    assert torch.equal(w.grad, torch.full((4,), 3.0)) and red.stats["launches"] == 0


def test_launch_ranks_runs_the_script_as_n_ranks(capfd):
    """`python bench.py --gpus N` without a launcher re-executes itself through
    torch.distributed.run; here the same helper launches a two-rank CPU stand-in."""
    import json
    import sys
    from pixelsplat_amd import parallel as P

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_probe.py")
    rc = P.launch_ranks(2, probe, ["--steps", "3"], timeout=300)
    assert rc == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone prints the JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["t"] == 2.0 and rec["out"] == ["--steps", "3"]
    assert rec["grad"] == [0.0, 1.5, 3.0, 4.5, 6.0]     # mean of (1x, 2x) arange


def _reduce_now_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pixelsplat_amd import parallel as P

    P.init_from_env("gloo")
    ws = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 7)]
    red = P.GradientReducer(ws, world, bucket_bytes=1024)
    # static gradient tensors, as a replayed hipGraph leaves them (no accumulate hook fires)
    ws[0].grad = torch.arange(5.0) * (rank + 1)
    ws[1].grad = torch.full((300,), float(rank))
    keep = [w.grad for w in ws[:2]]
    red.reduce_now()
    red.finish()
    out[rank] = ([w.grad.clone() for w in ws], [w.grad is k for w, k in zip(ws, keep)], dict(red.stats))
    P.shutdown(world)


def test_gradient_reducer_hook_free_variant_for_graph_replays():
    """reduce_now(): gradients copied into the buckets, reduced, means copied BACK into the same
    gradient tensors (a replayed graph keeps writing into them); a parameter without a gradient
    counts as zero and receives the mean."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_reduce_now_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        grads, same_storage, stats = res[rank]
        torch.testing.assert_close(grads[0], torch.arange(5.0) * 1.5)
        torch.testing.assert_close(grads[1], torch.full((300,), 0.5))
        torch.testing.assert_close(grads[2], torch.zeros(7))
        assert same_storage == [True, True]
        assert stats["launches"] == stats["buckets"] >= 2


def _edge_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from pixelsplat_amd import parallel as P

    P.init_from_env("gloo")
    res = {}
    # (1) no parameters, only the synthetic payload (AttributeError before)
    red0 = P.GradientReducer([], world, extra_payload_bytes=4096, bucket_bytes=1024)
    red0.launch_extra_payload()
    red0.finish()
    res["payload_launches"] = red0.stats["launches"]
    # (2) ranks that seed differently before building the model: broadcast from rank 0
    torch.manual_seed(100 + rank)
    ws = [torch.nn.Parameter(torch.randn(6)), torch.nn.Parameter(torch.randn(3))]
    red = P.GradientReducer(ws, world, broadcast_parameters=True)
    res["weights"] = [w.detach().clone() for w in ws]
    # (3) reduce_now() twice with a parameter that never has a gradient of its own: the mean the
    # first finish() handed it must NOT be reduced again as if it were a fresh gradient
    ws[0].grad = torch.full((6,), float(rank + 1))
    seen = []
    for _ in range(3):
        ws[0].grad.fill_(float(rank + 1))       # what a replayed graph would write
        red.reduce_now()
        red.finish()
        seen.append((ws[0].grad.clone(), ws[1].grad.clone()))
    res["seen"] = seen
    # (4) uneven ranks: every rank's step time, the slowest one, whole-job throughput
    P.barrier(world)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))
    mine = time.perf_counter() - t0
    res["times"] = P.gather_over_ranks(mine, world)
    res["slowest"] = P.max_over_ranks(mine, world)
    res["value"] = P.aggregate_throughput(28, 1, world, res["slowest"])
    res["comm"] = P.comm_info(world)
    res["exposed"] = red.exposed_ms()
    out[rank] = res
    P.shutdown(world)


def test_four_ranks_uneven_times_and_reducer_edge_cases():
    """Four gloo ranks: `value` = ALL ranks' views over the SLOWEST rank's time with uneven
    per-rank step times, the per-rank times and the communicator's own rank count as the bench
    line reports them, and the three GradientReducer edge cases of ADVICE r2."""
    world, port = 4, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_edge_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        r = res[rank]
        assert r["payload_launches"] == 4                   # 4096 B in 1024-B pieces
        for a, b in zip(r["weights"], res[0]["weights"]):
            assert torch.equal(a, b)                       # rank 0's initial values everywhere
        for g0, g1 in r["seen"]:                           # mean of 1, 2, 3, 4 -- every time
            torch.testing.assert_close(g0, torch.full((6,), 2.5))
            torch.testing.assert_close(g1, torch.zeros(3))
        times = r["times"]
        assert len(times) == world and times == res[0]["times"]
        assert all(times[i] < times[i + 1] for i in range(world - 1))     # rank 3 is the slow one
        assert r["slowest"] == max(times) >= 0.2
        assert r["value"] == world * 28 / max(times)        # NOT the mean of per-rank rates
        assert r["comm"]["rccl_nranks"] == world
        assert len(r["exposed"]) == 3 and all(e >= 0 for e in r["exposed"])


# ---- forced one-rank communicator (PIXELSPLAT_FORCE_COMM=1) ------------------------------
def _forced_worker(rank, world, port, out):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    os.environ.update(PIXELSPLAT_FORCE_COMM="1", PIXELSPLAT_DIST_BACKEND="gloo")
    from pixelsplat_amd import parallel as P

    r, w, _ = P.init_from_env()
    assert (r, w) == (0, 1) and P.active(w) and P.force_comm()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3),
                              torch.nn.Linear(3, 3))        # the last layer stays unused
    x = torch.randn(7, 6)
    ref = [g.clone() for g in torch.autograd.grad(net[:3](x).square().sum(), list(net[:3].parameters()))]
    red = P.GradientReducer(list(net.parameters()), w, bucket_bytes=64, extra_payload_bytes=1024)
    assert red.active and red.stats["buckets"] > 1
    for _ in range(2):                                       # step 2 runs on the re-bucketed layout
        for p in net.parameters():
            p.grad = None
        net[:3](x).square().sum().backward()
        red.launch_extra_payload()
        red.finish()
        got = [p.grad.clone() for p in net[:3].parameters()]
        assert all(torch.equal(a, b) for a, b in zip(got, ref))          # mean over one rank
        assert all(float(p.grad.abs().max()) == 0.0 for p in net[3].parameters())
    launches_hooks = red.stats["launches"]
    red.remove()                                             # hook-free variant (graph replays)
    for p, g in zip(net[:3].parameters(), ref):
        p.grad = g.clone()
    red.reduce_now()
    red.finish()
    assert all(torch.equal(p.grad, g) for p, g in zip(net[:3].parameters(), ref))
    info = P.comm_info(w)
    out["res"] = (red.stats["steps"], launches_hooks, red.stats["launches"], red.stats["rebucketed"],
                  info["rccl_nranks"], P.max_over_ranks(2.5, w), P.gather_over_ranks(1.5, w),
                  len(red.exposed_ms()))
    P.barrier(w)
    P.shutdown(w)
    assert not P.active(w)


def test_forced_one_rank_communicator_runs_the_whole_reducer():
    """PIXELSPLAT_FORCE_COMM=1: a single rank builds a real process group, the hooks launch real
    collectives and the values do not change -- the mode the RCCL leg is run in on a one-GPU box
    (tests/test_zz_rccl_one_rank_gpu.py)."""
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_forced_worker, args=(1, 0, out), nprocs=1, join=True)
        steps, l_hooks, l_all, rebucketed, nranks, mx, gathered, n_exposed = out["res"]
    assert steps == 3 and rebucketed and nranks == 1
    assert l_hooks > 0 and l_all > l_hooks and n_exposed == 3
    assert mx == 2.5 and gathered == [1.5]


# ---- one launch mode for the whole job (bench.py --launch auto, VERDICT r3 #7 / ADVICE r3) --------
def _launch_mode_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pixelsplat_amd import parallel as P

    _, w, _ = P.init_from_env("gloo")
    # stand-in for bench.py's sequence: warm-up (collectives), capture (none), agreement, timed steps
    P.barrier(w)
    rounds = []
    for failing_rank in (None, 1, 0):
        capture_ok = rank != failing_rank                      # the mocked hipGraph capture
        rounds.append(P.choose_launch_mode("auto", capture_ok, w))
    rounds.append(P.choose_launch_mode("eager", True, w))      # no collective issued for "eager"
    # the slowest rank sets the job's time whatever the launch mode: rank r takes (1 + r) s for 10 steps
    elapsed = P.max_over_ranks(1.0 + rank, w)
    out[rank] = (rounds, elapsed, P.aggregate_throughput(28, 10, w, elapsed),
                 P.gather_over_ranks(float(rank), w))
    P.barrier(w)
    P.shutdown(w)


def test_failed_capture_on_one_rank_sends_every_rank_to_eager():
    world, port = 3, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_launch_mode_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        rounds, elapsed, value, ranks = res[rank]
        assert rounds[0] == ("hipgraph", None)                 # every capture worked
        for failing, r in ((1, rounds[1]), (0, rounds[2])):    # one rank's capture failed: ALL eager
            assert r[0] == "eager"
            assert r[1] == ("this rank's capture failed" if rank == failing
                            else "another rank's capture failed")
        assert rounds[3] == ("eager", None)
        assert elapsed == 3.0                                  # the slowest rank's time ...
        assert value == 3 * 28 * 10 / 3.0                      # ... prices the whole job's views
        assert ranks == [0.0, 1.0, 2.0]


def test_exposed_wait_history_is_bounded():
    """finish() must not keep one timing record per step for ever."""
    from pixelsplat_amd import parallel as P

    os.environ["PIXELSPLAT_FORCE_COMM"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ["MASTER_PORT"] = str(_free_port())
    try:
        P.init_from_env("gloo")
        p = torch.nn.Parameter(torch.ones(5))
        red = P.GradientReducer([p], 1)
        for _ in range(300):
            p.grad = None
            (p * 2).sum().backward()
            red.finish()
        assert len(red.exposed_ms()) == 256 and len(red.exposed_ms(last=20)) == 20
        P.shutdown(1)
    finally:
        os.environ.pop("PIXELSPLAT_FORCE_COMM", None)
