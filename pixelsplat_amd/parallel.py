"""One process per GPU, data parallel over independent scenes.

The kernels of the hot path never communicate: every rank renders its own batch of scenes.
The one collective of a data-parallel training step is the gradient all-reduce of the
parameters -- the reference gets it from Lightning's `ddp_find_unused_parameters_true`
(/root/reference/src/main.py:94-98: torch DDP, 25 MB buckets, NCCL) -- and `GradientReducer`
below is that step for this path on RCCL over xGMI: flat fp32 buckets filled by
post-accumulate hooks, each all-reduced asynchronously the moment its last gradient lands (so
the reduction of path (A)'s 6.6 M parameters runs under the rasterizer's backward), unused
parameters contributing zeros, one wait at the end of the step.

Also here: the launch contract plumbing `bench.py` and training scripts share -- rendezvous from
the torchrun environment (RCCL = backend "nccl" on ROCm; gloo on CPU for tests), self-launch of
N ranks, barrier, max-over-ranks timing, per-rank seeds (the reference seeds per rank too:
/root/reference/src/main.py:106, src/dataset/data_module.py:83-88).
"""
from __future__ import annotations

import collections

import os

import torch


def env_rank() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def force_comm() -> bool:
    """PIXELSPLAT_FORCE_COMM=1: build the communicator, install the gradient hooks and launch the
    collectives even for ONE rank.  A one-rank all-reduce changes no value, but it runs everything
    else -- `init_process_group("nccl", device_id=...)`, the buckets' asynchronous collectives on
    RCCL's own stream, their ordering against the compute stream, hipGraph capture and replay
    beside a live communicator -- which is how the RCCL leg of the N-GPU path is executed on a
    box with a single GPU (RCCL refuses two ranks on one device)."""
    return os.environ.get("PIXELSPLAT_FORCE_COMM") == "1"


def active(world: int) -> bool:
    """Is there a communicator to talk to?  More than one rank, or a forced one-rank group."""
    if world > 1:
        return True
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _free_port() -> int:
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for 1 rank unless
    `force_comm()`)."""
    rank, world, local = env_rank()
    if world > 1 or force_comm():
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:      # forced one-rank group without a launcher
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend is None:
            # PIXELSPLAT_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate
            # devices) -- how the N > 1 bench path is exercised on a single-GPU box
            backend = os.environ.get("PIXELSPLAT_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            if torch.cuda.is_available():
                local = local % torch.cuda.device_count()
                torch.cuda.set_device(local)
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)
    return rank, world, local


def barrier(world: int) -> None:
    if active(world):
        import torch.distributed as dist

        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value: float, world: int, device: torch.device | str = "cpu") -> float:
    if not active(world):
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def agree_all(ok: bool, world: int, device: torch.device | str = "cpu") -> bool:
    """True iff `ok` on EVERY rank (one MIN all-reduce).  Call it only at points where all ranks have
    issued the same collectives so far -- bench.py asks after the capture warm-up and after the capture,
    phases that contain no partial collective history (a rank that fails INSIDE a phase with collectives
    in flight must abort the job instead: its peers are waiting in a collective it will never join)."""
    return max_over_ranks(0.0 if ok else 1.0, world, device) < 0.5


def choose_launch_mode(requested: str, local_capture_ok: bool, world: int,
                       device: torch.device | str = "cpu") -> tuple[str, str | None]:
    """One launch mode for the whole job: ("hipgraph", None) only if every rank captured, else
    ("eager", reason).  `requested` = "eager" short-circuits without a collective."""
    if requested == "eager":
        return "eager", None
    if agree_all(local_capture_ok, world, device):
        return "hipgraph", None
    return "eager", ("this rank's capture failed" if not local_capture_ok
                     else "another rank's capture failed")


def gather_over_ranks(value: float, world: int, device: torch.device | str = "cpu") -> list[float]:
    """Every rank's value, in rank order (on every rank)."""
    if not active(world):
        return [float(value)]
    import torch.distributed as dist

    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def comm_info(world: int, device: torch.device | str = "cpu") -> dict:
    """What the communicator actually is, for the bench line: backend, the number of ranks the
    collective library itself spans (checked by all-reducing a one per rank through it), the
    RCCL version torch was built against and the NCCL_* / RCCL_* knobs of the environment."""
    info = dict(rccl_nranks=None if active(world) else 1, rccl_version=None, env={})
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:       # CPU-only build / no RCCL: stays None
        pass
    info["env"] = {k: v for k, v in os.environ.items()
                   if k.startswith(("NCCL_", "RCCL_")) or k == "HSA_ENABLE_IPC_MODE_LEGACY"}
    if active(world):
        import torch.distributed as dist

        one = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        info["rccl_nranks"] = int(round(float(one.item())))
        info["devices_visible"] = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return info


def rank_seed(base: int, rank: int) -> int:
    """Distinct synthetic batches per rank (weak scaling: per-GPU work fixed)."""
    return base + rank


def aggregate_throughput(units_per_rank_step: int, steps: int, world: int, elapsed_max: float) -> float:
    """Whole-job units/s: all ranks' units over the slowest rank's time."""
    return world * units_per_rank_step * steps / elapsed_max


def shutdown(world: int) -> None:
    if active(world):
        import torch.distributed as dist

        dist.destroy_process_group()


def launch_ranks(n: int, script: str, argv: list[str], timeout: float | None = None) -> int:
    """Re-runs `script argv` as `n` ranks of one node (`python bench.py --gpus N` without a
    launcher): the same command line `python -m torch.distributed.run --nnodes=1
    --nproc-per-node n --master-addr 127.0.0.1 --master-port P script argv` the driver uses,
    with a free port.  Returns the launcher's exit code; the ranks' stdout passes through."""
    import subprocess
    import sys

    port = _free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


class GradientReducer:
    """Bucketed, asynchronous all-reduce of parameter gradients -- what torch DDP does for the
    reference (main.py:94-98), hand-rolled so that it serves operators that are not one
    nn.Module.forward (the fused blocks of path (A) are called piecewise) and so that the
    launch point is explicit.

    * Buckets are flat fp32 buffers of at most `bucket_bytes`, filled in REVERSE parameter order
      (gradients arrive roughly in reverse order of use).  A post-accumulate hook copies each
      gradient into its slice, re-points `p.grad` at the slice (gradient-as-bucket-view: the
      optimizer reads reduced values without a copy back) and, when the bucket is complete,
      launches ONE `all_reduce(SUM, async)` on it after pre-dividing by the world size (mean).
      On RCCL the collective runs on the communicator's own stream, ordered after the gradient
      copies and overlapping whatever the compute stream does next.
    * `finish()` -- once per step, after the last backward -- launches the buckets that are still
      incomplete with zeros for the parameters that got no gradient (`find_unused_parameters`
      semantics: a parameter unused on this rank still takes part, other ranks may have used
      it), waits for every collective, and re-arms the hooks.
    * `extra_payload_bytes` adds a dummy buffer that `launch_extra_payload()` reduces in
      `bucket_bytes` pieces (asynchronously, awaited by `finish()`): it models the gradients of
      the rest of the network (the reference reduces ~0.48 GB per step) when only
      the hot path's own parameters exist.
    * world == 1: hooks are not installed, `finish()` is a no-op -- unless a one-rank communicator
      was forced (`force_comm()`), in which case everything runs as for N ranks.
    """

    def __init__(self, params, world: int, bucket_bytes: int = 25 << 20, average: bool = True,
                 extra_payload_bytes: int = 0, broadcast_parameters: bool = False,
                 device: torch.device | str | None = None, rebucket_after_first_step: bool = True):
        """`broadcast_parameters`: copy rank 0's parameter values to every rank first (torch DDP
        does; needed when ranks seed before building the model).  `device`: where the synthetic
        payload lives when there are no parameters.  `rebucket_after_first_step`: after the first
        `finish()` the parameters that received no gradient in that step move to buckets of their
        own at the end, so that the buckets of the USED parameters complete -- and launch -- from
        the hooks, under the rest of the backward, instead of waiting for `finish()` (DDP rebuilds
        its buckets after the first iteration for the same reason)."""
        self.world = world
        self.active = active(world)
        self.average = average
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in params if p.requires_grad]
        self.buckets: list[dict] = []
        self._where: dict[int, tuple[int, int]] = {}
        self._works: list = []
        self._hooks = []
        self._next = 0                    # next bucket to launch (index order)
        self._given: set[int] = set()     # parameters whose .grad the reducer supplied (unused on this rank)
        # (start, end) event pairs / seconds around finish()'s wait: the last 256 steps only (a long
        # training run must not keep two live HIP events per step for ever)
        self._exposed = collections.deque(maxlen=256)
        self._rebucket = rebucket_after_first_step
        self.extra = None
        self.extra_chunk = max(1, bucket_bytes // 4)
        self.stats = dict(buckets=0, bytes_per_step=0, launches=0, steps=0, launches_before_finish=0,
                          rebucketed=False)
        if not self.active:
            return
        if device is None:
            device = self.params[0].device if self.params else (
                torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                else torch.device("cpu"))
        if extra_payload_bytes > 0:
            self.extra = torch.zeros(extra_payload_bytes // 4, dtype=torch.float32, device=device)
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("GradientReducer: fp32 parameters only (the path is fp32)")
        if broadcast_parameters and self.params:
            import torch.distributed as dist

            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p, src=0)
        self._build_buckets([list(reversed(self.params))])
        self.install_hooks()

    # ---- buckets -----------------------------------------------------------------------
    def _build_buckets(self, ordered_lists) -> None:
        """Flat buffers over the given parameter lists (a bucket never spans two lists); existing
        bucket contents and `.grad` aliases move to the new views."""
        old = {id(p): v for b in self.buckets for p, v in zip(b["params"], b["views"])}
        groups = []
        for plist in ordered_lists:
            cur, cur_n = [], 0
            for p in plist:
                n = p.numel()
                if cur and (cur_n + n) * 4 > self.bucket_bytes:
                    groups.append(cur)
                    cur, cur_n = [], 0
                cur.append(p)
                cur_n += n
            if cur:
                groups.append(cur)
        self.buckets, self._where = [], {}
        for bi, grp in enumerate(groups):
            total = sum(p.numel() for p in grp)
            flat = torch.zeros(total, dtype=torch.float32, device=grp[0].device)
            off, views = 0, []
            for p in grp:
                self._where[id(p)] = (bi, len(views))
                view = flat[off:off + p.numel()].view_as(p)
                prev = old.get(id(p))
                if prev is not None:
                    view.copy_(prev)
                    if p.grad is not None and p.grad.data_ptr() == prev.data_ptr():
                        p.grad = view
                views.append(view)
                off += p.numel()
            self.buckets.append(dict(flat=flat, params=grp, views=views, ready=[False] * len(grp),
                                     pending=len(grp), launched=False))
        self.stats["buckets"] = len(self.buckets)
        self.stats["bytes_per_step"] = 4 * sum(b["flat"].numel() for b in self.buckets) + \
            (0 if self.extra is None else 4 * self.extra.numel())

    def install_hooks(self) -> None:
        if self.active and not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def reset(self) -> None:
        """Forget a half-finished step (e.g. after a failed hipGraph capture): waits for what was
        launched, re-arms every bucket."""
        for w in self._works:
            w.wait()
        self._works.clear()
        for b in self.buckets:
            b.pop("copy_back", None)
            b["ready"] = [False] * len(b["params"])
            b["pending"] = len(b["params"])
            b["launched"] = False
        self._next = 0

    def _launch(self, b: dict, in_finish: bool = False) -> None:
        import torch.distributed as dist

        if self.average:
            b["flat"].div_(self.world)
        self._works.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True))
        b["launched"] = True
        self.stats["launches"] += 1
        if not in_finish:
            self.stats["launches_before_finish"] += 1

    def _on_grad(self, p: torch.Tensor) -> None:
        bi, slot = self._where[id(p)]
        b = self.buckets[bi]
        if b["launched"]:
            raise RuntimeError("GradientReducer: a gradient arrived after its bucket was reduced "
                               "(two backward passes through the same parameter in one step: "
                               "call finish() between them or accumulate before reducing)")
        view = b["views"][slot]
        self._given.discard(id(p))
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        if not b["ready"][slot]:
            b["ready"][slot] = True
            b["pending"] -= 1
        # buckets launch in INDEX order only (as torch DDP does): collectives are matched by
        # issue order, and which bucket completes first may differ between ranks (a parameter
        # unused on one rank holds its bucket back there until finish())
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def reduce_now(self) -> None:
        """Hook-free variant for steps replayed from hipGraphs (the gradients are static tensors
        written by the graph, no accumulate hook fires): copies every gradient into its bucket
        and launches the buckets' all-reduces asynchronously.  `finish()` waits and copies the
        means back into the gradient tensors.  Parameters without a gradient of their own -- None,
        or a tensor a previous `finish()` supplied because the parameter was unused on this rank
        -- count as zero."""
        if not self.active or not self.params:
            return
        for b in self.buckets:
            if b["launched"]:
                raise RuntimeError("GradientReducer.reduce_now(): the previous step was not finished")
            dst, src, zero = [], [], []
            for slot, (p, view) in enumerate(zip(b["params"], b["views"])):
                own = p.grad is not None and id(p) not in self._given
                if not own:
                    zero.append(view)
                elif p.grad.data_ptr() != view.data_ptr():
                    dst.append(view)
                    src.append(p.grad)
                b["ready"][slot] = own
            # one multi-tensor launch per bucket instead of one copy per parameter (62 parameters
            # on path (A): the per-tensor copies were most of what the reducer added to a step)
            if zero:
                torch._foreach_zero_(zero)
            if dst:
                torch._foreach_copy_(dst, src)
            b["pending"] = 0
            b["copy_back"] = True
            self._launch(b)
        self._next = len(self.buckets)

    def launch_extra_payload(self) -> None:
        """All-reduce of the synthetic payload (the rest of the network's gradients), in
        bucket-sized pieces, asynchronously; call where that backward would run."""
        if not self.active or self.extra is None:
            return
        import torch.distributed as dist

        for off in range(0, self.extra.numel(), self.extra_chunk):
            self._works.append(dist.all_reduce(self.extra[off:off + self.extra_chunk],
                                               op=dist.ReduceOp.SUM, async_op=True))
            self.stats["launches"] += 1
            self.stats["launches_before_finish"] += 1

    def finish(self) -> None:
        """End of the step: reduce what is still incomplete (unused parameters count as zero),
        wait for every collective, re-arm."""
        if not self.active:
            return
        for b in self.buckets:
            if not b["launched"]:
                for slot, ok in enumerate(b["ready"]):
                    if not ok:
                        b["views"][slot].zero_()
                self._launch(b, in_finish=True)
        # exposed communication: how long the compute stream (or the host, without a GPU) stands
        # in the waits below, i.e. what the rest of the step did not hide
        on_gpu = bool(self._works) and torch.cuda.is_available() and (
            (self.buckets and self.buckets[0]["flat"].is_cuda) or
            (self.extra is not None and self.extra.is_cuda))
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        else:
            import time
            t0 = time.perf_counter()
        for w in self._works:
            w.wait()
        if on_gpu:
            e1.record()
            self._exposed.append((e0, e1))
        elif self._works:
            self._exposed.append(time.perf_counter() - t0)
        self._works.clear()
        for b in self.buckets:
            if b.pop("copy_back", False):      # reduce_now(): static gradient tensors keep their
                dst, src = [], []              # storage: copy back (one multi-tensor launch)
                for p, view, ok in zip(b["params"], b["views"], b["ready"]):
                    if ok and p.grad.data_ptr() != view.data_ptr():
                        dst.append(p.grad)
                        src.append(view)
                if dst:
                    torch._foreach_copy_(dst, src)
            # a parameter unused on THIS rank still receives the other ranks' mean gradient
            for p, view, ok in zip(b["params"], b["views"], b["ready"]):
                if not ok:
                    p.grad = view
                    self._given.add(id(p))
        if self._rebucket and self.stats["steps"] == 0 and self.buckets:
            # one-time: the parameters NO rank touched in this step get their own trailing
            # buckets (the decision is all-reduced: every rank must end up with the same layout)
            import torch.distributed as dist

            flags = torch.tensor([float(ok) for b in self.buckets for ok in b["ready"]],
                                 dtype=torch.float32, device=self.buckets[0]["flat"].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            flags = iter(flags.tolist())
            used, unused = [], []
            for b in self.buckets:
                for p in b["params"]:
                    (used if next(flags) > 0 else unused).append(p)
            if used and unused:
                self._build_buckets([used, unused])
                self.stats["rebucketed"] = True
        for b in self.buckets:
            b["ready"] = [False] * len(b["params"])
            b["pending"] = len(b["params"])
            b["launched"] = False
        self._next = 0
        self.stats["steps"] += 1

    def exposed_ms(self, last: int | None = None) -> list[float]:
        """Per finished step: milliseconds the compute stream (GPU) or the host (CPU backend)
        stood waiting for the collectives in `finish()`.  Synchronises the device."""
        if torch.cuda.is_available() and any(not isinstance(e, float) for e in self._exposed):
            torch.cuda.synchronize()
        vals = [e * 1e3 if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in self._exposed]
        return vals if last is None else vals[-last:] if last > 0 else []

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
