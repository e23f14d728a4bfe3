"""Tuned algorithm selection for the plain library GEMMs that stay next to the HIP kernels
(path (A): `x W_in^T`, `out W_o_t` and their input gradients; the encoder head's two linear
layers).  hipBLASLt's default heuristic runs the [57 344 x 128] x [128 x 592] pair at 115 / 99 us;
the solutions PyTorch's TunableOp found on gfx950 (rocBLAS / hipBLASLt ids in the csv next to
this file, produced with `python -m pixelsplat_amd.gemm_tuning --tune`) take 73 / 66 us:
0.26 ms of the 10.1 ms step.  `enable()` only LOADS that table -- no tuning at run time; a
GEMM shape that is not in it, or a library version that does not match the table's validators,
falls back to the default heuristic.
"""
from __future__ import annotations

import os

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(HERE, "gfx950_rocm7_torch2.10.csv")


def enable(path: str | None = None) -> bool:
    """Turns TunableOp on in look-up-only mode with the committed table.  Returns whether the
    table was accepted (False: the default heuristics stay in charge; nothing else changes)."""
    import torch

    if not torch.cuda.is_available():
        return False
    try:
        tun = torch.cuda.tunable
        tun.enable(True)
        tun.tuning_enable(False)
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)
        else:   # this build always writes its table at exit: send that to the temp directory
            import tempfile
            tun.set_filename(os.path.join(tempfile.gettempdir(),
                                          f"pixelsplat_tunableop_{os.getpid()}.csv"))
        return bool(tun.read_file(path or TABLE))
    except Exception:   # no TunableOp in this build, malformed / foreign table: keep the defaults
        try:
            torch.cuda.tunable.enable(False)
        except Exception:
            pass
        return False


def main(argv: list) -> None:
    """python -m pixelsplat_amd.gemm_tuning --tune [--out NAME] [-- bench.py arguments]:
    regenerates a table on the GPU by running bench.py with TunableOp tuning switched on."""
    import subprocess
    import sys

    if "--tune" not in argv:
        print(main.__doc__)
        return
    # arguments after "--" go to bench.py (another configuration: --context-views 3 --batch 4);
    # --out NAME picks the file to write (default _tuned -> _tuned0.csv)
    extra = argv[argv.index("--") + 1:] if "--" in argv else []
    name = argv[argv.index("--out") + 1] if "--out" in argv else "_tuned"
    out = os.path.join(HERE, name + ".csv")
    env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1",
               PYTORCH_TUNABLEOP_FILENAME=out, PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="15",
               PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="3", PIXELSPLAT_NO_TUNED_GEMMS="1")
    root = os.path.dirname(os.path.dirname(HERE))
    subprocess.check_call([sys.executable, os.path.join(root, "bench.py"), "--steps", "3",
                           "--warmup", "1", "--no-cpu-baseline", "--launch", "eager"] + extra,
                          env=env)
    print("wrote", out.replace(".csv", "0.csv"), "- review and merge it into", TABLE)
