# experiment only: upper bound of what the 132 MB backward-temp memset costs the step (results of > 4-tile
# Gaussians are WRONG in this mode; timing only)
import ctypes as C, os, runpy, sys
sys.path.insert(0, os.getcwd())
import torch
import pixelsplat_amd.raster as R
from pixelsplat_amd import _lib
def _no_memset(cfg, capacity, dev):
    lib = _lib.load(); d = cfg.desc()
    temp = torch.empty(lib.ps_raster_backward_temp_bytes(C.byref(d), capacity), dtype=torch.uint8, device=dev)
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
    return temp, ev
R._zeroed_backward_temp = _no_memset
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
