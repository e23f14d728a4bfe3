"""`{"build": "<file>:<hash> ...", "git": "<sha>"}` of the library the process would load
(PIXELSPLAT_HIP_LIB honoured) -- what every counter summary under profiles/ records, and what
bench.py compares with the loaded library before it pairs live kernel times with committed
counters.  The git SHA comes from .git_sha (written before a gpurun call: the GPU box has no .git)
or from git itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stamp() -> dict:
    from pixelsplat_amd import _lib
    info = _lib.load().ps_build_info().decode()
    sha = None
    p = os.path.join(ROOT, ".git_sha")
    if os.path.isdir(os.path.join(ROOT, ".git")):
        try:
            sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
        except Exception:
            sha = None
    if sha is None and os.path.exists(p):
        sha = open(p).read().strip()
    return dict(build=info.split("|", 1)[1].strip() if "|" in info else info, git=sha)


if __name__ == "__main__":
    import json
    print(json.dumps(stamp()))
