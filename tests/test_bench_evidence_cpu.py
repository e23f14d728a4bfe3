"""The evidence plumbing of bench.py, on the CPU: a committed counter summary is paired with a live
kernel time only when the translation unit behind the kernel has the SAME hash in the summary's build
stamp and in the loaded library; the library's own stamp (ps_build_info) is the hash of the sources it
was built from; tools/kernel_table.py digests the committed summaries."""
import ctypes as C
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def bench():
    return _load("bench_under_test", os.path.join(ROOT, "bench.py"))


def test_counters_pair_only_with_the_same_build(bench, tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    for tag, h in (("old", "aaaaaaaaaaaa"), ("new", "bbbbbbbbbbbb")):
        doc = {"build": f"raster_sort:111111111111 raster_tiles:{h} epipolar_attention:cccccccccccc", "git": tag,
               "kernels": {"ps::tiles_backward_kernel(PsRasterDesc)": {"bytes": 3.0e9 if tag == "new" else 9.0e9},
                           "ps::tiles_forward_kernel(PsRasterDesc)": {"bytes": 2.5e9}}}
        (prof / f"{tag}_c2_pmc_traffic.json").write_text(json.dumps(doc))
        sq = {"build": doc["build"], "kernels": {"ps::tiles_backward_kernel(PsRasterDesc)":
                                                  {"valu_busy_ms_at_2.4GHz": 1.4 if tag == "new" else 9.9}}}
        (prof / f"{tag}_c2_pmc_sq.json").write_text(json.dumps(sq))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_TAG", "c2")

    monkeypatch.setattr(bench, "_LIB_HASHES", {"raster_tiles": "bbbbbbbbbbbb"})
    total, src = bench.pmc_traffic("tiles_backward")
    assert total == 3.0e9 and "new_c2_pmc_traffic.json" in src and "bbbbbbbbbbbb" in src
    assert bench.pmc_valu_busy_ms("tiles_backward") == 1.4

    monkeypatch.setattr(bench, "_LIB_HASHES", {"raster_tiles": "aaaaaaaaaaaa"})      # an older library
    assert bench.pmc_traffic("tiles_backward")[0] == 9.0e9

    monkeypatch.setattr(bench, "_LIB_HASHES", {"raster_tiles": "dddddddddddd"})      # code nobody profiled
    assert bench.pmc_traffic("tiles_backward") == (None, None)
    assert bench.pmc_valu_busy_ms("tiles_backward") is None
    monkeypatch.setattr(bench, "_LIB_HASHES", {})                                     # library without a stamp
    assert bench.pmc_traffic("tiles_forward") == (None, None)
    # a unit the summary does not list, and a group that is not one kernel
    monkeypatch.setattr(bench, "_LIB_HASHES", {"depth_sampler": "eeeeeeeeeeee"})
    assert bench.pmc_traffic("depth_sampler_forward") == (None, None)
    assert bench.pmc_traffic("depth_sort") == (None, None)
    monkeypatch.setattr(bench, "PMC_TAG", None)                                       # not a benchmarked config
    monkeypatch.setattr(bench, "_LIB_HASHES", {"raster_tiles": "bbbbbbbbbbbb"})
    assert bench.pmc_traffic("tiles_backward") == (None, None)


def test_every_single_kernel_group_names_its_translation_unit(bench):
    assert set(bench.SINGLE_KERNEL_GROUPS) == set(bench.GROUP_UNIT)
    csrc = os.path.join(ROOT, "pixelsplat_amd", "csrc")
    for group, unit in bench.GROUP_UNIT.items():
        src = open(os.path.join(csrc, unit + ".hip")).read()
        for kernel in bench.SINGLE_KERNEL_GROUPS[group]:
            name = kernel.split("<")[0]
            # names of removed kernels may stay listed (older summaries carry them); at least one is live
            if name in src:
                break
        else:
            raise AssertionError(f"{group}: none of {bench.SINGLE_KERNEL_GROUPS[group]} is defined in {unit}.hip")


def test_library_stamp_is_the_hash_of_its_sources():
    from pixelsplat_amd import _lib, build

    lib = _lib.load()
    lib.ps_build_info.restype = C.c_char_p
    info = lib.ps_build_info().decode()
    units = build.SOURCES + [(f, []) for f in sorted(os.listdir(build.CSRC))
                             if f.endswith(".hip") and f not in {s for s, _ in build.SOURCES}]
    want = build.unit_hashes(units)
    assert want in info, "libpixelsplat_hip.so was not built from the sources in the tree: run python -m pixelsplat_amd.build"
    stamp = dict(tok.split(":") for tok in want.split())
    assert {"raster_tiles", "raster_backward", "epipolar_attention", "gaussian_adapter", "depth_sampler"} <= set(stamp)
    assert all(len(h) == 12 for h in stamp.values())


def test_kernel_table_reads_the_committed_summaries():
    kt = _load("kernel_table_under_test", os.path.join(ROOT, "tools", "kernel_table.py"))
    tags = sorted({f[:-len("_pmc_sq.json")] for f in os.listdir(kt.PROF) if f.endswith("_pmc_sq.json")
                   and os.path.exists(os.path.join(kt.PROF, f[:-len("_pmc_sq.json")] + "_kernel_stats.csv"))
                   and os.path.exists(os.path.join(kt.PROF, f[:-len("_pmc_sq.json")] + "_pmc_traffic.json"))})
    assert tags, "no committed profile set"
    text = kt.table(tags[-1])
    rows = [ln for ln in text.splitlines() if ln.startswith("ps::")]
    assert any(r.startswith("ps::tiles_backward_kernel") for r in rows) and len(rows) > 20
    first = [r for r in rows if r.startswith("ps::tiles_backward_kernel")][0].split()
    assert float(first[2]) > 100.0 and float(first[3]) > 0.5      # avg us, traffic GB of the dominant kernel


def test_committed_gpu_test_log_is_of_this_build():
    """the committed GPU-suite log (profiles/r5_gpu_tests.log, written by
    tools/profile_all_configs.sh on the GPU box) starts with the `ps_build_info()` of the library that ran it.
    Every translation unit's hash in that stamp must equal the hash of the sources in the tree (a kernel change
    after the last full GPU suite -- round 4's failure -- turns this test red here, on the CPU), and the log must
    end in at least 125 passed tests and no failure."""
    import re

    from pixelsplat_amd import build

    logs = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_gpu_tests\.log", f)),
                  key=lambda f: int(f[1:f.index("_")]))
    assert logs, "no committed GPU-suite log: run tools/profile_all_configs.sh r<N> tests on the GPU box"
    path = os.path.join(ROOT, "profiles", logs[-1])      # the newest round's
    text = open(path).read()
    first = text.splitlines()[0]
    assert first.startswith("build:"), first[:80]
    stamped = dict(tok.split(":", 1) for tok in first.split("|", 1)[-1].split() if ":" in tok)
    units = build.SOURCES + [(f, []) for f in sorted(os.listdir(build.CSRC))
                             if f.endswith(".hip") and f not in {s_ for s_, _ in build.SOURCES}]
    want = dict(tok.split(":") for tok in build.unit_hashes(units).split())
    assert set(want) <= set(stamped), sorted(set(want) - set(stamped))
    stale = {u: (stamped[u], h) for u, h in want.items() if stamped[u] != h}
    assert not stale, f"the GPU suite was last run on other code than the tree holds: {stale}"
    tail = text.strip().splitlines()[-1]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 125 and "failed" not in tail and "error" not in tail.lower(), tail


def test_a_failing_rank_still_leaves_one_json_line(bench, capsys, monkeypatch):
    """where the process survives (an exception), rank 0 prints ONE line with value 0 and an
    `error` field; other ranks print nothing; nothing is printed twice."""
    monkeypatch.setattr(bench, "_STATE", {"json_out": None, "printed": False})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench._error_line(RuntimeError("peer closed the connection"))
    rec = json.loads(capsys.readouterr().out.strip())
    assert rec["value"] == 0.0 and rec["n_gpus"] == 8 and "peer closed" in rec["error"] and rec["unit"] == "views/s"
    monkeypatch.setenv("RANK", "3")
    bench._error_line(RuntimeError("x"))
    assert capsys.readouterr().out == ""
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(bench, "_STATE", {"json_out": None, "printed": True})     # the real line is already out
    bench._error_line(RuntimeError("x"))
    assert capsys.readouterr().out == ""


def test_newest_counter_summaries_carry_a_commit_of_this_history():
    """round 5's counter files were stamped with a round-4 commit (a stale .git_sha).  The
    newest round's profiles/r*_pmc_*.json must name a commit that is an ancestor of (or equal to) HEAD and not older
    than the last commit that touched the kernels -- tools/gpu_call.sh refreshes .git_sha before every gpurun call."""
    import re
    import subprocess

    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (GPU box snapshot)")
    prof = os.path.join(ROOT, "profiles")
    files = [f for f in os.listdir(prof) if re.fullmatch(r"r\d+_\w*pmc_\w+\.json", f)]
    assert files, "no committed counter summary"
    newest = max(int(re.match(r"r(\d+)_", f).group(1)) for f in files)
    if newest < 6:
        pytest.skip("counter summaries of rounds before the stamp was fixed")

    def git(*a):
        return subprocess.run(["git", "-C", ROOT, *a], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    last_csrc = git("log", "-1", "--format=%H", "--", "pixelsplat_amd/csrc", "include").stdout.strip()
    for f in sorted(f for f in files if f.startswith(f"r{newest}_")):
        rec = json.load(open(os.path.join(prof, f)))
        sha = (rec.get("git") or "").replace("-dirty", "")
        assert re.fullmatch(r"[0-9a-f]{40}", sha), f"{f}: no commit stamp ({rec.get('git')!r})"
        assert git("merge-base", "--is-ancestor", sha, "HEAD").returncode == 0, f"{f}: {sha} is not in this history"
        # the summary's per-unit source hashes are what pairs it with a build (bench.py); the commit must at least
        # not predate the last kernel change, unless the tree was dirty with exactly that change when it was taken
        if "-dirty" not in (rec.get("git") or ""):
            assert git("merge-base", "--is-ancestor", last_csrc, sha).returncode == 0, \
                f"{f}: stamped {sha[:12]}, older than the last kernel commit {last_csrc[:12]}"
