"""bench.py host logic that needs no GPU: the PMC `traffic` figure is only reported while the kernel sources it was
measured on are unchanged."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _fake_tree(tmp_path):
    csrc = tmp_path / "parrot_amd" / "csrc"
    csrc.mkdir(parents=True)
    for name in ("skinny.hip", "skinny.h", "plans.hip", "plans_common.h", "att_fwd_body.h"):
        (csrc / name).write_text("// " + name + "\n")
    (tmp_path / "profiles").mkdir()
    return str(tmp_path)


def test_traffic_figure_fresh_stale_absent(tmp_path):
    import bench
    root = _fake_tree(tmp_path)
    assert bench.pmc_traffic_figure(root) == (None, None)  # nothing measured: null, no note
    blob = {"hbm_bytes_per_launch": 12345678, "session": "unit test", "source_digest": bench.kernel_source_digest(root)}
    path = os.path.join(root, "profiles", "r06_pmc_traffic.json")
    json.dump(blob, open(path, "w"))
    value, note = bench.pmc_traffic_figure(root)
    assert value == 12345678 and "unit test" in note and "r06_pmc_traffic.json" in note
    # a kernel source changes -> the figure is refused and the note says why
    with open(os.path.join(root, "parrot_amd", "csrc", "plans.hip"), "a") as f:
        f.write("// edited\n")
    value, note = bench.pmc_traffic_figure(root)
    assert value is None and "stale" in note
    # an unreadable file never yields a number
    open(path, "w").write("{not json")
    value, note = bench.pmc_traffic_figure(root)
    assert value is None and "unreadable" in note


def test_repo_has_no_stale_traffic_figure():
    """Whatever profiles/ holds for this tree: bench.py either reports a figure whose digest matches, or null."""
    import bench
    value, note = bench.pmc_traffic_figure()
    if value is not None:
        import re
        name = re.search(r"profiles/(\S+\.json)", note).group(1)  # the file the note says the figure comes from
        blob = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert blob["source_digest"] == bench.kernel_source_digest()
    else:
        assert note is None or "stale" in note or "unreadable" in note


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun launches two ranks itself: the collective spans
    both (`ranks_seen` = all-reduce of ones) and the line says n_gpus 2.  CPU/gloo, no GPU step (--plumbing-only)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PARROT_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--plumbing-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["allreduce_ms"] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 2 inside a 1-rank environment must fail, not silently report one GPU."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
