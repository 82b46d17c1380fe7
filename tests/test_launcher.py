"""The N > 1 launcher of bench.py (`python bench.py --gpus N` without torchrun) and the GPU count it relies on:
nothing here may touch the GPU runtime in the parent (a parent that has initialised HIP and then starts ranks is
what takes a node down), a failing rank must end the others at once, and the KFD-topology count must honour the
visibility variables.  CPU only."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _fake_topology(tmp_path, simd_counts):
    for i, sc in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (0 if sc else 96, sc))
    return str(tmp_path)


def test_visible_gpu_count_reads_the_kfd_topology(tmp_path):
    from caro_ai_amd import parallel
    root = _fake_topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024])  # 2 CPU nodes + 8 GPUs
    assert parallel.visible_gpu_count(root, env={}) == 8
    assert parallel.visible_gpu_count(root, env={"HIP_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert parallel.visible_gpu_count(root, env={"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "0,1"}) == 2
    assert parallel.visible_gpu_count(root, env={"CUDA_VISIBLE_DEVICES": "3"}) == 1
    assert parallel.visible_gpu_count(root, env={"HIP_VISIBLE_DEVICES": "0,9,1"}) == 1   # cut at the first invalid entry
    assert parallel.visible_gpu_count(root, env={"HIP_VISIBLE_DEVICES": ""}) == 0
    assert parallel.visible_gpu_count(root, env={"HIP_VISIBLE_DEVICES": "GPU-abc,GPU-def"}) == 2
    assert parallel.visible_gpu_count(str(tmp_path / "nowhere"), env={}) == 0


def _run_launcher(child_body, n, extra_env=None, timeout=60):
    """bench.self_launch in a fresh interpreter in which every torch.cuda entry point that could initialise the
    runtime raises: the launcher must get through without them"""
    child = os.path.join(ROOT, "gpurun_out", "_launcher_child_%d.py" % os.getpid())
    os.makedirs(os.path.dirname(child), exist_ok=True)
    with open(child, "w") as f:
        f.write(textwrap.dedent(child_body))
    prog = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        import torch
        def boom(*a, **k):
            raise AssertionError("the launcher touched torch.cuda")
        for name in ("device_count", "is_available", "init", "current_device", "set_device", "get_device_name",
                     "get_device_properties", "synchronize"):
            setattr(torch.cuda, name, boom)
        import bench
        t0 = time.monotonic()
        rc = bench.self_launch(%d, argv=[], script=%r, poll_s=0.05, timeout_s=30.0)
        assert not torch.cuda.is_initialized()
        print("RC", rc, "SECONDS %%.1f" %% (time.monotonic() - t0))
    """) % (ROOT, n, child)
    env = dict(os.environ, CARO_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    env.update(extra_env or {})
    try:
        return subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=timeout)
    finally:
        os.unlink(child)


def test_self_launch_starts_its_ranks_without_touching_the_gpu_runtime():
    r = _run_launcher("""
        import os
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["LOCAL_RANK"] == os.environ["RANK"]
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print("rank", os.environ["RANK"], "of", os.environ["WORLD_SIZE"], flush=True)
    """, 3)
    assert r.returncode == 0, r.stderr
    assert "RC 0" in r.stdout
    assert sorted(l for l in r.stdout.splitlines() if l.startswith("rank")) == ["rank %d of 3" % i for i in range(3)]


def test_self_launch_ends_the_other_ranks_when_one_fails():
    """rank 1 dies at once; the others would sit in their rendezvous for ten minutes: the launcher terminates them
    and reports rank 1's exit code within seconds (bench.py:181)"""
    t0 = time.monotonic()
    r = _run_launcher("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)
    """, 3)
    assert r.returncode == 0, r.stderr
    assert "RC 7" in r.stdout, (r.stdout, r.stderr)
    assert "terminating the other ranks" in r.stderr
    assert time.monotonic() - t0 < 30


def test_self_launch_refuses_more_ranks_than_gpus(tmp_path):
    topo = _fake_topology(tmp_path, [0, 1024])  # one CPU node, one GPU
    r = _run_launcher("print('never')", 2, extra_env={"CARO_SHARE_GPU": "", "CARO_KFD_TOPOLOGY": topo})
    assert r.returncode == 0, r.stderr
    assert "RC 2" in r.stdout and "never" not in r.stdout
    r = _run_launcher("print('one')", 1, extra_env={"CARO_SHARE_GPU": "", "CARO_KFD_TOPOLOGY": topo})
    assert "RC 0" in r.stdout and "one" in r.stdout


# ------------------------------------------------------------------ bench.py --selfcheck (VERDICT r5 task 5)
_SELFCHECK_CHILD = """
    import os, sys
    sys.path.insert(0, %r)
    os.environ["CARO_DIST_BACKEND"] = "gloo"
    from caro_ai_amd import parallel
    import bench
    parallel.init()
    rec = bench.run_selfcheck("cpu", engine=False)
    print("rank", os.environ["RANK"], "checks", ",".join(rec["checks"]), flush=True)
"""


def test_selfcheck_passes_on_three_gloo_ranks():
    r = _run_launcher(_SELFCHECK_CHILD % ROOT, 3, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "RC 0" in r.stdout, (r.stdout, r.stderr)
    lines = sorted(l for l in r.stdout.splitlines() if l.startswith("rank"))
    assert len(lines) == 3 and all("payload_all_gather" in l and "broadcast_weights" in l and "allreduce_grads" in l for l in lines)


@pytest.mark.parametrize("fault", ["identity", "header_all_gather", "payload_all_gather", "empty_flush", "gather_tuples",
                                   "allreduce_float64", "allreduce_counts", "broadcast_weights", "allreduce_grads"])
def test_selfcheck_failure_names_the_collective_and_ends_every_rank(fault):
    """the failure path: the last rank's contribution to ONE check is sabotaged (CARO_SELFCHECK_FAULT); every rank must
    leave with exit code 4 -- the launcher reports it -- and the failing check is named on stderr; the checks before it
    have passed"""
    r = _run_launcher(_SELFCHECK_CHILD % ROOT, 2, extra_env={"CARO_SELFCHECK_FAULT": fault}, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "RC 4" in r.stdout, (r.stdout, r.stderr)
    assert "selfcheck FAILED" in r.stderr and fault + ":" in r.stderr, r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("rank")]  # nobody got past the check
