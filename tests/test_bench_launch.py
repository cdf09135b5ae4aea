"""bench.py --gpus N: the attempts it would run as fresh child processes (dry run: nothing touches a GPU), and the
launcher's bookkeeping with children that are not GPU programs at all."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dry(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMP_NUM_THREADS")}
    env["FROG_BENCH_LAUNCH_DRY_RUN"] = "1"
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    return r


def test_gpus_n_becomes_attempts_of_n_fresh_children():
    r = _dry(["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ranks"] == [0, 1, 2, 3]
    hosts = [(a["host"], a["transport"]) for a in out["attempts"]]
    assert hosts == [("preflight", "rccl"), ("native", "rccl"), ("torch", "nccl")]
    for i, a in enumerate(out["attempts"]):
        cmd = a["command"]
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py")
        assert cmd[2:8] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]          # the caller's arguments, unchanged
        assert cmd[cmd.index("--child") + 1] == a["host"] and cmd[cmd.index("--child-attempt") + 1] == str(i)


def test_rehearsal_backend_and_host_selection():
    r = _dry(["--gpus", "2"], {"FROG_BENCH_BACKEND": "gloo", "FROG_BENCH_HOSTS": "native"})
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert [(a["host"], a["transport"]) for a in out["attempts"]] == [("native", "shm")]


def test_a_rank_of_a_launched_run_checks_its_world_size():
    # WORLD_SIZE set (what torch.distributed.run gives its children) and --gpus disagreeing: an error, not a launch
    r = _dry(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_the_launcher_is_importable_without_torch_or_the_device_library():
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and 'frog_amd' not in sys.modules; "
            "print(json.dumps(bench.child_command(['--gpus', '2'], 'native', 'rccl', '/tmp/x', 1)))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout)
    assert cmd[-8:] == ["--child", "native", "--child-transport", "rccl", "--child-dir", "/tmp/x", "--child-attempt", "1"]


def test_orchestrator_with_failing_and_hanging_children(tmp_path, monkeypatch):
    """The launcher's bookkeeping on children that never touch a GPU: a preflight that fails puts the host-staged attempt
    in the queue and skips the RCCL ones; a child that hangs is killed at its time-out and reported with its stderr; the
    line that comes out carries every attempt."""
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "fake_child.py"
    fake.write_text(
        "import json, os, sys, time\n"
        "a = sys.argv; host = a[a.index('--child') + 1]; tr = a[a.index('--child-transport') + 1]\n"
        "rank = int(os.environ['RANK'])\n"
        "if host == 'preflight':\n"
        "    sys.stderr.write('rccl says no\\n'); sys.exit(5)\n"
        "if host == 'native' and tr == 'shm':\n"
        "    if rank == 0: print(json.dumps({'metric': 'm', 'value': 42.0, 'replicas_identical': True}))\n"
        "    sys.exit(0)\n"
        "time.sleep(60)\n")
    monkeypatch.setattr(bench, "child_command", lambda argv, host, tr, d, k: [sys.executable, str(fake), "--child", host, "--child-transport", tr])
    monkeypatch.setenv("FROG_BENCH_PREFLIGHT_TIMEOUT", "20")
    monkeypatch.setenv("FROG_BENCH_ATTEMPT_TIMEOUT", "3")
    monkeypatch.delenv("FROG_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("FROG_BENCH_HOSTS", raising=False)
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.orchestrate(None, [], 2, [0, 1], str(tmp_path / "run"))
    assert rc == 0
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert line["value"] == 42.0
    tried = line["hosts_tried"]
    assert [(t["host"], t["transport"]) for t in tried] == [("preflight", "rccl"), ("native", "rccl"), ("torch", "nccl"), ("native", "shm")]
    assert tried[0]["ok"] is False and "rccl says no" in tried[0]["stderr_tail"]["0"]
    assert tried[1].get("skipped") and tried[2].get("skipped") and tried[3]["ok"] is True

    # a hang: the RCCL preflight passes, the native attempt never returns
    fake.write_text(
        "import json, os, sys, time\n"
        "a = sys.argv; host = a[a.index('--child') + 1]; tr = a[a.index('--child-transport') + 1]\n"
        "rank = int(os.environ['RANK'])\n"
        "if host == 'preflight':\n"
        "    if rank == 0: print(json.dumps({'preflight': 'rccl', 'known_answers': True, 'latencies_rank0': {}}))\n"
        "    sys.exit(0)\n"
        "if host == 'native' and tr == 'rccl':\n"
        "    sys.stderr.write('stuck in a collective\\n'); sys.stderr.flush(); time.sleep(60)\n"
        "if rank == 0: print(json.dumps({'metric': 'm', 'value': 7.0 if tr == 'shm' else 9.0, 'replicas_identical': True}))\n")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.orchestrate(None, [], 2, [0, 1], str(tmp_path / "run2"))
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    tried = line["hosts_tried"]
    # the native RCCL attempt hung: the other RCCL host is not given the chance to sit out its own time-out, the host-staged
    # transport runs at once
    assert rc == 0 and line["value"] == 7.0
    assert [(t["host"], t["transport"]) for t in tried] == [("preflight", "rccl"), ("native", "rccl"), ("torch", "nccl"), ("native", "shm")]
    assert tried[1]["ok"] is False and tried[1]["return_codes"] == [-9, -9] and tried[1]["timed_out"] is True
    assert "stuck in a collective" in tried[1]["stderr_tail"]["0"] and tried[1]["seconds"] < 10
    assert "hung" in tried[2]["skipped"] and tried[3]["ok"] is True


def test_orchestrator_falls_back_when_every_rccl_attempt_fails_after_a_good_preflight(tmp_path, monkeypatch):
    """Preflight passes, both RCCL hosts fail fast: the host-staged attempt still produces the line."""
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "fake_child.py"
    fake.write_text(
        "import json, os, sys\n"
        "a = sys.argv; host = a[a.index('--child') + 1]; tr = a[a.index('--child-transport') + 1]\n"
        "rank = int(os.environ['RANK'])\n"
        "if host == 'preflight':\n"
        "    if rank == 0: print(json.dumps({'preflight': 'rccl', 'known_answers': True, 'latencies_rank0': {}}))\n"
        "    sys.exit(0)\n"
        "if tr != 'shm': sys.stderr.write('boom\\n'); sys.exit(3)\n"
        "if rank == 0: print(json.dumps({'metric': 'm', 'value': 5.0, 'replicas_identical': True}))\n")
    monkeypatch.setattr(bench, "child_command", lambda argv, host, tr, d, k: [sys.executable, str(fake), "--child", host, "--child-transport", tr])
    monkeypatch.delenv("FROG_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("FROG_BENCH_HOSTS", raising=False)
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.orchestrate(None, [], 2, [0, 1], str(tmp_path / "run"))
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert rc == 0 and line["value"] == 5.0
    assert [(t["host"], t["transport"], t["ok"]) for t in line["hosts_tried"]] == [
        ("preflight", "rccl", True), ("native", "rccl", False), ("torch", "nccl", False), ("native", "shm", True)]
    assert "boom" in line["hosts_tried"][1]["stderr_tail"]["0"]


def test_a_failed_rank_stops_the_others_waiting(tmp_path):
    """Rendezvous: a rank that dies before a collective tells the others, who stop waiting at once with its reason."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    a, b = bench.Rendezvous(str(tmp_path), 0, 0, 2), bench.Rendezvous(str(tmp_path), 0, 1, 2)
    b.fail("FrogError: frog_create: device index out of range")
    t0 = time.time()
    try:
        a.all_ready("context", timeout=30.0)
        raise AssertionError("rank 0 went on without rank 1")
    except RuntimeError as exc:
        assert "rank 1 of this attempt failed" in str(exc) and "device index out of range" in str(exc)
    assert time.time() - t0 < 5.0
    # and the ordinary case: both arrive, both go on; values travel
    c, d = bench.Rendezvous(str(tmp_path), 1, 0, 2), bench.Rendezvous(str(tmp_path), 1, 1, 2)
    d.put("context_1", b"1")
    c.all_ready("context", timeout=5.0)
    d.put("result_1", json.dumps({"x": 2}).encode())
    assert c.gather_json("result", {"x": 1}, timeout=5.0) == [{"x": 1}, {"x": 2}]
