"""bench.py --gpus N without a launcher around it: the command it would start (dry run: nothing touches a GPU)."""
import json
import os
import socket
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


def test_gpus_n_becomes_a_torch_distributed_run_of_n_children():
    r = _dry(["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = out["launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 <= port < 65536
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:        # the port was free when it was chosen
        sk.bind(("127.0.0.1", port))
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]      # the caller's arguments, unchanged
    assert out["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(out["env"]["OMP_NUM_THREADS"]) >= 1


def test_a_rank_of_a_launched_run_does_not_launch_again():
    # WORLD_SIZE set (what torch.distributed.run gives its children) and --gpus disagreeing: an error, not a second launch
    r = _dry(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert "launch" not in r.stdout


def test_launch_command_is_importable_without_torch_or_the_device_library():
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and 'frog_amd' not in sys.modules; "
            "print(json.dumps(bench.launch_command(2, ['--gpus', '2'], port=29555)))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout)
    assert cmd[cmd.index("--master-port") + 1] == "29555"
