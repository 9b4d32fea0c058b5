"""bench.py --dry-run: the multi-GPU bookkeeping (which client trains on which rank / device, what the all-reduce moves) without a GPU.
Reference: FedavgServer._sample_clients assigns device cuda:(i % n_gpu) by position in the sorted sampled list (src/server/fedavgserver.py:310-311)
and update() aggregates once per round (:812-819)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("gpus,cpr", [(8, 1), (8, 2), (4, 3), (2, 1), (1, 1)])
def test_plan_matches_the_reference_device_assignment(gpus, cpr):
    d = run("--gpus", str(gpus), "--clients-per-rank", str(cpr))
    eff_cpr = cpr if gpus > 1 else 1
    assert d["n_gpus"] == gpus and d["clients_per_rank"] == eff_cpr and d["sampled_clients"] == gpus * eff_cpr
    for p, c in enumerate(d["plan"]):                          # position p -> cuda:(p % n_gpu), queued in order
        assert c["client"] == p and c["rank"] == p % gpus and c["device"] == f"cuda:{p % gpus}" and c["queue_position"] == p // gpus
    per_rank = {}
    for c in d["plan"]:
        per_rank.setdefault(c["rank"], []).append(c["queue_position"])
    assert sorted(per_rank) == list(range(gpus)) and all(q == list(range(eff_cpr)) for q in per_rank.values())
    # one fp32 all-reduce of the whole flat parameter buffer per round (ViT-S img+txt, vocab 7732, 32 tokens: 45.9 M parameters)
    assert 45_000_000 < d["params"] < 47_000_000
    ar = d["allreduce"]
    if gpus > 1:
        assert ar["collectives_per_round"] == 1 and ar["message_bytes"] == 4 * d["params"]
        assert abs(ar["ring_one_link_ms"] - 2 * (gpus - 1) / gpus * 4 * d["params"] / 153e9 * 1e3) < 0.01
        assert d["launch"][:4] == ["python", "-m", "torch.distributed.run", "--nnodes=1"] and f"--nproc-per-node={gpus}" in d["launch"]
        assert "127.0.0.1" in d["launch"] and d["launch_env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"} and "never exec" in d["launch_how"]
    else:
        assert ar["collectives_per_round"] == 0 and ar["message_bytes"] == 0
    assert d["pairs_per_step_all_ranks"] == gpus * 64


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_spawns_exactly_the_dry_run_command_as_a_child_before_any_gpu_call(monkeypatch):
    """`bench.py --gpus 8` outside a torchrun job: the launcher must start `python -m torch.distributed.run ... bench.py --gpus 8 ...` as a CHILD
    process (never os.exec*: on this pool replacing a process that has initialised the GPU takes the machine down), with the rendezvous on
    127.0.0.1 and dmabuf IPC in the environment, and must not have initialised the GPU runtime itself when it does so.  The argv and the
    environment are compared with what `--dry-run` prints."""
    import argparse
    import io
    import torch
    b = _bench_module()
    seen = {}

    class FakeChild:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen.update(cmd=list(cmd), env=dict(env), cuda_initialised=torch.cuda.is_initialized())
            self.stdout = io.StringIO('rank noise\n{"metric": "x"}\n')

        def wait(self):
            return 0
    monkeypatch.setattr(b.subprocess, "Popen", FakeChild)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    for name in [n for n in dir(os) if n.startswith("exec") or n.startswith("spawn") or n == "posix_spawn"]:
        monkeypatch.setattr(os, name, lambda *a, **k: (_ for _ in ()).throw(AssertionError("the launcher must not exec")))
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    rc = b.launch_ranks(argparse.Namespace(gpus=8), argv)
    monkeypatch.undo()
    assert rc == 0 and seen["cuda_initialised"] is False
    dry = run("--gpus", "8", "--steps", "20", "--warmup", "5")
    cmd = seen["cmd"]
    port = cmd[cmd.index("--master-port") + 1]
    assert 1024 < int(port) < 65536
    norm = ["python" if c == sys.executable else ("bench.py" if c == os.path.join(ROOT, "bench.py") else ("<free port>" if c == port else c)) for c in cmd]
    assert norm == dry["launch"], (norm, dry["launch"])
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    for k, v in dry["launch_env"].items():
        assert seen["env"][k] == v
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _negotiate_worker(rank, world, port, fail_rank, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench_module()
        closed = []

        class FakeComm:
            def close(self):
                closed.append(rank)

        def make():
            if fail_rank == "hang":                      # a collective that never completes: every rank sits in it
                import time
                time.sleep(3600)
            if rank == fail_rank or fail_rank == "all":
                raise RuntimeError("librccl.so could not be loaded")
            return FakeComm()
        comm, err = b.negotiate_comm(dist, make, "cpu", timeout_s=(1.0 if fail_rank == "hang" else 60.0))
        if fail_rank == "hang":
            assert b.COMM_STUCK[0] is True
        q.put((rank, comm is None, err, closed, b.aggregate_path_name(world, comm, dist.get_backend())))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL", traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [1, "all", None, "hang"])
def test_a_failed_cabi_communicator_on_any_rank_moves_every_rank_to_torch_distributed(fail_rank):
    """bench.py at N > 1: if Comm.from_torch_dist() fails on ANY rank -- or never returns ("hang": the creation runs in a daemon thread with a
    time limit) -- every rank must give up its own C-ABI communicator and the line must still be produced on the torch.distributed path with
    the reason recorded (cabi_comm_error); two gloo ranks on the CPU."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, none, err, closed, path in res:
        assert none != "FAIL", err
        if fail_rank is None:
            assert none is False and err is None and closed == [] and path.startswith("C ABI fc_aggregate")
        else:
            assert none is True and err and path == "HIP blend + torch.distributed.all_reduce (gloo)"
            if fail_rank == 1:
                assert closed == ([0] if rank == 0 else []) and (("librccl" in err) if rank == 1 else ("another rank" in err))
            if fail_rank == "hang":
                assert "did not return within" in err and closed == []
