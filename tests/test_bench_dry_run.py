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
        assert "127.0.0.1" in d["launch"]
    else:
        assert ar["collectives_per_round"] == 0 and ar["message_bytes"] == 0
    assert d["pairs_per_step_all_ranks"] == gpus * 64
