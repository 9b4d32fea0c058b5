"""bench.py's multi-rank path (one client per rank, `_aggregate` as blend + all-reduce) on ONE device:
`--gpus 2` spawns two ranks (FC_BENCH_ONE_DEVICE=1 puts both on cuda:0), and the aggregated global model they produce is checked
against the oracle's sequential blend (/root/reference/src/server/fedavgserver.py:656-664) of the weights each rank dumped."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, extra):
    d = str(tmp_path / "dump")
    env = dict(os.environ, FC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline", "--dump-agg", d] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    return p, d


@pytest.mark.parametrize("agg", ["torch", "cabi"])
def test_bench_two_ranks_one_device_matches_sequential_blend(tmp_path, agg):
    from oracle import aggregate_oracle as AO
    p, d = _run(tmp_path, ["--agg", agg])
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 128 and rec["allreduce_bytes"] > 180e6
    assert len(rec["per_rank_ms_per_step"]) == 2 and rec["aggregate_ms"] > 0 and rec["value"] > 0
    # the line validates itself at N > 1 (the driver's multi-GPU run is the first execution of RCCL with more than one rank)
    assert rec["rccl_ranks"] == 2 and rec["aggregate_GBps"] > 0 and rec["allreduce_bus_GBps"] > 0
    assert rec["agg_all_ranks_equal"] is True and rec["agg_slice_recomputed_ok"] is True and rec["agg_checksum_agree"] is True
    assert rec["agg_paths_agree"] in (None, True)            # None: only one path exists here (RCCL refuses two ranks on one device)
    plan = json.load(open(os.path.join(d, "plan.json")))
    g0 = torch.load(os.path.join(d, "global_before.pt"))
    g1 = torch.load(os.path.join(d, "global_after.pt"))
    cl = [torch.load(os.path.join(d, f"client{r}.pt")) for r in range(2)]
    assert not torch.equal(cl[0], cl[1])                       # the two ranks trained different clients
    seg = plan["segments"]
    view = lambda flat: {k: flat[seg[k][0]: seg[k][0] + seg[k][1]] for k in plan["keys"]}
    coef = {k: {i: plan["coef"][k][i] for i in range(2)} for k in plan["keys"]}
    exp = AO.sequential_blend(view(g0), {0: view(cl[0]), 1: view(cl[1])}, [0, 1], coef)
    got = view(g1)
    for k, v in exp.items():
        err = float((got[k] - v).abs().max())
        assert err <= 3e-6 * max(1.0, float(v.abs().max())), (k, err)
    # ranges outside the plan (alignment padding between segments) are untouched
    planned = torch.zeros(g0.numel(), dtype=torch.bool)
    for k in plan["keys"]:
        planned[seg[k][0]: seg[k][0] + seg[k][1]] = True
    assert torch.equal(g1[~planned], g0[~planned])             # (ViT-S segments are multiples of 64 elements: the toy-model check with real padding is tests/test_gpu_aggregate.py)
    if agg == "cabi" and not rec["aggregate_path"].startswith("C ABI"):
        pytest.skip("checked on the torch path: RCCL refuses two ranks on one device, the C-ABI communicator needs two GPUs")


def test_bench_more_sampled_clients_than_ranks_queue_and_preaccumulate(tmp_path):
    """Four sampled clients on two ranks (--clients-per-rank 2: positions p % world, fedavgserver.py:310-311): each rank trains its two
    clients one after the other and blends both into its partial before the one all-reduce; the global model equals the oracle's
    sequential blend of the four dumped clients, and the line's own checks hold."""
    from oracle import aggregate_oracle as AO
    p, d = _run(tmp_path, ["--agg", "torch", "--clients-per-rank", "2"])
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["clients_per_rank"] == 2 and rec["agg_checksum_agree"] is True and rec["agg_all_ranks_equal"] is True
    assert rec["allreduce_message_MB"] > 180 and "ring_one_link" in rec["xgmi_estimate_ms"] and rec["rccl_env"]
    plan = json.load(open(os.path.join(d, "plan.json")))
    g0, g1 = torch.load(os.path.join(d, "global_before.pt")), torch.load(os.path.join(d, "global_after.pt"))
    cl = [torch.load(os.path.join(d, f"client{c}.pt")) for c in range(4)]
    assert all(not torch.equal(cl[0], cl[c]) for c in (1, 2, 3))
    seg = plan["segments"]
    view = lambda flat: {k: flat[seg[k][0]: seg[k][0] + seg[k][1]] for k in plan["keys"]}
    coef = {k: {i: plan["coef"][k][i] for i in range(4)} for k in plan["keys"]}
    exp = AO.sequential_blend(view(g0), {c: view(cl[c]) for c in range(4)}, [0, 1, 2, 3], coef)
    got = view(g1)
    for k, v in exp.items():
        err = float((got[k] - v).abs().max())
        assert err <= 3e-6 * max(1.0, float(v.abs().max())), (k, err)


def test_bench_gpus_flag_without_enough_devices_fails_loudly(tmp_path):
    if torch.cuda.device_count() >= 64:
        pytest.skip("box has 64 GPUs")
    env = {k: v for k, v in os.environ.items() if k != "FC_BENCH_ONE_DEVICE"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "only" in p.stderr


def test_bench_single_gpu_line_carries_the_contract_and_the_reported_legs():
    """The N = 1 line: the driver's contract keys, the dropout line, the client-round / sustained legs and the fp32-mode leg (the <= 1e-4 parity mode on
    the matrix cores: slower than the bf16 headline, faster than 100 ms -- it was 200 ms on the VALU)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 8 and rec["dtype"] == "bf16" and rec["vs_baseline"] is None and "workload" in rec["config"]
    assert abs(rec["value"] - 64 / (rec["ms_per_step"] * 1e-3)) <= 0.01 * rec["value"]
    assert rec["dropout_0p1"]["ms_per_step"] > 0 and rec["sustained"]["seconds"] >= 5
    # the legs that match how the reference is run (scripts/flickr.sh: --B 112 --E 5; a --with_aux uni-modal client of a FedCola run)
    cr, c112, caux = rec["client_round"], rec["client_round_b112_e5"], rec["client_round_img_aux"]
    assert cr["ms_per_step"] > 0 and "error" not in c112 and "error" not in caux, (c112, caux)
    assert c112["B"] == 112 and c112["E"] == 5 and c112["steps_per_round"] == 40 and c112["ms_per_step"] > cr["ms_per_step"]
    assert caux["unit"] == "images/s" and 0.0 <= caux["acc1"] <= 1.0 and caux["ms_per_step"] > 0
    f32 = rec["fp32_mode"]
    assert f32["dtype"] == "fp32" and rec["ms_per_step"] < f32["ms_per_step"] < 100.0


def test_bench_keeps_its_line_when_the_cabi_communicator_cannot_be_created(tmp_path):
    """N > 1 with fc_comm_create failing on every rank (as a missing librccl would): the run continues on torch.distributed's all-reduce, the
    line says which path aggregated and why (cabi_comm_error), and still validates itself."""
    d = str(tmp_path / "dump")
    env = dict(os.environ, FC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", FC_BENCH_INJECT_COMM_FAILURE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline", "--agg", "cabi", "--dump-agg", d]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "injected failure" in rec["cabi_comm_error"] and rec["aggregate_path"].startswith("HIP blend + torch.distributed.all_reduce")
    assert rec["rccl_world"] == 2 and rec["dist_backend"] in ("gloo", "nccl") and rec["rccl_version"]
    assert rec["agg_all_ranks_equal"] is True and rec["agg_slice_recomputed_ok"] is True
