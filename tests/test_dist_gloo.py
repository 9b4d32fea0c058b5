"""world_size-2 CPU (gloo) test of the distributed aggregation path: each rank owns the sampled clients at positions
p % world == rank, forms its local partial (CPU stand-in for the HIP blend), one all-reduce sums the partials, and every
rank must end with the reference's sequentially blended global models (golden agg.json)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, idx, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import golden_util as G
    import host_util as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = G.load("agg.json")[idx]
        srv = H.make_server(rec)
        # drop the models of clients this rank does not own (they were "trained on another GPU")
        for pos, i in enumerate(rec["ids"]):
            if srv._owner_rank(pos, world) != rank:
                srv.clients[i].model = None
        for c in srv.clients:
            if c.id not in rec["ids"]:
                c.model = None
        H.run_aggregation(srv, rec, local_partial=H.cpu_local_partial)
        H.check_aggregation(srv, rec, tol=3e-6)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("idx", [0, 1, 4])
def test_two_rank_aggregation_matches_reference(idx):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, idx, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _many_worker(rank, world, port, idx, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import golden_util as G
    import host_util as H
    from fedcola_amd import aggregate as agg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = G.load("agg.json")[idx]
        srv = H.make_server(rec)
        for pos, i in enumerate(rec["ids"]):
            if srv._owner_rank(pos, world) != rank:
                srv.clients[i].model = None
        for c in srv.clients:
            if c.id not in rec["ids"]:
                c.model = None
        ids = rec["ids"]
        sizes = {i: len(srv.clients[i]) for i in ids}
        items, calls = [], []
        for n, ds in enumerate(srv.global_models.keys()):
            srv.global_model = srv.global_models[ds]
            srv.task, srv.modality = H.DS[ds]
            srv.dataset = ds
            srv.out_modality_scale = rec["out_modality_scales"][n]
            plan, flats, *_ = srv._aggregate_plan(ids, sizes)
            items.append((srv.global_model, plan, flats))

        def counted(t):
            calls.append(t.numel())
            dist.all_reduce(t)
        agg.aggregate_many(items, rank=rank, world=world, all_reduce=counted, local_partial=H.cpu_local_partial)
        assert calls == [sum(gm.flat.numel() for gm, _, _ in items)], calls      # ONE collective over the concatenated buffer
        H.check_aggregation(srv, rec, tol=3e-6)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("idx,world", [(1, 2), (4, 2), (0, 4), (2, 4)])
def test_three_global_models_share_one_all_reduce(idx, world):
    """fedavgserver.py:812-819 aggregates one global model per dataset; across ranks the three partials travel in ONE all-reduce
    (aggregate_many) and every rank ends with the reference's models (golden agg.json).  world = 4 with five or six sampled clients:
    ranks 0 and 1 queue two clients each (positions p % 4, fedavgserver.py:310-311) and pre-accumulate them in their partial, ranks 2 / 3
    own one or none -- the shape of the driver's `--gpus 4/8 --clients-per-rank 2` runs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, idx, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _cream_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import types
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fedcola_amd.server.creamflserver import CreamflServer
        srv = object.__new__(CreamflServer)
        srv.device = "cpu"
        clients = []
        for i, mod in enumerate(["img", "txt", "img+txt", "img"]):
            c = types.SimpleNamespace(id=i, modality=mod, pub_features=None)
            clients.append(c)
        srv._clients = clients
        ids = [0, 1, 2, 3]
        # each rank computed the features of the clients it owns (positions p % world == rank); a stale copy of another rank's
        # client must be replaced
        for pos, i in enumerate(ids):
            if pos % world == rank and clients[i].modality != "img+txt":
                clients[i].pub_features = torch.full((5, 3), float(10 * i + 1))
        if rank == 0:
            clients[1].pub_features = torch.full((5, 3), -7.0)        # stale: client 1 belongs to rank 1 this round
        srv._exchange_pub_features(ids)
        for i in (0, 1, 3):
            assert torch.equal(clients[i].pub_features, torch.full((5, 3), float(10 * i + 1))), (rank, i, clients[i].pub_features)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_creamfl_public_features_are_exchanged_between_ranks():
    """CreamflServer.update read c.pub_features of clients trained on other ranks (creamflserver.py:352-366)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _uid_fail_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fedcola_amd.comm import Comm

        def boom():
            raise RuntimeError("librccl.so could not be loaded")
        Comm.unique_id = staticmethod(boom)              # what a node without RCCL does on rank 0
        try:
            Comm.from_torch_dist()
            q.put((rank, "FAIL: no exception"))
        except RuntimeError as e:
            q.put((rank, "ok" if "rank 0 could not create an RCCL id" in str(e) and "librccl" in str(e) else f"FAIL: {e}"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_a_failed_rccl_id_on_rank_0_raises_on_every_rank_instead_of_hanging_them():
    """Comm.from_torch_dist broadcasts the 128-byte RCCL id from rank 0.  If rank 0 cannot create it (no librccl), the other ranks must not be
    left waiting in the broadcast: the failure travels in the broadcast's slot and every rank raises (bench.py then takes torch.distributed)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
