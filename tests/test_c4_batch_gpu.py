"""The batch of independent what-if networks at a size that crosses the lockstep group cap:
more than 128 contexts on one device (psn_solve_batch splits them into two groups driven by
two host threads on two streams), 16 of them full 1 000-machine / 10 000-pod networks.  Every
member's optimum equals the CPU oracle's and every member carries an optimality certificate;
the same batch, sharded graph b -> rank b mod G as the multi-GPU run shards it, reassembles
to the same per-graph results."""
import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import batch
from oracle import capi, synth

pytestmark = pytest.mark.gpu

N_FULL = 16     # full config-1 networks (533 200 forward arcs each)
N_SMALL = 116   # small clusters of assorted sizes
SMALL = [(20 + 3 * (i % 17), 150 + 25 * (i % 23)) for i in range(N_SMALL)]


def _member(i):
    """(product cluster, oracle graph) of batch member i; full-size members are spread through the
    batch so that both lockstep groups get some."""
    if i % 8 == 0 and i // 8 < N_FULL:
        seed = 1 + i // 8
        return pa.synth_cluster(1, seed), synth.build_graph(synth.make_cluster(1, seed))
    j = i - min(N_FULL, (i + 7) // 8)
    nm, npods = SMALL[j]
    return pa.synth_cluster(0, 100 + j, nm, npods), synth.build_graph(synth.make_cluster(0, 100 + j, nm, npods))


def test_batch_across_two_lockstep_groups_matches_oracle():
    B = N_FULL + N_SMALL
    assert B > 128
    solvers, graphs, want, clusters = [], [], [], []
    n_full = 0
    for i in range(B):
        cl, og = _member(i)
        n_full += cl.layout.n_machines == 1000
        net = pa.build_network(cl)
        s = pa.Solver()
        s.upload(net)
        s.eval_costs(cl, net)
        rc, tc, _, _, _ = capi.cost_scaling(og)
        assert rc == 0
        solvers.append(s); graphs.append(og); want.append(tc); clusters.append(cl)
    assert n_full == N_FULL
    res = batch.solve_batch(solvers, workers=2)
    assert [r.total_cost for r in res] == want
    assert all(r.status == 0 for r in res)
    places = []
    for s, og, r, cl in zip(solvers, graphs, res, clusters):
        resid, price = s.residual(), s.prices()
        assert capi.check(og, resid, price, scale=s.price_scale())[:2] == (0, r.total_cost)
        assert capi.check_eps(og, resid, price, 1, scale=s.price_scale()) == 0
        p = s.placements()
        P = cl.layout.n_tasks
        assert int((p >= 0).sum()) == int((og.cap - resid)[og.fwd_pos[:P]].sum())  # routed tasks = flow on task -> class arcs
        places.append(p)
    # the multi-GPU sharding rule on one device: ranks solve b mod G, results land in graph order
    for world in (2, 8):
        for rank in range(world):
            mine = batch.shard(B, world, rank)
            assert mine == [b for b in range(B) if b % world == rank]
    # a second, warm pass over the whole batch changes nothing
    res2 = batch.solve_batch(solvers, warm_start=True, workers=2)
    assert [r.total_cost for r in res2] == want
    for s in solvers:
        s.close()


def test_group_of_two_is_not_slower_path_but_same_answer():
    """Two contexts only: the batch call solves them one after the other (a lockstep group pays
    for itself from four members on); the answers are those of the oracle either way."""
    solvers, want = [], []
    for seed in (3, 4):
        cl = pa.synth_cluster(0, seed)
        net = pa.build_network(cl)
        s = pa.Solver()
        s.upload(net)
        s.eval_costs(cl, net)
        rc, tc, _, _, _ = capi.cost_scaling(synth.build_graph(synth.make_cluster(0, seed)))
        assert rc == 0
        solvers.append(s); want.append(tc)
    res = batch.solve_batch(solvers)
    assert [r.total_cost for r in res] == want
    # launches of a single solve are counted per context; a lockstep group would report the group's
    assert all(r.launches > 0 for r in res)
    for s in solvers:
        s.close()


def test_gather_placements_through_the_c_abi():
    """psn_solve_batch + psn_gather_placements: the whole batch of what-if rounds from one process,
    contexts spread over as many GPUs as the node shows (all on one when there is one), rows gathered
    into host memory and into memory of GPU 0; every row equals the context's own placements."""
    import ctypes as C

    import torch

    from poseidon_amd import _lib
    lib = pa.load_library()
    ndev = torch.cuda.device_count()
    B = 6
    solvers, clusters = [], []
    for b in range(B):
        cl = pa.synth_cluster(0, 30 + b, 40 + 4 * b, 300 + 50 * b)
        s = pa.Solver(device=b % ndev)
        net = s.build_network(cl)
        s.eval_costs(cl, net)
        solvers.append(s); clusters.append(cl)
    res = batch.solve_batch(solvers)
    for r, cl, b in zip(res, clusters, range(B)):
        rc, tc, _, _, _ = capi.cost_scaling(synth.build_graph(synth.make_cluster(0, 30 + b, 40 + 4 * b, 300 + 50 * b)))
        assert rc == 0 and r.total_cost == tc
    stride = max(cl.layout.n_tasks for cl in clusters)
    ctxs = (C.c_void_p * B)(*[s._ctx for s in solvers])
    lays = (_lib.Layout * B)(*[cl.layout for cl in clusters])
    host = np.full((B, stride), -9, np.int32)
    assert lib.psn_gather_placements(ctxs, B, lays, stride, -1, host.ctypes.data) == _lib.PSN_OK
    dev = torch.full((B, stride), -9, dtype=torch.int32, device="cuda:0")
    assert lib.psn_gather_placements(ctxs, B, lays, stride, 0, C.c_void_p(dev.data_ptr())) == _lib.PSN_OK
    torch.cuda.synchronize()
    for b, (s, cl) in enumerate(zip(solvers, clusters)):
        want = s.placements()
        P = cl.layout.n_tasks
        assert np.array_equal(host[b, :P], want) and (host[b, P:] == -9).all()
        assert np.array_equal(dev[b, :P].cpu().numpy(), want)
    # a stride shorter than a row is refused
    assert lib.psn_gather_placements(ctxs, B, lays, 10, -1, host.ctypes.data) == _lib.PSN_EINVAL
    for s in solvers:
        s.close()
