"""CPU tests of the host side: the shared library loads and exports every declared entry
point, the C++ network builder reproduces the oracle's arrays, argument errors come back
as codes (no GPU call is made here)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import _lib, network
from oracle import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden", "psn_synth_v1.json")


def test_library_exports_every_declared_symbol():
    lib = pa.load_library()
    declared = set()
    for hdr in ("psn.h", "psn_builder.h", "psn_service.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"\b(psn_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.EXPORTS) <= declared
    assert lib.psn_version() >= 100
    assert lib.psn_strerror(-4) and b"infeasible" in lib.psn_strerror(-4)


@pytest.mark.parametrize("cfg,seed", [(0, 1), (0, 2), (0, 3), (1, 1)])
def test_builder_matches_oracle(cfg, seed):
    cl = pa.synth_cluster(cfg, seed)
    net = pa.build_network(cl)
    ocl = synth.make_cluster(cfg, seed)
    g = synth.build_graph(ocl)
    for k in ("cap_cpu", "cap_mem", "used_cpu", "used_mem", "req_cpu", "req_mem"):
        assert np.array_equal(getattr(cl, k), getattr(ocl, k)), k
    assert np.array_equal(cl.task_class, ocl.pod_shape) and np.array_equal(cl.task_job, ocl.pod_job)
    assert np.array_equal(net.row_off, g.row_off) and np.array_equal(net.head, g.head)
    assert np.array_equal(net.rev, g.rev) and np.array_equal(net.supply, g.supply)
    # slot arcs are priced on the device; everything else must already be final
    slot = np.zeros(net.m2, bool)
    N, K = cl.layout.n_machines, cl.layout.k_slots
    for e in range(cl.layout.n_classes):
        b = int(net.ec_arc_base[e])
        slot[b:b + N * K] = True
    slot |= slot[net.rev]
    assert np.array_equal(net.cap[~slot], g.cap[~slot]) and np.array_equal(net.cost[~slot], g.cost[~slot])
    assert (net.cap[slot] == 0).all() and (net.cost[slot] == 0).all()
    # filling in the oracle's prices gives the golden checksum
    net.cap[slot] = g.cap[slot]
    net.cost[slot] = g.cost[slot]
    gold = [e for e in json.load(open(GOLD))["entries"] if e["config"] == cfg and e["seed"] == seed][0]
    assert "%016x" % net.checksum() == gold["csr_fnv1a64"] == "%016x" % capi.csr_checksum(g)


def test_builder_rejects_bad_arguments():
    lib = pa.load_library()
    lay = network.make_layout(10, 100)
    n = C.c_int32(0)
    m2 = C.c_int64(0)
    assert lib.psn_network_size(None, C.byref(n), C.byref(m2)) == _lib.PSN_EINVAL
    bad = _lib.Layout(0, 64, 2, 100, 8)
    assert lib.psn_network_size(C.byref(bad), C.byref(n), C.byref(m2)) == _lib.PSN_EINVAL
    assert lib.psn_network_size(C.byref(lay), C.byref(n), C.byref(m2)) == 0
    assert n.value == 1 + 10 + 64 + 2 + 100 and m2.value == 2 * (200 + 64 * 10 * 8 + 10 + 2)
    cl = pa.synth_cluster(0, 1, 10, 100)
    cl.task_class[3] = 99  # out of range class
    with pytest.raises(pa.PsnError) as ei:
        pa.build_network(cl)
    assert ei.value.code == _lib.PSN_EINVAL


def test_units_follow_the_reference_conversions():
    """cpu "2" -> 2000 millicores, memory 1024 B -> 1 Kb
    (/root/reference/pkg/k8sclient/podwatcher_test.go:166-176); cpu "10" -> 10000,
    1e10 B -> 9765625 Kb (/root/reference/pkg/k8sclient/nodewatcher_test.go:184-203)."""
    from poseidon_amd.firmament import cpu_millicores, memory_kb
    assert cpu_millicores("2") == 2000 and memory_kb("1024") == 1
    assert cpu_millicores("10") == 10000 and memory_kb("10000000000") == 9765625
    assert cpu_millicores("10m") == 10 and memory_kb("10Mi") == 10240


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PsnError) as ei:
        pa.Solver()
    assert ei.value.code == _lib.PSN_EDEVICE


def test_churn_generator_matches_the_oracle_spec():
    """The churn rounds the benchmark feeds to the service are drawn by the package's own generator;
    it must produce exactly what the oracle's restatement of the spec produces."""
    from poseidon_amd import network
    for cfg, seed, rnd in [(0, 1, 1), (0, 3, 2), (1, 2, 1)]:
        cl = pa.synth_cluster(cfg, seed)
        ch = network.churn_round(cl, cfg, seed, rnd)
        _, och = synth.churn(synth.make_cluster(cfg, seed), cfg, seed, rnd)
        for a, b in [(ch.tasks_replaced, och.pods_replaced), (ch.new_class, och.new_shape), (ch.new_job, och.new_job),
                     (ch.machines_restat, och.machines_restat), (ch.new_used_cpu, och.new_used_cpu),
                     (ch.new_used_mem, och.new_used_mem), (ch.machines_out, och.machines_out)]:
            assert np.array_equal(a, b)


def test_committed_profile_belongs_to_this_build():
    """bench.py quotes HBM traffic from the latest committed counter pass and refuses one taken from
    another build (its kernel list must cover psn_kernel_list()): checked here as well, so that a
    change of the launch sequence without a new profile fails before any benchmark run does."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("psn_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ.pop("PSN_BENCH_ALLOW_STALE_PROFILE", None)
    assert bench.check_profile_matches_library() is None
    traffic, src = bench.pmc_traffic("k_push")
    assert traffic > 0 and src["pmc_dispatches"] > 0 and src["trace_launches"] > 0 and src["trace_avg_launch_us"] > 0
    assert set(pa.load_library().psn_kernel_list().decode().split(",")) >= {"k_push", "k_gu_expand", "k_relabel_compact", "k_flip"}
