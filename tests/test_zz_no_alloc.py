"""The entry point that allocates nothing, psn_get_placements_device_noalloc, against the entry
point it is a variant of (psn_get_placements_device): same placements, bit for bit, on two calls
in a row on one GPU, the second on a smaller network; no allocation is made by the calls; a
layout beyond the stated limit is refused before anything is launched."""
import ctypes as C

import numpy as np
import pytest
import torch

import poseidon_amd as pa
from poseidon_amd import _lib

pytestmark = pytest.mark.gpu


def _solved(cfg, seed, nm=None, nt=None):
    cl = pa.synth_cluster(cfg, seed, nm, nt)
    net = pa.build_network(cl)
    s = pa.Solver(device=0)
    s.upload(net)
    s.eval_costs(cl, net)
    s.solve()
    return s, cl


def test_no_allocation_entry_point_matches_the_allocating_one():
    lib = pa.load_library()
    big, cl_big = _solved(1, 3)          # 1 000 machines / 10 000 tasks: 640 000 scratch entries
    small, cl_small = _solved(0, 2)      # 100 machines / 1 000 tasks, second call
    try:
        for s, cl in ((big, cl_big), (small, cl_small)):
            P = cl.layout.n_tasks
            want = torch.full((P,), -7, dtype=torch.int32, device="cuda")
            got = torch.full((P + 8,), -7, dtype=torch.int32, device="cuda")  # tail guards an overrun
            s.placements_into(want.data_ptr())  # the existing entry point (allocates its scratch on first use)
            torch.cuda.synchronize()
            before = lib.psn_debug_alloc_count()
            rc = lib.psn_get_placements_device_noalloc(s._ctx, C.byref(cl.layout), C.c_void_p(got.data_ptr()))
            assert rc == _lib.PSN_OK, lib.psn_last_error(s._ctx)
            assert lib.psn_debug_alloc_count() == before
            torch.cuda.synchronize()
            assert torch.equal(got[:P], want)
            assert bool((got[P:] == -7).all())
            assert int((want >= 0).sum()) > 0
        # the first network again after the smaller one used the shared scratch
        P = cl_big.layout.n_tasks
        want = torch.empty(P, dtype=torch.int32, device="cuda")
        got = torch.empty(P, dtype=torch.int32, device="cuda")
        big.placements_into(want.data_ptr())
        assert lib.psn_get_placements_device_noalloc(big._ctx, C.byref(cl_big.layout), C.c_void_p(got.data_ptr())) == _lib.PSN_OK
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    finally:
        big.close()
        small.close()


def test_no_allocation_entry_point_refuses_what_exceeds_its_scratch():
    lib = pa.load_library()
    cl = pa.synth_cluster(0, 5, 60, 20_000)  # 64 classes x 20 000 tasks > 1 048 576
    net = pa.build_network(cl)
    with pa.Solver(device=0) as s:
        s.upload(net)
        s.eval_costs(cl, net)
        s.solve()
        dst = torch.full((20_000,), -7, dtype=torch.int32, device="cuda")
        rc = lib.psn_get_placements_device_noalloc(s._ctx, C.byref(cl.layout), C.c_void_p(dst.data_ptr()))
        assert rc == _lib.PSN_EINVAL and b"scratch" in lib.psn_last_error(s._ctx)
        torch.cuda.synchronize()
        assert bool((dst == -7).all())  # nothing was launched
        ref = torch.empty(20_000, dtype=torch.int32, device="cuda")
        s.placements_into(ref.data_ptr())  # the allocating entry point still serves it
        assert int((ref >= 0).sum()) > 0
