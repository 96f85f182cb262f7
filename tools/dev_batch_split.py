"""Development aid: where a batch step of 256 config-1 networks spends its time (pricing, solve, extraction)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import poseidon_amd as pa
from poseidon_amd import batch as pbatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bcl, bnet, bsol = [], [], []
for b in range(B):
    c1 = pa.synth_cluster(1, 1 + b)
    n1 = pa.build_network(c1)
    s1 = pa.Solver(device=0)
    s1.upload(n1)
    bcl.append(c1); bnet.append(n1); bsol.append(s1)
bplace = torch.empty((B, 10_000), dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s1, c1, n1 in zip(bsol, bcl, bnet):
        s1.eval_costs(c1, n1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rs = pbatch.solve_batch(bsol)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for i, s1 in enumerate(bsol):
        s1.placements_into(bplace[i].data_ptr())
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pricing {1e3 * (t1 - t0):.1f} ms, solve {1e3 * (t2 - t1):.1f} ms, extraction {1e3 * (t3 - t2):.1f} ms", flush=True)
