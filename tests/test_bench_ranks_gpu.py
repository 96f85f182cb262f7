"""Rehearsal of the multi-rank benchmark on one GPU: `bench.py --gpus 4` launched the way the
driver launches it (torch.distributed.run, one fresh process per rank), with the collectives over
gloo and the four ranks sharing the device (PSN_BENCH_BACKEND=gloo).  The what-if batch of 256
networks is sharded network b -> rank b mod 4, each rank solves its 64 with the product solver and
the placements and costs are gathered into graph order: the printed line must carry the same
number of graphs, the same placed total and the same checksum of the 256 optimal costs as the
1-rank run of the same command, and that checksum is the one of the committed golden list."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, port):
    env = dict(os.environ, PSN_BENCH_BACKEND="gloo", PSN_BENCH_ALLOW_STALE_PROFILE="1", MASTER_ADDR="127.0.0.1")
    args = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--config", "0", "--no-cpu-baseline", "--no-c3",
            "--batch-graphs", "256", "--batch-check", "1"]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_four_ranks_over_gloo_print_the_same_batch_as_one_rank():
    one = _run(1, 0)
    four = _run(4, 29631)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "psn_c4_costs.json")))
    want = sum(gold["optimal_cost"]) % (1 << 61)
    for line, world in ((one, 1), (four, 4)):
        b = line["what_if_batch"]
        assert line["n_gpus"] == world and b["n_gpus"] == world
        assert b["graphs"] == 256 and b["scaling"] == "strong"
        assert b["cost_checksum"] == want
        assert b["placed_total"] == one["what_if_batch"]["placed_total"] > 0
        assert [p["networks"] for p in b["per_rank"]] == [256 // world] * world
        assert all(p["solve_ms"] > 0 for p in b["per_rank"])
    assert four["config"]["graphs_per_step"] == 4 and len(four["optimal_cost"]) == 4
    assert "rehearsal" in four["data"]
