"""Development aid (diagnostic build with -DPSN_STAMPS, loaded through PSN_LIB): where the waves of
the last heavy global-update sweep spent their cycles."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cl = pa.synth_cluster(2, 1)
net = pa.build_network(cl)
with pa.Solver() as s:
    s.upload(net); s.eval_costs(cl, net)
    r = s.solve()
    buf = np.zeros((4096, 16), np.uint64)
    s._lib.psn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert s._lib.psn_debug_stamps(s._ctx, buf.ctypes.data) == 0
print("solve_ms", r.solve_ms)
b = buf[buf[:, 5] > 0].astype(np.int64)
print("waves recorded", len(b), "nseg", b[0, 13], "k", b[0, 14])
rt0 = b[:, 12].min()
print("  100 MHz clock: wave begins (us since first)", np.percentile((b[:, 12] - rt0) / 100.0, [10, 50, 90, 100]),
      " wave ends", np.percentile((b[:, 15] - rt0) / 100.0, [10, 50, 90, 100]))
life_rt = (b[:, 15] - b[:, 12]) / 100.0
life_cy = (b[:, 5] - b[:, 0])
print("  wave life us: mean", life_rt.mean(), " cycles per us (s_memtime / 100 MHz clock):", (life_cy / np.maximum(life_rt, 0.01)).mean())
names = ["begin", "wave items done", "lane items done", "after barrier", "hubs flushed", "end"]
print("  cycles since the wave's own begin (the counters of different XCDs are not aligned):")
for i, nm in enumerate(names):
    x = b[:, i] - b[:, 0]
    print(f"  {nm:18s} mean {x.mean():10.0f}  p10 {np.percentile(x, 10):10.0f}  p50 {np.percentile(x, 50):10.0f}  p90 {np.percentile(x, 90):10.0f}  max {x.max():10.0f}")
# waves of one workgroup share a CU: spread of begin stamps inside the kernel per XCD group
blk = b.reshape(-1, 4, 16)
print("  begin spread inside a workgroup (max-min)", np.percentile(blk[:, :, 0].max(1) - blk[:, :, 0].min(1), [50, 90, 99]))
grp = (b[:, 0] // (1 << 30))
for gval in np.unique(grp)[:8]:
    sel = b[grp == gval]
    print(f"  clock group {gval}: waves {len(sel)}  first begin .. last end = {sel[:, 5].max() - sel[:, 0].min()} cycles; begin spread {sel[:, 0].max() - sel[:, 0].min()}")
print("  items/wave", b[:, 6].mean(), "steps/wave", b[:, 7].mean())
for i, nm in ((8, "item+dist+price"), (9, "arc rows"), (10, "gathers"), (11, "commit")):
    per = b[:, i] / np.maximum(1, b[:, 7] if i != 8 else b[:, 6])
    print(f"  {nm:16s} cycles per {'item' if i == 8 else 'step'}: mean {per.mean():9.0f} p50 {np.percentile(per, 50):9.0f} p90 {np.percentile(per, 90):9.0f};  per wave total mean {b[:, i].mean():9.0f}")
