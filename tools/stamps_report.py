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
# imbalance: by XCD (workgroups are dealt round-robin over the 8 XCDs), by items per wave, by position in the grid
if len(b) == 4096:
    widx = np.arange(4096)
    xcd = (widx // 4) % 8
    print("  wave life us by XCD (mean / max):", " ".join(f"{life_rt[xcd == x].mean():.1f}/{life_rt[xcd == x].max():.1f}" for x in range(8)))
    for n_it in np.unique(b[:, 6]):
        sel = b[:, 6] == n_it
        print(f"  waves with {n_it} items: {int(sel.sum())}, life us mean {life_rt[sel].mean():.1f} p90 {np.percentile(life_rt[sel], 90):.1f} max {life_rt[sel].max():.1f}")
    q = [life_rt[i * 512:(i + 1) * 512].mean() for i in range(8)]
    print("  wave life us by eighth of the grid (mean):", " ".join(f"{x:.1f}" for x in q))
    for nm, col in (("steps", 7), ("item+dist+price cycles", 8), ("arc rows cycles", 9), ("gathers cycles", 10), ("commit cycles", 11)):
        print(f"  {nm} per wave by eighth of the grid (mean):", " ".join(f"{b[i * 512:(i + 1) * 512, col].mean():.0f}" for i in range(8)))
    tail = b[:, 5] - b[:, 1]
    print("  cycles after the wave items (lane items, barrier, hubs, flush) by eighth:", " ".join(f"{tail[i * 512:(i + 1) * 512].mean():.0f}" for i in range(8)))
    per_item = life_rt / np.maximum(b[:, 6], 1)
    print("  life per item us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(per_item, [10, 50, 90, 100])))
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
