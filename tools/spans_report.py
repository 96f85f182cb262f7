"""Development aid (diagnostic build with -DPSN_STAMPS via PSN_LIB): per launch of the push /
relabel kernels, the time from the first wave's begin to the last wave's end, and the gaps
between consecutive launches of a replay."""
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
    EP, SL, W = 8, 64, 4096
    buf = np.zeros(EP * SL * W * 2 + EP * SL * 2, np.uint64)
    s._lib.psn_debug_spans.argtypes = [C.c_void_p, C.c_void_p]
    assert s._lib.psn_debug_spans(s._ctx, buf.ctypes.data) == 0
print("solve_ms", r.solve_ms, "rounds", r.rounds)
w = buf[: EP * SL * W * 2].astype(np.int64).reshape(EP, SL, W, 2)
work = buf[EP * SL * W * 2:].astype(np.int64).reshape(EP, SL, 2)
has = w[:, :, :, 1] > 0
beg = np.where(has, w[:, :, :, 0], 1 << 62).min(axis=2)
end = np.where(has, w[:, :, :, 1], 0).max(axis=2)
life = np.where(has, w[:, :, :, 1] - w[:, :, :, 0], 0)
b = np.stack([beg, end, work[:, :, 0], work[:, :, 1]], axis=2)
valid = has.any(axis=2)
push, rel, gaps_pr, gaps_rp = [], [], [], []
for ep in range(EP):
    for k in range(12):
        sp, sr = 16 + 2 * k, 17 + 2 * k
        if valid[ep, sp]:
            lw = life[ep, sp][has[ep, sp]]
            print(f"  epoch {ep} round {k}: push items {b[ep, sp, 3]:6d} span {(b[ep, sp, 1] - b[ep, sp, 0]) / 100.0:6.1f} us, wave life mean {lw.mean() / 100.0:5.1f} max {lw.max() / 100.0:5.1f}, "
                  f"begin spread {(np.where(has[ep, sp], w[ep, sp, :, 0], 0).max() - b[ep, sp, 0]) / 100.0:5.1f}"
                  + (f" | relabel span {(b[ep, sr, 1] - b[ep, sr, 0]) / 100.0:5.1f} gap before {(b[ep, sr, 0] - b[ep, sp, 1]) / 100.0:4.1f}" if valid[ep, sr] else ""))
            push.append(((b[ep, sp, 1] - b[ep, sp, 0]) / 100.0, b[ep, sp, 3]))
            if valid[ep, sr]:
                rel.append((b[ep, sr, 1] - b[ep, sr, 0]) / 100.0)
                gaps_pr.append((b[ep, sr, 0] - b[ep, sp, 1]) / 100.0)
                if k + 1 < 12 and valid[ep, sp + 2]:
                    gaps_rp.append((b[ep, sp + 2, 0] - b[ep, sr, 1]) / 100.0)
for ep, k in ((7, 5), (7, 9), (4, 3)):
    sp = 16 + 2 * k
    if valid[ep, sp]:
        lw = life[ep, sp] / 100.0
        hs = has[ep, sp]
        t0 = b[ep, sp, 0]
        print(f"  epoch {ep} round {k}: {b[ep, sp, 3]} wave items; wave life histogram (us):",
              " ".join(f"<{e}:{int(((lw < e) & (lw >= lo) & hs).sum())}" for lo, e in ((0, 1), (1, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 14), (14, 20), (20, 99))))
        longw = np.where(hs & (lw >= 8))[0]
        print("     waves living >= 8 us: count", len(longw), "indices", longw[:40], " their begins (us after first):", np.round((w[ep, sp, longw[:12], 0] - t0) / 100.0, 1))
        idle = hs & (np.arange(W) >= b[ep, sp, 3])
        if idle.any():
            print("     waves without an item: life mean %.2f us, begin (us after first) p50 %.1f max %.1f" % (lw[idle].mean(), np.percentile((w[ep, sp, idle, 0] - t0) / 100.0, 50), ((w[ep, sp, idle, 0] - t0) / 100.0).max()))
for ep, k in ((6, 5), (7, 9)):
    sr = 17 + 2 * k
    if valid[ep, sr]:
        lw = life[ep, sr] / 100.0
        hs = has[ep, sr]
        t0 = b[ep, sr, 0]
        bg = (w[ep, sr, :, 0] - t0) / 100.0
        print(f"  relabel epoch {ep} round {k}: waves {int(hs.sum())}, life mean {lw[hs].mean():.2f} p90 {np.percentile(lw[hs], 90):.2f} max {lw[hs].max():.2f} us; "
              f"begin p50 {np.percentile(bg[hs], 50):.2f} p90 {np.percentile(bg[hs], 90):.2f} max {bg[hs].max():.2f}; span {(b[ep, sr, 1] - t0) / 100.0:.2f}")
        order = np.argsort(-lw * hs)[:8]
        print("     longest-lived waves", order, np.round(lw[order], 2), "begin", np.round(bg[order], 2))
ctrl_cyc = np.array([b[ep, 17 + 2 * k, 3] for ep in range(EP) for k in range(12) if valid[ep, 17 + 2 * k]])
if len(ctrl_cyc):
    print("  relabel: kernel entry -> control words read, cycles: mean %.0f p50 %.0f p90 %.0f" % (ctrl_cyc.mean(), np.percentile(ctrl_cyc, 50), np.percentile(ctrl_cyc, 90)))
push = np.array(push)
print("push launches", len(push), " relabel launches", len(rel))
for lo, hi in ((0, 64), (64, 1024), (1024, 4096), (4096, 1 << 30)):
    sel = push[(push[:, 1] >= lo) & (push[:, 1] < hi)]
    if len(sel):
        print(f"  push with {lo}..{hi} wave items: {len(sel)} launches, first-begin..last-end mean {sel[:, 0].mean():.1f} us, p50 {np.percentile(sel[:, 0], 50):.1f}")
print("  relabel first-begin..last-end us: mean %.1f p50 %.1f p90 %.1f" % (np.mean(rel), np.percentile(rel, 50), np.percentile(rel, 90)))
print("  gap push end -> relabel begin us: mean %.1f p50 %.1f p90 %.1f" % (np.mean(gaps_pr), np.percentile(gaps_pr, 50), np.percentile(gaps_pr, 90)))
print("  gap relabel end -> next push begin us: mean %.1f p50 %.1f p90 %.1f" % (np.mean(gaps_rp), np.percentile(gaps_rp, 50), np.percentile(gaps_rp, 90)))
print("  sum of spans: push %.1f ms relabel %.1f ms gaps %.1f ms" % (push[:, 0].sum() / 1e3, np.sum(rel) / 1e3, (np.sum(gaps_pr) + np.sum(gaps_rp)) / 1e3))
