"""Correlate PSN_TRACE records (tools/trace_corr.py) with a rocprofv3 kernel trace."""
import csv, glob, sys, numpy as np
d = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr = np.load(d + '/trace.npy')
rows = list(csv.DictReader(open(glob.glob(d + '/prof/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if ('k_push' in r['Kernel_Name'] or 'k_gu_expand' in r['Kernel_Name'])]
out = []; ti = 0; pk = 0; ek = 0
for name, dur in ks:
    if 'k_push' in name: k = pk; pk = (pk + 1) % R; kind = 1
    else: k = ek; ek = (ek + 1) % 12; kind = 2
    if ti < len(tr) and tr[ti][0] == kind and tr[ti][1] == k: out.append((kind, k, int(tr[ti][2]), int(tr[ti][3]), dur)); ti += 1
    else: out.append((kind, k, 0, 0, dur))
print('matched', ti, 'of', len(tr))
out = np.array(out)
for kind, nm in ((2, 'expand'), (1, 'push')):
    o = out[out[:, 0] == kind]; idle = o[o[:, 2] + o[:, 3] == 0]; o = o[o[:, 2] + o[:, 3] > 0]
    print(nm, 'calls with work', len(o), 'total ms', round(o[:, 4].sum() / 1e3, 2), '| idle calls', len(idle), 'total ms', round(idle[:, 4].sum() / 1e3, 2))
    for lo, hi in ((0, 16), (16, 256), (256, 2048), (2048, 6000), (6000, 12000), (12000, 10**9)):
        sel = o[(o[:, 3] >= lo) & (o[:, 3] < hi)]
        if len(sel): print(f'   nseg in [{lo},{hi}): calls {len(sel)} avg_us {sel[:,4].mean():.1f} total_ms {sel[:,4].sum()/1e3:.1f} avg_nlow {sel[:,2].mean():.0f} avg_nseg {sel[:,3].mean():.0f}')
gi = [i for i, o in enumerate(out) if o[0] == 2 and o[1] == 0]
for g0 in (gi[10], gi[60], gi[120]):
    print('--- GU sample')
    for o in out[g0:g0 + 12 + 2 * R]:
        if o[2] + o[3] > 0: print('   ', 'push  ' if o[0] == 1 else 'expand', int(o[1]), 'nlow', int(o[2]), 'nseg', int(o[3]), 'us', round(o[4], 1))
