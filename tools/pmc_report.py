"""Join PSN_TRACE records with a rocprofv3 --pmc counter_collection.csv (per-dispatch counters)."""
import csv, glob, sys, collections, numpy as np
d = sys.argv[1]; R = 8
tr = np.load(d + '/trace.npy')
f = glob.glob(d + '/prof/*/*_counter_collection.csv')[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = int(r['Dispatch_Id'])
    e = disp.setdefault(k, {'name': r['Kernel_Name']})
    e[r['Counter_Name']] = float(r['Counter_Value'])
ds = [disp[k] for k in sorted(disp)]
names = sorted({c for e in ds for c in e if c != 'name'})
out = []; ti = 0; pk = 0; ek = 0
for e in ds:
    nm = e['name']
    if 'k_push' in nm: k = pk; pk = (pk + 1) % R; kind = 1
    elif 'k_gu_expand' in nm: k = ek; ek = (ek + 1) % 12; kind = 2
    else: continue
    if ti < len(tr) and tr[ti][0] == kind and tr[ti][1] == k: rec = (kind, k, int(tr[ti][2]), int(tr[ti][3])); ti += 1
    else: rec = (kind, k, 0, 0)
    out.append((rec, e))
print('matched', ti, 'of', len(tr)); print(names)
def show(title, sel):
    sel = list(sel)
    if not sel: return
    print(title, 'n =', len(sel))
    for c in names:
        vals = [e.get(c, 0.0) for _, e in sel]
        print(f'   {c:40s} mean {np.mean(vals):14.1f}')
show('expand EC-frontier (k=1, 4000<nseg<6000)', [(r, e) for r, e in out if r[0] == 2 and 4000 < r[3] < 6000 and r[2] == 0])
show('expand machine-frontier (nseg>9000, nlow<100)', [(r, e) for r, e in out if r[0] == 2 and r[3] > 9000 and r[2] < 100])
show('expand task-frontier (nlow>90000, nseg<100)', [(r, e) for r, e in out if r[0] == 2 and r[2] > 90000 and r[3] < 100])
show('push 500<nseg<1500', [(r, e) for r, e in out if r[0] == 1 and 500 < r[3] < 1500])
show('push nseg>3000', [(r, e) for r, e in out if r[0] == 1 and r[3] > 3000])
