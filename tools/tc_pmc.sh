# Development aid: per-dispatch counters of a traced solve (expansion k after each k_gu_init).
set -e
export TMPDIR=/tmp PSN_TC_CONFIG=${1:-2}
ctrs=${2:-"TCC_ATOMIC_sum TCC_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS"}
rm -rf gpurun_out/tcpmc
timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/tcpmc -- python3 tools/trace_corr.py run gpurun_out/tcpmc_trace.npy > gpurun_out/tcpmc_run.log 2>&1
python3 - <<'PY'
import csv,glob,collections
import numpy as np
f=sorted(glob.glob('gpurun_out/tcpmc/*/*counter_collection.csv'))[-1]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k=int(r['Dispatch_Id']); d.setdefault(k,{'name':r['Kernel_Name']})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(d)
acc=collections.defaultdict(list)
i=0
while i<len(ids):
    if 'k_gu_init' in d[ids[i]]['name']:
        for k in range(12):
            x=d[ids[i+1+k]]
            if 'k_gu_expand' in x['name']: acc[k].append(x)
        i+=13
    else: i+=1
names=[c for c in acc[0][0] if c!='name']
print('iter n '+' '.join(names))
for k in range(8):
    print(k,len(acc[k]),' '.join(f"{np.median([x[n] for x in acc[k]])/1e3:10.1f}k" for n in names))
PY
