set -e
export TMPDIR=/tmp PSN_LIB=$PWD/bin_tmp/libpsn_ub.so
cfg=${1:-2}; shift_=${2:-8}; variants=${3:-0,1,5}; ctrs=${4:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"}
rm -rf gpurun_out/ubpmc
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/ubpmc -- python3 tools/ubench/gu_bench.py run $cfg $shift_ $variants > gpurun_out/ubpmc_run.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/ubpmc/*/*counter_collection.csv'))[-1]
rows=list(csv.DictReader(open(f)))
# group rows by dispatch id
d=collections.OrderedDict()
for r in rows:
    k=int(r['Dispatch_Id'])
    d.setdefault(k,{'name':r['Kernel_Name']})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(d)
res=collections.defaultdict(list)
i=0
while i<len(ids):
    if 'k_ub_seed' not in d[ids[i]]['name']: i+=1; continue
    j=i+2; nb=0
    while 'k_ub_expand' not in d[ids[j]]['name']: j+=1; nb+=1
    nm=d[ids[j]]['name']; v=nm[nm.index('<')+1:nm.index('>')]
    res[(v,nb)].append(d[ids[j]])
    i=j+1
names=None
for k in sorted(res):
    x=res[k][0]
    if names is None:
        names=[c for c in x if c!='name']; print('variant iter '+' '.join(n.replace('SQ_','') for n in names))
    print(k[0],k[1],' '.join(f'{x[n]/1e6:9.2f}M' for n in names))
PY
