"""Summarise a rocprofv3 --kernel-trace --stats CSV directory (per-kernel totals + duration percentiles)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3:
        print(r['Name'][:45].ljust(46), r['Calls'].rjust(6), 'avg_us', str(round(float(r['AverageNs'])/1e3,2)).rjust(8), 'tot_ms', str(round(int(r['TotalDurationNs'])/1e6,2)).rjust(7), r['Percentage'].rjust(6), 'min', r['MinNs'], 'max', r['MaxNs'])
rows = list(csv.DictReader(open(glob.glob(d + '/*/*_kernel_trace.csv')[0])))
for name in ('k_push', 'k_gu_expand', 'k_saturate', 'k_relabel_compact'):
    x = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if name in r['Kernel_Name'])
    n = len(x)
    if n: print(name, n, 'p10', x[n//10], 'p50', x[n//2], 'p90', x[9*n//10], 'p99', x[99*n//100], 'max', x[-1], 'top10%_ms', round(sum(x[9*n//10:])/1e6, 2))
