"""Per-kernel totals of rocprofv3 --pmc passes (one directory per pass, --output-format csv).

  python tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/<name>.csv

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes
for wide coalesced reads, so `hbm_read_bytes_x2` doubles it; gather-dominated kernels sit between the
raw and the doubled figure.
"""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name][r["Counter_Name"]] += 1
counters = sorted({c for v in tot.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + [c + "_total" for c in counters] + [c + "_per_dispatch" for c in counters])
for name in sorted(tot, key=lambda k: -sum(tot[k].values())):
    n = max(calls[name].values())
    w.writerow([name, n] + [round(tot[name].get(c, 0.0), 1) for c in counters]
               + [round(tot[name].get(c, 0.0) / max(1, calls[name].get(c, 0)), 3) for c in counters])
