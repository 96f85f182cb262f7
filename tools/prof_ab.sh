#!/bin/bash
# usage: prof_ab.sh lib1 lib2 ...: per-kernel average durations under rocprofv3 for each build
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_$(basename $lib .so)
  rm -rf $out
  PSN_LIB=$GRAFT_REPO_ROOT/$lib REPEAT=5 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -u $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 > $out.log 2>&1 || exit 1
  echo "== $lib"; grep -o '"ms_per_step": [0-9.]*' $out.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_relabel_compact", "k_push", "k_gu_expand", "k_gu_init", "k_gu_apply")):
        print("   %-22s calls %6s avg %8.2f us total %8.2f ms" % (n.split("(")[0].replace("psn::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
