cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_grp
PSN_BATCH_GROUP=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_grp -- python3 -u $R/tools/dev_group.py 8 1 > $R/gpurun_out/prof_grp.log 2>&1 || { tail -5 $R/gpurun_out/prof_grp.log; exit 1; }
tail -4 $R/gpurun_out/prof_grp.log
f=$(find $R/gpurun_out/prof_grp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("psn::", "")
    if n.startswith("k_"):
        print("   %-28s calls %6s avg %8.2f us total %8.2f ms max %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["MaxNs"]) / 1e3))
PY
