# Development aid: headline step with two builds of the library, alternating.
#   bash tools/ab_bench.sh <libA> <libB> [rounds] [steps]
set -e
A=$1; B=$2; R=${3:-3}; S=${4:-8}
for r in $(seq 1 $R); do
  for L in $A $B; do
    PSN_LIB=$PWD/$L timeout -k 10 300 python bench.py --no-c3 --no-cpu-baseline --batch-graphs 0 --steps $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['counters']; print('$L', d['ms_per_step'], d['solve_ms_min'], d['solve_ms_max'], c['rounds'], c['gu_iterations'], c['global_updates'], d['optimal_cost'])"
  done
done
