set -e
run() { timeout -k 10 120 python -u tools/dev_stats.py ${CFG:-2} ${N:-40} "$1"; }
run c2_new
CFG=1 run c1_new
CFG=0 run c0_new
CFG=0 PSN_TAIL_UNITS=12 run c0_t12
CFG=0 ALPHA=4 run c0_a4
PSN_TAIL_POWER=0.7 PSN_TAIL_FACTOR=0.13 run c2_p70
CFG=1 PSN_TAIL_POWER=0.7 PSN_TAIL_FACTOR=0.13 run c1_p70
SEED=2 run c2_seed2
SEED=3 run c2_seed3
SEED=2 PSN_LIB=$PWD/ab_libs/c.so run c2_seed2_old
SEED=3 PSN_LIB=$PWD/ab_libs/c.so run c2_seed3_old
echo shapes new; timeout -k 10 300 python -u tools/dev_shapes.py child
echo shapes old; PSN_LIB=$PWD/ab_libs/c.so timeout -k 10 300 python -u tools/dev_shapes.py child
