# Development aid: cold solves under several PSN_TAIL_DIV (tail cut = sources / div).
set -e
for cfg in ${CONFIGS:-0 1 2}; do
for t in ${DIVS:-1000000 512 256 128 64 32}; do
  PSN_TAIL_DIV=$t timeout -k 10 300 python tools/dev_cfg.py $cfg ${STEPS:-8} ${SEED:-1}
done; done
