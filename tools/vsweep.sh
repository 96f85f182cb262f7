# Development aid: cold solves and keep-running churn rounds under several PSN_TAIL_VERTICES.
set -e
for v in ${VS:-0 16 32 64}; do
  echo "== tail_vertices $v (items at sources/128)"
  for c in 0 1 2; do PSN_TAIL_VERTICES=$v timeout -k 10 300 python tools/dev_cfg.py $c 6; done
  for c in 1 2; do PSN_TAIL_VERTICES=$v PSN_DEBUG_SOLVE=1 timeout -k 10 200 python tools/dev_service.py $c 1 3 nocheck 2>&1 | grep "psn_solve" | tail -3 | cut -c1-150; done
done
