#!/bin/bash
# the driver's short command (--steps 20 --warmup 5) against longer warm-ups and the LPT row order
for extra in "" "--tune rowblock_lpt=1"; do
for wk in "5 20" "160 20" "5 32" "160 32" "160 1920"; do
  set -- $wk
  echo "== warmup $1 steps $2 $extra"
  python bench.py --warmup $1 --steps $2 --no-cpu-baseline $extra | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('  ms_per_step %.3f us  kernel_us %.3f  frac %.3f  region_us/step %.3f  launches %d' % (d['ms_per_step']*1e3, r['kernel_us'], r['frac'], r['region_us_per_step'], r['launches']))"
done; done
