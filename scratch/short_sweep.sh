#!/bin/bash
# the driver's short command against launch-shape overrides: kernel_us / frac of each
for t in "" "--tune rowblock_waves=4" "--tune rowblock_many_qi=1" "--tune rowblock_many_qi=4" "--tune rowblock_many_qi=5" "--tune rowblock_many_qi=10" "--tune rowblock_lpt=1" "--tune rowblock_many_fpw=2" "--tune rowblock_waves=4 --tune rowblock_many_qi=4"; do
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline $t 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('%-60s kernel_us %.3f frac %.3f region %.2f' % ('$t', r['roofline']['kernel_us'], r['roofline']['frac'], r['roofline']['region_us_per_step']))"
  done
done
