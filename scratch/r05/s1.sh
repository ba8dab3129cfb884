#!/bin/bash
# round 5, session 1: the tiled row-group kernel (grouptile) -- parity tests, then one-launch timings against the LDS-atomic form
mkdir -p gpurun_out/r05s1; o=gpurun_out/r05s1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgroup_on_grid or fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -5 $o/pytest.txt
timeout -k 10 500 python scratch/depth1.py "g5:EvI:16:rowgroup_form=0" "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=0" "g5:EvI:64:rowgroup_form=1" \
   "g1:EvI:64:rowgroup_form=0" "g1:EvI:64:rowgroup_form=1" "g1:EvI:16:rowgroup_form=0,kernel=rowgroup" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" > $o/depth1.txt 2>&1
cat $o/depth1.txt
