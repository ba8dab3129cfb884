#!/bin/bash
mkdir -p gpurun_out/r05s27; o=gpurun_out/r05s27; : > $o/t.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -6 $o/pytest.txt
timeout -k 10 300 python scratch/depth1.py "g5:EvI:64:rowgroup_form=0" "g5:EvI:64:rowgroup_form=1,grouptile_split=0" "g5:EvI:64:" "g5:EvI:64:grouptile_seg=128" "g5:EvI:32:rowgroup_form=0" "g5:EvI:32:" "g5:EvI:16:rowgroup_form=1,grouptile_split=1" "g5:EvI:16:rowgroup_form=1,grouptile_split=1,grouptile_seg=128" "g5:EvI:128:rowgroup_form=0" "g5:EvI:128:" "g20:EvI:64:rowgroup_form=0" "g20:EvI:64:" 2>&1 | grep -v amdgpu >> $o/t.txt
cat $o/t.txt
