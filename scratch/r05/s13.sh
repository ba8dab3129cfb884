#!/bin/bash
mkdir -p gpurun_out/r05s13; o=gpurun_out/r05s13; : > $o/depth1.txt; : > $o/a1h.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -5 $o/pytest.txt
timeout -k 10 300 python scratch/depth1.py "g5:EvI:16:" "g5:EvI:64:" "g1:EvI:64:" "g1:EvI:16:kernel=rowgroup" "g1:EvI:64:rowgroup_form=1,grouptile_seg=128" >> $o/depth1.txt 2>&1
grep -v amdgpu.ids $o/depth1.txt
timeout -k 10 400 python scratch/kernel_choice.py a1h EvI 16,128 auto,rowgroup 2>&1 | grep nf= >> $o/a1h.txt
cat $o/a1h.txt
