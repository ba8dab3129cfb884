#!/bin/bash
mkdir -p gpurun_out/r05s26; o=gpurun_out/r05s26; : > $o/t.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgroup_on_grid or fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -3 $o/pytest.txt
timeout -k 10 300 python scratch/depth1.py "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=1" "g1:EvI:64:rowgroup_form=1" "g1:EvI:64:rowgroup_form=1,grouptile_seg=128" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" 2>&1 | grep -v amdgpu >> $o/t.txt
for t in "rowgroup_form=1" "rowgroup_form=1,grouptile_seg=256"; do TUNE=$t timeout -k 10 300 python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o/t.txt; done
cat $o/t.txt
