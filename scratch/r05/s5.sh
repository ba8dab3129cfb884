#!/bin/bash
mkdir -p gpurun_out/r05s5; o=gpurun_out/r05s5; : > $o/tl.txt; : > $o/depth1.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgroup_on_grid or fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -3 $o/pytest.txt
timeout -k 10 500 python scratch/depth1.py "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=1,grouptile_fields=16" "g5:EvI:64:rowgroup_form=1,grouptile_fields=32" \
   "g1:EvI:64:rowgroup_form=1,grouptile_fields=16" "g1:EvI:64:rowgroup_form=1,grouptile_fields=32" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" $EXTRA >> $o/depth1.txt 2>&1
cat $o/depth1.txt
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_tl.so
for a in "g5 EvI 16 grouptile_fields=16" "g1 EvI 64 grouptile_fields=16"; do
  timeout -k 10 200 python scratch/r05/timeline.py $a >> $o/tl.txt 2>&1
done
grep -v amdgpu.ids $o/tl.txt
