#!/bin/bash
mkdir -p gpurun_out/r05s12; o=gpurun_out/r05s12; : > $o/depth1.txt; : > $o/a1h.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgroup_on_grid or fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -3 $o/pytest.txt
for v in _b4u16 _b4u16w8 _s128n4; do
echo "== variant [$v]" >> $o/depth1.txt
ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip$v.so timeout -k 10 300 python scratch/depth1.py "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=1,grouptile_fields=16" \
   "g1:EvI:64:rowgroup_form=1,grouptile_fields=16" "g1:EvI:64:rowgroup_form=1,grouptile_fields=32" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" >> $o/depth1.txt 2>&1
done
grep -v amdgpu.ids $o/depth1.txt
for v in _b4u16w8 _s128n4; do
for t in "rowgroup_form=1" "rowgroup_form=1,grouptile_fields=16"; do
  ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip$v.so TUNE=$t timeout -k 10 400 python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$v $t] /" >> $o/a1h.txt
done; done
cat $o/a1h.txt
