#!/bin/bash
# round 5, session 2: grouptile variants (unroll 4 / 8, v reads merged / separate)
mkdir -p gpurun_out/r05s2; o=gpurun_out/r05s2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgroup_on_grid or fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -3 $o/pytest.txt
for v in "" _w4u4 _w4u8; do
  echo "== variant [$v]" >> $o/depth1.txt
  ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip$v.so timeout -k 10 300 python scratch/depth1.py "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=1" \
   "g1:EvI:64:rowgroup_form=1" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" >> $o/depth1.txt 2>&1
done
cat $o/depth1.txt
