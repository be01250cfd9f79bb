cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/q4; mkdir -p $O
for cfg in q4_3_4 q4abl2 q4abl16; do
  echo "=== $cfg" | tee -a $O/summary_abl.txt
  for shape in "12800 768 3072 resid" "12800 768 768 resid"; do
    OAKE_LAB_LIB=oadp_amd/liboake_$cfg.so timeout 120 python tools/gemm_trace.py $shape 7 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/summary_abl.txt
  done
done
