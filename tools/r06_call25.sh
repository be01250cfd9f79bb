# round 6, GPU call 25: cycle anatomy of gemm_w8_kernel (s_memtime per phase, lab build), and the late-piece form in cycles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8_trace; mkdir -p $O
for L in oadp_amd/liboake_hip_lab.so oadp_amd/liboake_w8late.so; do
  echo "== $L"
  for shape in "12800 3072 768 gelu" "12800 3072 768 bias" "25600 768 3072 resid" "25600 768 768 resid"; do
    OAKE_LAB_LIB=$L timeout 120 python tools/gemm_trace.py $shape 13 2>&1 | grep -v amdgpu.ids
  done
done 2>&1 | tee $O/trace.txt
echo "== gemm_pp_kernel (variant 10: four short phases, the stamped form), c_fc's shape" | tee -a $O/trace.txt
timeout 120 python tools/gemm_trace.py 12800 3072 768 gelu 10 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.txt
