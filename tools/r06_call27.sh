# round 6, GPU call 27: what a load phase of gemm_w8_kernel is made of — measurement builds (wrong results) without the
# pieces of LOAD0 (1), of LOAD1 (2), of both (3), without LOAD0's fragment reads (4); cycle stamps per phase
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8_trace; mkdir -p $O
for L in oadp_amd/liboake_hip_lab.so oadp_amd/liboake_w8abl1.so oadp_amd/liboake_w8abl2.so oadp_amd/liboake_w8abl3.so oadp_amd/liboake_w8abl4.so; do
  echo "== $L"
  OAKE_LAB_LIB=$L timeout 120 python tools/gemm_trace.py 25600 768 3072 bias 13 2>&1 | grep -v amdgpu.ids | head -3
done 2>&1 | tee $O/trace_ablate.txt
