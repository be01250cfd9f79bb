# round 6, GPU call 26: gemm_w8_kernel with uniform (SGPR) piece addresses — parity, cycle anatomy, launch times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8_trace; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2 | tee $O/pytest_sgpr_addr.txt
for shape in "12800 3072 768 gelu" "25600 768 3072 resid" "25600 768 768 resid"; do
  timeout 120 python tools/gemm_trace.py $shape 13 2>&1 | grep -v amdgpu.ids | head -4
done 2>&1 | tee $O/trace_sgpr_addr.txt
timeout 300 python tools/gemm_ablate.py 4,13 3 12800 2>&1 | grep -v amdgpu.ids | grep "c_fc" | tee $O/ablate_sgpr_addr.txt
timeout 300 python tools/gemm_ablate.py 4,13 3 25600 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate_sgpr_addr.txt
