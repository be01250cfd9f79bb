# round 6, GPU call 24: gemm_w8_kernel with its LDS-DMA pieces issued BEHIND the landed fragment reads of a load phase
# (-DOAKE_W8_DMA_LATE=1) against the build (pieces right behind the ds_read issue)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8_dma_late; mkdir -p $O
for r in 1 2; do for L in oadp_amd/liboake_hip_lab.so oadp_amd/liboake_w8late.so; do
  echo "== $L run $r"; OAKE_LAB_LIB=$L timeout 300 python tools/gemm_ablate.py 13 3 12800 2>&1 | grep -v amdgpu.ids | grep "c_fc\|c_proj"
done; done 2>&1 | tee $O/ablate.txt
OAKE_LAB_LIB=oadp_amd/liboake_w8late.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "320_row" 2>&1 | tail -2 | tee $O/pytest.txt
