# round 6, GPU call 4: the one-compute-wave-per-SIMD loop (gemm_q4_kernel, variant 7) with its prefetch knobs, against production
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/q4; mkdir -p $O
echo "=== production (variant 10: four phases, stamped)" | tee $O/summary.txt
for shape in "12800 768 3072 resid" "12800 768 768 resid"; do timeout 120 python tools/gemm_trace.py $shape 10 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/summary.txt; done
for cfg in 2_6 4_6 4_4 4_2 3_4; do
  echo "=== q4 AD_BF1 = $cfg" | tee -a $O/summary.txt
  for shape in "12800 768 3072 resid" "12800 768 768 resid"; do
    OAKE_LAB_LIB=oadp_amd/liboake_q4_$cfg.so timeout 120 python tools/gemm_trace.py $shape 7 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/summary.txt
  done
  OAKE_LAB_LIB=oadp_amd/liboake_q4_$cfg.so timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_gemm_tile_configs and 7-" 2>&1 | tail -1 | tee -a $O/summary.txt
done
