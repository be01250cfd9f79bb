# round 6, GPU call 23: the order of a wave's MFMAs inside a K-step — serpentine (the build) against row by row
# (liboake_rows.so = -DOAKE_MFMA_ORDER=0), bit-identical results; bench in every mode, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/mfma_order; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or qkv" 2>&1 | tail -2 | tee $O/pytest.txt
for r in 1 2 3; do for L in oadp_amd/liboake_rows.so oadp_amd/liboake_hip.so; do
  OAKE_LIB=$L OAKE_BENCH_FULL_LINE=1 timeout 300 python bench.py --mode globals --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('globals', '$L', 'run $r', d['value'], 'one lane', d['one_lane_images_per_sec'], ' '.join(f\"{n} {k[n]['ms_per_step']:.3f}\" for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_out_proj', 'qkv_attn')))"
done; done 2>&1 | tee $O/ab_globals.txt
for m in blocks objects; do for L in oadp_amd/liboake_rows.so oadp_amd/liboake_hip.so; do
  OAKE_LIB=$L OAKE_BENCH_FULL_LINE=1 timeout 600 python bench.py --mode $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$m', '$L', d['value'], 'one lane', d['one_lane_images_per_sec'], ' '.join(f\"{n} {k[n]['ms_per_step']:.3f}\" for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_out_proj', 'qkv_attn')))"
done; done 2>&1 | tee $O/ab_modes.txt
