# round 6, GPU call 2: deferred GELU (variant 12 / -DOAKE_DEFER_GELU=1) parity + stamps + A/B; tile walk A/B with the cheap decode
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_16bit_epilogues or layernorm_folded or qkv or walk or refuses" 2>&1 | tail -5
  OAKE_LIB=oadp_amd/liboake_dg.so timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "encode_image and not fused_qkv" 2>&1 | tail -5 ) > $O/call2_pytest.txt 2>&1
cat $O/call2_pytest.txt
for v in 10 12; do timeout 120 python tools/gemm_trace.py 12800 3072 768 gelu $v 2>&1 | grep -v "amdgpu.ids"; done | tee $O/gemm_trace_defer_gelu.txt
timeout 300 python tools/gemm_ablate.py 4,12 5 2>&1 | grep -v "amdgpu.ids" | tee $O/gemm_ablate_defer_gelu.txt
DG=OAKE_LIB=oadp_amd/liboake_dg.so
python tools/ab_env.py 4 w0:OAKE_QKV_WALK=0 w4:OAKE_QKV_WALK=4 dg_w0:$DG,OAKE_QKV_WALK=0 dg_w4:$DG,OAKE_QKV_WALK=4 > $O/ab2_globals.log 2>&1; tail -5 $O/ab2_globals.log
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 3 w0:OAKE_QKV_WALK=0 w4:OAKE_QKV_WALK=4 dg_w0:$DG,OAKE_QKV_WALK=0 > $O/ab2_objects.log 2>&1; tail -4 $O/ab2_objects.log
AB_BENCH_ARGS="--mode blocks --no-cpu-baseline --steps 20 --warmup 4" python tools/ab_env.py 3 w0:OAKE_QKV_WALK=0 w4:OAKE_QKV_WALK=4 dg_w0:$DG,OAKE_QKV_WALK=0 > $O/ab2_blocks.log 2>&1; tail -4 $O/ab2_blocks.log
