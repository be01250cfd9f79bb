# round 6, GPU call 19: the final binary (gemm_w8_kernel for c_fc / K,V everywhere, out_proj / c_proj in blocks and objects
# passes): every fuzzer on a fresh seed, and rank 0's FULL objects share of the 118 k sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
bash tools/fuzz_all.sh 263 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_all_final_seed263.log | tail -30
timeout 900 python tools/sweep_shard.py --total 118000 --world 8 --rank 0 --modes objects --sample 32 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/sweep_shard_rank0of8_118k_objects_final_w8.log
