# round 6, GPU call 12: the whole GPU suite on the final binary, every fuzzer, and rank 0's FULL objects share of the 118 k sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu_full.txt
bash tools/fuzz_all.sh 251 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_all_final_seed251.log
timeout 900 python tools/sweep_shard.py --total 118000 --world 8 --rank 0 --modes objects --sample 32 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/sweep_shard_rank0of8_118k_objects_final.log
