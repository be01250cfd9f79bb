mkdir -p gpurun_out/head
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "object_token" 2>&1 | tail -5 > gpurun_out/head/pytest_objtoken.txt
cat gpurun_out/head/pytest_objtoken.txt
AB_BENCH_ARGS='--mode objects --no-cpu-baseline --steps 8' python tools/ab_env.py 2 p128 p137:OAKE_PASS_CROPS=137 p136:OAKE_PASS_CROPS=136 p68:OAKE_PASS_CROPS=68 p171:OAKE_PASS_CROPS=171 p120:OAKE_PASS_CROPS=120 2>&1 | tee gpurun_out/head/ab_pass_crops.log
