mkdir -p gpurun_out/head
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -4 > gpurun_out/head/pytest_attention2.txt
cat gpurun_out/head/pytest_attention2.txt
python -m pytest tests/test_encoder_gpu.py tests/test_hook_branch_gpu.py tests/test_hooks_golden_gpu.py tests/test_oake_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/head/pytest_encoder2.txt
cat gpurun_out/head/pytest_encoder2.txt
AB_BENCH_ARGS='--mode objects --no-cpu-baseline --steps 8' python tools/ab_env.py 3 base:OAKE_LIB=oadp_amd/liboake_base.so new 2>&1 | tee gpurun_out/head/ab_kv_split.log
python bench.py --mode objects --no-cpu-baseline --steps 8 > gpurun_out/head/bench_objects2.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/head/bench_objects2.json').read().strip().splitlines()[-1])
print(d['value'], d['crops_per_sec']); print({k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['share']>0.004})"
