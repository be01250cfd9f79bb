mkdir -p gpurun_out/head
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/head/pytest_gpu_all2.txt
cat gpurun_out/head/pytest_gpu_all2.txt
python tools/sweep_shard.py --total 16000 > gpurun_out/head/sweep_shard_16k_defaults.log 2>&1
grep '"mode"' gpurun_out/head/sweep_shard_16k_defaults.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'], [x for x in d['log_tail'] if 'train:' in x])"
tail -1 gpurun_out/head/sweep_shard_16k_defaults.log | cut -c1-300
python bench.py > gpurun_out/head/bench_default2.json 2> gpurun_out/head/bench_default2.err
python -c "
import json
d=json.loads(open('gpurun_out/head/bench_default2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['mfma_roofline_frac_e2e'])
for m,v in d['modes'].items(): print(m, v['value'], v.get('crops_per_sec'), v['roofline']['frac'])"
