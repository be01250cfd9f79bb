export OAKE_BENCH_FULL_LINE=1
set -x
mkdir -p gpurun_out/mb
for r in 1 2; do
for mb in 512 256 128 384 192; do
 python bench.py --mode objects --max-batch $mb --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('objects mb',$mb, d['value'], d['crops_per_sec'], d.get('one_lane_images_per_sec'), {k:v['ms_per_step'] for k,v in d['kernels'].items() if v['share']>0.05})" >> gpurun_out/mb/sweep.log
done
for mb in 512 256 128; do
 python bench.py --mode blocks --max-batch $mb --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks mb',$mb, d['value'], d['crops_per_sec'], d.get('one_lane_images_per_sec'))" >> gpurun_out/mb/sweep.log
done
done
cat gpurun_out/mb/sweep.log
