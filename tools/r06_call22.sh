# round 6, GPU call 22: the objects sub-record of the default bench line showed gemm_c_proj at 175 us per launch (101 in a
# stand-alone objects run): reproduce, and compare with the automatic choice without the 320-row kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/subrecord; mkdir -p $O
for v in -1 -2 -1; do
  OAKE_GEMM_VARIANT=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/default_v$v.json
  python - <<PY
import json
d = json.load(open('$O/default_v$v.json'))
for m, o in d['modes'].items():
    print('default run, variant $v,', m, o['images_per_sec'], 'one lane', o['one_lane_images_per_sec'], o['kernel'], o['avg_launch_us'], 'sum', o['kernels_sum_ms'], 'step', o['one_lane_step_ms'], o['top'])
PY
done 2>&1 | tee $O/summary.txt
OAKE_BENCH_FULL_LINE=1 timeout 600 python bench.py --mode objects --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('stand-alone objects', d['value'], d['one_lane_images_per_sec'], ' '.join(f\"{n} {k[n]['ms_per_step']:.2f}\" for n in list(k)[:5]))" | tee -a $O/summary.txt
