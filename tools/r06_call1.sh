# round 6, GPU call 1: parity of the changed paths, K-loop ablation, tile-walk A/B + FETCH pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -x -q -m gpu -k "qkv or refuses or walk or abi" 2>&1 | tail -5
  timeout 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "fused_qkv or pass_limit or pass_cap or objects" 2>&1 | tail -5 ) > $O/call1_pytest.txt 2>&1
cat $O/call1_pytest.txt
bash tools/kloop_ablate.sh $O/kloop > $O/kloop.log 2>&1
tail -60 $O/kloop/summary.txt
# tile walk A/B (three interleaved rounds each)
python tools/ab_env.py 3 walk0:OAKE_QKV_WALK=0 walk4:OAKE_QKV_WALK=4 walk2:OAKE_QKV_WALK=2 walk6:OAKE_QKV_WALK=6 > $O/ab_qkv_walk_globals.log 2>&1; tail -5 $O/ab_qkv_walk_globals.log
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 2 walk0:OAKE_QKV_WALK=0 walk4:OAKE_QKV_WALK=4 walk2:OAKE_QKV_WALK=2 walk6:OAKE_QKV_WALK=6 > $O/ab_qkv_walk_objects.log 2>&1; tail -5 $O/ab_qkv_walk_objects.log
AB_BENCH_ARGS="--mode blocks --no-cpu-baseline --steps 20 --warmup 4" python tools/ab_env.py 2 walk0:OAKE_QKV_WALK=0 walk4:OAKE_QKV_WALK=4 > $O/ab_qkv_walk_blocks.log 2>&1; tail -3 $O/ab_qkv_walk_blocks.log
# HBM-side reads of the fused kernel per walk (one lane, PMC pass on its own)
cd /tmp
for w in 0 4 2; do
  for m in globals objects; do
    OAKE_QKV_WALK=$w OAKE_BENCH_LANES=1 OAKE_BENCH_RAMP_S=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch_walk${w}_$m -o p -- python $GRAFT_REPO_ROOT/bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline --no-modes --no-profile > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_fetch_walk${w}_$m.err
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/qkv_walk_fetch.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r06/pmc_fetch_walk*')):
    if d.endswith('.err'): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if 'qkv_attn' in k or 'Li8E' in k or 'Li5E' in k:
            print(d.split('/')[-1], k, 'n=%d' % len(v), 'reads %.1f MB/launch (FETCH_SIZE x2)' % (2 * 1024 * sum(v) / len(v) / 1e6))
PY
find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +3M -delete
du -sh $O
