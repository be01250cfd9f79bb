# round 6, GPU call 6: crops straight into the padded conv1 batch (no pad_nchw), short last tap group of the horizontal pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
( timeout 1500 python -m pytest tests/test_resample_gpu.py tests/test_edge_cases_gpu.py tests/test_oake_gpu.py tests/test_jpeg.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -4
  timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_hooks_golden_gpu.py tests/test_hook_branch_gpu.py tests/test_integration_doc_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $O/call6_pytest.txt 2>&1
cat $O/call6_pytest.txt
for s in 211 223; do timeout 600 python tools/resample_fuzz.py 150 $s 2>&1 | tail -3; done | tee $O/fuzz_resample_padded.log
timeout 600 python tools/blocks_fuzz.py 40 211 2>&1 | tail -1 | tee -a $O/fuzz_resample_padded.log
timeout 900 python tests/fuzz_pipeline.py 12 211 2>&1 | tail -1 | tee -a $O/fuzz_resample_padded.log
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 3 dense:OAKE_PADDED_CROPS=0 padded:OAKE_PADDED_CROPS=1 > $O/ab_padded_crops_objects.log 2>&1; tail -3 $O/ab_padded_crops_objects.log
OAKE_BENCH_FULL_LINE=1 python bench.py --mode objects --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_objects_padded.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_objects_padded.json') if l.startswith('{')][-1])
print(d['value'], {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if v['ms_per_step'] < 1.5})
PY
