# round 6, GPU call 20: the pass planner (api.hip plan_pass_size) — encoder suite, then blocks 640x480 with the planner's
# choice (3 x 512 + 192 crops) against four equal passes (cap lowered to 432), interleaved; objects / blocks 1700x1134 once
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/planner; mkdir -p $O
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_gpu.py -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
for r in 1 2; do for c in 432 0; do
  if [ $c = 0 ]; then E=""; else E="OAKE_PASS_CROPS=$c"; fi
  env $E OAKE_BENCH_FULL_LINE=1 timeout 600 python bench.py --mode blocks --no-cpu-baseline 2>/dev/null | tail -1 > $O/blocks_cap${c}_r$r.json
  python - <<PY
import json
d = json.load(open('$O/blocks_cap${c}_r$r.json'))
k = d.get('kernels', {})
print('blocks cap $c run $r', d['value'], d['unit'], ' '.join(f"{n} {k[n]['ms_per_step']:.3f}x{k[n]['launches_per_step']:.0f}" for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_out_proj', 'qkv_attn') if n in k))
PY
done; done 2>&1 | tee $O/ab_blocks.txt
for m in objects "blocks --image-size 1700x1134 --steps 8 --warmup 2"; do
  OAKE_BENCH_FULL_LINE=1 timeout 900 python bench.py --mode $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$m', d['value'], d['unit'], ' '.join(f\"{n} {k[n]['ms_per_step']:.3f}x{k[n]['launches_per_step']:.0f}\" for n in ('gemm_c_fc', 'gemm_c_proj', 'qkv_attn') if n in k))"
done 2>&1 | tee $O/other_modes.txt
