# round 6, GPU call 21: blocks 640x480 with larger encoder passes (the cap is a memory bound: 25 600 token rows by default)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/planner; mkdir -p $O
for r in 1 2; do for c in "512 25600" "1024 51200" "2048 0"; do
  set -- $c
  OAKE_PASS_ROWS=$2 OAKE_BENCH_FULL_LINE=1 timeout 600 python bench.py --mode blocks --max-batch $1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/blocks_mb$1_r$r.json
  python - <<PY
import json
d = json.load(open('$O/blocks_mb$1_r$r.json'))
k = d.get('kernels', {})
print('blocks max_batch $1 pass_rows $2 run $r', d['value'], 'one lane', d['one_lane_images_per_sec'], ' '.join(f"{n} {k[n]['ms_per_step']:.3f}x{k[n]['launches_per_step']:.0f}" for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_out_proj', 'qkv_attn') if n in k))
PY
done; done 2>&1 | tee $O/ab_blocks_pass_rows.txt
