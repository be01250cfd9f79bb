for i in 1 2 3; do
  for v in 0 1; do
    OAKE_GEMM_MFMA32=$v python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('mfma32=$v', d['value'], 'c_proj', k['gemm_c_proj']['ms_per_step'], 'out_proj', k['gemm_out_proj']['ms_per_step'], 'c_fc', k['gemm_c_fc']['ms_per_step'], 'qkv', k['gemm_qkv']['ms_per_step'])"
  done
done
