cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "padded" 2>&1 | tail -2
for s in 231 237; do timeout 900 python tests/fuzz_pipeline.py 12 $s 2>&1 | grep -v "^\[" | tail -6; done | tee $O/fuzz_pipeline_padded.txt
