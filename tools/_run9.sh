mkdir -p gpurun_out/head
python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "pass_cap or 1024" -s 2>&1 | grep -E "passes of|passed|failed|Error|max\|" | head
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/head/pytest_gpu_all3.txt
cat gpurun_out/head/pytest_gpu_all3.txt
