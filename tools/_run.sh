python -m pytest tests/test_gpu_parity.py tests/test_dictgen_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|^FAILED|assert" | head -20
