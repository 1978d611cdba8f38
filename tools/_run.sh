python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bgr or cpp_cylinder" 2>&1 | grep -E "passed|failed|Error|error|^FAILED|assert|^E " | head -30
python bench.py --cpu-frames 0 --pose-frames 0 --latency-calls 0 --host-frames 512 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d.get('pcie_inclusive'), d.get('pcie_inclusive_bgr'))"
