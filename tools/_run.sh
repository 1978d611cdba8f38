python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "params or every_stage or golden or fuzz or whole_wave or long_thin" 2>&1 | grep -E "passed|failed|Error|error|^FAILED|assert|^E " | head -30
B="python bench.py --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0"
$B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('default', d['value'], d['stage_ms_per_step'])"
