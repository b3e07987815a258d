timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_mvit_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
uptime
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
uptime | sed 's/.*load/load/'
done
