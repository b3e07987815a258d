NC=$(nproc)
echo "cores $NC"
PIDS=""
for i in $(seq 1 $((NC * 3))); do
  python -c "while True: pass" &
  PIDS="$PIDS $!"
done
sleep 2
python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('contended graphs   ', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
python bench.py --no-cpu-baseline --no-kernel-timing --no-graphs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('contended no-graphs', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
kill $PIDS
wait 2>/dev/null
python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quiet graphs       ', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
python bench.py --no-cpu-baseline --no-kernel-timing --no-graphs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quiet no-graphs    ', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
