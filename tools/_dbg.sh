cd /root/repo
for P in 1 3; do
  for G in "" "--no-graph"; do
    echo "pipeline $P $G"; python bench.py --pipeline $P --steps 200 --warmup 10 --no-cpu-baseline --no-extras $G 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config'].get('verified'), d['roofline']['solo_launch_ms'])"
  done
done
echo "2k"; for G in "" "--no-graph"; do python bench.py --workload 2k_qam16_1_2 --pipeline 1 --steps 400 --warmup 10 --no-cpu-baseline --no-extras $G 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config'].get('verified'))"; done
