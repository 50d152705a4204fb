timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
for s in 1 8; do echo "inflight $s"; timeout 200 python bench.py --no-cpu-baseline --inflight $s 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['all_kernels_us'])"; done
