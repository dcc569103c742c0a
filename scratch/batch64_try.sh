for i in 1 2 3; do
  for b in 32 64; do
    python bench.py --batch $b --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', $b, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['frames_per_launch'])"
  done
done
