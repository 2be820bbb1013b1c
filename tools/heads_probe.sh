for h in 0 2 1 0 2; do
  AQG_HEADS_IN_TRUNK=$h python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('heads_in_trunk $h:', round(d['value'],1), 'games/s')"
done
