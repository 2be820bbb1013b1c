for cfg in "4 0" "4 256" "2 0" "2 256" "8 0" "8 256" "6 256" "3 0"; do set -- $cfg; python bench.py --steps 1 --warmup 1 --large-games 0 --no-cpu-baseline --sets $1 --trunk-grid $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('sets $1 grid $2:', round(d['value']), 'games/s', round(d['leaf_evals_per_s']/1e6,1), 'M evals/s')
"; done
