run() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['c3']; print(d['value'], 'c3', c['value'], c['ms_per_step'], 'c5', d['c5']['ms_per_iteration'])"; }
run; run; run
