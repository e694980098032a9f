for p in 2 3 4 3 2; do
timeout -k 10 200 python bench.py --pipeline $p --no-train --no-next --no-alt --no-c1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline', j['config']['pipeline_depth'], round(j['value']), round(j['ms_per_step'],3))" || exit 1
done
