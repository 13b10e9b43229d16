for c in 2 3 5 2m; do timeout -k 10 300 python bench.py --config $c --no-cpu 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$c', j['value'], j['ms_per_step'], j['batches_in_flight'])"; done
