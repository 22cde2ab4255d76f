for k in 20 20 200 200; do python bench.py --steps $k --warmup 5 --no-cpu-baseline --also "" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['steps'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done
