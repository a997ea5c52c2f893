for sc in 20 22 24 26; do python bench.py --no-cpu-baseline --scale $sc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale $sc', round(d['value'],1), round(d['roofline']['kernel_ms'],4))"; done
