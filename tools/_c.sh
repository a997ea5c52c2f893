timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/sp.log 2>&1; tail -2 gpurun_out/sp.log
for ch in 21 20 19 18; do GRAPHTAP_PB_CH=$ch GRAPHTAP_PB_STATS=1 python bench.py --no-cpu-baseline --steps 5 2>&1 | grep "nbins\|slots after\|balance" | cut -c1-220; done
python bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1
for rep in 1 2 3; do
GRAPHTAP_PB_SPLIT=entries python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('entries CH 21', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
for ch in 21 20 19 18; do GRAPHTAP_PB_CH=$ch python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bins    CH $ch', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['value_checksum'], d['config']['ingress_s'])"; done; done
