F="--steps 60 --warmup 10 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2"
p() { python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], 'single', (j.get('single_stream') or {}).get('value'))"; }
for r in 1 2; do
 (cd _r04 && python bench.py $F 2>/dev/null | tail -1 | p "r04 bf16   ")
 (python bench.py $F --no-configs 2>/dev/null | tail -1 | p "r05 bf16   ")
 (cd _r04 && python bench.py --precision fp16x3 $F 2>/dev/null | tail -1 | p "r04 fp16x3 ")
 (python bench.py --precision fp16x3 $F --no-configs 2>/dev/null | tail -1 | p "r05 fp16x3 ")
done
