for r in 1 2 3; do for l in 2 3; do python bench.py --lanes $l --steps 80 --warmup 10 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2 --no-configs --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16 lanes $l:', j['value'])"; done; done
for r in 1 2; do for l in 2 3; do python bench.py --precision fp16x3 --lanes $l --steps 30 --warmup 6 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2 --no-configs --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('fp16x3 lanes $l:', j['value'])"; done; done
