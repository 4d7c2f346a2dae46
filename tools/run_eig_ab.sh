for v in 0 1; do
  if [ $v = 1 ]; then export QC_EIG_ONESIDED=1; else unset QC_EIG_ONESIDED; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference > gpurun_out/b_eig$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b_eig$v.json')); print('onesided=$v ms/step %.3f'%d['ms_per_step'], d['iter_breakdown_ms'])"
done
