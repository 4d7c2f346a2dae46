for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference > gpurun_out/b_h2o_q$q.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b_h2o_q$q.json")); fb=d["roofline"]["fock_build"]
print("queues $q", "ms/step %.3f"%d["ms_per_step"], d["iter_breakdown_ms"], "fock %.3f sum %.3f"%(fb["ms"],fb["sum_class_kernels_ms"]))
PY
done
