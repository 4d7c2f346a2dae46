python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 || exit 1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference > gpurun_out/b_h2o.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload c6h6_ccpvdz > gpurun_out/b_c6h6.json 2>/dev/null
python - <<PY
import json
for f in ("gpurun_out/b_h2o.json","gpurun_out/b_c6h6.json"):
    d=json.load(open(f)); print(d["config"]["workload"], "ms/step %.3f"%d["ms_per_step"], d["iter_breakdown_ms"])
    fb=d["roofline"]["fock_build"]; print(" fock ms %.3f sum %.3f TF %.3f"%(fb["ms"],fb["sum_tier_kernels_serial_ms"],d["roofline"]["fp64_valu"]["achieved"]))
    for c in fb["top_classes_serial"][:8]: print("   ",c)
PY
