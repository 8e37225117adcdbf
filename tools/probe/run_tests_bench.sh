timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ps.py tests/test_gpu_fit.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -4 && timeout -k 10 300 python bench.py --no-configs > gpurun_out/bench_r3g.json 2> gpurun_out/bench_r3g.err && python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r3g.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["sum_t_roof_over_sum_t"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:22]: print(k, v["ms_per_step"], v.get("frac"))
PY
