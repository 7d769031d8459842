#!/bin/bash
# bench lines of BASELINE.json configs 3 and 5 (rollout only) -> gpurun_out/$1/
out=gpurun_out/${1:-r2}; mkdir -p $out
python bench.py --task anymal_c_rough --training-iters 0 > $out/bench_rough.json 2> $out/bench_rough.err &&
python bench.py --task cassie --training-iters 0 > $out/bench_cassie.json 2> $out/bench_cassie.err &&
python - <<PY
import json
for t in ("rough", "cassie"):
    d = json.load(open("$out/bench_%s.json" % t))
    print(t, "%.3e" % d["value"], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", d["roofline"]["kernel_ms"], d["config"].get("policy"))
PY
