#!/bin/bash
# round 4, session t: the eight-wave kernels inside bench.py (events per step, a wait between steps) against the four-wave ones, same box, same process shape
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in 4 8 4 8; do
  SJGPU_PIPE_WAVES=$w SJGPU_MINIFY_WAVES=$w timeout 600 python bench.py --legs config2_minify,config3_amazon_ndjson --no-cpu-baseline > gpurun_out/r4t_bench_w$w.json 2> gpurun_out/r4t_bench_w$w.err
  python3 - <<PY
import json
d = json.load(open("gpurun_out/r4t_bench_w$w.json"))
r = d["roofline"]
print("waves $w: headline", r["gpu_ms_per_step"], d["ms_per_step"], r["kernel"], "| minify", d["legs"]["config2_minify"]["roofline"]["gpu_ms_per_step"], d["legs"]["config2_minify"]["roofline"]["kernel"],
      "| ndjson", d["legs"]["config3_amazon_ndjson"]["roofline"]["gpu_ms_per_step"], d["legs"]["config3_amazon_ndjson"]["roofline"]["kernel"][:40])
PY
done
