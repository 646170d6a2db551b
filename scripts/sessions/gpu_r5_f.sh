#!/bin/bash
# round 5, session F: the cheaper scan (transposition 145 -> 119 VALU per block, classes 43 -> 37, the two lane carries by v_mbcnt / DPP), k_resolve_segments'
# scans by DPP, k_stage1_emit with four waves per workgroup -- against the library of the commit before (build/ab/libsjgpu_base.so) on the same box, then the
# parity tests that exercise those kernels
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/lib_ab.py base=build/ab/libsjgpu_base.so new=simdjson_amd/lib/libsjgpu.so new_emit1=simdjson_amd/lib/libsjgpu.so,SJGPU_EMIT_WAVES=1 --rounds 2 > $O/r5f_lib_ab.jsonl 2> $O/r5f_lib_ab.err; echo "ab rc=$?"
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5f_lib_ab.jsonl") if l.startswith("{")]
keys = [k for k in rows[0] if k.endswith(":us")]
names = sorted({r["variant"] for r in rows})
print("%-36s" % "us per call (best of the rounds)", *["%12s" % n for n in names])
for k in keys:
    print("%-36s" % k[:-3], *["%12.1f" % min(r[k] for r in rows if r["variant"] == n) for n in names])
for k in [k for k in rows[0] if k.endswith(":digest") or k.endswith(":flags")]:
    vals = {json.dumps(r[k]) for r in rows}
    print("digest", k, "SAME" if len(vals) == 1 else "DIFFERENT: " + str(vals))
PY
timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -p no:cacheprovider --durations=30 -k "built_from or golden or straddling or quote_parity or adversarial_shapes or control_character or fuzz or streaming_modes or dense_non_ascii or deterministic or full_size_device_resident or full_size_adversarial or strings_with_every or string_stream_random or tape_of_random or tape_numbers or minify_and_validate or ranges_equal or backslash_runs" > $O/r5f_pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $O/r5f_pytest.log | cut -c1-160
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/r5f_bench_n2_dry.json 2> $O/r5f_bench_n2_dry.err; echo "n2 dry rc=$?"; tail -c 600 $O/r5f_bench_n2_dry.json
