#!/bin/bash
# round 5, session H (lab): what k_stage1_summarize waits for.  Variants that are WRONG on purpose (their digests differ): L1 writes no masks, L2 skips the
# UTF-8 noting and parking, L3 both, L6/L7 82 VGPRs = five waves per SIMD instead of six
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_QUICK=1
timeout 1200 python scripts/lib_ab.py new=simdjson_amd/lib/libsjgpu.so L1_nostore=build/ab/libsjgpu_L1_nostore.so L2_noutf8=build/ab/libsjgpu_L2_noutf8.so L3_both=build/ab/libsjgpu_L3_both.so L6_occ5=build/ab/libsjgpu_L6_occ8.so --rounds 2 > $O/r5h_lib_ab.jsonl 2> $O/r5h_lib_ab.err; echo "ab rc=$?"
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5h_lib_ab.jsonl") if l.startswith("{")]
names = []
for r in rows:
    if r["variant"] not in names: names.append(r["variant"])
keys = [k for k in rows[0] if k.endswith(":us")]
print("%-30s" % "us per call (best)", *["%10s" % n[:10] for n in names])
for k in keys:
    print("%-30s" % k[:-3], *["%10.1f" % min(r[k] for r in rows if r["variant"] == n) for n in names])
for k in [k for k in rows[0] if k.endswith(":slots_us")]:
    for slot in range(3):
        print("%-30s" % (k[:-9] + " slot %d" % slot), *["%10.1f" % min(r[k][slot] for r in rows if r["variant"] == n) for n in names])
PY
