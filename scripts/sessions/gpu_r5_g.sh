#!/bin/bash
# round 5, session G: session F found the cheaper scan SLOWER (NDJSON split 328 -> 346 us, escape_heavy 254 -> 283) although it issues fewer instructions:
# which step costs?  Libraries of the tree before (base) and with the steps added one by one -- A the transposition by rotations, B the class functions,
# C the lane carries by v_mbcnt / DPP wave_shr, E the DPP scans of k_resolve_segments, then the whole new tree (+ the tail note, + emit's workgroup shape)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_QUICK=1
timeout 1200 python scripts/lib_ab.py base=build/ab/libsjgpu_base.so S1_A=build/ab/libsjgpu_S1_A.so S2_AB=build/ab/libsjgpu_S2_AB.so S3_ABC=build/ab/libsjgpu_S3_ABC.so S4_ABCE=build/ab/libsjgpu_S4_ABCE.so new_emit1=simdjson_amd/lib/libsjgpu.so,SJGPU_EMIT_WAVES=1 new_emit4=simdjson_amd/lib/libsjgpu.so --rounds 2 > $O/r5g_lib_ab.jsonl 2> $O/r5g_lib_ab.err; echo "ab rc=$?"
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5g_lib_ab.jsonl") if l.startswith("{")]
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
for k in [k for k in rows[0] if k.endswith(":digest") or k.endswith(":flags")]:
    vals = {json.dumps(r[k]) for r in rows}
    print("digest", k, "SAME" if len(vals) == 1 else "DIFFERENT: " + str(vals))
PY
