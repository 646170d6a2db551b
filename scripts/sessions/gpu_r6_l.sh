#!/bin/bash
# round 6, session L: the same box, the same library: scripts/lib_ab.py's figure for the split pipeline against bench.py's legs (are the 5 % between them the box or the harness?)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="amazon_ndjson:split:stage1,escape_heavy:split:stage1"
timeout 900 python scripts/lib_ab.py tree=build/ab/libsjgpu_tree.so tree2=build/ab/libsjgpu_tree2.so --rounds 8 --reps 10 > $O/r6l_lib_ab.txt 2> $O/r6l_lib_ab.err; grep -v "^{" $O/r6l_lib_ab.txt | head -12
for i in 1 2; do
timeout 600 python bench.py --legs config3_amazon_ndjson,config4_escape_heavy --no-cpu-baseline > $O/r6l_bench_$i.json 2> $O/r6l_bench_$i.err; python3 - <<PY
import json
d=json.load(open("bench_detail.json"))
for k,v in d["legs"].items(): print("bench run $i", k, v["ms_per_step"], v["first_reps_ms_per_step"], v["roofline"]["kernel_ms_slots"])
PY
done
timeout 600 python bench.py --legs none --workload amazon_ndjson --no-cpu-baseline > $O/r6l_bench_nd.json 2>/dev/null; python3 -c "
import json; d=json.load(open('bench_detail.json')); print('bench ndjson headline', d['ms_per_step'], d['first_reps_ms_per_step'], d['roofline']['kernel_ms_slots'])"
