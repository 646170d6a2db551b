#!/bin/bash
# scripts/gpu_pmc.sh -- PMC passes (each counter set in its own rocprofv3 run, kernel-trace only).
# Usage: bash scripts/gpu_pmc.sh "<bench args>" tag ["fetch write sq1 sq2"]
set -u
ARGS=${1:-"--op stage1"}
TAG=${2:-stage1}
SETS=${3:-"fetch write sq1 sq2"}
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name -o $name -- \
     python $GRAFT_REPO_ROOT/bench.py --legs none $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$name.log 2>&1)
  echo "pmc $name rc=$?"
}
for set in $SETS; do
  case $set in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    sq1) run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR ;;
    sq2) run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE ;;
  esac
done
find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head
python3 - <<PY
import csv, glob, collections
import json
summary={}
for f in sorted(glob.glob("gpurun_out/pmc_$TAG/*/*counter_collection.csv")+glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r.get("Kernel_Name","?")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k,v in agg.items():
        if "fill" in k or "copy" in k: continue
        print("  ", k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
        summary.setdefault(k, {}).update({c: sum(x)/len(x) for c,x in v.items()})
json.dump(summary, open("gpurun_out/pmc_$TAG/summary.json","w"), indent=1)
PY
