#!/bin/bash
# scripts/gpu_pmc_cmd.sh -- PMC passes over ANY command (each counter set in its own rocprofv3 run, kernel-trace only).
# Usage (from the repo root, via gpurun): bash scripts/gpu_pmc_cmd.sh <tag> "<set set ...>" -- <command...>
#   sets: fetch write sq1 sq2 tcc tcp ta
# Output: gpurun_out/pmc_<tag>/summary.json + a table on stdout (mean per dispatch, summed over XCDs).
set -u
TAG=$1; SETS=$2; shift 3
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
run() { # name, counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$TAG/$name -o $name -- "${CMD[@]}" > $ROOT/gpurun_out/pmc_$TAG/$name.log 2>&1)
  echo "pmc $name rc=$?"
}
CMD=("$@")
for set in $SETS; do
  case $set in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    sq1) run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR ;;
    sq2) run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE ;;
    tcc) run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum ;;
    tcc2) run tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum ;;
    tcp) run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum ;;
    ta) run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum ;;
  esac
done
python3 - <<PY
import csv, glob, collections, json
summary = {}
for f in sorted(glob.glob("gpurun_out/pmc_$TAG/*/*counter_collection.csv") + glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r.get("Kernel_Name", "?")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "fill" in k or "copy" in k.lower():
            continue
        summary.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(summary, open("gpurun_out/pmc_$TAG/summary.json", "w"), indent=1)
for k, v in summary.items():
    print(k)
    print("   ", {c: float("%.4g" % x) for c, x in v.items()})
PY
