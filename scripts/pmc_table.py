"""Tables from the summary.json files scripts/gpu_pmc_cmd.sh leaves (means per dispatch, summed over the 8 XCDs):
python scripts/pmc_table.py gpurun_out/pmc_<tag> [...]   -> fetch / write GB, VALU, issue share, waits, LDS per kernel"""
import json
import sys


def short(name):
    name = name.replace("sjgpu::(anonymous namespace)::", "").replace("void ", "").replace("sjgpu::", "")
    return name.split("(")[0][:34]


def table(path):
    d = json.load(open(path + "/summary.json"))
    print(f"== {path}")
    print(f"{'kernel':36s}{'fetch GB':>9s}{'write GB':>9s}{'VALU M':>9s}{'VALU %':>7s}{'cycles M':>9s}{'wait %':>7s}{'LDS M':>8s}{'bankconf M':>11s}")
    tot_f = tot_w = 0.0
    for k, v in d.items():
        f = v.get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e9
        w = v.get("WRITE_SIZE", 0.0) * 1024 / 1e9
        tot_f += f
        tot_w += w
        valu = v.get("SQ_INSTS_VALU", 0.0)
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8
        share = 100 * valu * 4 / (1024 * cyc) if cyc else float("nan")
        wait = 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in v else float("nan")
        print(f"{short(k):36s}{f:9.3f}{w:9.3f}{valu / 1e6:9.1f}{share:7.0f}{cyc / 1e6:9.2f}{wait:7.0f}{v.get('SQ_INSTS_LDS', 0.0) / 1e6:8.1f}{v.get('SQ_LDS_BANK_CONFLICT', 0.0) / 1e6:11.1f}")
    print(f"{'all kernels':36s}{tot_f:9.3f}{tot_w:9.3f}   -> {tot_f + tot_w:.3f} GB per call")
    return d


for p in sys.argv[1:]:
    table(p.rstrip("/"))
    print()
