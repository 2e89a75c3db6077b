#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_final.sh (gpurun_out/pmc_f/<workload>_<pass>/p_counter_collection.csv)
into the per-launch summary kept under profiles/: raw counters (median and max dispatch) and the derived rates --
instructions per chain-sweep, fp64 VALU rate against the 78.6 TFLOP/s vector peak, VALU busy, traffic beyond L2
(FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE counting half the bytes on gfx950: MI355X_MICROARCH.md), TCC hit rate, LDS
bank conflicts.

    python tools/pmc_report.py [gpurun_out/pmc_f] [--traffic-json] > profiles/r01_g_pmc_summary.txt
(--traffic-json also rewrites profiles/traffic.json, the PMC traffic bench.py quotes in roofline.traffic)
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
SRC = _pos[0] if _pos else os.path.join(ROOT, "gpurun_out", "pmc_f")

# workload -> (description, chain-sweeps of the launch looked at, kernel, proposals per sweep or None, which dispatch)
UNITS = {
    "free": ("bench.py --steps 3 --warmup 1 --no-cpu: launch = 100 sweeps x 4096 chains (the median dispatch)", 100 * 4096, "free_stage_kernel", 3666, "median"),
    "edge": ("tools/bench_param.py edge: launch = 40 sweeps x 4096 chains (N_e=80)", 40 * 4096, "edge_stage_kernel", None, "median"),
    "peak": ("tools/bench_param.py peak: the 100-sweep launch of the timed sample stage = the largest dispatch "
             "(the median dispatch is a 10-sweep adjust_windows launch); 1000+1 deltas, 4096 chains", 100 * 4096,
             "peak_stage_kernel", None, "max"),
}


def per_dispatch(path, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if want in row["Kernel_Name"]:
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    med = {c: sorted(v.values())[len(v) // 2] for c, v in acc.items()}
    mx = {c: max(v.values()) for c, v in acc.items()}
    return med, mx


def main():
    print("rocprofv3 --pmc passes on the committed library (scripts/pmc_final.sh; one counter set per run).")
    print("Values are per dispatch, summed over XCDs/instances.  FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE counts half the")
    print("bytes on gfx950 (MI355X_MICROARCH.md), hence (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes beyond L2.")
    for w, (desc, cs, kern, pps, which) in UNITS.items():
        print()
        print("==", kern, "--", desc)
        use = {}
        for p in ("p1", "p2", "p3", "p4"):
            path = os.path.join(SRC, f"{w}_{p}", "p_counter_collection.csv")
            if not os.path.exists(path):
                print(f"  ({path} missing)")
                continue
            med, mx = per_dispatch(path, kern)
            for c in sorted(med):
                print(f"  {c:26s} median {med[c]:.6g}   max {mx[c]:.6g}")
            use.update(med if which == "median" else mx)
        need = ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_ADD_F64",
                "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_ACTIVE_INST_VALU", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum",
                "TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
        if any(k not in use for k in need):
            continue
        gui = use["GRBM_GUI_ACTIVE"] / 8.0                    # cycles per XCD
        t = gui / 2.4e9
        print(f"  derived ({which} launch, {cs} chain-sweeps, ~{t * 1e3:.1f} ms at 2.4 GHz):")
        print(f"    instructions per chain-sweep: VALU {use['SQ_INSTS_VALU'] / cs:.0f}  SALU {use['SQ_INSTS_SALU'] / cs:.0f}  "
              f"LDS {use['SQ_INSTS_LDS'] / cs:.0f}  VMEM_RD {use['SQ_INSTS_VMEM_RD'] / cs:.0f}")
        if pps:
            print(f"    per proposal ({pps} per sweep): VALU {use['SQ_INSTS_VALU'] / cs / pps:.1f}  SALU {use['SQ_INSTS_SALU'] / cs / pps:.1f}")
        fl = 64 * (use["SQ_INSTS_VALU_ADD_F64"] + use["SQ_INSTS_VALU_MUL_F64"] + 2 * use["SQ_INSTS_VALU_FMA_F64"])
        print(f"    fp64 VALU: {fl:.3g} flop per launch = {fl / t / 1e12:.2f} TFLOP/s = {fl / t / 78.6e12 * 100:.1f} % of the 78.6 TFLOP/s "
              f"vector fp64 peak (-ffp-contract=off: separate multiplies and adds)")
        quad = 1024 * gui / 4
        print(f"    VALU busy: SQ_ACTIVE_INST_VALU {use['SQ_ACTIVE_INST_VALU']:.3g} quad-cycles of {quad:.3g} SIMD quad-cycles = "
              f"{use['SQ_ACTIVE_INST_VALU'] / quad * 100:.0f} %")
        fb = (2 * use["FETCH_SIZE"] + use["WRITE_SIZE"]) * 1024
        hit = use["TCC_HIT_sum"] / (use["TCC_HIT_sum"] + use["TCC_MISS_sum"])
        print(f"    beyond-L2 traffic: {fb / 1e9:.1f} GB per launch = {fb / t / 1e12:.2f} TB/s ({fb / cs / 1e3:.0f} KB per chain-sweep); "
              f"TCC hit rate {hit * 100:.0f} %")
        print(f"    LDS: bank-conflict cycles {use['SQ_LDS_BANK_CONFLICT']:.3g} of {use['SQ_LDS_IDX_ACTIVE']:.3g} active = "
              f"{use['SQ_LDS_BANK_CONFLICT'] / max(use['SQ_LDS_IDX_ACTIVE'], 1) * 100:.1f} %")
        if w == "free" and "--traffic-json" in sys.argv:
            json.dump({"_comment": "HBM/fabric-side traffic of free_stage_kernel per launch from rocprofv3 --pmc (separate passes; "
                                   "profiles/r01_g_pmc_summary.txt), for bench.py's roofline.traffic. bytes = (2*FETCH_SIZE + "
                                   "WRITE_SIZE)*1024: FETCH_SIZE counts half the bytes on gfx950, calibrated on this access pattern in "
                                   "profiles/r01_gather_probe.txt.",
                       "workload": {"n_tau": 64, "n_omega": 2000, "n_grid": 100001, "chains": 4096, "sweeps_per_launch": 100},
                       "fetch_size_kb_median": use["FETCH_SIZE"], "write_size_kb_median": use["WRITE_SIZE"],
                       "valu_active_quad_cycles_median": use["SQ_ACTIVE_INST_VALU"], "valu_insts_median": use["SQ_INSTS_VALU"],
                       "salu_insts_median": use["SQ_INSTS_SALU"],
                       "source": "profiles/r01_g_pmc_summary.txt"},
                      open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=2)


if __name__ == "__main__":
    main()
