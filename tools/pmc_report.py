#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_final.sh (gpurun_out/pmc_f/<workload>_<pass>/p_counter_collection.csv)
into the per-launch summary kept under profiles/: raw counters (median and max dispatch) and the derived rates --
instructions per chain-sweep, fp64 VALU rate against the 78.6 TFLOP/s vector peak, VALU busy, traffic beyond L2
(FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE counting half the bytes on gfx950: MI355X_MICROARCH.md), TCC hit rate, LDS
bank conflicts.

    python tools/pmc_report.py [gpurun_out/pmc_r02] [--traffic-json] [--source=profiles/...] > profiles/r02_pmc_summary.txt
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
    "hbm1": ("tools/bench_param.py free_hbm --d-omega 1e-5: sac/free on the 1000001-column table (488 MiB), launch = 100 sweeps x 4096 chains "
             "(bench.py hbm_regime[0]); the largest dispatch", 100 * 4096, "free_stage_kernel", 3666, "max"),
    "hbm4": ("tools/bench_param.py free_hbm --d-omega 2.5e-6: sac/free on the 4000001-column table (1953 MiB), launch = 100 sweeps x 4096 chains "
             "(bench.py hbm_regime[1]); the largest dispatch", 100 * 4096, "free_stage_kernel", 3666, "max"),
    "edge": ("tools/bench_param.py edge: launch = 40 sweeps x 4096 chains (N_e=80)", 40 * 4096, "edge_stage_kernel", None, "median"),
    "peak": ("tools/bench_param.py peak: the 100-sweep launch of the timed sample stage = the largest dispatch "
             "(the median dispatch is a 10-sweep adjust_windows launch); 1000+1 deltas, 4096 chains (bench.py hbm_regime[2])", 100 * 4096,
             "peak_stage_kernel", None, "max"),
    "pkhb": ("tools/bench_param.py peak_hbm: bench.py's hbm_regime[2] (BASELINE configs[3]) with its seed and stage sequence; the largest dispatch "
             "= a 100-sweep launch x 4096 chains", 100 * 4096, "peak_stage_kernel", None, "max"),
    "pkh4": ("tools/bench_param.py peak_hbm --d-omega 2.5e-6: the same on a 4000001-column table (1953 MiB, bench.py hbm_regime[3]); the largest "
             "dispatch = a 100-sweep launch x 4096 chains", 100 * 4096, "peak_stage_kernel", None, "max"),
    "pk2": ("tools/bench_param.py peak2: sac/peak fermionic finite-T, both frequency axes with the weight-transfer moves (XFER kernels), "
            "2 x (1000+1) deltas, two 488 MiB tables; the largest dispatch = a 100-sweep launch x 4096 chains", 100 * 4096,
            "peak_stage_kernel", None, "max"),
    "pack": ("tools/bench_param.py free_packed: N_tau=22, 1000 deltas, 8192 chains, two chains per wavefront; launch = 100 sweeps x 8192 chains",
             100 * 8192, "free_stage_packed_kernel", 1833, "median"),
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
    print("rocprofv3 --pmc passes on the committed library (scripts/r03_pmc.sh; one counter set per run).")
    print("Values are per dispatch, summed over XCDs/instances.  FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE counts half the")
    print("bytes on gfx950 (MI355X_MICROARCH.md), hence (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes beyond L2.")
    traffic, free_use = {}, None
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
        if "GRBM_GUI_ACTIVE" not in use:
            continue
        has = lambda *ks: all(k in use for k in ks)
        gui = use["GRBM_GUI_ACTIVE"] / 8.0                    # cycles per XCD
        t = gui / 2.4e9
        print(f"  derived ({which} launch, {cs} chain-sweeps, ~{t * 1e3:.1f} ms at 2.4 GHz):")
        if has("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
            print(f"    instructions per chain-sweep: VALU {use['SQ_INSTS_VALU'] / cs:.0f}  SALU {use['SQ_INSTS_SALU'] / cs:.0f}  "
                  f"LDS {use['SQ_INSTS_LDS'] / cs:.0f}  VMEM_RD {use['SQ_INSTS_VMEM_RD'] / cs:.0f}")
            if pps:
                print(f"    per proposal ({pps} per sweep): VALU {use['SQ_INSTS_VALU'] / cs / pps:.1f}  SALU {use['SQ_INSTS_SALU'] / cs / pps:.1f}")
        if has("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"):
            fl = 64 * (use["SQ_INSTS_VALU_ADD_F64"] + use["SQ_INSTS_VALU_MUL_F64"] + 2 * use["SQ_INSTS_VALU_FMA_F64"])
            print(f"    fp64 VALU: {fl:.3g} flop per launch = {fl / t / 1e12:.2f} TFLOP/s = {fl / t / 78.6e12 * 100:.1f} % of the 78.6 TFLOP/s "
                  f"vector fp64 peak (-ffp-contract=off: separate multiplies and adds)")
        if has("SQ_ACTIVE_INST_VALU"):
            quad = 1024 * gui / 4
            print(f"    VALU busy: SQ_ACTIVE_INST_VALU {use['SQ_ACTIVE_INST_VALU']:.3g} quad-cycles of {quad:.3g} SIMD quad-cycles = "
                  f"{use['SQ_ACTIVE_INST_VALU'] / quad * 100:.0f} %")
        if has("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
            fb = (2 * use["FETCH_SIZE"] + use["WRITE_SIZE"]) * 1024
            hit = use["TCC_HIT_sum"] / (use["TCC_HIT_sum"] + use["TCC_MISS_sum"])
            print(f"    beyond-L2 traffic: {fb / 1e9:.1f} GB per launch = {fb / t / 1e12:.2f} TB/s = {fb / t / 8e12:.2f} of the 8 TB/s HBM peak "
                  f"({fb / cs / 1e3:.0f} KB per chain-sweep); TCC hit rate {hit * 100:.0f} %")
            traffic[w] = {"fetch_size_kb": use["FETCH_SIZE"], "write_size_kb": use["WRITE_SIZE"], "bytes_per_launch": fb,
                          "launch_ms_at_2.4GHz": t * 1e3, "tcc_hit_rate": hit}
        if has("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            print(f"    LDS: bank-conflict cycles {use['SQ_LDS_BANK_CONFLICT']:.3g} of {use['SQ_LDS_IDX_ACTIVE']:.3g} active = "
                  f"{use['SQ_LDS_BANK_CONFLICT'] / max(use['SQ_LDS_IDX_ACTIVE'], 1) * 100:.1f} %")
        if w == "free":
            free_use = dict(use)
    if "--traffic-json" in sys.argv and free_use:
        use = free_use
        src = [a for a in sys.argv[1:] if a.startswith("--source=")]
        source = src[0].split("=", 1)[1] if src else "profiles/r02_pmc_summary.txt"
        json.dump({"_comment": "HBM/fabric-side traffic per launch from rocprofv3 --pmc (separate passes; scripts/r03_pmc.sh, summary in 'source'), "
                               "for bench.py's roofline.traffic and hbm_regime[].traffic. bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE counts "
                               "half the bytes on gfx950, calibrated on this access pattern in profiles/r01_gather_probe.txt. "
                               "valu_active_quad_cycles = SQ_ACTIVE_INST_VALU (quad-cycles a SIMD's vector ALU was busy), for roofline.bound = valu-issue.",
                   "workload": {"n_tau": 64, "n_omega": 2000, "n_grid": 100001, "chains": 4096, "sweeps_per_launch": 100},
                   "fetch_size_kb_median": use["FETCH_SIZE"], "write_size_kb_median": use["WRITE_SIZE"],
                   "valu_active_quad_cycles_median": use["SQ_ACTIVE_INST_VALU"], "valu_insts_median": use["SQ_INSTS_VALU"],
                   "salu_insts_median": use["SQ_INSTS_SALU"], "vmem_rd_insts_median": use.get("SQ_INSTS_VMEM_RD"),
                   "source": source,
                   # sac_build_id() of the library that ran under the profiler (the PMC script writes it next to the passes):
                   # bench.py quotes these numbers only when the library it runs has the same id
                   "build_id": (open(os.path.join(SRC, "build_id.txt")).read().strip()
                                if os.path.exists(os.path.join(SRC, "build_id.txt")) else None),
                   "hbm_regime": {"free_1000001": traffic.get("hbm1"), "free_4000001": traffic.get("hbm4"), "peak_1000001": traffic.get("pkhb"), "peak_4000001": traffic.get("pkh4")},
                   "other": {"peak_two_axes": traffic.get("pk2"), "peak_bench_param": traffic.get("peak"), "edge": traffic.get("edge"), "free_packed_ntau22": traffic.get("pack")}},
                  open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=2)


if __name__ == "__main__":
    main()
