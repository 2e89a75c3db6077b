"""Command line twin of the reference's program entries (free:1264-1268, peak:1571-1580, edge:2068-2075,
cross_val.jl:514-531): run from a directory holding in_free.in / in_peak.in / in_edge.in.

    python -m sac_amd free  [--chains N] [--seed S]
    python -m sac_amd peak  [A_0 N_p] [--chains N]
    python -m sac_amd edge  [A_c A_r p] [--chains N]
    python -m sac_amd cv free|peak|edge REP a|b [--chains N]
"""
import argparse
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m sac_amd")
    ap.add_argument("program", choices=["free", "peak", "edge", "cv"])
    ap.add_argument("args", nargs="*")
    ap.add_argument("--chains", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    kw = dict(n_chains=a.chains, seed=a.seed, device=a.device)
    if a.program == "free":
        from .free import FreeSAC
        return FreeSAC.from_infile("in_free.in", **kw).run("in_free.in")
    if a.program == "peak":
        from .peak import PeakSAC
        A_0 = float(a.args[0]) if len(a.args) > 0 else None
        N_p = int(a.args[1]) if len(a.args) > 1 else None
        return PeakSAC.from_infile("in_peak.in", A_0_in=A_0, N_p_in=N_p, **kw).run("in_peak.in")
    if a.program == "edge":
        from .edge import EdgeSAC
        v = [float(x) for x in a.args[:3]] + [None] * (3 - len(a.args[:3]))
        return EdgeSAC.from_infile("in_edge.in", A_c_in=v[0], A_r_in=v[1], p_in=v[2], **kw).run("in_edge.in")
    from .cross_val import edge_cv, free_cv, peak_cv
    if len(a.args) < 3:
        ap.error("cv needs: free|peak|edge REP a|b")
    fn = {"free": free_cv, "peak": peak_cv, "edge": edge_cv}[a.args[0]]
    return fn(int(a.args[1]), a.args[2], **kw)


if __name__ == "__main__":
    main()
    sys.exit(0)
