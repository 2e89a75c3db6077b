"""Copies DATA of the reference's committed sac_edge.jl runs into tests/golden/edge/<name>/:
t.in, the eight parameter rows of in_edge.in, the "Edge:" lines of log.txt (known answers of the
deterministic initial scan, edge:553-556), anneal.csv and sample.csv (statistical oracle).
The two example_double_* runs are left out: their logged endpoints come from an older init_config_jk!."""
import os
import shutil

SRC = "/root/reference/sac/edge/out_files"
RUNS = {
    "hchain_single": "hchain_beta2048_single/Nw80/Ac_0.000/p_0.500",
    "t1_single": "t1_single/Nw80/Ac_0.000/p_0.500",
    "t2_symm": "t2_double_symm/Nw80/Ac_0.000/p_0.500",
    "t3_in": "t3_double_in/Nw80/Ac_0.000/p_0.500/Ar_0.500",
    "t4_out": "t4_double_out/Nw80/Ac_0.000/p_0.500/Ar_0.700",
    "tj_in_05": "1D_tJ/04_pi2_double_in/Nw80/Ac_0.000/p_0.500/Ar_0.500",
    "tj_in_07": "1D_tJ/04_pi2_double_in/Nw80/Ac_0.000/p_0.500/Ar_0.700",
}


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "edge")
    for name, rel in RUNS.items():
        out = os.path.join(root, name)
        os.makedirs(out, exist_ok=True)
        src = os.path.join(SRC, rel)
        for f in ("t.in", "anneal.csv", "sample.csv"):
            shutil.copyfile(os.path.join(src, f), os.path.join(out, f))
        with open(os.path.join(src, "in_edge.in")) as f:
            rows = f.readlines()[:8]
        with open(os.path.join(out, "in_edge.in"), "w") as f:
            f.writelines(rows)
        with open(os.path.join(src, "log.txt")) as f:
            edges = [ln.split(" - ", 1)[1] for ln in f if "Edge:" in ln]
        with open(os.path.join(out, "init_edges.txt"), "w") as f:
            f.writelines(edges)
        if name == "hchain_single":          # the run's final spectra (statistical oracle for write_spec, edge:1514-1637)
            for f in ("sw000_1.dat", "dw000_1.dat"):
                shutil.copyfile(os.path.join(src, f), os.path.join(out, f))
        os.system(f"chmod -R u+w {out}")


if __name__ == "__main__":
    main()
