"""Layout of every struct that crosses the C ABI, three ways: (1) include/sac_hip.h compiled by gcc into a probe that prints
sizeof / offsetof, (2) the ctypes / numpy mirrors of sac_amd/lib.py (what the GPU tests call through), (3) the Julia mirrors of
julia/SacHip.jl (what a maintainer's `ccall`s use), whose field order and types are parsed from the source and laid out by the
C rules Julia follows for isbits structs.  Julia is not installed in the build image, so (3) is the one mechanical check the
Julia driver files get; it also checks that every `ccall` in julia/*.jl names a symbol the header declares."""
import ctypes
import os
import re
import subprocess

import numpy as np

from sac_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sac_hip.h")

C_SIZES = {"int32_t": 4, "uint32_t": 4, "int64_t": 8, "uint64_t": 8, "double": 8}
JL_TYPES = {"Int32": ("int32_t", 4), "UInt32": ("uint32_t", 4), "Int64": ("int64_t", 8), "UInt64": ("uint64_t", 8),
            "Float64": ("double", 8)}
PAIRS = {   # C struct: (Julia struct, ctypes class or numpy dtype)
    "sac_config_t": ("SacConfig", lib.Config), "sac_stats_t": ("SacStats", lib.Stats),
    "sac_peak_stats_t": ("SacPeakStats", lib.PeakStats), "sac_move_rec_t": ("SacMoveRec", lib.MOVE_REC),
    "sac_table_spec_t": ("SacTableSpec", lib.TableSpec), "sac_proposal_t": ("SacProposal", lib.PROPOSAL),
    "sac_peak_proposal_t": ("SacPeakProposal", lib.PEAK_PROPOSAL), "sac_edge_config_t": ("SacEdgeConfig", lib.EdgeConfig),
    "sac_edge_results_t": ("SacEdgeResults", lib.EdgeResults), "sac_edge_proposal_t": ("SacEdgeProposal", lib.EDGE_PROPOSAL),
}


def header_structs():
    """{struct: [(field, ctype, element count)]} parsed from the header, declaration order."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, rest = decl.split(None, 1)
            for d in rest.split(","):
                m = re.fullmatch(r"\s*(\w+)((?:\[\d+\])*)\s*", d)
                count = int(np.prod([int(x) for x in re.findall(r"\[(\d+)\]", m.group(2))])) if m.group(2) else 1
                fields.append((m.group(1), ctype, count))
        out[name] = fields
    return out


def c_probe(structs, tmp_path):
    """sizeof / offsetof of every field as gcc lays the header out."""
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for s, fields in structs.items():
        src.append(f'  printf("{s} sizeof %zu\\n", sizeof({s}));')
        for f, _, _ in fields:
            src.append(f'  printf("{s} {f} %zu %zu\\n", offsetof({s}, {f}), sizeof((({s} *)0)->{f}));')
    src += ["  return 0;", "}"]
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(c)])
    sizes, offs = {}, {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        p = ln.split()
        if p[1] == "sizeof":
            sizes[p[0]] = int(p[2])
        else:
            offs.setdefault(p[0], []).append((p[1], int(p[2]), int(p[3])))
    return sizes, offs


def julia_structs():
    """{struct: [(field, ctype, count)]} parsed from julia/SacHip.jl."""
    text = open(os.path.join(ROOT, "julia", "SacHip.jl")).read()
    out = {}
    for name, body in re.findall(r"^struct (\w+)[^\n]*\n(.*?)^end", text, flags=re.S | re.M):
        fields = []
        for ln in body.splitlines():
            ln = ln.split("#")[0].strip()
            if not ln:
                continue
            f, t = ln.split("::")
            m = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", t.strip())
            if m:
                fields.append((f.strip(), JL_TYPES[m.group(2)][0], int(m.group(1))))
            elif t.strip() in JL_TYPES:
                fields.append((f.strip(), JL_TYPES[t.strip()][0], 1))
            else:
                fields = None       # not an ABI struct (e.g. TimeData)
                break
        if fields:
            out[name] = fields
    return out


def c_layout(fields):
    """Offsets / sizes by the C rules (natural alignment, struct padded to its largest member)."""
    off, res, amax = 0, [], 1
    for f, t, n in fields:
        a = C_SIZES[t]
        amax = max(amax, a)
        off = (off + a - 1) // a * a
        res.append((f, off, a * n))
        off += a * n
    return (off + amax - 1) // amax * amax, res


def test_struct_layouts_agree_three_ways(tmp_path):
    structs = header_structs()
    assert set(PAIRS) <= set(structs), set(PAIRS) - set(structs)
    sizes, offs = c_probe({k: structs[k] for k in PAIRS}, tmp_path)
    jl = julia_structs()
    for cname, (jname, mirror) in PAIRS.items():
        want = offs[cname]
        # (2) the Python mirror
        if isinstance(mirror, np.dtype):
            assert mirror.itemsize == sizes[cname], cname
            got = [(f, mirror.fields[f][1], mirror.fields[f][0].itemsize) for f in mirror.names]
        else:
            assert ctypes.sizeof(mirror) == sizes[cname], cname
            got = [(f, getattr(mirror, f).offset, getattr(mirror, f).size) for f, _ in mirror._fields_]
        norm = lambda rows: [(f.lower(), o, s) for f, o, s in rows]      # noqa: E731  (EdgeResults calls amp_accept_rates the same)
        assert norm(got) == norm(want), (cname, got, want)
        # (3) the Julia mirror: same field names in the same order, same element types and counts, hence the same layout
        assert jname in jl, f"julia/SacHip.jl lacks struct {jname}"
        jf = jl[jname]
        assert [(f, t, n) for f, t, n in jf] == [(f, t, n) for f, t, n in structs[cname]], (jname, jf, structs[cname])
        size, rows = c_layout(jf)
        assert size == sizes[cname] and rows == want, (jname, rows, want)


def test_every_julia_ccall_names_a_declared_symbol():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(sac_[a-z_0-9]+)\s*\(", text))
    used = set()
    jdir = os.path.join(ROOT, "julia")
    for fn in sorted(os.listdir(jdir)):
        if fn.endswith(".jl"):
            used |= set(re.findall(r"ccall\(\(:(\w+),", open(os.path.join(jdir, fn)).read()))
    assert used, "no ccall found in julia/*.jl"
    assert used <= declared, used - declared
    # the binding module covers the whole header
    assert declared <= used, declared - used


def test_julia_drivers_exist_with_the_reference_entry_points():
    """The four driver files the north star names: same program names and the reference's stage / schedule function names."""
    want = {"sac_free_hip.jl": ("function run(", "adjust_windows", "sample", "fast_anneal", "main_anneal", "final_anneal", "write_spec"),
            "sac_peak_hip.jl": ("function run(", "adjust_windows", "sample", "intial_sampling", "run_anneal", "final_anneal", "write_spec"),
            "sac_edge_hip.jl": ("function run(", "run_bins", "anneal(", "anneal_and_sample", "anneal_and_scan", "write_res", "write_spec"),
            "cross_val_hip.jl": ("free_cv", "peak_cv", "edge_cv", "GSAC.csv")}
    for fn, needles in want.items():
        src = open(os.path.join(ROOT, "julia", fn)).read()
        for n in needles:
            assert n in src, f"{fn}: {n} missing"
        # balanced block structure (function/for/if/while/do/begin/let/struct/module ... end), the cheapest syntax check
        code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"#[^\n]*", "", src))
        opens = len(re.findall(r"(?<![\w.!])(?:function|for|if|while|do|begin|let|struct|module|try)(?![\w!])", code))
        opens -= len(re.findall(r"(?<![\w.!])mutable struct(?![\w!])", code)) * 0
        ends = len(re.findall(r"(?<![\w.!:\[])end(?![\w!])", code))
        assert opens == ends, f"{fn}: {opens} block openers, {ends} `end`s"
