"""bench.py and __graft_entry__.py are run by the driver, not by the other tests: they must at least compile cleanly (a
SyntaxWarning such as "'str' object is not callable" is an error waiting for the first run) and keep the contract's flags."""
import os
import subprocess
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_entry_files_compile_without_warnings():
    for name in ("bench.py", "__graft_entry__.py", "tools/bench_param.py", "tools/time_to_solution.py", "tools/pmc_report.py"):
        with open(os.path.join(ROOT, name)) as f:
            src = f.read()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            compile(src, name, "exec")


def test_bench_flags_of_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, text=True, check=True).stdout
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out
