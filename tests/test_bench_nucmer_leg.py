"""CPU-only: bench.py's "reference" CPU-baseline leg (nucmer --mum + delta-filter -1 under one Pool of all cores, the reference's
runner: pyani/run_multiprocessing.py:130-144, pyani/anim.py:240-289) driven with STUB executables on PATH.

MUMmer is not installed in this image or on the GPU box, so the leg never ran (What's weak #11): the day nucmer appears
its first run must not also be its first test.  The stubs only check the command lines pyani would issue and leave the files pyani
expects; nothing here measures anything."""
import importlib.util
import os
import stat
import sys

import numpy as np

from tests.conftest import ROOT

NUCMER_STUB = """#!/bin/sh
# stub of `nucmer --mum -p PREFIX REF QRY` (pyani/anim.py:262-279)
[ "$1" = "--mum" ] && [ "$2" = "-p" ] || { echo "unexpected nucmer arguments: $*" >&2; exit 2; }
[ -s "$4" ] && [ -s "$5" ] || { echo "missing FASTA input" >&2; exit 3; }
printf '%s %s\\nNUCMER\\n' "$4" "$5" > "$3.delta"
echo "$4 $5" >> "$STUB_LOG"
"""
FILTER_STUB = """#!/bin/sh
# stub of `delta-filter -1 PREFIX.delta > PREFIX.filter` (pyani/anim.py:280-288)
[ "$1" = "-1" ] && [ -s "$2" ] || { echo "unexpected delta-filter arguments: $*" >&2; exit 2; }
cat "$2"
"""


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_reference_baseline_leg_runs_pyanis_command_lines_under_one_pool(tmp_path, monkeypatch):
    bench = _bench_module()
    from pyani_amd import synth
    for name, text in (("nucmer", NUCMER_STUB), ("delta-filter", FILTER_STUB)):
        p = tmp_path / name
        p.write_text(text)
        p.chmod(p.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    log = tmp_path / "calls.log"
    monkeypatch.setenv("PATH", f"{tmp_path}{os.pathsep}{os.environ['PATH']}")
    monkeypatch.setenv("STUB_LOG", str(log))
    data = [synth.genome(7, 4, g, 3000) for g in range(4)]
    sample = [(0, 1), (1, 0), (0, 2), (3, 1)]          # two "related", two "unrelated"
    rec = bench._nucmer_baseline(sample, 2, data, n_rel_job=10, n_unrel_job=90, threads=2)
    assert rec["kind"] == "reference" and rec["cores"] == 2 and rec["unit"] == "genome-pairs/s"
    assert rec["value"] > 0 and np.isfinite(rec["value"])
    # value = job pairs / (extrapolated CPU seconds / cores), from the per-kind means of the stubs' wall times
    assert abs(rec["value"] - 100 / rec["job_seconds_extrapolated"]) < 1e-6 * rec["value"]
    assert abs(rec["job_seconds_extrapolated"] * 2 - rec["cpu_seconds_extrapolated"]) < 1e-9
    calls = log.read_text().splitlines()
    assert len(calls) == len(sample)
    names = sorted(tuple(os.path.basename(x).rsplit(".", 1)[0] for x in c.split()) for c in calls)
    assert names == sorted((synth.genome_name(q), synth.genome_name(s)) for q, s in sample)      # reference first, query second


def test_baseline_leg_is_chosen_only_when_both_executables_exist(tmp_path, monkeypatch):
    """anim_cpu_baseline takes the reference leg when `nucmer` AND `delta-filter` are on PATH (bench.py); with one missing it must
    not try."""
    import shutil
    p = tmp_path / "nucmer"
    p.write_text(NUCMER_STUB)
    p.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path))
    assert shutil.which("nucmer") and not shutil.which("delta-filter")
