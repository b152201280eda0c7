"""Build the CPU checkers under oracle/ (test / measurement infrastructure — never loaded by pyani_amd/).

  liboracle.so    oracle/tetra_oracle.c   gcc -ffp-contract=off     TETRA restatement (tetra.py:78-194)
  libanimcpu.so   oracle/anim_cpu.cpp     g++ -pthread              host statement of the ANIm pair search (own-cpu baseline)
  libanibcpu.so   oracle/anib_cpu.cpp     g++ -pthread              host statement of fragment mode (ANIb)
  _build/nucmer_oracle   oracle/nucmer_oracle.cpp   g++               restatement of MUMmer 3.23's nucmer pipeline (the ANIm search oracle)
  libblastnoracle.so     oracle/blastn_oracle.cpp   g++ -pthread      restatement of BLAST+'s blastn for pyani's ANIb command line (the ANIb search oracle)
"""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
ORACLE_LIB = HERE / "liboracle.so"
ANIM_CPU_LIB = HERE / "libanimcpu.so"


# The CPU statements are also what bench.py times as the "own-cpu" baseline: full optimisation and the vector ISA both this container
# and the GPU box's host have (AVX2; no -march=native: the library is built here and travels), no fast-math (results are compared).
CPU_OPT = ["-O3", "-mavx2", "-mbmi2", "-ffp-contract=off"]


def _newer(target: Path, sources) -> bool:
    return target.exists() and all(Path(s).stat().st_mtime <= target.stat().st_mtime for s in sources)


def _run(cmd):
    print("+", " ".join(str(c) for c in cmd), file=sys.stderr, flush=True)
    subprocess.run([str(c) for c in cmd], check=True)


def build_oracle(force=False) -> Path:
    srcs = sorted(HERE.glob("*.c"))
    if force or not _newer(ORACLE_LIB, srcs):
        _run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-o", ORACLE_LIB, *srcs, "-lm"])
    return ORACLE_LIB


def build_anim_cpu(force=False) -> Path:
    core = ROOT / "pyani_amd" / "csrc" / "pg_anim_core.h"
    src = HERE / "anim_cpu.cpp"
    if force or not _newer(ANIM_CPU_LIB, [src, core, core.with_name("pg_nucmer_core.h")]):
        _run(["g++", *CPU_OPT, "-std=c++17", "-pthread", "-fPIC", "-shared", f"-I{core.parent}", "-o", ANIM_CPU_LIB, src])
    return ANIM_CPU_LIB


ANIB_CPU_LIB = HERE / "libanibcpu.so"


def build_anib_cpu(force=False) -> Path:
    csrc = ROOT / "pyani_amd" / "csrc"
    src = HERE / "anib_cpu.cpp"
    if force or not _newer(ANIB_CPU_LIB, [src, csrc / "pg_anib_core.h", csrc / "pg_anim_core.h"]):
        _run(["g++", *CPU_OPT, "-std=c++17", "-pthread", "-fPIC", "-shared", f"-I{csrc}", "-o", ANIB_CPU_LIB, src])
    return ANIB_CPU_LIB


NUCMER_ORACLE = HERE / "_build" / "nucmer_oracle"


def build_nucmer_oracle(force=False) -> Path:
    """oracle/nucmer_oracle.cpp -> oracle/_build/nucmer_oracle (a command-line program: ref.fna qry.fna [--maxmatch] [--delta])."""
    src = HERE / "nucmer_oracle.cpp"
    if force or not _newer(NUCMER_ORACLE, [src]):
        NUCMER_ORACLE.parent.mkdir(exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-o", NUCMER_ORACLE, src])
    return NUCMER_ORACLE


BLASTN_ORACLE_LIB = HERE / "libblastnoracle.so"


def build_blastn_oracle(force=False) -> Path:
    """oracle/blastn_oracle.cpp -> oracle/libblastnoracle.so: the independent restatement of blastn for pyani's ANIb command line
    (includes nothing from pyani_amd/csrc)."""
    src = HERE / "blastn_oracle.cpp"
    if force or not _newer(BLASTN_ORACLE_LIB, [src]):
        _run(["g++", "-O2", "-std=c++17", "-pthread", "-fPIC", "-shared", "-Wall", "-o", BLASTN_ORACLE_LIB, src])
    return BLASTN_ORACLE_LIB


def build_all(force=False):
    return (build_oracle(force), build_anim_cpu(force), build_anib_cpu(force), build_nucmer_oracle(force),
            build_blastn_oracle(force))


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
