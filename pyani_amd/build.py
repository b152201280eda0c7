"""Build the native pieces in-tree (no cmake, no JIT cache): explicit hipcc / g++ / gcc command lines.

  libpyani_gpu.so   pyani_amd/csrc/*.hip + *.cpp   hipcc --offload-arch=gfx950   (the product)
  libpgsynth.so     pyani_amd/csrc/synth.cpp       g++                            (synthetic test/bench data)

hipcc cross-compiles gfx950 without a GPU, so both build in the CPU-only container.  (The CPU checkers under oracle/ have
their own recipe, oracle/oracle_build.py: nothing in this package builds, loads or calls them.)
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "pyani_amd" / "csrc"
GPU_LIB = ROOT / "pyani_amd" / "libpyani_gpu.so"
SYNTH_LIB = ROOT / "pyani_amd" / "libpgsynth.so"


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def _run(cmd):
    print("+", " ".join(str(c) for c in cmd), file=sys.stderr, flush=True)
    subprocess.run([str(c) for c in cmd], check=True)


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libpyani_gpu.so)")


def build_gpu(force=False):
    srcs = sorted(CSRC.glob("pg_*.hip")) + sorted(CSRC.glob("pg_*.cpp"))
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + sorted((ROOT / "include").glob("*.h"))
    if not force and _newer(GPU_LIB, deps):
        return GPU_LIB
    _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
          "-Wall", "-Wno-unused-result", f"-I{ROOT / 'include'}", f"-I{CSRC}", "-o", GPU_LIB, *srcs, "-lpthread"])
    return GPU_LIB


def build_synth(force=False):
    src = CSRC / "synth.cpp"
    if not force and _newer(SYNTH_LIB, [src]):
        return SYNTH_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", SYNTH_LIB, src])
    return SYNTH_LIB


def build_all(force=False):
    return build_gpu(force), build_synth(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
