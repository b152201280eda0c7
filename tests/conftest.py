"""pytest configuration: markers + shared fixtures for the parity tests."""
import gzip
import json
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library's development knobs (PYANI_ANIM_*, PYANI_EXT_*: alternative launch shapes that must give the same results) are
    # honoured only under this switch (pg_internal.h, pg_dev_env); the tests that use them run in this process and its children
    import os
    os.environ.setdefault("PYANI_DEV_KNOBS", "1")


@pytest.fixture(scope="session")
def goldens():
    with open(GOLD / "tetra_goldens.json") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def genome_dir(tmp_path_factory):
    """All golden FASTA inputs, decompressed once per session: {group: {stem: Path}}."""
    base = tmp_path_factory.mktemp("genomes")
    out = {}
    for grp_dir in sorted((GOLD / "genomes").iterdir()):
        d = base / grp_dir.name
        d.mkdir()
        out[grp_dir.name] = {}
        for gz in sorted(grp_dir.glob("*.fna.gz")):
            dst = d / gz.name[:-3]
            with gzip.open(gz, "rb") as fi, open(dst, "wb") as fo:
                shutil.copyfileobj(fi, fo)
            out[grp_dir.name][dst.stem] = dst
    out["edge"] = {p.stem: p for p in sorted((GOLD / "edge").glob("*.fna"))}
    return out


@pytest.fixture(scope="session")
def synth_ci_dir(tmp_path_factory):
    from pyani_amd import synth
    cfg = synth.SETS["CI"]
    d = tmp_path_factory.mktemp("synthCI")
    paths = synth.write_set(d, cfg["seed"], cfg["n"], cfg["L"])
    return {p.stem: p for p in paths}


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_bind
    return oracle_bind.load()
