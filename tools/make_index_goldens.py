#!/usr/bin/env python3
"""Golden vectors for the `pyani index` path: the MD5 hashes, class and label lines the reference's
tests hold for the six Blochmannia genomes (tests/test_targets/subcmd_index/) -> tests/golden/ref_targets/subcmd_index.json.
Data only (hashes and description strings); run in the build container where /root/reference exists."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/tests/test_targets/subcmd_index")
out = {"md5": {}, "labels": [], "classes": []}
for f in sorted(REF.glob("*.md5")):
    out["md5"][f.name[:-len(".md5")]] = f.read_text().split()[0]
# the genome copies this repo holds (tests/golden/genomes/blochmannia, made from tests/fixtures/legacy/ANI_input) are
# byte-identical to the index test's inputs (tests/test_input/subcmd_index) only for some genomes: name those
import hashlib
legacy = Path("/root/reference/tests/fixtures/legacy/ANI_input")
out["same_bytes_as_golden_genome"] = sorted(n for n, h in out["md5"].items()
                                            if (legacy / n).is_file() and hashlib.md5((legacy / n).read_bytes()).hexdigest() == h)
out["labels"] = sorted(line for line in (REF / "labels.txt").read_text().splitlines() if line)
out["classes"] = sorted(line for line in (REF / "classes.txt").read_text().splitlines() if line)
dst = ROOT / "tests/golden/ref_targets/subcmd_index.json"
dst.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
print("wrote", dst, len(out["md5"]), "hashes")
