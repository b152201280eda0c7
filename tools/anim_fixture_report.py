#!/usr/bin/env python3
"""Run the GPU ANIm engine on every genome pair for which the reference's tests hold real MUMmer output AND both FASTA
files (15 Blochmannia pairs + NC_002696<->NC_011916) and print engine vs fixture side by side (parse_delta tuples)."""
import gzip
import json
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_amd.engine import Engine  # noqa: E402

gold = json.loads((ROOT / "tests/golden/anim_goldens.json").read_text())["parse_delta"]
tmp = Path(tempfile.mkdtemp())
paths = {}
for grp in ("blochmannia", "caulobacter"):
    for gz in sorted((ROOT / "tests/golden/genomes" / grp).glob("*.fna.gz")):
        dst = tmp / gz.name[:-3]
        with gzip.open(gz, "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        paths[dst.stem] = dst
eng = Engine(0)
ids = {stem: eng.add_fasta(p)[0] for stem, p in paths.items()}
only = [a for a in sys.argv[2:]]
pairs = []
for key in sorted(gold):
    if only and not any(o in key for o in only):
        continue
    if key.endswith(".filter") and "/" in key:
        a, b = key.split("/")[1][:-7].split("_vs_")
        if a in ids and b in ids:
            pairs.append((key, a, b))
t0 = time.time()
res = eng.anim_pairs([ids[a] for _, a, _ in pairs], [ids[b] for _, _, b in pairs])
dt = time.time() - t0
rows = []
for (key, a, b), r in zip(pairs, res):
    g = gold[key]
    rows.append(dict(pair=f"{a[:13]}|{b[:13]}", status=int(r["status"]), ref_aln=int(r["ref_aln_len"]), fix_ref_aln=g[0],
                     qry_aln=int(r["qry_aln_len"]), fix_qry_aln=g[1], identity=float(r["identity"]), fix_identity=g[2],
                     errs=int(r["sim_errors"]), fix_errs=g[3], n=int(r["n_alignments"])))
    x = rows[-1]
    print(f"{x['pair']:28s} st={x['status']} id {x['identity']:.6f} vs {x['fix_identity']:.6f} (d={x['identity']-x['fix_identity']:+.1e})  "
          f"ref_aln {x['ref_aln']} vs {x['fix_ref_aln']} ({100*(x['ref_aln']-x['fix_ref_aln'])/x['fix_ref_aln']:+.2f}%)  errs {x['errs']} vs {x['fix_errs']}  n={x['n']}")
print(f"{len(pairs)} ordered pairs in {dt:.2f} s")
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(rows, indent=1))
