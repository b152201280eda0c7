"""Row-level comparison of the PRODUCT's fragment search with the independent blastn oracle and with the BLAST+ tables.

  python tools/anib_product_vs_oracle.py [--gpu] [--pairs Q_vs_S ...] [--out profiles/r06_anib_product_vs_blastn_restatement.json]

Product rows: with --gpu from pg_anib_pair_rows (the HIP path, through the C ABI); without it from oracle/anib_cpu.cpp — the product's
own header compiled for the host, which the GPU tests hold equal to the kernels row for row (so the numbers are the GPU's; the host
build lets the table be made in the CPU-only container).  Oracle rows: oracle/blastn_oracle.cpp.  BLAST+ rows: tests/golden/anib/.
Per pair: the rows parse_blast_tab USES (first row of a fragment with coverage > 70 % and identity > 30 %, pyani/anib.py:641-649)
identical on both sides / present on one side only, and parse_blast_tab's tuple on all three sides."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))

import blastn_oracle  # noqa: E402
import blastn_oracle_agreement as A  # noqa: E402


def tuples(a):
    return [(int(r["frag"]), int(r["length"]), int(r["mismatch"]), int(r["gaps"]), int(r["qstart"]), int(r["qend"]), int(r["sstart"]),
             int(r["send"]), int(r["srec"]), int(r["qlen"]), float("%.3f" % (100.0 * int(r["nident"]) / int(r["length"])))) for r in a]


def side_by_side(up, uo):
    same = sum(1 for k in uo if k in up and up[k][:9] == uo[k][:9])
    pm1 = sum(1 for k in uo if k in up and up[k][:9] != uo[k][:9] and up[k][1] == uo[k][1] and up[k][3:9] == uo[k][3:9])
    rp, ro = A.reduce_used(up), A.reduce_used(uo)
    return {"used_rows_product": len(up), "used_rows_other": len(uo), "identical": same, "identical_fraction": same / max(1, len(uo)),
            "same_extent_and_gaps_mismatch_count_differs": pm1, "only_product": len(set(up) - set(uo)), "only_other": len(set(uo) - set(up)),
            "tuple_product": rp, "tuple_other": ro, "identity_pp_diff": rp[2] - ro[2], "aln_length_rel_diff": (rp[0] - ro[0]) / max(1, ro[0]),
            "sim_errors_rel_diff": (rp[1] - ro[1]) / max(1, ro[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--pairs", nargs="*")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    stems = ["NC_002696", "NC_010338", "NC_011916", "NC_014100"]
    pairs = a.pairs or [f"{q}_vs_{s}" for q in stems for s in stems if q != s]
    gdir = A.GOLD / "genomes" / "caulobacter"
    G = {s: A.read_fasta_gz(gdir / f"{s}.fna.gz") for s in stems}
    eng = ids = None
    if a.gpu:
        from pyani_amd.engine import Engine
        eng = Engine(0)
        ids = {s: eng.add_genome(*G[s]) for s in stems}
    else:
        import anib_cpu
    report = {}
    for p in pairs:
        q, s = p.split("_vs_")
        prod = tuples(eng.anib_pair_rows(ids[q], ids[s]) if a.gpu else anib_cpu.anib_cpu_pair(G[q], G[s]))
        orac = tuples(blastn_oracle.blastn_pair(G[q], G[s]))
        blast = A.blast_rows(A.GOLD / "anib" / f"{p}.blast_tab.gz", A.record_names(gdir / f"{s}.fna.gz"))
        up, uo, ub = A.used_rows(prod), A.used_rows(orac), A.used_rows(blast)
        report[p] = {"product_rows_from": "GPU (pg_anib_pair_rows)" if a.gpu else "host build of the product's header (oracle/anib_cpu.cpp)",
                     "vs_oracle": side_by_side(up, uo), "vs_blast_plus": side_by_side(up, ub)}
        print(p, json.dumps(report[p]), flush=True)
    if a.out:
        Path(a.out).write_text(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
