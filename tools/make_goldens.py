#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):
  * imports /root/reference/pyani/tetra.py unmodified via importlib, with tools/bio_shim standing in for
    Biopython's FASTA reader (the shim holds no hot-path arithmetic);
  * records the reference's internal k-mer count dicts by handing it a recording ``collections`` namespace;
  * writes inputs (gzipped copies of the data files the reference's own tests hold, plus small hand-made
    edge-case FASTA files and the repo's seeded synthetic genomes) and expected outputs (integer counts,
    Z-scores and correlation matrices as C99 hex floats => exact) as JSON fixtures.

Fixtures are DATA only — no reference source text is written anywhere.

Usage: python tools/make_goldens.py [--skip-large]
"""
import argparse
import gzip
import importlib.util
import json
import random
import shutil
import sys
import tarfile
import types
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "bio_shim"))

KMERS = {k: ["".join(p) for p in __import__("itertools").product("ACGT", repeat=k)] for k in (2, 3, 4)}


def load_reference_tetra():
    spec = importlib.util.spec_from_file_location("ref_tetra", REF / "pyani" / "tetra.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # recording stand-in for the `collections` name inside the reference module
    created = []

    def recording_defaultdict(factory):
        d = defaultdict(factory)
        created.append(d)
        return d

    mod.collections = types.SimpleNamespace(defaultdict=recording_defaultdict)
    return mod, created


def gz_copy(src: Path, dst: Path):
    dst.parent.mkdir(parents=True, exist_ok=True)
    with open(src, "rb") as fi, open(dst, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as fo:
            shutil.copyfileobj(fi, fo)


def gunzip_to(src: Path, dst: Path):
    with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)


def run_reference(ref, created, fasta: Path):
    del created[:]
    z = ref.calculate_tetra_zscore(fasta)
    mono, di, tri, tet = created[:4]
    return {
        "c2": [int(di.get(k, 0)) for k in KMERS[2]],
        "c3": [int(tri.get(k, 0)) for k in KMERS[3]],
        "c4": [int(tet.get(k, 0)) for k in KMERS[4]],
        "order": list(z.keys()),
        "z": {k: float(v).hex() for k, v in z.items()},
    }, z


def write_edge_cases(edge: Path):
    edge.mkdir(parents=True, exist_ok=True)
    rnd = random.Random(20250228)

    def rs(n, alphabet="ACGT"):
        return "".join(rnd.choice(alphabet) for _ in range(n))

    def fasta(path, recs, width=60):
        with open(path, "w") as fh:
            for title, s in recs:
                fh.write(f">{title}\n")
                for i in range(0, len(s), width):
                    fh.write(s[i:i + width] + "\n")

    fasta(edge / "e01_tiny_records.fna",
          [("r_empty", ""), ("r1", "A"), ("r2", "CG"), ("r3", "TGA"), ("r4", "ACGT"), ("r5", "GATTA"),
           ("r6", "CAAGT"), ("r_long", rs(3000))])
    body = rs(4000)
    mixed = "".join(c.lower() if rnd.random() < 0.4 else c for c in body)
    amb = list(mixed)
    for _ in range(60):
        amb[rnd.randrange(len(amb))] = rnd.choice("NRYKMSWBDHVnryk-*")
    fasta(edge / "e02_lower_iupac.fna", [("mixedcase iupac", "".join(amb)), ("second", rs(2500).lower())], width=61)
    fasta(edge / "e03_n_runs.fna",
          [("allN", "N" * 137), ("runs", rs(900) + "N" * 50 + rs(3) + "N" + rs(2) + "NN" + rs(1) + "N" + rs(2200)),
           ("edgeN", "N" + rs(1500) + "N")])
    # two genomes over a 3-letter alphabet: identical (incomplete) key sets -> correlation over < 256 keys
    fasta(edge / "e04_acg_only_a.fna", [("acg_a", rs(6000, "ACG"))])
    fasta(edge / "e05_acg_only_b.fna", [("acg_b1", rs(3500, "ACG")), ("acg_b2", rs(3100, "ACG"))])
    # homopolymer + the sd == 0 branch (CAAGT alone: exp=1, sd=0 -> z = 1/(den*den))
    fasta(edge / "e06_sd_zero.fna", [("only", "CAAGT")])
    fasta(edge / "e07_homopolymer.fna", [("polyA", "A" * 333), ("polyGC", "GC" * 200)])
    # whitespace inside sequence lines, CRLF, blank lines, leading junk before the first header
    with open(edge / "e08_whitespace.fna", "w", newline="") as fh:
        fh.write("junk before header is ignored\n")
        s = rs(2600)
        fh.write(">ws record one\r\n")
        for i in range(0, 1300, 50):
            fh.write(s[i:i + 25] + " " + s[i + 25:i + 50] + "\r\n")
        fh.write("\n>ws2\n" + s[1300:] + "\n\n")


def corr_json(ref, zs):
    df = ref.calculate_correlations(zs)
    labels = list(df.index)
    return {"labels": labels, "matrix": [[float(df.loc[a, b]).hex() for b in labels] for a in labels]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true", help="skip the 2.4-4 Mb genomes (faster)")
    args = ap.parse_args()
    ref, created = load_reference_tetra()
    tmp = ROOT / "gpurun_out" / "_goldens_tmp"
    tmp.mkdir(parents=True, exist_ok=True)
    out = {}

    # ---- (1) data files held by the reference's own tests ------------------------------------------------
    recover_caulobacter_genomes()
    recovered = []
    for stem in ("NC_010338", "NC_014100"):
        gunzip_to(GOLD / "genomes" / "caulobacter" / f"{stem}.fna.gz", tmp / f"{stem}.fna")
        recovered.append(tmp / f"{stem}.fna")
    groups = {
        "caulobacter": [REF / "tests/fixtures/sequences/NC_002696.fna", REF / "tests/fixtures/sequences/NC_011916.fna"] + recovered,
        "blochmannia": sorted((REF / "tests/fixtures/legacy/ANI_input").glob("*.fna")),
        "concordance": sorted((REF / "tests/fixtures/concordance").glob("*.fna")),
    }
    if args.skip_large:
        groups.pop("caulobacter"), groups.pop("concordance")
    for grp, files in groups.items():
        zs = {}
        for f in files:
            if f not in recovered:
                gz_copy(f, GOLD / "genomes" / grp / (f.name + ".gz"))
            rec, z = run_reference(ref, created, f)
            out[f"{grp}/{f.stem}"] = rec
            zs[f.stem] = z
            print("ref tetra", grp, f.stem, len(z), flush=True)
        out[f"{grp}/__corr__"] = corr_json(ref, zs)

    # reference's own committed targets for this path (data)
    tgt = GOLD / "ref_targets"
    tgt.mkdir(parents=True, exist_ok=True)
    shutil.copyfile(REF / "tests/fixtures/targets/tetra/zscore.json", tgt / "tetra_zscore_NC_002696.json")
    shutil.copyfile(REF / "tests/fixtures/targets/tetra/correlation.tab", tgt / "tetra_correlation_2x2.tab")
    shutil.copyfile(REF / "tests/target_TETRA_output/TETRA_correlations.tab", tgt / "TETRA_correlations_caulobacter_4x4.tab")
    shutil.copyfile(REF / "tests/test_targets/legacy_scripts/TETRA_mpl/TETRA_correlations.tab",
                    tgt / "TETRA_correlations_blochmannia_6x6.tab")
    shutil.copyfile(REF / "tests/fixtures/concordance/jspecies_output.tab", tgt / "jspecies_output.tab")

    # ---- (2) hand-made edge cases ------------------------------------------------------------------------
    edge = GOLD / "edge"
    write_edge_cases(edge)
    zs_acg = {}
    for f in sorted(edge.glob("*.fna")):
        rec, z = run_reference(ref, created, f)
        out[f"edge/{f.stem}"] = rec
        if f.stem.startswith(("e04", "e05")):
            zs_acg[f.stem] = z
    out["edge/__corr_acg__"] = corr_json(ref, zs_acg)

    # ---- (3) the repo's seeded synthetic CI set (N=8, L=50 000) ------------------------------------------
    from pyani_amd import synth
    cfg = synth.SETS["CI"]
    zs = {}
    for g in range(cfg["n"]):
        seq, off = synth.genome(cfg["seed"], cfg["n"], g, cfg["L"])
        p = tmp / f"{synth.genome_name(g)}.fna"
        synth.write_fasta(p, seq, off, synth.genome_name(g))
        rec, z = run_reference(ref, created, p)
        out[f"synthCI/{p.stem}"] = rec
        zs[p.stem] = z
    out["synthCI/__corr__"] = corr_json(ref, zs)

    with open(GOLD / "tetra_goldens.json", "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote", GOLD / "tetra_goldens.json", len(out), "entries")




# ---------------------------------------------------------------------------------------------------------------
# ANIm: real MUMmer output held by the reference's tests (the only pin for the alignment search) + the reference's
# known answers for parse_delta.  Run with:  python tools/make_goldens.py --anim-only
def load_reference_anim():
    """The reference's own pyani/anim.py, imported as part of its package (tools/bio_shim supplies stand-ins for the two
    absent third-party imports: Bio.SeqIO and intervaltree — neither holds pyani arithmetic)."""
    sys.path.insert(0, str(REF))
    import pyani.anim as ref_anim
    return ref_anim


def decode_blastdb_nsq(nsq: Path, nin: Path):
    """One-sequence NCBI BLAST nucleotide database (formatdb v4) -> (sequence bytes, title).  The reference's JSpecies test
    data keeps NC_010338 / NC_014100 only in this form (tests/test_JSpecies/pyani_tests/*.nsq; their FASTA files are
    missing blobs).  .nin: big-endian version, dbtype, title, date, nseq, LITTLE-endian total length, max length, then the
    header / sequence / ambiguity offset arrays; .nsq: ncbi2na, 4 bases per byte, first base in the high bits."""
    import struct
    import numpy as np
    b = nin.read_bytes()
    o = 8
    tl, = struct.unpack(">i", b[o:o + 4]); o += 4 + tl
    dl, = struct.unpack(">i", b[o:o + 4]); o += 4 + dl
    nseq, = struct.unpack(">i", b[o:o + 4]); o += 4
    total, = struct.unpack("<q", b[o:o + 8]); o += 8 + 4
    n = nseq + 1
    seq_off = struct.unpack(f">{n}i", b[o + 4 * n:o + 8 * n])
    amb_off = struct.unpack(f">{n}i", b[o + 8 * n:o + 12 * n])
    assert nseq == 1 and amb_off[0] == seq_off[1], "expected one sequence without ambiguity runs"
    raw = np.fromfile(nsq, dtype=np.uint8)[seq_off[0]:seq_off[1]]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((len(raw), 4), dtype=np.uint8)
    for k in range(4):
        out[:, k] = lut[(raw >> (6 - 2 * k)) & 3]
    return out.reshape(-1)[:total].tobytes()


def fasta_body(path: Path) -> bytes:
    return b"".join(l.strip() for l in open(path, "rb") if not l.startswith(b">")).upper()


def gz_bytes(data: bytes, dst: Path):
    dst.parent.mkdir(parents=True, exist_ok=True)
    with open(dst, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as fo:
            fo.write(data)


def recover_caulobacter_genomes():
    """The two Caulobacter genomes whose FASTA files are missing blobs, recovered from JSpecies' BLAST databases; the decoder
    is validated on the two genomes that exist in both forms, and the lengths equal the .delta headers."""
    js = REF / "tests/test_JSpecies"
    pt = js / "pyani_tests"
    for stem in ("NC_002696", "NC_011916"):
        assert decode_blastdb_nsq(pt / f"{stem}.fna.nsq", pt / f"{stem}.fna.nin") == fasta_body(pt / f"{stem}.fna"), stem
    # single-record NC_002696 of the JSpecies runs == the two records of fixtures/anim/sequences joined
    assert fasta_body(pt / "NC_002696.fna") == fasta_body(REF / "tests/fixtures/anim/sequences/NC_002696.fna")
    titles = {"NC_010338": ("gi|167643973|ref|NC_010338.1|", "Caulobacter sp. K31 chromosome, complete genome", 5477872),
              "NC_014100": ("gi|295687459|ref|NC_014100.1|", "Caulobacter segnis ATCC 21756 chromosome, complete genome", 4655622)}
    for stem, (sid, title, length) in titles.items():
        seq = decode_blastdb_nsq(pt / f"{stem}.fna.nsq", pt / f"{stem}.fna.nin")
        assert len(seq) == length and title.encode() in (pt / f"{stem}.fna.nhr").read_bytes()
        lines = [f">{sid} {title}".encode()] + [seq[i:i + 70] for i in range(0, len(seq), 70)]
        gz_bytes(b"\n".join(lines) + b"\n", GOLD / "genomes" / "caulobacter" / f"{stem}.fna.gz")


def make_deltadir_cases(ref_anim, tmp):
    """process_deltadir of the REFERENCE (pyani/anim.py:415-497 + pyani_tools.ANIResults, imported) on (1) its own deltadir
    fixture and (2) a small synthetic directory whose stems are prefixes of one another ('g', 'g-2', 'g.1': the sorted-Path
    order differs from the sorted-string order there) and which has one-directional and empty-mirror cases.  Stored: the
    inputs of case 2 (tiny .filter texts) and all five matrices of both cases as hex floats."""
    import logging
    rng = random.Random(5)

    def matrices(res):
        out = {}
        for df, stem in res.data:
            out[stem] = {"labels": list(df.index), "rows": [[float(df.loc[a, b]).hex() for b in df.columns] for a in df.index]}
        return out

    cases = {}
    ddir = REF / "tests/fixtures/anim/deltadir"
    names = sorted(d.name for d in ddir.iterdir())
    lengths = {n_: 4_000_000 + 1000 * i for i, n_ in enumerate(names)}
    cases["caulobacter_deltadir"] = {"lengths": list(lengths.items()), "matrices": matrices(ref_anim.process_deltadir(ddir, lengths))}
    stems = ["g", "g-2", "g.1", "a", "zz"]
    lengths = {s: 5000 + 37 * i for i, s in enumerate(stems)}
    root = tmp / "deltadir_case"
    if root.exists():
        shutil.rmtree(root)
    files = {}
    for q in stems:
        for s in stems:
            if q == s or (q, s) in (("a", "zz"), ("g.1", "g")):       # two directions missing
                continue
            lines = [f"/x/{q}.fna /x/{s}.fna", "NUCMER", f">{q}_1 {s}_1 {lengths[q]} {lengths[s]}"]
            pos = 1
            for _ in range(rng.randint(1, 4)):
                ln = rng.randint(70, 600)
                qs = rng.randint(1, lengths[s] - 700)
                lines += [f"{pos} {pos + ln - 1} {qs} {qs + ln + rng.randint(-3, 3)} {rng.randint(0, 30)} 0 0", "0"]
                pos += ln - rng.randint(0, 40)
            files[f"{q}/{q}_vs_{s}.filter"] = "\n".join(lines) + "\n"
    for rel, text in files.items():
        (root / rel).parent.mkdir(parents=True, exist_ok=True)
        (root / rel).write_text(text)
    cases["prefix_stems"] = {"lengths": list(lengths.items()), "files": files, "matrices": matrices(ref_anim.process_deltadir(root, lengths))}
    (GOLD / "ref_targets" / "anim_process_deltadir_cases.json").write_text(json.dumps(cases, indent=0, sort_keys=True))
    logging.getLogger().info("process_deltadir cases written")


def make_anib_goldens():
    """The BLAST+ tables the reference's tests hold for the four Caulobacter genomes (tests/fixtures/anib/blastn): data files."""
    for f in sorted((REF / "tests/fixtures/anib/blastn").glob("*.blast_tab")):
        gz_copy(f, GOLD / "anib" / (f.name + ".gz"))
    shutil.copyfile(REF / "tests/fixtures/anib/dataframes/blastn_result.csv", GOLD / "ref_targets" / "anib_blastn_result.csv")


def make_anim_goldens():
    import tarfile
    sys.path.insert(0, str(ROOT / "oracle"))
    import anim_oracle
    ref_anim = load_reference_anim()
    out_dir = GOLD / "anim"
    tuples = {}
    tmp = ROOT / "gpurun_out" / "_goldens_tmp"
    tmp.mkdir(parents=True, exist_ok=True)

    def store(src_bytes, rel):
        """Fixture file + the tuple the REFERENCE's parse_delta (pyani/anim.py:292-411, imported) returns for it."""
        dst = out_dir / (rel + ".gz")
        gz_bytes(src_bytes, dst)
        plain = tmp / "cur.delta"
        plain.write_bytes(src_bytes)
        try:
            tuples[rel] = list(ref_anim.parse_delta(plain))
        except ZeroDivisionError:
            tuples[rel] = None
        mine = None
        try:
            mine = list(anim_oracle.parse_delta(dst))
        except ZeroDivisionError:
            pass
        assert mine == tuples[rel], (rel, mine, tuples[rel])     # the restatement equals the reference on every fixture

    for d in sorted((REF / "tests/fixtures/anim/deltadir").iterdir()):
        for f in sorted(d.glob("*_vs_*")):
            if f.suffix in (".delta", ".filter"):
                store(f.read_bytes(), f"caulobacter/{f.name}")
    store((REF / "tests/fixtures/anim/test.delta").read_bytes(), "test.delta")
    with tarfile.open(REF / "tests/test_targets/legacy_scripts/ANIm_mpl_Linux_3.1/nucmer_output.tar.gz") as tf:
        for m in tf.getmembers():
            if m.isfile() and m.name.endswith((".delta", ".filter")):
                store(tf.extractfile(m).read(), f"blochmannia/{Path(m.name).name}")
    # ---- out-of-sample MUMmer output (never looked at while the engine's constants were fitted) -----------
    js = REF / "tests/test_JSpecies"
    for f in sorted((js / "Group_2").glob("*.delta")):            # raw nucmer output, 2-record draft vs 1 record
        a, b = f.name[:-len(".delta")].split("_vs_")
        store(f.read_bytes(), f"group2/{a[:-4]}_vs_{b[:-4]}.delta")
    for f in sorted((js / "Group_2").glob("*.fna")):
        gz_copy(f, GOLD / "genomes" / "group2" / (f.name + ".gz"))
    for f in sorted((js / "pyani_tests").glob("*.fna_vs_*.fna.delta")):   # JSpecies' own nucmer runs on the 4 Caulobacter genomes
        a, b = f.name[:-len(".delta")].split("_vs_")
        store(f.read_bytes(), f"jspecies/{a[:-4]}_vs_{b[:-4]}.delta")
    recover_caulobacter_genomes()
    shutil.copyfile(REF / "tests/fixtures/anim/dataframes/deltadir_result.csv", GOLD / "ref_targets" / "anim_deltadir_result.csv")
    shutil.copyfile(js / "jspecies_results.tab", GOLD / "ref_targets" / "jspecies_results_caulobacter.tab")
    make_deltadir_cases(ref_anim, tmp)
    known = {"test.delta": [4016947, 4017751, 0.9994621994447228, 2191]}  # tests/test_anim.py:96-100
    assert tuples["test.delta"] == known["test.delta"]
    with open(GOLD / "anim_goldens.json", "w") as fh:
        json.dump({"reference_known_answers": known, "parse_delta": tuples,
                   "generated_by": "pyani/anim.py:parse_delta imported from the reference (tools/make_goldens.py)"},
                  fh, indent=0, sort_keys=True)
    print("wrote", GOLD / "anim_goldens.json", len(tuples), "files")


if __name__ == "__main__":
    if "--anim-only" in sys.argv:
        make_anim_goldens()
    else:
        main()
        make_anim_goldens()
        make_anib_goldens()
