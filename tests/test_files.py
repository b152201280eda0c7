"""Host ingest helpers: file discovery, `pyani index` hashing, sequence lengths — against the targets the
reference's tests hold for `pyani index` on the six Blochmannia genomes (tests/test_targets/subcmd_index/; extracted by
tools/make_index_goldens.py) and against the sequence lengths MUMmer wrote into the reference's .delta fixtures."""
import gzip
import json
import shutil

import pytest

from tests.conftest import GOLD


@pytest.fixture()
def index_dir(tmp_path):
    want = json.loads((GOLD / "ref_targets" / "subcmd_index.json").read_text())
    keep = set(want["same_bytes_as_golden_genome"])   # the genome copies held here that are byte-identical to the index inputs
    want["md5"] = {n: h for n, h in want["md5"].items() if n in keep}
    want["labels"] = [x for x in want["labels"] if x.split("\t")[0] in want["md5"].values()]
    want["classes"] = [x for x in want["classes"] if x.split("\t")[0] in want["md5"].values()]
    assert len(want["md5"]) >= 3 and len(want["labels"]) == len(want["md5"])
    for name in want["md5"]:
        with gzip.open(GOLD / "genomes" / "blochmannia" / (name + ".gz"), "rb") as fi, open(tmp_path / name, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    (tmp_path / "notes.txt").write_text("not a genome\n")
    return tmp_path, want


def test_index_directory_matches_reference_targets(index_dir):
    from pyani_amd import files
    d, want = index_dir
    got = files.index_directory(d, threads=3)
    assert [p.name for p, _ in got] == sorted(want["md5"])
    for p, h in got:
        assert h == want["md5"][p.name] == files.create_hash(p)
        assert (d / (p.name + ".md5")).read_text() == f"{h}\t{p}\n"
    assert sorted((d / "labels.txt").read_text().splitlines()) == want["labels"]
    assert sorted((d / "classes.txt").read_text().splitlines()) == want["classes"]
    # a second run re-uses the hash files and leaves the class / label files alone
    first = sorted(want["md5"])[0]
    (d / (first + ".md5")).write_text("feedfacefeedfacefeedfacefeedface\tsomewhere\n")
    (d / "labels.txt").write_text("mine\n")
    again = dict((p.name, h) for p, h in files.index_directory(d))
    assert again[first] == "feedfacefeedfacefeedfacefeedface" and (d / "labels.txt").read_text() == "mine\n"


def test_discovery_hash_pairs_and_lengths(index_dir):
    from pyani_amd import files
    d, want = index_dir
    assert [p.name for p in files.get_fasta_paths(d)] == sorted(want["md5"])
    assert files.get_fasta_files(d) == files.get_input_files(d, ".fasta", ".fas", ".fa", ".fna", ".fsa_nt")
    with pytest.raises(files.PyaniFilesException):
        files.get_fasta_and_hash_paths(d)                       # no hash files yet
    files.index_directory(d)
    pairs = files.get_fasta_and_hash_paths(d)
    assert all(h.name == f.name + ".md5" for f, h in pairs)
    f0, h0 = pairs[0]
    h0.rename(f0.with_suffix(".md5"))                            # the alternative name without the FASTA suffix
    assert files.get_fasta_and_hash_paths(d)[0] == (f0, f0.with_suffix(".md5"))
    with pytest.raises(files.PyaniIndexException):
        files.create_hash(d / "missing.fna")
    lengths = files.get_sequence_lengths(files.get_fasta_paths(d), threads=2)
    # sequence lengths as MUMmer recorded them in the headers of the reference's .delta fixtures (single-record genomes)
    for f in sorted((GOLD / "anim" / "blochmannia").glob("*.delta.gz")):
        a, b = f.name[:-len(".delta.gz")].split("_vs_")
        with gzip.open(f, "rt") as fh:
            hdr = next(line for line in fh if line.startswith(">")).split()
        if a in lengths and b in lengths:
            assert (lengths[a], lengths[b]) == (int(hdr[2]), int(hdr[3])), f.name
    assert lengths["GCF_000011745.1_ASM1174v1_genomic"] == 791654
    odd = d / "odd.fa"
    odd.write_text("; comment before the first record\n>r1 x\nAC GT\r\nNNN\n\n>r2\n\nacgtn\n")
    assert files.get_sequence_lengths([odd]) == {"odd": 12}
