"""Host ingest helpers around the GPU engine.

The names below are the ones pyani's callers use (`pyani/pyani_files.py:59-142`, `pyani index` in
`scripts/subcommands/subcmd_index.py:53-112`, `download.create_hash` at `download.py:585-605`) and they keep the same
return shapes and exceptions, so the parity tests can read like the reference's.  The implementation is organised
differently: ONE directory scan (`_scan`) feeds every discovery function, ONE resolver (`_hash_file_of`) knows the two places
an MD5 side file may live, and hashing / length counting run on a thread pool (hashlib and `bytes.translate` release the
GIL).  Packing sequences for the GPU is `Engine.add_fasta_batch` (multithreaded C++, `pg_add_fasta_batch`), which also
returns each genome's total length — a run that uploads its genomes needs `get_sequence_lengths` only for files it does not
upload.
"""
import hashlib
import logging
import mmap
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

LOG = logging.getLogger(__name__)
LEGACY_SUFFIXES = (".fasta", ".fas", ".fa", ".fna", ".fsa_nt")   # pyani_files.py:59-66
INDEX_SUFFIXES = (".fna", ".fa", ".fasta", ".fas")               # pyani_files.py:69-86
_WHITESPACE = b" \t\r\n"


class PyaniFilesException(Exception):
    """A genome file without its companion files (same role as pyani.pyani_files.PyaniFilesException)."""


class PyaniIndexException(Exception):
    """A file that was to be hashed is not there (same role as pyani.download.PyaniIndexException)."""


def _scan(dirname, suffixes: Sequence[str], files_only: bool) -> List[Path]:
    """Directory entries with one of the suffixes, sorted by path."""
    wanted = frozenset(suffixes)
    hits = [p for p in Path(dirname).iterdir() if p.suffix in wanted and (p.is_file() or not files_only)]
    hits.sort()
    return hits


def get_input_files(dirname: Path, *ext) -> List[Path]:
    """Entries of `dirname` ending in any of `ext`, sorted (pyani_files.py:118-125)."""
    return _scan(dirname, ext, files_only=False)


def get_fasta_files(dirname: Path = Path(".")) -> List[Path]:
    """Legacy FASTA discovery: the five historical suffixes (pyani_files.py:59-66)."""
    return _scan(dirname, LEGACY_SUFFIXES, files_only=False)


def get_fasta_paths(dirname: Path = Path("."), extlist: Optional[List] = None) -> List[Path]:
    """Regular files with a FASTA suffix, sorted (pyani_files.py:69-86)."""
    return _scan(dirname, extlist or INDEX_SUFFIXES, files_only=True)


def _hash_file_of(fasta: Path) -> Optional[Path]:
    """Where `pyani index` (genome.fna.md5) or a download (genome.md5) left the MD5 of `fasta`; None if neither exists."""
    for candidate in (fasta.with_name(fasta.name + ".md5"), fasta.with_suffix(".md5")):
        if candidate.is_file():
            return candidate
        LOG.warning("no MD5 file at %s", candidate)
    return None


def get_fasta_and_hash_paths(dirname: Path = Path(".")) -> List[Tuple[Path, Path]]:
    """[(FASTA, its MD5 file)] for every genome of the directory; PyaniFilesException names the first genome that has none
    (pyani_files.py:89-115)."""
    pairs = []
    for fasta in get_fasta_paths(dirname):
        md5 = _hash_file_of(fasta)
        if md5 is None:
            raise PyaniFilesException(f"{fasta} has neither {fasta.name}.md5 nor {fasta.stem}.md5 beside it")
        pairs.append((fasta, md5))
    return pairs


def create_hash(fname: Path) -> str:
    """MD5 hex digest of the file's bytes (download.py:585-605); PyaniIndexException if it cannot be opened."""
    digest = hashlib.md5()  # nosec: an identifier, not a security measure
    try:
        with open(fname, "rb") as fh:
            if os.fstat(fh.fileno()).st_size:
                with mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as view:
                    digest.update(view)
    except FileNotFoundError as exc:
        LOG.error("cannot hash %s: no such file", fname)
        raise PyaniIndexException(str(fname)) from exc
    return digest.hexdigest()


def _sequence_length(fname: Path) -> int:
    """Bases of all records together = bytes of the sequence lines minus white space (the sum of len(SeqRecord))."""
    data = Path(fname).read_bytes()
    first = 0 if data.startswith(b">") else data.find(b"\n>") + 1
    if first == 0 and not data.startswith(b">"):
        return 0
    total = 0
    for record in data[first:].split(b"\n>"):
        _, _, body = record.partition(b"\n")
        total += len(body.translate(None, _WHITESPACE))
    return total


def _first_description(fname: Path) -> str:
    """Header of the first record without the '>' (SeqRecord.description)."""
    with open(fname, "r") as fh:
        header = next((line for line in fh if line.startswith(">")), None)
    if header is None:
        raise PyaniFilesException(f"{fname} holds no FASTA record")
    return header[1:].rstrip("\r\n")


def _pool(threads: Optional[int]) -> ThreadPoolExecutor:
    return ThreadPoolExecutor(max(1, threads or min(32, os.cpu_count() or 1)))


def get_sequence_lengths(fastafilenames: Iterable[Path], threads: Optional[int] = None) -> Dict[str, int]:
    """{file stem: total sequence length}; ambiguity symbols count (pyani_files.py:128-142)."""
    files = [Path(f) for f in fastafilenames]
    with _pool(threads) as pool:
        return {f.stem: n for f, n in zip(files, pool.map(_sequence_length, files))}


def _indexed_hash(fasta: Path) -> str:
    """The genome's MD5: read back from `<name>.md5` when `pyani index` ran before, else computed and written there."""
    side = fasta.with_name(fasta.name + ".md5")
    if side.is_file():
        LOG.info("%s: hash file found, not re-hashing", fasta)
        return side.read_text().split()[0]
    md5 = create_hash(fasta)
    side.write_text(f"{md5}\t{fasta}\n")
    return md5


def index_directory(indir: Path, classfname: str = "classes.txt", labelfname: str = "labels.txt",
                    threads: Optional[int] = None) -> List[Tuple[Path, str]]:
    """`pyani index` (subcmd_index.py:53-112): an MD5 side file per genome and — unless present — a class and a label file
    with one `<hash>\\t<stem>\\t<description after its first word>` line per genome.  Returns [(path, hash)] by path."""
    genomes = get_fasta_paths(Path(indir))
    with _pool(threads) as pool:
        hashes = list(pool.map(_indexed_hash, genomes))
    table = "".join(f"{h}\t{g.stem}\t{_first_description(g).split(' ', 1)[-1]}\n" for g, h in zip(genomes, hashes))
    for name in (classfname, labelfname):
        out = Path(indir) / name
        if out.exists():
            LOG.warning("keeping the existing %s", out)
        else:
            out.write_text(table)
    return list(zip(genomes, hashes))
