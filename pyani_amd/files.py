"""Host ingest helpers around the GPU engine: pyani's file discovery, MD5 indexing and sequence
lengths — `pyani/pyani_files.py:59-142`, `pyani index` (`scripts/subcommands/subcmd_index.py:53-112`),
`download.create_hash` (`download.py:585-605`) — with the same names, returns and error behaviour.

Hashing and length counting are spread over a thread pool (hashlib and bytes operations release the GIL); packing the
sequences for the GPU is `Engine.add_fasta_batch` (multithreaded C++, `pg_add_fasta_batch`), which also returns each
genome's total length, so a run that uploads its genomes needs `get_sequence_lengths` only for files it does not upload.
"""
import hashlib
import logging
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple


class PyaniFilesException(Exception):
    """General exception for file handling (mirrors pyani.pyani_files.PyaniFilesException)."""


class PyaniIndexException(Exception):
    """Raised when a file to be hashed does not exist (mirrors pyani.download.PyaniIndexException)."""


def get_input_files(dirname: Path, *ext) -> List[Path]:
    """Sorted files of the directory whose suffix is one of `ext` (pyani_files.py:118-125)."""
    return sorted(fname for fname in Path(dirname).iterdir() if fname.suffix in ext)


def get_fasta_files(dirname: Path = Path(".")) -> List[Path]:
    """FASTA files of a directory by the legacy extension list (pyani_files.py:59-66)."""
    return get_input_files(dirname, ".fasta", ".fas", ".fa", ".fna", ".fsa_nt")


def get_fasta_paths(dirname: Path = Path("."), extlist: Optional[List] = None) -> List[Path]:
    """Sorted full paths of the regular files with a FASTA suffix (pyani_files.py:69-86)."""
    extlist = extlist or [".fna", ".fa", ".fasta", ".fas"]
    return sorted(fname for fname in Path(dirname).iterdir() if fname.is_file() and fname.suffix in extlist)


def get_fasta_and_hash_paths(dirname: Path = Path(".")) -> List[Tuple[Path, Path]]:
    """(FASTA file, hash file) pairs; the hash is `<name>.<ext>.md5`, else `<name>.md5`, else PyaniFilesException
    (pyani_files.py:89-115)."""
    logger = logging.getLogger(__name__)
    outfiles = []
    for infile in get_fasta_paths(dirname):
        hashfile = infile.with_name(f"{infile.name}.md5")
        if not hashfile.is_file():
            logger.warning("Hashfile %s does not exist...", hashfile)
            hashfile = infile.with_suffix(".md5")
            logger.warning("... trying %s.", hashfile)
        if not hashfile.is_file():
            raise PyaniFilesException(f"Alternate hashfile {hashfile} does not exist.")
        outfiles.append((infile, hashfile))
    return outfiles


def create_hash(fname: Path) -> str:
    """MD5 of the file's bytes (download.py:585-605); PyaniIndexException if the file is missing."""
    hash_md5 = hashlib.md5()  # nosec: an identifier, not a security measure
    try:
        with Path(fname).open("rb") as fhandle:
            for chunk in iter(lambda: fhandle.read(1 << 20), b""):
                hash_md5.update(chunk)
    except FileNotFoundError:
        logging.getLogger(__name__).error("Input file %s is not a file or symlink", fname)
        raise PyaniIndexException
    return hash_md5.hexdigest()


def _sequence_length(fname: Path) -> int:
    """Total bases of all records: every non-header line without its white space (what len(SeqRecord) sums to)."""
    total = 0
    with open(fname, "rb") as fh:
        started = False
        for line in fh:
            if line.startswith(b">"):
                started = True
            elif started:
                total += len(line.translate(None, b" \t\r\n"))
    return total


def _first_description(fname: Path) -> str:
    """Header line of the first record without '>' (SeqRecord.description)."""
    with open(fname, "r") as fh:
        for line in fh:
            if line.startswith(">"):
                return line[1:].rstrip("\r\n")
    raise PyaniFilesException(f"{fname} holds no FASTA record")


def _threads(threads: Optional[int]) -> int:
    return max(1, threads or min(32, os.cpu_count() or 1))


def get_sequence_lengths(fastafilenames: Iterable[Path], threads: Optional[int] = None) -> Dict[str, int]:
    """{file stem: total sequence length}; ambiguity symbols are not discounted (pyani_files.py:128-142)."""
    files = [Path(f) for f in fastafilenames]
    with ThreadPoolExecutor(_threads(threads)) as ex:
        return dict(zip((f.stem for f in files), ex.map(_sequence_length, files)))


def index_directory(indir: Path, classfname: str = "classes.txt", labelfname: str = "labels.txt",
                    threads: Optional[int] = None) -> List[Tuple[Path, str]]:
    """`pyani index` (subcmd_index.py:53-112): `<genome>.<ext>.md5` next to every FASTA file (an existing one is re-used),
    plus the class and label files `<hash>\\t<stem>\\t<description after the first word>` unless they exist already.
    Returns [(path, hash)] in sorted path order."""
    logger = logging.getLogger(__name__)
    indir = Path(indir)
    fpaths = get_fasta_paths(indir)

    def one(fpath: Path) -> str:
        hashfname = fpath.with_name(f"{fpath.name}.md5")
        if hashfname.is_file():
            logger.info("%s already indexed (using existing hash)", fpath)
            return hashfname.read_text().split()[0]
        datahash = create_hash(fpath)
        hashfname.write_text(f"{datahash}\t{fpath}\n")
        return datahash

    with ThreadPoolExecutor(_threads(threads)) as ex:
        hashes = list(ex.map(one, fpaths))
    lines = ["\t".join([h, p.stem, _first_description(p).split(" ", 1)[-1]]) for p, h in zip(fpaths, hashes)]
    for name in (classfname, labelfname):
        target = indir / name
        if target.exists():
            logger.warning("%s exists, not overwriting", target)
        else:
            target.write_text("\n".join(lines) + "\n")
    return list(zip(fpaths, hashes))
