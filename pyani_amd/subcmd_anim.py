"""`pyani anim` without the database: the driver of pyani/scripts/subcommands/subcmd_anim.py:128-459 with its nucmer /
delta-filter jobs replaced by ONE batched GPU call.  No CLI, no SQLAlchemy — the rows and matrices it would store come
back as plain data.

What is kept of the reference's behaviour:
  * inputs: the FASTA files of `indir` in sorted order (pyani_files.get_fasta_paths); genome ids 1..N in that order (what
    add_run_genomes assigns in a fresh database), all N(N-1) permutations are compared (subcmd_anim.py:233);
  * output files: `<outdir>/nucmer_output/<q>/<q>_vs_<s>.filter` (`.delta` with nofilter), anim.py:271-288;
  * `recovery=True` (subcmd_anim.py:269-285, generate_joblist:308-375): comparisons whose output file already exists are
    NOT recomputed — the file (MUMmer's or ours) is parsed instead — the others are run;
  * results: one Comparison row per ordered pair (subcmd_anim.py:434-456) and update_comparison_matrices' five NaN-initialised
    frames indexed by genome id, with their to_json() strings (pyani_orm.py:618-666);
  * errors: a pair the engine could not process raises PyaniANImException (the reference: a failed job ->
    PyaniException "Multiprocessing run failed in ANIm"); a pair without any alignment raises ZeroDivisionError exactly as
    parse_delta does on its empty .filter file, unless skip_zero=True (then the pair has no row and its cells stay NaN).
"""
from itertools import permutations
from pathlib import Path
from typing import Dict, List, NamedTuple, Optional, Tuple

import pandas as pd

from . import anim, files
from .engine import Engine, default_engine

ALIGNDIR = "nucmer_output"     # pyani_config.ALIGNDIR["ANIm"]


class AnimRun(NamedTuple):
    genome_ids: Dict[str, int]                                  # stem -> genome id (1-based, sorted-path order)
    lengths: Dict[str, int]
    results: Dict[Tuple[str, str], Tuple[int, int, float, int]]  # (query stem, subject stem) -> parse_delta tuple
    comparisons: List[dict]                                     # Comparison rows
    matrices: Dict[str, pd.DataFrame]                           # identity / coverage / aln_lengths / sim_errors / hadamard
    json: Dict[str, str]                                        # Run.df_* strings
    recovered: List[Path]                                       # output files reused in recovery mode
    written: List[Path]                                         # output files written by this run


WRITE_CHUNK = 256      # ordered pairs per pg_anim_alignments_batch call of write_output (its result is held on the host)


def outfile_path(outdir: Path, qstem: str, sstem: str, nofilter: bool = False) -> Path:
    """anim.py:271-288: suffixes are appended as strings (stems may contain dots)."""
    return Path(outdir) / ALIGNDIR / qstem / (f"{qstem}_vs_{sstem}" + (".delta" if nofilter else ".filter"))


def run_anim(indir, outdir=None, recovery: bool = False, nofilter: bool = False, maxmatch: bool = False,
             write_output: bool = False, skip_zero: bool = False, engine: Optional[Engine] = None,
             devices: Optional[List[int]] = None, workers: Optional[int] = None, distributed: bool = False, group=None) -> AnimRun:
    """ANIm over every FASTA file of `indir`.  outdir is needed for recovery / write_output only.
    devices / workers: run on several GPUs of this node (pyani's `--workers`, subcmd_anim.py:392-396, counts GPUs here): the
    comparisons are pulled from a work queue by one engine per device (pyani_amd/multi.py); ignored when `engine` is given.
    distributed=True (or group=<a torch.distributed process group>): one process per GPU — the call is COLLECTIVE: EVERY rank of
    the group must make it with the same arguments (pyani_amd.parallel.DistributedEngine deals the comparisons over the ranks and
    assembles them with one RCCL all-gather; every rank returns the whole run).  In that mode rank 0 alone reads recovery files and
    writes output files, and the list of comparisons still to run is broadcast from rank 0, so every rank provably deals the same
    list.  Never implied: an initialised process group alone changes nothing (a library user calling run_anim on rank 0 only must
    not deadlock in a collective; passing a DistributedEngine as `engine` is the other explicit way in)."""
    if write_output and outdir is None:
        raise ValueError("write_output needs an output directory")     # before any work is done
    own = None
    if engine is None and (devices is not None or workers):
        from . import multi
        engine = multi.engine_for(devices, workers)
        own = engine if isinstance(engine, multi.MultiEngine) else None
    try:
        eng = engine or default_engine()
        coll = None
        if distributed or group is not None:
            from . import parallel
            eng = parallel.engine_for_process_group(eng, group)
        if type(eng).__name__ == "DistributedEngine" and eng.world > 1:
            coll = eng
        return _run_anim(indir, outdir, recovery, nofilter, maxmatch, write_output, skip_zero, eng, coll)
    finally:
        if own is not None:
            own.close()


def _bcast(coll, obj):
    """rank 0's `obj` on every rank of the collective run (torch.distributed.broadcast_object_list)."""
    import torch.distributed as dist
    box = [obj if coll.rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(coll.group, 0) if coll.group is not None else 0, group=coll.group)
    return box[0]


def _run_anim(indir, outdir, recovery, nofilter, maxmatch, write_output, skip_zero, eng, coll=None) -> AnimRun:
    lead = coll is None or coll.rank == 0          # collective run: rank 0 alone touches the output directory
    paths = files.get_fasta_paths(Path(indir))
    stems = [p.stem for p in paths]
    if len(set(stems)) != len(stems):
        raise ValueError("two input files share a stem (pyani keys every result by Path.stem)")
    by_stem = dict(zip(stems, paths))
    genome_ids = {s: k + 1 for k, s in enumerate(stems)}
    # (subject, query) tuples as generate_joblist unpacks them (subcmd_anim.py:323): the QUERY is nucmer's reference
    todo = [(q, s) for s, q in permutations(stems, 2)]
    results: Dict[Tuple[str, str], Tuple[int, int, float, int]] = {}
    recovered: List[Path] = []
    if recovery:
        if outdir is None:
            raise ValueError("recovery mode needs the output directory of the earlier run")
        old, err = [], None
        if lead:
            try:
                existing = set(sorted((Path(outdir) / ALIGNDIR).glob("*/*.delta" if nofilter else "*/*.filter")))
                old = [(q, s, outfile_path(outdir, q, s, nofilter)) for q, s in todo]
                old = [(q, s, f) for q, s, f in old if f in existing]
                if old:
                    recs = eng.anim_reduce([anim.read_delta(f) for _, _, f in old], apply_filter=False)
                    for (q, s, f), rec in zip(old, recs):
                        try:
                            results[(q, s)] = anim._tuple(rec)
                        except ZeroDivisionError:
                            if not skip_zero:
                                raise
                        recovered.append(f)
            except Exception as exc:  # noqa: BLE001
                if coll is None:
                    raise
                err = exc
        if coll is not None:      # every rank continues from rank 0's view of the output directory (or fails with it)
            results, recovered, done_keys, err_text = _bcast(coll, (results, recovered, [(q, s) for q, s, _ in old], None if err is None else repr(err)))
            if err_text is not None:
                raise err if err is not None else RuntimeError(f"run_anim: recovery failed on rank 0: {err_text}")
            done = set(done_keys)
        else:
            done = {(q, s) for q, s, _ in old}
        todo = [k for k in todo if k not in done]
    written: List[Path] = []
    scratch_store = eng.genome_count() == 0
    lengths: Dict[str, int] = {}
    try:
        ids = {}
        for p, (gid, total, _) in zip(paths, eng.add_fasta_batch(paths)):
            ids[p.stem], lengths[p.stem] = gid, total
        if todo:
            recs = eng.anim_pairs([ids[q] for q, _ in todo], [ids[s] for _, s in todo], filter_1to1=not nofilter, maxmatch=maxmatch)
            for (q, s), rec in zip(todo, recs):
                try:
                    results[(q, s)] = anim._tuple(rec)
                except ZeroDivisionError:
                    if not skip_zero:
                        raise
            if write_output and lead:
                # the files nucmer / delta-filter would have left, indel lists included: batched calls, a traceback pass on the GPU each
                for c0 in range(0, len(todo), WRITE_CHUNK):
                    part = todo[c0:c0 + WRITE_CHUNK]
                    off, recs, ioff, ind = eng.anim_alignments_batch([ids[q] for q, _ in part], [ids[s] for _, s in part],
                                                                     maxmatch=maxmatch, with_indels=True)
                    for k, (q, s) in enumerate(part):
                        f = outfile_path(outdir, q, s, nofilter)
                        f.parent.mkdir(parents=True, exist_ok=True)
                        lo, hi = int(off[k]), int(off[k + 1])
                        anim.write_delta(f, by_stem[q], by_stem[s], recs[lo:hi], filtered=not nofilter,
                                         indels=[ind[int(ioff[a]):int(ioff[a + 1])] for a in range(lo, hi)])
                        written.append(f)
    finally:
        if scratch_store:
            eng.clear_genomes()
    if coll is not None and write_output:
        written = _bcast(coll, written)      # (also the point where the other ranks wait for rank 0's files)
    mats = anim.assemble_run_matrices(results, lengths, genome_ids=genome_ids)
    return AnimRun(genome_ids, lengths, results, anim.comparison_rows(results, lengths, genome_ids, maxmatch=maxmatch), mats,
                   anim.run_matrices_to_json(mats), recovered, written)
