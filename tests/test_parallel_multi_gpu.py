"""The in-product multi-GPU path (pyani_amd/multi.py): one process, one engine per device, comparisons pulled from a work queue.
CPU: the chunking keeps a pair with its reverse and loses nothing.  GPU: two engines on device 0 give exactly what one gives
(ANIm and fragment mode), and run_anim(devices=...) == run_anim()."""
import numpy as np
import pytest


def test_chunks_keep_a_pair_with_its_reverse_and_cover_the_list():
    from pyani_amd.multi import _chunks_by_hub
    rng = np.random.RandomState(7)
    n = 37
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    rng.shuffle(pairs)
    pairs = pairs[:900] + [(3, 4), (3, 4)]            # an incomplete grid with a repeated pair
    a, b = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    chunks = _chunks_by_hub(a, b, 50)
    flat = np.concatenate(chunks)
    assert sorted(flat.tolist()) == list(range(len(pairs))) and len(chunks) > 5
    where = {}
    for k, c in enumerate(chunks):
        for i in c:
            where.setdefault(frozenset((int(a[i]), int(b[i]))), set()).add(k)
    assert all(len(ks) == 1 for ks in where.values())          # both directions (and repeats) of a pair share a chunk


def test_static_shares_keep_a_pair_with_its_reverse_cover_the_list_and_are_even():
    """The default deal of MultiEngine / DistributedEngine since round 5: one scrambled share per device, one engine call each
    (pyani_amd.multi._static_parts_by_hub; measured on MI355X: profiles/r05_deal_probe.json)."""
    from pyani_amd.multi import _static_parts_by_hub
    n = 200
    a, b = np.array([x for x in range(n) for y in range(n) if x != y]), np.array([y for x in range(n) for y in range(n) if x != y])
    K = 8                                              # families at a fixed stride, as in the synthetic sets
    for parts in (2, 3, 8):
        shares = _static_parts_by_hub(a, b, parts)
        assert len(shares) == parts and sorted(np.concatenate(shares).tolist()) == list(range(len(a)))
        where = {}
        for k, c in enumerate(shares):
            for i in c:
                where.setdefault(frozenset((int(a[i]), int(b[i]))), set()).add(k)
        assert all(len(ks) == 1 for ks in where.values())
        sizes = [len(c) for c in shares]
        related = [int(((a[c] % K) == (b[c] % K)).sum()) for c in shares]      # the expensive pairs
        assert max(sizes) <= 1.15 * min(sizes) and max(related) <= 1.35 * (sum(related) / parts), (parts, sizes, related)
    assert len(_static_parts_by_hub(a[:6], b[:6], 8)) <= 6                    # fewer groups than parts: no empty share


@pytest.mark.gpu
def test_two_engines_on_one_gpu_equal_one_engine():
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    from pyani_amd.multi import MultiEngine
    n, L, seed = 12, 120_000, 31
    data = [synth.genome(seed, n, g, L) for g in range(n)]
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    with Engine(0) as one:
        ids = [one.add_genome(*d) for d in data]
        ra, qa = [ids[a] for a, _ in pairs], [ids[b] for _, b in pairs]
        want = one.anim_pairs(ra, qa)
        want_b = one.anib_pairs(ra[:40], qa[:40])
    with MultiEngine([0, 0], chunk_pairs=16) as two:
        ids2 = [two.add_genome(*d) for d in data]
        assert ids2 == ids and two.genome_count() == n
        got = two.anim_pairs(ra, qa)
        assert sum(two.last_chunks_per_engine) >= 6 and min(two.last_chunks_per_engine) >= 1
        got_b = two.anib_pairs(ra[:40], qa[:40])
    assert got.tobytes() == want.tobytes()
    assert got_b.tobytes() == want_b.tobytes()
    assert (want["status"] == 0).sum() >= len(pairs) // 3
    with MultiEngine([0, 0]) as two:                   # the default deal: one scrambled share per device, one call each
        assert [two.add_genome(*d) for d in data] == ids
        assert two.anim_pairs(ra, qa).tobytes() == want.tobytes() and two.last_chunks_per_engine == [1, 1]


@pytest.mark.gpu
def test_run_anim_with_devices_equals_single_engine(genome_dir, tmp_path):
    import shutil
    from pyani_amd import subcmd_anim
    d = tmp_path / "in"
    d.mkdir()
    for p in list(genome_dir["blochmannia"].values())[:4]:
        shutil.copy(p, d / p.name)
    one = subcmd_anim.run_anim(d)
    two = subcmd_anim.run_anim(d, devices=[0, 0])
    assert one.results == two.results and one.json == two.json
    with pytest.raises(ValueError):
        subcmd_anim.run_anim(d, write_output=True)      # refused before any work


def test_alignment_parts_are_merged_into_the_callers_order():
    """merge_alignment_parts (what MultiEngine.anim_alignments_batch does with its devices' chunks) against one flat result."""
    from pyani_amd.engine import Engine
    from pyani_amd.multi import merge_alignment_parts
    rng = np.random.RandomState(5)
    n = 23
    counts = rng.randint(0, 5, size=n)
    counts[4] = 0
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(counts)
    recs = np.zeros(int(off[-1]), dtype=Engine.ALN_DTYPE)
    recs["rs"] = np.arange(len(recs)) * 7 + 1
    icnt = rng.randint(0, 4, size=len(recs))
    ioff = np.zeros(len(recs) + 1, dtype=np.uint64); ioff[1:] = np.cumsum(icnt)
    ind = np.arange(int(ioff[-1]), dtype=np.int64) * 3 - 5
    order = rng.permutation(n)
    chunks = [order[:9], order[9:10], order[10:]]
    parts = []
    for idx in chunks:          # what an engine would return for the chunk's own pair list
        o = np.zeros(len(idx) + 1, dtype=np.uint64); o[1:] = np.cumsum(counts[idx])
        r = np.concatenate([recs[int(off[p]):int(off[p + 1])] for p in idx]) if len(idx) else recs[:0]
        ic = np.concatenate([icnt[int(off[p]):int(off[p + 1])] for p in idx]) if len(idx) else icnt[:0]
        io = np.zeros(len(r) + 1, dtype=np.uint64); io[1:] = np.cumsum(ic)
        iv = np.concatenate([ind[int(ioff[int(off[p])]):int(ioff[int(off[p + 1])])] for p in idx]) if len(idx) else ind[:0]
        parts.append((o, r, io, iv))
    o2, r2, io2, i2 = merge_alignment_parts(n, chunks, parts, True)
    assert o2.tolist() == off.tolist() and r2.tobytes() == recs.tobytes() and io2.tolist() == ioff.tolist() and i2.tolist() == ind.tolist()
    o3, r3, io3, i3 = merge_alignment_parts(n, chunks, [(p[0], p[1], None, None) for p in parts], False)
    assert o3.tolist() == off.tolist() and r3.tobytes() == recs.tobytes() and io3 is None and i3 is None


@pytest.mark.gpu
def test_alignment_records_and_tetra_over_two_engines_equal_one(genome_dir):
    """MultiEngine shards anim_alignments_batch (run_anim(write_output=True)) and TETRA over its devices: same records, same
    indel lists, same Z-scores and matrix as one engine (two engines on GPU 0)."""
    from pyani_amd.engine import Engine
    from pyani_amd.multi import MultiEngine
    paths = list(genome_dir["blochmannia"].values())[:4]
    with Engine(0) as one:
        ids = [one.add_fasta(p)[0] for p in paths]
        pairs = [(a, b) for a in ids for b in ids if a != b]
        want = one.anim_alignments_batch([a for a, _ in pairs], [b for _, b in pairs], with_indels=True)
        want_t = one.tetra_matrix(ids)
        want_c = one.tetra_counts(ids)
    with MultiEngine([0, 0]) as two:
        ids2 = [two.add_fasta(p)[0] for p in paths]
        assert ids2 == ids
        got = two.anim_alignments_batch([a for a, _ in pairs], [b for _, b in pairs], with_indels=True)
        got_t = two.tetra_matrix(ids)
        got_c = two.tetra_counts(ids)
    assert got[0].tolist() == want[0].tolist() and got[2].tolist() == want[2].tolist() and got[3].tolist() == want[3].tolist()
    for k in range(len(pairs)):      # the records of a pair: same set (a pair's own order is MUMmer's in both)
        a, b = int(want[0][k]), int(want[0][k + 1])
        assert got[1][a:b].tobytes() == want[1][a:b].tobytes()
    assert all(x.tobytes() == y.tobytes() for x, y in zip(got_t, want_t)) and all(x.tobytes() == y.tobytes() for x, y in zip(got_c, want_c))


def _run_anim_rank(rank, world, port, indir, out_dir):
    import json
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # (two RCCL ranks cannot share one GPU)
    try:
        from pyani_amd import subcmd_anim
        from pyani_amd.engine import Engine
        outdir = os.path.join(out_dir, "shared_out")      # ONE output directory for both ranks: rank 0 alone writes and recovers
        with Engine(0) as eng:
            run = subcmd_anim.run_anim(indir, outdir, write_output=True, engine=eng, distributed=True)
            again = subcmd_anim.run_anim(indir, outdir, recovery=True, engine=eng, distributed=True)
        assert again.results == run.results and len(again.recovered) == len(run.written) == len(run.results)
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
            json.dump({"results": {f"{a}|{b}": list(v) for (a, b), v in run.results.items()}, "json": run.json}, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_run_anim_under_a_process_group_equals_the_plain_run(genome_dir, tmp_path):
    """run_anim(distributed=True) as one process per GPU (here: two gloo ranks on GPU 0): the comparisons are dealt over the ranks and
    assembled with one all-gather (pyani_amd.parallel.DistributedEngine) — every rank returns the run a single process computes,
    result for result; both ranks share ONE output directory (rank 0 alone writes the .filter files, and a recovery run reads them
    on rank 0 and broadcasts what is left to do)."""
    import json
    import shutil
    import socket
    import torch.multiprocessing as mp
    from pyani_amd import subcmd_anim
    d = tmp_path / "in"
    d.mkdir()
    for p in list(genome_dir["blochmannia"].values())[:5]:
        shutil.copy(p, d / p.name)
    one = subcmd_anim.run_anim(d)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_run_anim_rank, args=(2, port, str(d), str(tmp_path)), nprocs=2, join=True)
    want = {f"{a}|{b}": list(v) for (a, b), v in one.results.items()}
    for rank in range(2):
        got = json.loads((tmp_path / f"rank{rank}.json").read_text())
        assert got["results"] == want and got["json"] == one.json, rank
