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
