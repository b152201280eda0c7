#!/usr/bin/env python3
"""Self-consistency fixture for the GPU ANIm pipeline: results of the HOST build of the same per-pair core
(tools/anim_debug/anim_debug: exhaustive sorted-table seeding + pg_anim_core.h + pg_nucmer_core.h) on seeded synthetic genomes.
-> tests/golden/anim_synth_host.json.

This does NOT pin parity with MUMmer (tests/golden/anim/ does that); it pins that the GPU seeding / clustering /
wave-cooperative extension compute exactly what the scalar statement of the algorithm computes, on inputs with every
divergence level of the SURVEY.md §8(d) generator.   Usage: python tools/make_anim_synth_host.py
"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_amd import synth  # noqa: E402

EXT = []
N, L, SEED = 6, 400_000, 20250228
tmp = ROOT / "gpurun_out" / "_synth_host"
tmp.mkdir(parents=True, exist_ok=True)
exe = ROOT / "tools" / "anim_debug" / "anim_debug"
subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/pyani_amd/csrc", str(exe) + ".cpp", "-o", str(exe)], check=True)
paths = []
for g in range(N):
    seq, off = synth.genome(SEED, N, g, L)
    p = tmp / f"{synth.genome_name(g)}.fna"
    synth.write_fasta(p, seq, off, synth.genome_name(g))
    paths.append(p)
out = {"n": N, "length": L, "seed": SEED, "pairs": {}}
for a in range(N):
    for b in range(N):
        if a == b:
            continue
        for mode, flag in (("filter", []), ("nofilter", ["--nofilter"])):
            r = subprocess.run([str(exe), str(paths[a]), str(paths[b])] + flag + EXT, capture_output=True, text=True, check=True)
            f = r.stdout.split()
            out["pairs"][f"{a},{b},{mode}"] = [int(f[0]), int(f[1]), float(f[2]).hex() if f[2] not in ("nan", "-nan") else "nan",
                                               int(f[3]), int(f[4])]
        print(a, b, out["pairs"][f"{a},{b},filter"], flush=True)
# draft-genome shape: the same genomes cut into many records of unequal length (contigs), some of them shorter than a
# seed; alignments must stop at record ends and the per-record bookkeeping (record_of, unions per sequence) must hold
import numpy as np  # noqa: E402


def resplit(seq, step, salt):
    cuts, p, k = [0], 0, 0
    while p < len(seq):
        p = min(len(seq), p + step + ((k * 7919 + salt) % step) - step // 2 + (12 if k % 9 == 4 else 0))
        if k % 11 == 5:
            p = min(len(seq), cuts[-1] + 15)          # a contig too short to seed
        cuts.append(p)
        k += 1
    return np.array(sorted(set(cuts)), dtype=np.uint64)


out["contigs"] = {"step": 3000, "pairs": {}}
cpaths = []
for g in range(3):
    seq, _ = synth.genome(SEED, N, g, L)
    off = resplit(seq, 3000, g)
    p = tmp / f"contigs_{g}.fna"
    synth.write_fasta(p, seq, off, f"c{g}")
    cpaths.append(p)
for a in range(3):
    for b in range(3):
        if a == b:
            continue
        r = subprocess.run([str(exe), str(cpaths[a]), str(cpaths[b])] + EXT, capture_output=True, text=True, check=True)
        f = r.stdout.split()
        out["contigs"]["pairs"][f"{a},{b}"] = [int(f[0]), int(f[1]), float(f[2]).hex(), int(f[3]), int(f[4])]
        print("contigs", a, b, out["contigs"]["pairs"][f"{a},{b}"], flush=True)
(ROOT / "tests" / "golden" / "anim_synth_host.json").write_text(json.dumps(out, indent=0, sort_keys=True))
