"""Small synthetic genome pairs whose RECORDS share content (repeat elements spread over contigs, as rRNA operons and IS elements are
in draft assemblies), records cut at different places in reference and query (so that clusters run across record junctions), some
records reverse-complemented.  Test DATA for the multi-record parity tests (CPU: host statement vs nucmer oracle; GPU: engine vs
nucmer oracle) and for tools/anim_fuzz_multirecord.py.  Deterministic for a given Python version (random.Random)."""
import random

COMP = str.maketrans("ACGT", "TGCA")


def mutate(rng, s, p_sub, p_indel):
    out = []
    for ch in s:
        x = rng.random()
        if x < p_sub:
            out.append(rng.choice([c for c in "ACGT" if c != ch]))
        elif x < p_sub + p_indel:
            if rng.random() < 0.5:
                continue
            out.append(ch)
            out.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4))))
        else:
            out.append(ch)
    return "".join(out)


def make_pair(rng, n_rec_max=4, L=24000):
    """reference + query, each 1..n_rec_max records; both carry copies of a few repeat elements, spread over their records."""
    repeats = ["".join(rng.choice("ACGT") for _ in range(rng.randint(120, 900))) for _ in range(rng.randint(1, 3))]
    base = "".join(rng.choice("ACGT") for _ in range(L))
    # the ancestor: the base sequence with repeat copies (some reverse-complemented, some slightly diverged) inserted
    parts, pos = [], 0
    cuts = sorted(rng.sample(range(500, L - 500), rng.randint(3, 8)))
    for c in cuts:
        parts.append(base[pos:c])
        rep = rng.choice(repeats)
        if rng.random() < 0.3:
            rep = rep[::-1].translate(COMP)
        if rng.random() < 0.5:
            rep = mutate(rng, rep, 0.01, 0.0)
        parts.append(rep)
        pos = c
    parts.append(base[pos:])
    anc = "".join(parts)

    def split(s, n):
        if n <= 1:
            return [s]
        cs = sorted(rng.sample(range(1000, len(s) - 1000), n - 1))
        return [s[a:b] for a, b in zip([0] + cs, cs + [len(s)])]

    p = rng.choice([0.0, 0.002, 0.01, 0.03, 0.08])
    ref = split(anc, rng.randint(1, n_rec_max))
    qry = split(mutate(rng, anc, p, p / 10), rng.randint(2, n_rec_max))
    if rng.random() < 0.5:      # records in another order, some reverse-complemented
        rng.shuffle(qry)
        qry = [r[::-1].translate(COMP) if rng.random() < 0.3 else r for r in qry]
    return ref, qry


def write_fasta(path, name, recs):
    with open(path, "w") as fh:
        for k, r in enumerate(recs):
            fh.write(f">{name}{k}\n")
            for i in range(0, len(r), 70):
                fh.write(r[i:i + 70] + "\n")


