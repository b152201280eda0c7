"""FASTA-only subset of Bio.SeqIO: parse(path, "fasta") -> records with .id/.description/.seq."""
_COMP = str.maketrans("ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn")


class Seq:
    def __init__(self, data):
        self._d = data

    def __str__(self):
        return self._d

    def __len__(self):
        return len(self._d)

    def reverse_complement(self):
        return Seq(self._d.translate(_COMP)[::-1])

    def upper(self):
        return Seq(self._d.upper())


class SeqRecord:
    def __init__(self, title, seq):
        self.description = title
        self.id = title.split(None, 1)[0] if title.split() else ""
        self.seq = Seq(seq)

    def __len__(self):
        return len(self.seq)


def parse(handle, fmt):
    if fmt != "fasta":
        raise ValueError("shim supports fasta only")
    opened = not hasattr(handle, "read")
    fh = open(handle, "r") if opened else handle
    try:
        title, chunks = None, []
        for line in fh:
            if line.startswith(">"):
                if title is not None:
                    yield SeqRecord(title, "".join(chunks))
                title, chunks = line[1:].rstrip(), []
            elif title is not None:
                chunks.append(line.strip().replace(" ", "").replace("\r", ""))
        if title is not None:
            yield SeqRecord(title, "".join(chunks))
    finally:
        if opened:
            fh.close()
