from .Seq import Seq
from .SeqRecord import SeqRecord


def _fasta(handle):
    name, chunks = None, []
    for line in handle:
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if name is not None:
                yield SeqRecord(Seq("".join(chunks)), id=name)
            toks = line[1:].split()
            name = toks[0] if toks else ""
            chunks = []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield SeqRecord(Seq("".join(chunks)), id=name)


def _fastq(handle):
    while True:
        head = handle.readline()
        if not head:
            return
        if not head.strip():
            continue
        seq = handle.readline().rstrip("\r\n")
        handle.readline()
        handle.readline()
        toks = head[1:].split()
        yield SeqRecord(Seq(seq), id=toks[0] if toks else "")


def parse(handle, fmt):
    if fmt == "fasta":
        return _fasta(handle)
    if fmt == "fastq":
        return _fastq(handle)
    raise ValueError(fmt)
