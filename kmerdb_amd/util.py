"""Host helpers mirrored from the reference's kmerdb/util.py (checksum, format sniffing)."""
import hashlib
import os


def checksum(filepath):
    """md5 and sha256 of the raw (possibly gzipped) file bytes -- kmerdb/util.py:35-50."""
    if type(filepath) is not str:
        raise TypeError("kmerdb_amd.util.checksum() expects a str as its argument")
    elif not os.path.exists(filepath):
        raise IOError("kmerdb_amd.util.checksum could not find '{}' on the filesystem".format(filepath))
    hash_md5 = hashlib.md5()
    hash_sha256 = hashlib.sha256()
    with open(filepath, "rb") as ifile:
        for chunk in iter(lambda: ifile.read(1 << 20), b""):
            hash_md5.update(chunk)
            hash_sha256.update(chunk)
    return (hash_md5.hexdigest(), hash_sha256.hexdigest())


def is_gz_file(filepath):
    """Content sniff, like kmerdb/util.py:80-88 (which tries gzip.open + readline)."""
    with open(filepath, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def is_fasta(fname):
    """kmerdb/util.py:120-126: format is chosen by filename suffix."""
    if type(fname) is not str:
        raise TypeError("kmerdb_amd.util.is_fasta() expects a str argument")
    return fname.endswith((".fna", ".fna.gz", ".fa.gz", ".fa", ".fasta", ".fasta.gz"))


def is_fastq(fname):
    """kmerdb/util.py:128-134."""
    if type(fname) is not str:
        raise TypeError("kmerdb_amd.util.is_fastq() expects a str argument")
    return fname.endswith((".fastq", ".fastq.gz", ".fq.gz", ".fq"))
