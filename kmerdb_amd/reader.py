"""Host feeder: FASTA / FASTQ (plain or gzip) -> flat residue buffers for the engine.

Takes over kmerdb/parse.py:50-85 (parse_sequence_file over Bio.SeqIO): gzip is sniffed by content, the
format is chosen by filename suffix, record ids are the first whitespace token of the header, FASTA lines
are concatenated, case is preserved.  Instead of one SeqRecord object per read it emits, per block,
    bases   uint8[nbytes]      residues of all records of the block, concatenated
    offsets uint64[nreads+1]   record r = bases[offsets[r]:offsets[r+1]]
which is exactly what kdb_submit() takes.  The splitting itself is native (kdb_parse_fastq / kdb_parse_fasta
in libkdbhip.so: memchr-driven, one pass); with pinned=True the residues land directly in pinned host
memory, so the engine's DMA reads them without another copy.
"""
import ctypes
import gzip
import os

import numpy as np

from . import _abi, util

BLOCK_BYTES = 128 << 20
_RING = 3      # a block may be overwritten once three further blocks were produced (see iter_blocks)


class Block(tuple):
    """(bases, offsets, ids) -- unpacks like the 3-tuple it always was -- plus .cont: record 0 is the next piece of the
    previous block's last record (a FASTA record longer than a block) and starts with that block's last `overlap` residues."""

    def __new__(cls, bases, offsets, ids, cont=False):
        self = tuple.__new__(cls, (bases, offsets, ids))
        self.cont = cont
        return self


def _host_threads():
    """Threads a BGZF file's members are inflated with: the CPUs this process can really use (affinity cut down to the cgroup's quota),
    less the four that split records, hash the file (md5, sha256) and feed the engine beside them; at most 16."""
    return max(1, min(16, util.effective_cpus() - 4))


def _split_threads():
    """Threads one block of FASTQ text is split with (kdb_parse_fastq_mt) while the next block is being read or inflated."""
    return max(1, min(4, util.effective_cpus() // 4))


class _BgzfFile:
    """Read-only file object over a BGZF file (bgzip, Bio.bgzf): .read(n) / .readinto(buf) like gzip.open(path, 'rb'), but the
    members are inflated in parallel by the native kdb_bgzf_inflate (gzip inflates them one by one), straight into the
    caller's buffer."""
    CHUNK = 8 << 20                       # compressed bytes per file read (about 4 x that of text)

    def __init__(self, path):
        self.f = open(path, "rb")
        self.lib = _abi.lib()
        self.comp = b""                   # compressed bytes read but not yet inflated
        self.spill = b""                  # inflated bytes of a member that did not fit the caller's buffer
        self.file_eof = False
        self.threads = _host_threads()
        self.one = np.empty(65536, dtype=np.uint8)

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _inflate(self, dst_addr, cap):
        """Inflate whole members of self.comp into [dst_addr, dst_addr + cap) -> bytes produced (0: none fits / none complete)."""
        consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _abi.check(self.lib.kdb_bgzf_inflate(ctypes.cast(ctypes.c_char_p(self.comp), ctypes.c_void_p), len(self.comp), dst_addr, cap,
                                             self.threads, ctypes.byref(consumed), ctypes.byref(produced)))
        if consumed.value:
            self.comp = self.comp[consumed.value:]
        return produced.value, consumed.value

    def readinto(self, mv):
        want, n = len(mv), 0
        if want == 0:
            return 0
        base = ctypes.addressof(ctypes.c_char.from_buffer(mv))
        while n < want:
            if self.spill:
                take = min(len(self.spill), want - n)
                mv[n:n + take] = self.spill[:take]
                self.spill = self.spill[take:]
                n += take
                continue
            if len(self.comp) < (1 << 16) and not self.file_eof:
                data = self.f.read(self.CHUNK)
                if not data:
                    self.file_eof = True
                self.comp += data
            if not self.comp:
                break
            if want - n >= 65536:
                produced, consumed = self._inflate(base + n, want - n)
            else:                                                   # less room than a member may need: through a member-sized buffer
                produced, consumed = self._inflate(self.one.ctypes.data, 65536)
                if produced:
                    self.spill = self.one[:produced].tobytes()
                    continue
            n += produced
            if not consumed:                                        # no complete member in what is buffered
                if self.file_eof:
                    raise ValueError("truncated BGZF file")
                data = self.f.read(self.CHUNK)
                if not data:
                    self.file_eof = True
                self.comp += data
        return n

    def read(self, n=-1):
        if n is None or n < 0:
            parts = []
            while True:
                x = self.read(32 << 20)
                if not x:
                    return b"".join(parts)
                parts.append(x)
        buf = bytearray(n)
        got = self.readinto(memoryview(buf))
        return bytes(buf[:got]) if got < n else bytes(buf)


class _GzFile:
    """Read-only file object over one gzip stream: .read(n) like gzip.open(path, 'rb').read(n); the inflating is done by a
    native thread that runs ahead of the reader (kdb_gz_open), outside the GIL."""

    def __init__(self, path):
        self.lib = _abi.lib()
        self.h = ctypes.c_void_p()
        _abi.check(self.lib.kdb_gz_open(path.encode(), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.kdb_gz_close(self.h)
            self.h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def readinto(self, mv):
        """Fill the writable buffer `mv` (fewer bytes only at the end of the stream) -> bytes written."""
        n = len(mv)
        if n == 0:
            return 0
        got = ctypes.c_size_t(0)
        addr = ctypes.addressof(ctypes.c_char.from_buffer(mv))
        _abi.check(self.lib.kdb_gz_read(self.h, addr, n, ctypes.byref(got)))
        return got.value

    def read(self, n=-1):
        if n is None or n < 0:
            parts = []
            while True:
                x = self.read(16 << 20)
                if not x:
                    return b"".join(parts)
                parts.append(x)
        buf = bytearray(n)
        got = self.readinto(memoryview(buf))
        return bytes(buf[:got]) if got < n else bytes(buf)


def is_bgzf(path):
    """gzip member whose extra field carries the 'BC' block-size subfield (BGZF)."""
    with open(path, "rb") as f:
        h = f.read(18)
    return len(h) >= 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[12:14] == b"BC"


def _open(path):
    if not util.is_gz_file(path):
        return open(path, "rb")
    if is_bgzf(path):
        try:
            return _BgzfFile(path)
        except Exception:                                            # no native library: plain gzip reads BGZF too
            pass
    try:
        return _GzFile(path)
    except _abi.KdbHipError:                                         # no native library (reading only): Python's gzip
        return gzip.open(path, "rb")


def _readinto(f, mv):
    """Fill `mv` from the file object `f` (short only at EOF) -> bytes written.  No intermediate bytes objects for sources
    that can write into a buffer (plain files, the native gzip stream)."""
    n, want = 0, len(mv)
    if hasattr(f, "readinto"):
        while n < want:
            got = f.readinto(mv[n:])
            if not got:
                break
            n += got
        return n
    while n < want:
        chunk = f.read(want - n)
        if not chunk:
            break
        mv[n:n + len(chunk)] = chunk
        n += len(chunk)
    return n


class _Prefetch:
    """A thread that reads the next blocks of text while the caller splits the current one: buffers of `reserve + block` bytes,
    the text of a block behind the first `reserve` bytes (where the caller puts what the block before left unparsed).  The
    heavy part of every source's readinto (file copy, native inflate) runs outside the GIL."""

    def __init__(self, f, block, reserve, depth=2):
        import queue
        import threading
        self.reserve = reserve
        self.free, self.full = queue.Queue(), queue.Queue()
        for _ in range(depth):
            self.free.put(bytearray(reserve + block))

        def run():
            try:
                while True:
                    b = self.free.get()
                    if b is None:
                        return
                    n = _readinto(f, memoryview(b)[reserve:])
                    self.full.put((b, n))
                    if n == 0:
                        return
            except BaseException as e:  # noqa: BLE001 - re-raised in get()
                self.full.put(e)

        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def get(self):
        item = self.full.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def release(self, b):
        self.free.put(b)

    def close(self):
        self.free.put(None)
        self.thread.join()


_ring_pool = []                      # [key, ring, busy]: rings live as long as the process (pinning memory is slow; `profile` parses many files)
_ring_lock = None


def _get_ring(nbytes, pinned, depth=_RING):
    """A ring that nobody else is using (several files are read at the same time by profile()); _release_ring gives it back."""
    import threading
    global _ring_lock
    if _ring_lock is None:
        _ring_lock = threading.Lock()
    key = (nbytes, bool(pinned), depth)
    with _ring_lock:
        for ent in _ring_pool:
            if ent[0] == key and not ent[2]:
                ent[2] = True
                return ent[1]
        if len(_ring_pool) > 12:
            _ring_pool[:] = [e for e in _ring_pool if e[2]]
    r = _Buffers(nbytes, pinned, depth)
    with _ring_lock:
        _ring_pool.append([key, r, True])
    return r


def _release_ring(ring):
    if _ring_lock is None:
        return
    with _ring_lock:
        for ent in _ring_pool:
            if ent[1] is ring:
                ent[2] = False


class _Buffers:
    """Ring of residue buffers (pinned if possible) and offset buffers."""

    def __init__(self, nbytes, pinned, depth=_RING):
        self.offs = {}
        self.bufs = []
        self.depth = depth
        for _ in range(depth):
            b = None
            if pinned:
                try:
                    from .engine import pinned_empty
                    b = pinned_empty(nbytes)
                except Exception:      # no device / no pinned memory: plain host memory works too (engine stages it)
                    b = None
                    pinned = False
            if b is None:
                b = np.empty(nbytes, dtype=np.uint8)
            self.bufs.append(b)
        self.pinned = pinned
        self.i = 0

    def next(self, need):
        b = self.bufs[self.i]
        if b.size < need:              # grow (rare: a block larger than expected), keeping the ring what it says it is
            nb = None
            if self.pinned:
                try:
                    from .engine import pinned_empty
                    nb = pinned_empty(need + need // 8)
                except Exception:      # no more pinned memory: the whole ring counts as pageable from now on (submit stages it)
                    self.pinned = False
            if nb is None:
                nb = np.empty(need, dtype=np.uint8)
            b = self.bufs[self.i] = nb
        self.i = (self.i + 1) % self.depth
        return b

    def offsets(self, slot_of, n):
        """uint64 scratch of at least n entries tied to ring slot (reused: a fresh np.empty page-faults every block)."""
        key = id(slot_of)
        a = self.offs.get(key)
        if a is None or a.size < n:
            a = self.offs[key] = np.empty(n, dtype=np.uint64)
        return a


def _ids_from_spans(text, spans, n):
    out = []
    for r in range(n):
        toks = bytes(text[int(spans[2 * r]) + 1:int(spans[2 * r + 1])]).split()
        out.append(toks[0].decode("utf-8", "replace") if toks else "")
    return out


def _parse_fastq(lib, text, at_eof, out, want_ids, ring=None):
    n = len(text)
    cap_reads = n // 6 + 2
    offsets = ring.offsets(out, cap_reads + 1) if ring is not None else np.empty(cap_reads + 1, dtype=np.uint64)
    spans = np.empty(2 * cap_reads, dtype=np.uint64) if want_ids else None
    nreads, nbases, consumed = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _abi.check(lib.kdb_parse_fastq(ctypes.cast(ctypes.c_char_p(text), ctypes.c_void_p), n, 1 if at_eof else 0, out.ctypes.data, out.size, offsets.ctypes.data, cap_reads,
                                   spans.ctypes.data if want_ids else None,
                                   ctypes.byref(nreads), ctypes.byref(nbases), ctypes.byref(consumed)))
    nr = nreads.value
    ids = _ids_from_spans(memoryview(text), spans, nr) if want_ids else None
    return out[:nbases.value], offsets[:nr + 1], ids, consumed.value


def _parse_fastq_buf(lib, work, n, at_eof, out, want_ids, ring, off=0):
    """_parse_fastq for the n bytes at offset `off` of the bytearray `work` (no copy)."""
    cap_reads = n // 6 + 2
    offsets = ring.offsets(out, cap_reads + 1)
    spans = np.empty(2 * cap_reads, dtype=np.uint64) if want_ids else None
    nreads, nbases, consumed = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _abi.check(lib.kdb_parse_fastq_mt(_addr(work, off), n, 1 if at_eof else 0, out.ctypes.data, out.size, offsets.ctypes.data, cap_reads,
                                      spans.ctypes.data if want_ids else None,
                                      ctypes.byref(nreads), ctypes.byref(nbases), ctypes.byref(consumed), _split_threads()))
    nr = nreads.value
    ids = _ids_from_spans(memoryview(work)[off:off + n], spans, nr) if want_ids else None
    return out[:nbases.value], offsets[:nr + 1], ids, consumed.value


def _parse_fasta(lib, text, out, want_ids):
    n = len(text)
    cap_reads = text.count(b">") + 1
    offsets = np.empty(cap_reads + 1, dtype=np.uint64)
    spans = np.empty(2 * cap_reads, dtype=np.uint64) if want_ids else None
    nreads, nbases = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _abi.check(lib.kdb_parse_fasta(ctypes.cast(ctypes.c_char_p(text), ctypes.c_void_p), n, out.ctypes.data, out.size,
                                   offsets.ctypes.data, cap_reads, spans.ctypes.data if want_ids else None,
                                   ctypes.byref(nreads), ctypes.byref(nbases)))
    nr = nreads.value
    ids = _ids_from_spans(memoryview(text), spans, nr) if want_ids else None
    return out[:nbases.value], offsets[:nr + 1], ids


class BlockReader:
    """Iterable over (bases, offsets, ids|None) blocks of a FASTA/FASTQ file (parse.py:50-85).

    `bases` is a view into a ring of three buffers: it stays valid until three further blocks have been
    produced (enough for Engine.submit / submit_pinned, whose double-buffered pipeline has consumed block i
    by the time block i+2 has been submitted).  Copy it if you need it longer.  `.pinned` says whether the
    ring is pinned host memory (then Engine.submit_pinned can DMA from it directly).

    FASTA files are read whole by default (one block).  With `overlap` = k - 1 they are STREAMED in blocks: a record
    that does not end inside a block is held back and parsed with the next one; a record longer than a block comes in
    pieces -- the blocks after the first have .cont set and begin with the last `overlap` residues of the piece before
    (Engine.submit(..., continues=True)).  After the iteration .total_reads / .min_len / .max_len / .sum_len hold the
    record statistics of the file (pieces joined)."""

    def __init__(self, path, want_ids=False, block_bytes=None, pinned=False, overlap=None, hold_ring=False):
        if type(path) is not str:
            raise TypeError("BlockReader expects a fasta/fastq filepath as a str")
        if not os.path.exists(path) or not os.access(path, os.R_OK):
            raise ValueError("the filepath must be readable on the filesystem")       # parse.py:57-58
        if not (util.is_fasta(path) or util.is_fastq(path)):
            raise ValueError("Could not determine the format of file '{0}'".format(path))   # parse.py:74
        self.path, self.want_ids, self.block_bytes = path, want_ids, int(block_bytes or BLOCK_BYTES)
        self.overlap = overlap
        self._want_pinned = pinned
        self.pinned = False
        self._lib = _abi.lib()
        self.total_reads, self.min_len, self.max_len, self.sum_len = 0, None, 0, 0
        self._open_len = None            # length so far of a record that continues in the next block
        # hold_ring: the ring goes back to the pool only at release() -- for a consumer whose DMA still reads the last
        # blocks when the iteration ends (Engine.submit_pinned): it calls release() after its engine has synced
        self._hold_ring, self._held = hold_ring, []

    def _done(self, ring):
        if self._hold_ring:
            self._held.append(ring)
        else:
            _release_ring(ring)

    def release(self):
        """Give the ring back (hold_ring=True): call once nothing reads the yielded buffers any more."""
        for r in self._held:
            _release_ring(r)
        self._held = []

    def _account(self, lens, cont_prefix=None, open_last=False):
        """record statistics from the piece lengths of a block.  cont_prefix: record 0 is the next piece of the open
        record and repeats that many of its residues; open_last: the last record goes on in the next block."""
        lens = [int(x) for x in lens]
        done = []
        if cont_prefix is not None and lens:
            self._open_len += lens[0] - cont_prefix
            lens = lens[1:]
            if not lens and open_last:
                return
            done.append(self._open_len)
            self._open_len = None
        if open_last and lens:
            self._open_len = lens[-1]
            lens = lens[:-1]
        done += lens
        if not done:
            return
        self.total_reads += len(done)
        self.sum_len += sum(done)
        self.min_len = min(done) if self.min_len is None else min(self.min_len, min(done))
        self.max_len = max(self.max_len, max(done))

    def __iter__(self):
        lib, want_ids = self._lib, self.want_ids
        if util.is_fasta(self.path):
            if self.overlap is None:
                with _open(self.path) as f:
                    text = f.read()
                # one block per file: a single buffer (rounded up so that files of similar size share it)
                need = max(len(text), 1)
                ring = _get_ring(1 << (need - 1).bit_length(), self._want_pinned, 1) if need >= (1 << 20) else _Buffers(need, False, 1)
                self.pinned = ring.pinned
                try:
                    bases, offsets, ids = _parse_fasta(lib, text, ring.next(need), want_ids)
                    self._account(np.diff(offsets.astype(np.int64)))
                    yield Block(bases, offsets, ids)
                finally:
                    self._done(ring)
                return
            yield from self._stream_fasta()
            return
        # FASTQ: a thread reads the next blocks of text (file copy / inflate: outside the GIL) straight into reusable buffers while
        # this one splits the current block from there into the ring -- no bytes object per block.  What a block leaves unparsed
        # (the start of a record) goes in front of the next block's text.  A gzip stream is read in smaller blocks: its
        # inflating thread runs 64 MiB ahead at most.
        block = self.block_bytes
        if not util.is_gz_file(self.path):
            block = min(block, max(os.path.getsize(self.path), 1 << 16))     # (small files: small buffers)
        elif not is_bgzf(self.path):
            block = min(block, 32 << 20)
        reserve = min(1 << 20, block)
        ring = _get_ring(block + reserve + (1 << 20), self._want_pinned)
        self.pinned = ring.pinned
        try:
            with _open(self.path) as f:
                pf = _Prefetch(f, block, reserve)
                try:
                    carry = b""
                    first = True
                    while True:
                        b, got = pf.get()
                        if not got:
                            break
                        if first and b[reserve:reserve + 1] in (b"\n", b"\r"):
                            # Bio.SeqIO's FASTQ reader refuses a file whose first line is not a title ("Records in Fastq files should start
                            # with '@' character"); blank lines BEHIND a record it absorbs, and so does the splitter
                            raise ValueError("Records in Fastq files should start with '@' character")
                        first = False
                        if len(carry) <= reserve:
                            work, off = b, reserve - len(carry)
                            work[off:reserve] = carry
                        else:                                      # (a record longer than the reserve: one copy)
                            work, off = bytearray(carry) + b[reserve:reserve + got], 0
                        n = len(carry) + got
                        bases, offsets, ids, consumed = _parse_fastq_buf(lib, work, n, False, ring.next(n), want_ids, ring, off)
                        carry = bytes(work[off + consumed:off + n])
                        pf.release(b)
                        if len(offsets) > 1:
                            self._account_fast(offsets)
                            yield Block(bases, offsets, ids)
                    if carry.strip():
                        work = bytearray(carry)
                        bases, offsets, ids, _ = _parse_fastq_buf(lib, work, len(work), True, ring.next(len(work)), want_ids, ring)
                        if len(offsets) > 1:
                            self._account_fast(offsets)
                            yield Block(bases, offsets, ids)
                finally:
                    pf.close()
        finally:
            self._done(ring)

    def _account_fast(self, offsets):
        """_account for a block of whole records (vectorised: FASTQ blocks hold a million records)."""
        lens = np.diff(offsets.astype(np.int64))
        if lens.size == 0:
            return
        self.total_reads += int(lens.size)
        self.sum_len += int(lens.sum())
        lo, hi = int(lens.min()), int(lens.max())
        self.min_len = lo if self.min_len is None else min(self.min_len, lo)
        self.max_len = max(self.max_len, hi)

    def _stream_fasta(self):
        lib, want_ids, ov = self._lib, self.want_ids, int(self.overlap)
        ring = _get_ring(self.block_bytes + (1 << 20) + ov, self._want_pinned)
        self.pinned = ring.pinned
        try:
            yield from self._stream_fasta_blocks(ring, lib, want_ids, ov)
        finally:
            self._done(ring)

    def _stream_fasta_blocks(self, ring, lib, want_ids, ov):
        in_record = False                 # a record has been emitted in part and goes on
        fa_state = 0                      # the parser's state at the end of the last chunk (1 in a record, 2 in the middle of a line)
        tail = b""                        # its last residues (up to `ov` of them): the next piece starts with them
        carry = b""
        with _open(self.path) as f:
            eof = False
            while not eof:
                chunk = f.read(self.block_bytes)
                eof = not chunk
                text = carry + chunk if carry else chunk
                carry = b""
                if not text:
                    break
                # A record that starts in this chunk and may not end in it is held back: parse up to its header only.
                # Without any later header the chunk is (part of) ONE record longer than a chunk: it goes out in pieces.
                cut = len(text)
                if not eof:
                    h = text.rfind(b"\n>")
                    if h >= 0:
                        cut = h + 1
                piece = text if cut == len(text) else text[:cut]
                whole_lines = eof or cut < len(text)
                cap_reads = piece.count(b">") + 2
                out = ring.next(len(piece) + len(tail))
                offsets = np.empty(cap_reads + 1, dtype=np.uint64)
                spans = np.empty(2 * cap_reads, dtype=np.uint64) if want_ids else None
                nreads, nbases, consumed, in_out = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_int(0)
                cont = in_record
                pre = len(tail) if cont else 0                      # the overlap goes in front of a continuation piece
                _abi.check(lib.kdb_parse_fasta_chunk(ctypes.cast(ctypes.c_char_p(piece), ctypes.c_void_p), len(piece), 1 if whole_lines else 0,
                                                     fa_state if cont else 0, out.ctypes.data + pre, out.size - pre, offsets.ctypes.data, cap_reads,
                                                     spans.ctypes.data if want_ids else None, ctypes.byref(nreads), ctypes.byref(nbases),
                                                     ctypes.byref(consumed), ctypes.byref(in_out)))
                nr, nb = nreads.value, nbases.value
                if consumed.value < len(text):
                    carry = text[consumed.value:]
                offs = offsets[:nr + 1].copy()
                # the last record goes on in the next chunk only if the chunk could not hold it back
                open_last = bool(in_out.value) and not eof and cut == len(text) and nr > 0
                first = 0
                region = out[pre:]                                  # where the parser wrote
                if cont:
                    if int(offs[1]) == 0 and not (nr == 1 and open_last):
                        # nothing new for the open record (it ended at the chunk boundary): it is complete as it stands
                        self._account([pre], cont_prefix=pre)
                        first = 1
                        cont = False
                        pre = 0
                    else:
                        out[:pre] = np.frombuffer(tail, dtype=np.uint8)
                        offs[1:] += np.uint64(pre)
                        nb += pre
                        region = out
                offs = offs[first:]
                nr -= first
                if nr <= 0:
                    if not open_last:
                        in_record, fa_state = False, 0
                    continue
                ids = None
                if want_ids:
                    ids = _ids_from_spans(memoryview(piece), spans[2 * first:], nr)
                    if cont:
                        ids[0] = None
                self._account(np.diff(offs.astype(np.int64)), cont_prefix=pre if cont else None, open_last=open_last)
                if open_last:
                    last = region[int(offs[nr - 1]):nb]             # what has gone out of the open record in this block (with its prefix)
                    tail = bytes(last[-ov:]) if ov else b""
                in_record = open_last
                fa_state = in_out.value if open_last else 0
                base0 = int(offs[0])
                yield Block(region[base0:nb], offs - np.uint64(base0), ids, cont)
        if self._open_len is not None:                              # the file ended inside the open record: it is complete now
            self._account([0], cont_prefix=0)


def iter_blocks(path, want_ids=False, block_bytes=None, pinned=False):
    """Generator form of BlockReader."""
    return iter(BlockReader(path, want_ids=want_ids, block_bytes=block_bytes, pinned=pinned))


# ---------------------------------------------------------------------------------------------------------------
# Sharded reading (multi-GPU): the file is cut into byte ranges of `block_bytes`; block i is the run of
# whole records whose header line starts inside range i, and belongs to rank i % world.  Every rank finds the same
# record boundaries on its own (FASTQ: a line starting with '@' whose second-next line starts with '+'; FASTA: a line
# starting with '>'), so the blocks partition the records exactly and no rank reads (plain files) or splits (gzip
# streams, which cannot be entered in the middle) what it does not own.
# ---------------------------------------------------------------------------------------------------------------
class _ForwardSource:
    """Byte source over a plain or gzip file that serves non-decreasing windows [lo, hi) of the (decompressed) stream."""

    def __init__(self, path):
        self.gz = util.is_gz_file(path)
        self.f = _open(path)
        self.buf = bytearray()
        self.start = 0                  # stream offset of buf[0]
        self.eof = False

    def close(self):
        self.f.close()

    def ensure(self, lo, hi):
        """Make the buffer cover [lo, hi) (clipped at EOF); everything before lo is dropped.  lo must not decrease."""
        assert lo >= self.start
        end = self.start + len(self.buf)
        if lo >= end:                   # skip ahead
            gap = lo - end
            self.buf = bytearray()
            if self.gz:
                while gap and not self.eof:
                    got = len(self.f.read(min(gap, 8 << 20)))
                    if got == 0:
                        self.eof = True
                    gap -= got
            else:
                self.f.seek(lo)
            self.start = lo if not self.eof or not self.gz else lo - gap
        elif lo > self.start:
            del self.buf[:lo - self.start]
            self.start = lo
        while not self.eof and self.start + len(self.buf) < hi:
            chunk = self.f.read(max(hi - self.start - len(self.buf), 1 << 20))
            if not chunk:
                self.eof = True
            else:
                self.buf += chunk
        return self.start + len(self.buf)

    def find_record_start(self, pos, lo, fastq):
        """Stream offset of the first record header whose line starts at or after `pos` (the EOF offset if none).
        The buffer keeps covering [lo, ...); lo <= max(pos - 1, 0)."""
        marker = 0x40 if fastq else 0x3E
        want = max(pos, 1) + (1 << 16)

        def more():
            nonlocal want
            if self.eof:
                return False
            want = max(want, self.start + len(self.buf)) + (want - pos)       # doubles the look-ahead
            self.ensure(lo, want)
            return True

        self.ensure(lo, want)
        # 1. the first line start at or after pos
        if pos == 0:
            i = 0 - self.start if self.start == 0 else None
            assert i is not None, "block 0 must be scanned from the start of the stream"
        else:
            while True:
                nl = self.buf.find(b"\n", pos - 1 - self.start)
                if nl >= 0:
                    i = nl + 1
                    break
                if not more():
                    return self.start + len(self.buf)
        # 2. walk the lines until one is a record header
        while True:
            b = self.buf
            n = len(b)
            if i >= n:
                if not more():
                    return self.start + len(self.buf)
                continue
            nl1 = b.find(b"\n", i)
            if b[i] == marker:
                if not fastq:
                    return self.start + i
                # '@' also starts quality lines; only a header is followed, two lines later, by a '+' line
                nl2 = b.find(b"\n", nl1 + 1) if nl1 >= 0 else -1
                if nl1 < 0 or nl2 < 0 or nl2 + 1 >= n:
                    if more():
                        continue
                elif b[nl2 + 1] == 0x2B:
                    return self.start + i
            if nl1 < 0:
                if not more():
                    return self.start + len(self.buf)
                continue
            i = nl1 + 1


class _BgzfShardSource(_ForwardSource):
    """_ForwardSource over a BGZF file that inflates only what is asked for: the members' offsets come from their headers
    and trailers (kdb_bgzf_scan), so skipping ahead is a seek, and a rank of a multi-GPU job inflates the members that
    hold its own blocks (plus the look-ahead to the next record start) instead of the whole stream."""

    def __init__(self, path):
        self.gz = True
        self.lib = _abi.lib()
        # typical members hold tens of KB; a file of tiny members makes kdb_bgzf_scan answer KDB_ERR_NOMEM and the index doubles
        # (sized for the worst case -- 28-byte members -- the two arrays asked for 0.57 x the file size of address space per rank)
        cap = os.path.getsize(path) // 4096 + 16
        while True:
            self.coff = np.empty(cap, dtype=np.uint64)
            self.uoff = np.empty(cap, dtype=np.uint64)
            n = ctypes.c_size_t(0)
            rc = self.lib.kdb_bgzf_scan(path.encode(), self.coff.ctypes.data, self.uoff.ctypes.data, cap, ctypes.byref(n))
            if rc == _abi.KDB_ERR_NOMEM:
                cap *= 2
                continue
            _abi.check(rc)
            break
        self.nmem = n.value - 1
        self.coff, self.uoff = self.coff[:n.value].astype(np.int64), self.uoff[:n.value].astype(np.int64)
        self.f = open(path, "rb")
        self.buf = bytearray()
        self.start = 0
        self.eof = self.nmem == 0
        self.mi = 0                                         # the next member to inflate
        self.threads = _host_threads()
        self.inflated = 0                                   # uncompressed bytes produced so far (tests: a rank inflates its share only)

    def _inflate_members(self, m0, m1):
        c0, c1 = int(self.coff[m0]), int(self.coff[m1])
        self.f.seek(c0)
        comp = self.f.read(c1 - c0)
        want = int(self.uoff[m1] - self.uoff[m0])
        out = np.empty(max(want, 1), dtype=np.uint8)
        consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _abi.check(self.lib.kdb_bgzf_inflate(ctypes.cast(ctypes.c_char_p(comp), ctypes.c_void_p), len(comp), out.ctypes.data, out.size,
                                             self.threads, ctypes.byref(consumed), ctypes.byref(produced)))
        if consumed.value != len(comp) or produced.value != want:
            raise ValueError("truncated or corrupt BGZF file")
        self.inflated += want
        return out[:want]

    def ensure(self, lo, hi):
        assert lo >= self.start
        end = self.start + len(self.buf)
        if lo >= end:                                       # skip ahead: straight to the member that holds `lo`
            m = int(np.searchsorted(self.uoff, lo, side="right")) - 1
            m = min(max(m, self.mi), self.nmem)
            self.buf = bytearray()
            self.mi = m
            self.start = int(self.uoff[m])
            self.eof = m >= self.nmem
        while not self.eof and self.start + len(self.buf) < hi:
            need = hi - self.start - len(self.buf)
            m1 = int(np.searchsorted(self.uoff, int(self.uoff[self.mi]) + max(need, 4 << 20), side="left"))
            m1 = min(max(m1, self.mi + 1), self.nmem)
            self.buf += self._inflate_members(self.mi, m1).tobytes()
            self.mi = m1
            self.eof = m1 >= self.nmem
        if lo > self.start:
            drop = min(lo - self.start, len(self.buf))
            del self.buf[:drop]
            self.start += drop
        return self.start + len(self.buf)


def _forward_source(path):
    if util.is_gz_file(path) and is_bgzf(path):
        try:
            return _BgzfShardSource(path)
        except (ValueError, OSError, MemoryError, _abi.KdbHipError):      # (not really BGZF throughout, no room for the member index, or no native library: the stream reader does it)
            pass
    return _ForwardSource(path)


def _addr(buf, off):
    """Address of buf[off] for a bytes / bytearray object (no copy)."""
    if isinstance(buf, bytearray):
        return ctypes.addressof(ctypes.c_char.from_buffer(buf, off)) if off < len(buf) else 0
    return ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value + off


class ShardedBlockReader:
    """iter_blocks for rank `rank` of `world`: yields (bases, offsets, ids|None) for the blocks this rank owns.
    Same buffer-lifetime rules as BlockReader."""

    def __init__(self, path, rank, world, want_ids=False, block_bytes=BLOCK_BYTES, pinned=False, hold_ring=False):
        if type(path) is not str:
            raise TypeError("ShardedBlockReader expects a fasta/fastq filepath as a str")
        if not os.path.exists(path) or not os.access(path, os.R_OK):
            raise ValueError("the filepath must be readable on the filesystem")       # parse.py:57-58
        if not (util.is_fasta(path) or util.is_fastq(path)):
            raise ValueError("Could not determine the format of file '{0}'".format(path))   # parse.py:74
        if not (0 <= rank < world):
            raise ValueError("rank {0} outside world of {1}".format(rank, world))
        self.path, self.rank, self.world = path, rank, world
        self.want_ids, self.block_bytes = want_ids, int(block_bytes)
        self._want_pinned = pinned
        self.pinned = False
        self._lib = _abi.lib()
        self._hold_ring, self._held = hold_ring, []

    def release(self):
        for r in self._held:
            _release_ring(r)
        self._held = []

    def __iter__(self):
        lib, B, fastq = self._lib, self.block_bytes, util.is_fastq(self.path)
        ring = _get_ring(B + (1 << 20), self._want_pinned)
        self.pinned = ring.pinned
        src = self._src = _forward_source(self.path)
        try:
            i = self.rank
            while True:
                lo = max(i * B - 1, 0)
                if src.ensure(lo, lo + 1) <= lo and src.eof:
                    break                                           # the stream ended before this block
                s0 = src.find_record_start(i * B, lo, fastq)
                e0 = src.find_record_start((i + 1) * B, lo, fastq) if s0 < (i + 1) * B else s0
                if e0 > s0:
                    src.ensure(lo, e0)
                    off, n = s0 - src.start, e0 - s0
                    out = ring.next(n)
                    if fastq:
                        yield self._fastq(lib, src.buf, off, n, out, ring)
                    else:
                        yield self._fasta(lib, src.buf, off, n, out)
                if src.eof and src.start + len(src.buf) <= (i + 1) * B:
                    break
                i += self.world
        finally:
            src.close()
            if self._hold_ring:
                self._held.append(ring)
            else:
                _release_ring(ring)

    def _fastq(self, lib, buf, off, n, out, ring):
        cap_reads = n // 6 + 2
        offsets = ring.offsets(out, cap_reads + 1)
        spans = np.empty(2 * cap_reads, dtype=np.uint64) if self.want_ids else None
        nreads, nbases, consumed = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        _abi.check(lib.kdb_parse_fastq(_addr(buf, off), n, 1, out.ctypes.data, out.size, offsets.ctypes.data, cap_reads,
                                       spans.ctypes.data if self.want_ids else None,
                                       ctypes.byref(nreads), ctypes.byref(nbases), ctypes.byref(consumed)))
        nr = nreads.value
        ids = _ids_from_spans(memoryview(buf)[off:off + n], spans, nr) if self.want_ids else None
        return out[:nbases.value], offsets[:nr + 1], ids

    def _fasta(self, lib, buf, off, n, out):
        cap_reads = buf.count(b">", off, off + n) + 1
        offsets = np.empty(cap_reads + 1, dtype=np.uint64)
        spans = np.empty(2 * cap_reads, dtype=np.uint64) if self.want_ids else None
        nreads, nbases = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _abi.check(lib.kdb_parse_fasta(_addr(buf, off), n, out.ctypes.data, out.size, offsets.ctypes.data, cap_reads,
                                       spans.ctypes.data if self.want_ids else None, ctypes.byref(nreads), ctypes.byref(nbases)))
        nr = nreads.value
        ids = _ids_from_spans(memoryview(buf)[off:off + n], spans, nr) if self.want_ids else None
        return out[:nbases.value], offsets[:nr + 1], ids


def iter_blocks_sharded(path, rank, world, want_ids=False, block_bytes=BLOCK_BYTES, pinned=False):
    return iter(ShardedBlockReader(path, rank, world, want_ids=want_ids, block_bytes=block_bytes, pinned=pinned))
