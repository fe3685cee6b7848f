"""Stand-in for Bio.bgzf (Biopython is not installed here; this is our code, not the reference's): just enough of
BgzfWriter for kmerdb's KDBWriter / KDBGWriter subclasses to run when the golden vectors are generated -- BGZF
framing per the SAM/BAM specification (gzip member with a 'BC' extra field holding the block size)."""
import struct
import zlib


class BgzfWriter:
    def __init__(self, filename=None, mode="w", fileobj=None, compresslevel=6):
        self._handle = fileobj if fileobj is not None else open(filename, "wb")
        self._text = "b" not in mode.lower()
        self._buffer = b""
        self.compresslevel = compresslevel

    def _write_block(self, block):
        assert len(block) <= 65536
        c = zlib.compressobj(self.compresslevel, zlib.DEFLATED, -15, zlib.DEF_MEM_LEVEL, 0)
        compressed = c.compress(block) + c.flush()
        del c
        assert len(compressed) < 65536
        crc = struct.pack("<I", zlib.crc32(block) & 0xFFFFFFFF)
        bsize = struct.pack("<H", len(compressed) + 25)
        uncompressed_length = struct.pack("<I", len(block))
        self._handle.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00\x42\x43\x02\x00" + bsize + compressed + crc
                           + uncompressed_length)

    def write(self, data):
        if isinstance(data, str):
            data = data.encode("latin-1")
        self._buffer += data
        while len(self._buffer) >= 65536:
            self._write_block(self._buffer[:65536])
            self._buffer = self._buffer[65536:]

    def flush(self):
        while len(self._buffer) >= 65536:
            self._write_block(self._buffer[:65535])
            self._buffer = self._buffer[65535:]
        self._write_block(self._buffer)
        self._buffer = b""
        self._handle.flush()

    def close(self):
        if self._buffer:
            self.flush()
        self._handle.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")
        self._handle.flush()
        self._handle.close()


class BgzfReader:
    pass


class BgzfBlocks:
    pass
