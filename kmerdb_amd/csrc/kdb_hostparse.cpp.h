// kdb_hostparse.cpp.h -- host-side record splitter (no GPU): FASTQ / FASTA text -> flat residue buffer + offsets.
// Takes over what Bio.SeqIO.parse does for kmerdb/parse.py:50-85 on the path: record ids are the first
// whitespace token of the header, FASTA lines are concatenated (blanks / CR stripped), case is preserved.
// memchr-driven; one pass over the text.
#pragma once
#include <cstdint>
#include <cstring>

namespace kdbhost {

// returns 0 ok; 1 malformed; 2 output capacity exceeded
inline int parse_fastq(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                       uint64_t *hdr /* optional [2*cap_reads]: header line start/end */, size_t *nreads_out, size_t *nbases_out,
                       size_t *consumed_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    offs[0] = 0;
    while (pos < n) {
        // skip blank lines between records
        if (text[pos] == '\n' || text[pos] == '\r') { pos++; continue; }
        const uint8_t *l0 = text + pos;
        const uint8_t *e0 = (const uint8_t *)memchr(l0, '\n', n - pos);
        if (!e0) break;
        const uint8_t *l1 = e0 + 1;
        const uint8_t *e1 = (const uint8_t *)memchr(l1, '\n', (size_t)(text + n - l1));
        if (!e1) break;
        const uint8_t *l2 = e1 + 1;
        const uint8_t *e2 = (const uint8_t *)memchr(l2, '\n', (size_t)(text + n - l2));
        if (!e2) break;
        const uint8_t *l3 = e2 + 1;
        const uint8_t *e3 = (const uint8_t *)memchr(l3, '\n', (size_t)(text + n - l3));
        if (!e3) { if (!at_eof) break; e3 = text + n; }          // last record of the file without trailing newline
        if (*l0 != '@') { *why = "FASTQ record does not start with '@'"; return 1; }
        if (l2 >= text + n || *l2 != '+') { *why = "FASTQ third line does not start with '+'"; return 1; }
        size_t slen = (size_t)(e1 - l1), qlen = (size_t)(e3 - l3);
        if (slen && l1[slen - 1] == '\r') slen--;
        if (qlen && l3[qlen - 1] == '\r') qlen--;
        if (slen != qlen) { *why = "FASTQ sequence and quality lengths differ"; return 1; }
        if (nreads >= cap_reads || nb + slen > bases_cap) { *why = "output capacity exceeded"; return 2; }
        memcpy(bases + nb, l1, slen);
        nb += slen;
        if (hdr) { size_t hl = (size_t)(e0 - l0); if (hl && l0[hl - 1] == '\r') hl--; hdr[2 * nreads] = (uint64_t)(l0 - text); hdr[2 * nreads + 1] = (uint64_t)(l0 - text) + hl; }
        offs[++nreads] = nb;
        pos = (size_t)(e3 - text) + (e3 < text + n ? 1 : 0);
    }
    if (at_eof) {
        // anything left must be blank
        for (size_t i = pos; i < n; i++)
            if (text[i] != '\n' && text[i] != '\r' && text[i] != ' ') { *why = "truncated FASTQ record at end of file"; return 1; }
        pos = n;
    }
    *nreads_out = nreads; *nbases_out = nb; *consumed_out = pos;
    return 0;
}

inline int parse_fasta(const uint8_t *text, size_t n, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                       uint64_t *hdr, size_t *nreads_out, size_t *nbases_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    bool in_record = false;
    offs[0] = 0;
    while (pos < n) {
        const uint8_t *l = text + pos;
        const uint8_t *e = (const uint8_t *)memchr(l, '\n', n - pos);
        size_t len = e ? (size_t)(e - l) : n - pos;
        pos += len + (e ? 1 : 0);
        if (len && l[len - 1] == '\r') len--;
        if (len && l[0] == '>') {
            if (in_record) offs[nreads] = nb;
            if (nreads >= cap_reads) { *why = "output capacity exceeded"; return 2; }
            if (hdr) { hdr[2 * nreads] = (uint64_t)(l - text); hdr[2 * nreads + 1] = (uint64_t)(l - text) + len; }
            nreads++;
            in_record = true;
            continue;
        }
        if (!in_record || !len) continue;                      // text before the first header is ignored
        if (nb + len > bases_cap) { *why = "output capacity exceeded"; return 2; }
        // copy the line, dropping blanks inside it (Biopython strips spaces and CR from FASTA sequence lines)
        if (!memchr(l, ' ', len) && !memchr(l, '\t', len) && !memchr(l, '\r', len)) {
            memcpy(bases + nb, l, len);
            nb += len;
        } else {
            for (size_t i = 0; i < len; i++) { const uint8_t c = l[i]; if (c != ' ' && c != '\t' && c != '\r') bases[nb++] = c; }
        }
    }
    if (in_record) offs[nreads] = nb;
    *nreads_out = nreads; *nbases_out = nb;
    return 0;
}

}  // namespace kdbhost
