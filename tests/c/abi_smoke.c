/* Plain C user of the C ABI (include/kdbhip.h): counts the records given on stdin-free argv and prints the
 * non-zero bins, so that the pytest wrapper can compare them with the oracle.  No Python, no torch. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kdbhip.h"

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s k canonical record [record ...]\n", argv[0]); return 2; }
    const int k = atoi(argv[1]), canonical = atoi(argv[2]);
    const size_t nreads = (size_t)(argc - 3);
    size_t nbytes = 0;
    for (size_t r = 0; r < nreads; r++) nbytes += strlen(argv[3 + r]);
    uint8_t *bases = (uint8_t *)malloc(nbytes + 1);
    uint64_t *offs = (uint64_t *)malloc((nreads + 1) * sizeof *offs);
    offs[0] = 0;
    for (size_t r = 0; r < nreads; r++) {
        const size_t len = strlen(argv[3 + r]);
        memcpy(bases + offs[r], argv[3 + r], len);
        offs[r + 1] = offs[r] + len;
    }
    kdb_engine *e = NULL;
    int rc = kdb_create(k, canonical, KDB_N_DROP, 0, NULL, &e);
    if (rc != KDB_OK) { fprintf(stderr, "kdb_create: %s\n", kdb_last_error()); return 1; }
    const uint64_t nbins = 1ull << (2 * k);
    uint64_t *counts = (uint64_t *)malloc(nbins * sizeof *counts);
    uint64_t total = 0, unique = 0;
    rc = kdb_submit(e, bases, nbytes, offs, nreads);
    if (rc == KDB_OK) rc = kdb_finish(e, counts, &total, &unique);
    if (rc != KDB_OK) { fprintf(stderr, "status %d: %s\n", rc, kdb_last_error()); kdb_destroy(e); return 10 + rc; }
    printf("total %llu unique %llu\n", (unsigned long long)total, (unsigned long long)unique);
    for (uint64_t i = 0; i < nbins; i++)
        if (counts[i]) printf("%llu %llu\n", (unsigned long long)i, (unsigned long long)counts[i]);
    kdb_destroy(e);
    free(counts); free(offs); free(bases);
    return 0;
}
