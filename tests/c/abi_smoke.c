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
    /* KDB_SMOKE_ENGINES=n: deal the records out over n engines (engine j on device j mod the device count) and sum
     * their vectors with kdb_reduce -- the single-process multi-GPU form of SURVEY 8(e) */
    const char *ne = getenv("KDB_SMOKE_ENGINES");
    const int n = ne ? atoi(ne) : 1;
    if (n < 1 || n > KDB_REDUCE_MAX) { fprintf(stderr, "KDB_SMOKE_ENGINES=%d\n", n); return 2; }
    int ndev = 0;
    if (kdb_device_count(&ndev) != KDB_OK || ndev < 1) { fprintf(stderr, "kdb_device_count: %s\n", kdb_last_error()); return 1; }
    kdb_engine *eng[KDB_REDUCE_MAX] = {NULL};
    int rc = KDB_OK;
    for (int j = 0; j < n && rc == KDB_OK; j++) rc = kdb_create(k, canonical, KDB_N_DROP, j % ndev, NULL, &eng[j]);
    if (rc != KDB_OK) { fprintf(stderr, "kdb_create: %s\n", kdb_last_error()); for (int j = 0; j < n; j++) kdb_destroy(eng[j]); return 1; }
    kdb_engine *e = eng[0];
    const uint64_t nbins = 1ull << (2 * k);
    uint64_t *counts = (uint64_t *)malloc(nbins * sizeof *counts);
    uint64_t total = 0, unique = 0;
    if (n == 1) rc = kdb_submit(e, bases, nbytes, offs, nreads);
    else {
        /* records r with r mod n == j go to engine j, one submit per record (offsets rebased to the record) */
        for (size_t r = 0; r < nreads && rc == KDB_OK; r++) {
            const uint64_t o2[2] = {0, offs[r + 1] - offs[r]};
            rc = kdb_submit(eng[r % (size_t)n], bases + offs[r], (size_t)o2[1], o2, 1);
        }
        if (rc == KDB_OK) rc = kdb_reduce(eng, n, 0);
    }
    if (rc == KDB_OK) rc = kdb_finish(e, counts, &total, &unique);
    if (rc != KDB_OK) { fprintf(stderr, "status %d: %s\n", rc, kdb_last_error()); for (int j = 0; j < n; j++) kdb_destroy(eng[j]); return 10 + rc; }
    printf("total %llu unique %llu\n", (unsigned long long)total, (unsigned long long)unique);
    for (uint64_t i = 0; i < nbins; i++)
        if (counts[i]) printf("%llu %llu\n", (unsigned long long)i, (unsigned long long)counts[i]);
    for (int j = 0; j < n; j++) kdb_destroy(eng[j]);
    free(counts); free(offs); free(bases);
    return 0;
}
