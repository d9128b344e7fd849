/* verify_shim.c -- ours; linked with the reference's verify.c (compiled with
 * -Dfprintf=smol_verify_fprintf) by build_ref_with_hip.sh. verify.c's fourth suite fails on
 * the reference itself and prints ~415 million mismatch lines (32 GB); this shim hashes
 * everything (FNV-1a-64), echoes only the first lines, and -- when VERIFY_LINE_BUDGET is set --
 * ends the run with exit code 100 once that many lines were produced, so that three builds can
 * be compared in seconds. With the budget unset the whole run is hashed (about 3 minutes). */

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t hash = 0xcbf29ce484222325ull;
static unsigned long long n_lines, n_bytes;
static long long budget = -1;

static void
summary (void)
{
    printf ("[verify_shim] lines=%llu bytes=%llu fnv1a64=%016llx\n", n_lines, n_bytes, (unsigned long long) hash);
    fflush (stdout);
}

int
smol_verify_fprintf (FILE *f, const char *fmt, ...)
{
    char buf [1024];
    va_list ap;
    int n, i;

    if (budget < 0)
    {
        const char *e = getenv ("VERIFY_LINE_BUDGET");
        budget = e ? atoll (e) : 0;
        atexit (summary);
    }

    va_start (ap, fmt);
    n = vsnprintf (buf, sizeof (buf), fmt, ap);
    va_end (ap);
    if (n < 0)
        return n;
    if (n >= (int) sizeof (buf))
        n = sizeof (buf) - 1;

    if (n_lines < 48)
        fwrite (buf, 1, (size_t) n, f);
    for (i = 0; i < n; i++)
    {
        hash = (hash ^ (unsigned char) buf [i]) * 0x100000001b3ull;
        if (buf [i] == '\n')
            n_lines++;
    }
    n_bytes += (unsigned) n;

    if (budget > 0 && n_lines >= (unsigned long long) budget)
        exit (100);
    return n;
}
