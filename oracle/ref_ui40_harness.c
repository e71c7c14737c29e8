/*
 * ref_ui40_harness.c -- exports the reference's own static-inline ui40
 * helpers (psascan/sa_use.h:17-46) so tests can compare the restatement and
 * the product's .sa5 reader against them.  This file contains no reference
 * code: it only includes the reference header where it lies.
 * TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/ (git-ignored).
 */
#include "psascan/sa_use.h"

unsigned long ref_ui40_sizeof(void) { return sizeof(ui40_t); }

uint64_t ref_ui40_from_bytes_convert(const uint8_t *pos) {
    return ui40_convert(from_bytes(pos));
}

/* ui40_fread (sa_use.h:31-46) over a file path; values written as u64 */
size_t ref_ui40_fread_path(const char *path, uint64_t *out, size_t nitems) {
    FILE *fp = fopen(path, "r");
    if (!fp) return 0;
    ui40_t *buf = (ui40_t *) malloc(sizeof(ui40_t) * (nitems ? nitems : 1));
    size_t n = ui40_fread(buf, nitems, fp);
    fclose(fp);
    for (size_t i = 0; i < n; ++i) out[i] = ui40_convert(buf[i]);
    free(buf);
    return n;
}
