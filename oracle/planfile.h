/*
 * planfile.h — on-disk container for one H·psi GEMM-pair plan (test infrastructure).
 *
 * Layout (little endian):
 *   char     magic[8] = "B2XPLAN1"
 *   uint64_t n_pairs, psi_len, sigma_len, max_work, arena_len, n_ranges, flags, n_meta
 *   b2x_pair pairs[n_pairs]                      (include/b2x.h)
 *   uint64_t ranges[n_ranges][2]                 (arena_off, len) — contiguous operator ranges
 *   double   meta[n_meta]                        (free-form: const_e, energy, site, sweep, ...)
 *   double   arena[arena_len]      if flags & 1
 *   double   psi[psi_len]          if flags & 2
 *   double   sigma_ref[sigma_len]  if flags & 4  (reference  sigma = H psi, scale 1)
 *   double   diag[psi_len]         if flags & 8
 *   double   psi_out[psi_len]      if flags & 16 (reference Davidson eigenvector)
 *
 * Used by: oracle/ref_dump.cpp + oracle/ref_replay.cpp (built against the reference into
 * oracle/_ref/), oracle/hpsi_oracle.c, tests/ (numpy reader in tests/planio.py).
 */
#ifndef B2X_PLANFILE_H
#define B2X_PLANFILE_H
#include "../include/b2x.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define B2XPF_ARENA 1u
#define B2XPF_PSI 2u
#define B2XPF_SIGMA 4u
#define B2XPF_DIAG 8u
#define B2XPF_PSIOUT 16u

typedef struct b2x_planfile {
    uint64_t n_pairs, psi_len, sigma_len, max_work, arena_len, n_ranges, flags, n_meta;
    b2x_pair *pairs;
    uint64_t *ranges;
    double *meta, *arena, *psi, *sigma_ref, *diag, *psi_out;
} b2x_planfile;

static inline int b2x_planfile_write(const char *fn, const b2x_planfile *p) {
    FILE *f = fopen(fn, "wb");
    if (!f)
        return -1;
    fwrite("B2XPLAN1", 1, 8, f);
    fwrite(&p->n_pairs, 8, 8, f);
    fwrite(p->pairs, sizeof(b2x_pair), p->n_pairs, f);
    fwrite(p->ranges, 16, p->n_ranges, f);
    fwrite(p->meta, 8, p->n_meta, f);
    if (p->flags & B2XPF_ARENA)
        fwrite(p->arena, 8, p->arena_len, f);
    if (p->flags & B2XPF_PSI)
        fwrite(p->psi, 8, p->psi_len, f);
    if (p->flags & B2XPF_SIGMA)
        fwrite(p->sigma_ref, 8, p->sigma_len, f);
    if (p->flags & B2XPF_DIAG)
        fwrite(p->diag, 8, p->psi_len, f);
    if (p->flags & B2XPF_PSIOUT)
        fwrite(p->psi_out, 8, p->psi_len, f);
    return fclose(f);
}

static inline double *b2xpf_rd_(FILE *f, uint64_t n) {
    double *d = (double *)malloc(n ? n * 8 : 8);
    if (n && fread(d, 8, n, f) != n) {
        free(d);
        return NULL;
    }
    return d;
}

static inline int b2x_planfile_read(const char *fn, b2x_planfile *p) {
    char magic[8];
    FILE *f = fopen(fn, "rb");
    if (!f)
        return -1;
    memset(p, 0, sizeof(*p));
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "B2XPLAN1", 8) != 0 ||
        fread(&p->n_pairs, 8, 8, f) != 8) {
        fclose(f);
        return -2;
    }
    p->pairs = (b2x_pair *)malloc(p->n_pairs ? p->n_pairs * sizeof(b2x_pair) : 8);
    if (fread(p->pairs, sizeof(b2x_pair), p->n_pairs, f) != p->n_pairs) {
        fclose(f);
        return -3;
    }
    p->ranges = (uint64_t *)malloc(p->n_ranges ? p->n_ranges * 16 : 8);
    if (fread(p->ranges, 16, p->n_ranges, f) != p->n_ranges) {
        fclose(f);
        return -3;
    }
    p->meta = b2xpf_rd_(f, p->n_meta);
    if (p->flags & B2XPF_ARENA)
        p->arena = b2xpf_rd_(f, p->arena_len);
    if (p->flags & B2XPF_PSI)
        p->psi = b2xpf_rd_(f, p->psi_len);
    if (p->flags & B2XPF_SIGMA)
        p->sigma_ref = b2xpf_rd_(f, p->sigma_len);
    if (p->flags & B2XPF_DIAG)
        p->diag = b2xpf_rd_(f, p->psi_len);
    if (p->flags & B2XPF_PSIOUT)
        p->psi_out = b2xpf_rd_(f, p->psi_len);
    fclose(f);
    return 0;
}

static inline void b2x_planfile_free(b2x_planfile *p) {
    free(p->pairs), free(p->ranges), free(p->meta), free(p->arena), free(p->psi);
    free(p->sigma_ref), free(p->diag), free(p->psi_out);
    memset(p, 0, sizeof(*p));
}
#endif
