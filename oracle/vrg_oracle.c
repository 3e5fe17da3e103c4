/*
 * vrg_oracle.c - CPU restatement of the reference's variational region growing.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (arterynetwork_amd/) never links, imports or falls back to it.
 *
 * It follows /root/reference/Code/variationalRegionGrowing.py statement by statement
 * (sequential flip processing, in-place label mutation, Python-list order semantics);
 * every function cites the lines it restates.  Parity is PINNED: tests/test_oracle_golden.py
 * checks it against tests/golden/*.npz, which tests/golden/make_goldens.py produced by
 * running the real reference (its two self-tests :284-314, BASELINE config 1 and
 * adversarial volumes) in the build container.
 *
 * Index convention: the reference indexes arrays as [x][y][z]; "lex" index here is
 * (x*ny + y)*nz + z, the order np.where returns (variationalRegionGrowing.py:44).
 *
 * density_mode 0: brute-force sums over voxels exactly as :149-155, :236-255, added up with numpy's
 *   pairwise np.sum scheme in float64 (no extended precision) so rounding follows the reference.
 * density_mode 1: the same sums regrouped by distinct intensity value (histogram over the
 *   sorted unique values); mathematically identical, rounding differs at the 1e-15 level.
 *   Used so that medium-size volumes finish in seconds; tests check mode 1 against mode 0.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VRGO_OK 0
#define VRGO_STOP_CONVERGED 1   /* :91-96  no flipped points          */
#define VRGO_STOP_TIME 2        /* :97-100 wall clock cap             */
#define VRGO_STOP_SIZE 3        /* :101-104 len(segmented) >= maxSegmentSize */
#define VRGO_STOP_ITERMAX 4     /* :118-121 loop exhausted            */
#define VRGO_ERR -1

typedef struct {
    int32_t *items; /* lex index or -1 (tombstone) */
    int32_t *pos;   /* dense map lex -> position, -1 if absent */
    int64_t n, cap;
} olist;

typedef struct {
    int64_t nflip, nseg, n_in, n_out, ni, no;
    double sum_in, sum_out;
} vrgo_trace;

typedef struct vrgo {
    int nx, ny, nz;
    int64_t V;
    double H;
    int density_mode;
    double *data;
    uint8_t *label;      /* valueMap   :21  */
    uint8_t *seg;        /* segmentedMap :45 */
    double *ip, *op;     /* innerProb / outerProb dense, :132-133 */
    olist inner, outer, segl;
    int64_t innerSize, outerSize; /* :51-52, :115-116 */
    int64_t iterNum;              /* :57 */
    int stop;                     /* 0 while running */
    /* levels (mode 1) */
    int64_t L;
    double *lev;       /* sorted unique values */
    int32_t *levidx;   /* per voxel level index */
    int64_t *hin, *hout;
    /* trace */
    vrgo_trace *trace;
    int64_t ntrace, captrace;
    /* scratch */
    int32_t *flips; int64_t nflips, capflips;
    int32_t *newin, *newout, *incl; int64_t nnewin, nnewout, nincl, capnew, capnewout, capincl;
    double A;
    double t0;
    double *scr; int64_t capscr;
} vrgo;

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---- Python-list stand-in: append keeps order, remove leaves a tombstone, compaction keeps order */
static int ol_init(olist *l, int64_t V) {
    l->cap = 1024; l->n = 0;
    l->items = (int32_t *)malloc(sizeof(int32_t) * l->cap);
    l->pos = (int32_t *)malloc(sizeof(int32_t) * V);
    if (!l->items || !l->pos) return -1;
    for (int64_t i = 0; i < V; i++) l->pos[i] = -1;
    return 0;
}
static void ol_free(olist *l) { free(l->items); free(l->pos); }
static int ol_append(olist *l, int32_t v) {
    if (l->n == l->cap) {
        l->cap *= 2;
        l->items = (int32_t *)realloc(l->items, sizeof(int32_t) * l->cap);
        if (!l->items) return -1;
    }
    l->items[l->n] = v; l->pos[v] = (int32_t)l->n; l->n++;
    return 0;
}
static int ol_remove(olist *l, int32_t v) { /* list.remove(v): v is unique in the list */
    int32_t p = l->pos[v];
    if (p < 0) return -1;                   /* Python would raise ValueError */
    l->items[p] = -1; l->pos[v] = -1;
    return 0;
}
static void ol_compact(olist *l) {
    int64_t w = 0;
    for (int64_t r = 0; r < l->n; r++) {
        int32_t v = l->items[r];
        if (v >= 0) { l->items[w] = v; l->pos[v] = (int32_t)w; w++; }
    }
    l->n = w;
}

/* get_neighbours(p, exclude_p=True, shape) :263-282 - offsets in lexicographic (dx,dy,dz)
 * order, centre removed, out-of-bounds rows dropped. Returns the count. */
static int neighbours(const vrgo *o, int32_t p, int32_t *out) {
    int z = p % o->nz, y = (p / o->nz) % o->ny, x = p / (o->nz * o->ny);
    int n = 0;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                if (!dx && !dy && !dz) continue;
                int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx < 0 || yy < 0 || zz < 0 || xx >= o->nx || yy >= o->ny || zz >= o->nz) continue;
                out[n++] = (int32_t)(((int64_t)xx * o->ny + yy) * o->nz + zz);
            }
    return n;
}

static inline double kern(const vrgo *o, double d) { /* A*exp(-0.5*H*d**2)  :154 */
    return o->A * exp(-0.5 * o->H * (d * d));
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* The distinct values of a[0..n) when there are at most DSET_MAX of them (quantised volumes: a few hundred to a few
 * thousand), through a small open-addressing set - one pass instead of sorting every voxel (a minute at 880x880x640,
 * which is what bench.py's cpu_baseline on the whole volume could not afford).  Returns the count, or -1 when there
 * are more (the caller then sorts everything).  out: room for DSET_MAX values, unsorted. */
enum { DSET_BITS = 17, DSET_MAX = 1 << 15 };
static int64_t distinct_small(const double *a, int64_t lo, int64_t hi, double *out) {
    const uint64_t EMPTY = 0x7ff8dead00000001ull;          /* a NaN pattern no finite intensity has */
    uint64_t *tab = (uint64_t *)malloc(sizeof(uint64_t) << DSET_BITS);
    if (!tab) return -1;
    for (int64_t i = 0; i < (1 << DSET_BITS); i++) tab[i] = EMPTY;
    int64_t n = 0;
    uint64_t last = EMPTY;
    for (int64_t i = lo; i < hi; i++) {
        double v = a[i];
        if (v == 0.0) v = 0.0;                              /* -0.0 and +0.0 are one value (as for the sort's comparison) */
        uint64_t k; memcpy(&k, &v, 8);
        if (k == last) continue;
        last = k;
        uint64_t h = (k * 0x9E3779B97F4A7C15ull) >> (64 - DSET_BITS);
        while (tab[h] != EMPTY && tab[h] != k) h = (h + 1) & ((1u << DSET_BITS) - 1);
        if (tab[h] == EMPTY) {
            if (n == DSET_MAX || v != v) { free(tab); return -1; }
            tab[h] = k; out[n++] = v;
        }
    }
    free(tab);
    return n;
}

/* mode 1 set-up: sorted unique intensity values and the per-voxel level index */
static int build_levels(vrgo *o) {
    int64_t L = -1;
    double *tmp = NULL;
    {   /* few distinct values: per chunk (one per thread in the all-cores build) a small set, then their union, sorted */
        enum { NCH = 64 };
        double *part = (double *)malloc(sizeof(double) * DSET_MAX * NCH);
        int64_t cnt[NCH];
        if (part) {
#ifdef _OPENMP
            #pragma omp parallel for schedule(dynamic, 1)
#endif
            for (int c = 0; c < NCH; c++) cnt[c] = distinct_small(o->data, o->V * c / NCH, o->V * (c + 1) / NCH, part + (size_t)c * DSET_MAX);
            int64_t tot = 0; int ok = 1;
            for (int c = 0; c < NCH; c++) { if (cnt[c] < 0) ok = 0; else tot += cnt[c]; }
            if (ok) {
                tmp = (double *)malloc(sizeof(double) * (tot ? tot : 1));
                if (tmp) {
                    int64_t m = 0;
                    for (int c = 0; c < NCH; c++) { memcpy(tmp + m, part + (size_t)c * DSET_MAX, sizeof(double) * cnt[c]); m += cnt[c]; }
                    qsort(tmp, m, sizeof(double), cmp_double);
                    L = 0;
                    for (int64_t i = 0; i < m; i++) if (i == 0 || tmp[i] != tmp[L - 1]) tmp[L++] = tmp[i];
                }
            }
            free(part);
        }
    }
    if (L < 0) {                                            /* many distinct values: sort every voxel's */
        free(tmp);
        tmp = (double *)malloc(sizeof(double) * o->V);
        if (!tmp) return -1;
        memcpy(tmp, o->data, sizeof(double) * o->V);
        qsort(tmp, o->V, sizeof(double), cmp_double);
        L = 0;
        for (int64_t i = 0; i < o->V; i++)
            if (i == 0 || tmp[i] != tmp[L - 1]) tmp[L++] = tmp[i];
    }
    o->L = L;
    o->lev = (double *)malloc(sizeof(double) * L);
    memcpy(o->lev, tmp, sizeof(double) * L);
    free(tmp);
    o->levidx = (int32_t *)malloc(sizeof(int32_t) * o->V);
    o->hin = (int64_t *)calloc(L, sizeof(int64_t));
    o->hout = (int64_t *)calloc(L, sizeof(int64_t));
#ifdef _OPENMP
    #pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < o->V; i++) {
        int64_t lo = 0, hi = L - 1; double v = o->data[i];
        while (lo < hi) { int64_t m = (lo + hi) / 2; if (o->lev[m] < v) lo = m + 1; else hi = m; }
        o->levidx[i] = (int32_t)lo;
    }
    return 0;
}

/* innerValues = dataArray[(valueMap==0)|(valueMap==1)], outerValues = dataArray[(==2)|(==3)]
 * (:149-150, :249-250), kept as class histograms in mode 1. */
static void recount_hist(vrgo *o) {
    memset(o->hin, 0, sizeof(int64_t) * o->L);
    memset(o->hout, 0, sizeof(int64_t) * o->L);
#ifdef _OPENMP
    /* all-cores build (libvrg_oracle_omp.so, bench.py's cpu_baseline): private histograms per thread,
     * added up afterwards - integer counts, so the result is the serial one */
    #pragma omp parallel
    {
        int64_t *hi = (int64_t *)calloc((size_t)o->L * 2, sizeof(int64_t)), *ho = hi + o->L;
        #pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < o->V; i++) {
            uint8_t l = o->label[i];
            if (l <= 1) hi[o->levidx[i]]++;
            else if (l <= 3) ho[o->levidx[i]]++;
        }
        #pragma omp critical
        for (int64_t l = 0; l < o->L; l++) { o->hin[l] += hi[l]; o->hout[l] += ho[l]; }
        free(hi);
    }
#else
    for (int64_t i = 0; i < o->V; i++) {
        uint8_t l = o->label[i];
        if (l <= 1) o->hin[o->levidx[i]]++;
        else if (l <= 3) o->hout[o->levidx[i]]++;
    }
#endif
}

/* np.sum of a contiguous float64 vector: numpy's pairwise summation (blocks of 128, eight
 * running partial sums), so that rounding - and therefore exact-tie sign tests - follow the reference. */
static double np_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum(a, n2) + np_sum(a + n2, n - n2);
    }
}

static double *scratch(vrgo *o, int64_t n) {
    if (n > o->capscr) {
        o->capscr = n + 1024;
        o->scr = (double *)realloc(o->scr, sizeof(double) * o->capscr);
    }
    return o->scr;
}

/* np.sum(A * np.exp(-0.5 * H * (values - v)**2))  (:154-155, :240-242, :254-255) */
static double kern_sum(vrgo *o, const double *values, int64_t n, double v) {
    double *t = scratch(o, n);
    for (int64_t i = 0; i < n; i++) t[i] = kern(o, values[i] - v);
    return np_sum(t, n);
}

/* exact densities of one point over the whole inner / outer regions (:152-155, :252-255) */
static void exact_probs(vrgo *o, int32_t p, const double *innerValues, int64_t nin,
                        const double *outerValues, int64_t nout) {
    double v = o->data[p];
    if (o->density_mode == 0) {
        o->ip[p] = kern_sum(o, innerValues, nin, v);
        o->op[p] = kern_sum(o, outerValues, nout, v);
    } else {
        double si = 0, so = 0;
        for (int64_t l = 0; l < o->L; l++) {
            if (!o->hin[l] && !o->hout[l]) continue;
            double k = kern(o, o->lev[l] - v);
            si += (double)o->hin[l] * k;
            so += (double)o->hout[l] * k;
        }
        o->ip[p] = si; o->op[p] = so;
    }
}

static int gather_region_values(vrgo *o, double **iv, int64_t *nin, double **ov, int64_t *nout) {
    if (o->density_mode != 0) { recount_hist(o); *iv = *ov = NULL; *nin = *nout = 0; return 0; }
    int64_t a = 0, b = 0;
    for (int64_t i = 0; i < o->V; i++) { uint8_t l = o->label[i]; if (l <= 1) a++; else if (l <= 3) b++; }
    *iv = (double *)malloc(sizeof(double) * (a ? a : 1));
    *ov = (double *)malloc(sizeof(double) * (b ? b : 1));
    if (!*iv || !*ov) return -1;
    a = b = 0;
    for (int64_t i = 0; i < o->V; i++) {
        uint8_t l = o->label[i];
        if (l <= 1) (*iv)[a++] = o->data[i]; else if (l <= 3) (*ov)[b++] = o->data[i];
    }
    *nin = a; *nout = b;
    return 0;
}

static void recount_sizes(vrgo *o) { /* :49-52, :113-116 */
    int64_t a = 0, b = 0; double sa = 0, sb = 0;
#ifdef _OPENMP
    /* all-cores build: 256 fixed chunks, each summed in voxel order, added up in chunk order - the intensity
     * sums then do not depend on the thread count (they differ from the serial build's by rounding only) */
    enum { NCH = 256 };
    int64_t ca[NCH], cb[NCH]; double csa[NCH], csb[NCH];
    #pragma omp parallel for schedule(static)
    for (int c = 0; c < NCH; c++) {
        int64_t lo = o->V * c / NCH, hi = o->V * (c + 1) / NCH, x = 0, y = 0; double sx = 0, sy = 0;
        for (int64_t i = lo; i < hi; i++) {
            uint8_t l = o->label[i];
            if (l <= 1) { x++; sx += o->data[i]; } else if (l <= 3) { y++; sy += o->data[i]; }
        }
        ca[c] = x; cb[c] = y; csa[c] = sx; csb[c] = sy;
    }
    for (int c = 0; c < NCH; c++) { a += ca[c]; b += cb[c]; sa += csa[c]; sb += csb[c]; }
#else
    for (int64_t i = 0; i < o->V; i++) {
        uint8_t l = o->label[i];
        if (l <= 1) { a++; sa += o->data[i]; } else if (l <= 3) { b++; sb += o->data[i]; }
    }
#endif
    o->innerSize = a; o->outerSize = b;
    if (o->ntrace == o->captrace) {
        o->captrace = o->captrace ? o->captrace * 2 : 64;
        o->trace = (vrgo_trace *)realloc(o->trace, sizeof(vrgo_trace) * o->captrace);
    }
    vrgo_trace *t = &o->trace[o->ntrace++];
    t->nflip = o->nflips; t->nseg = 0; t->n_in = a; t->n_out = b;
    t->ni = o->inner.n; t->no = o->outer.n; t->sum_in = (double)sa; t->sum_out = (double)sb;
    int64_t ns = 0; for (int64_t i = 0; i < o->segl.n; i++) ns += o->segl.items[i] >= 0;
    t->nseg = ns;
}

static int push32(int32_t **a, int64_t *n, int64_t *cap, int32_t v) {
    if (*n == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        *a = (int32_t *)realloc(*a, sizeof(int32_t) * *cap);
        if (!*a) return -1;
    }
    (*a)[(*n)++] = v;
    return 0;
}

/* update(..., flipedPoints=None) - init mode, :129-155 */
static int update_init(vrgo *o) {
    int32_t nb[26];
    int64_t nseg = o->segl.n;
    for (int64_t s = 0; s < nseg; s++) {                     /* for point in segmentedList :134 */
        int32_t p = o->segl.items[s];
        int n = neighbours(o, p, nb);                        /* :135 */
        for (int k = 0; k < n; k++) if (o->label[nb[k]] == 4) o->label[nb[k]] = 3;   /* :137 */
        for (int k = 0; k < n; k++) {                        /* :138 */
            int32_t q = nb[k];
            if (o->seg[q] == 0) {                            /* :139 */
                if (o->label[p] != 1) { if (ol_append(&o->inner, p)) return VRGO_ERR; o->label[p] = 1; }  /* :140-142 */
                if (o->label[q] != 2) { if (ol_append(&o->outer, q)) return VRGO_ERR; o->label[q] = 2; }  /* :143-145 */
            }
        }
    }
    double *iv, *ov; int64_t nin, nout;
    if (gather_region_values(o, &iv, &nin, &ov, &nout)) return VRGO_ERR;     /* :149-150 */
    for (int64_t i = 0; i < o->inner.n; i++) exact_probs(o, o->inner.items[i], iv, nin, ov, nout); /* :151-155 */
    for (int64_t i = 0; i < o->outer.n; i++) exact_probs(o, o->outer.items[i], iv, nin, ov, nout);
    free(iv); free(ov);
    return VRGO_OK;
}

static inline int any_label(const vrgo *o, const int32_t *nb, int n, uint8_t want) {
    for (int k = 0; k < n; k++) if (o->label[nb[k]] == want) return 1;
    return 0;
}

/* 4 -> 3 on a neighbour set, remembering the converted voxels (:167-169, :178-180, :207-209) */
static int include_excluded(vrgo *o, const int32_t *nb, int n) {
    for (int k = 0; k < n; k++)
        if (o->label[nb[k]] == 4) {
            o->label[nb[k]] = 3;
            if (push32(&o->incl, &o->nincl, &o->capincl, nb[k])) return -1;
        }
    return 0;
}

/* update(..., flipedPoints, innerBnd, outerBnd, innerProb, outerProb) - incremental mode, :156-259 */
static int update_incremental(vrgo *o) {
    int32_t nb[26], nb2[26];
    o->nnewin = o->nnewout = o->nincl = 0;
    for (int64_t f = 0; f < o->nflips; f++) {                /* for point in flipedPoints :165 */
        int32_t p = o->flips[f];
        int n = neighbours(o, p, nb);                        /* :166 */
        if (include_excluded(o, nb, n)) return VRGO_ERR;     /* :167-169 */
        if (o->label[p] == 1) {                              /* :170 originally inner bound */
            if (ol_remove(&o->inner, p)) return VRGO_ERR;    /* :171 */
            if (ol_remove(&o->segl, p)) return VRGO_ERR;     /* :172 */
            o->seg[p] = 0; o->label[p] = 2;                  /* :173-174 */
            if (ol_append(&o->outer, p)) return VRGO_ERR;    /* :175 */
            for (int k = 0; k < n; k++) {                    /* :176 */
                int32_t q = nb[k];
                int n2 = neighbours(o, q, nb2);              /* :177 */
                if (include_excluded(o, nb2, n2)) return VRGO_ERR;   /* :178-180 */
                uint8_t lq = o->label[q];
                if (lq == 3) {                               /* :181-182 */
                } else if (lq == 2) {                        /* :183 */
                    if (!any_label(o, nb2, n2, 1)) {         /* :186 */
                        o->label[q] = 3;                     /* :187 */
                        if (ol_remove(&o->outer, q)) return VRGO_ERR;   /* :188 */
                        o->ip[q] = 0; o->op[q] = 0;          /* :189-190 */
                    }
                } else if (lq == 1) {                        /* :191-192 */
                } else {                                     /* inside :193 */
                    o->label[q] = 1;                         /* :194 */
                    if (push32(&o->newin, &o->nnewin, &o->capnew, q)) return VRGO_ERR;  /* :195 */
                    if (ol_append(&o->inner, q)) return VRGO_ERR;                       /* :196 */
                }
            }
        } else if (o->label[p] == 2) {                       /* :198 originally outer bound */
            if (ol_remove(&o->outer, p)) return VRGO_ERR;    /* :199 */
            if (ol_append(&o->segl, p)) return VRGO_ERR;     /* :200 */
            o->seg[p] = 1; o->label[p] = 1;                  /* :201-202 */
            if (ol_append(&o->inner, p)) return VRGO_ERR;    /* :204 */
            for (int k = 0; k < n; k++) {                    /* :205 */
                int32_t q = nb[k];
                int n2 = neighbours(o, q, nb2);              /* :206 */
                if (include_excluded(o, nb2, n2)) return VRGO_ERR;   /* :207-209 */
                uint8_t lq = o->label[q];
                if (lq == 3) {                               /* :210 */
                    o->label[q] = 2;                         /* :211 */
                    if (push32(&o->newout, &o->nnewout, &o->capnewout, q)) return VRGO_ERR;        /* :212 */
                    if (ol_append(&o->outer, q)) return VRGO_ERR;   /* :213 */
                } else if (lq == 2) {                        /* :217-218 */
                } else if (lq == 1) {                        /* :219 */
                    if (!any_label(o, nb2, n2, 2)) {         /* :223 */
                        o->label[q] = 0;                     /* :224 */
                        if (ol_remove(&o->inner, q)) return VRGO_ERR;   /* :226 */
                        o->ip[q] = 0; o->op[q] = 0;          /* :227-228 */
                    }
                }                                            /* else inside: pass :229-230 */
            }
        }                                                    /* neither 1 nor 2: nothing else happens */
    }
    /* :232-235 */
    int64_t nia = 0, noa = 0;
    double *innerAdded = (double *)malloc(sizeof(double) * (o->nflips ? o->nflips : 1));
    double *outerAdded = (double *)malloc(sizeof(double) * (o->nflips ? o->nflips : 1));
    double *addedPoints = (double *)malloc(sizeof(double) * (o->nincl ? o->nincl : 1));
    if (!innerAdded || !outerAdded || !addedPoints) return VRGO_ERR;
    for (int64_t f = 0; f < o->nflips; f++) {
        int32_t p = o->flips[f];
        if (o->label[p] == 1) innerAdded[nia++] = o->data[p];
        else if (o->label[p] == 2) outerAdded[noa++] = o->data[p];
    }
    for (int64_t i = 0; i < o->nincl; i++) addedPoints[i] = o->data[o->incl[i]];
    /* :236-247 correction for every voxel of the final band lists */
    double *memo = NULL; uint8_t *have = NULL;
    if (o->density_mode != 0) {
        memo = (double *)malloc(sizeof(double) * 3 * o->L);
        have = (uint8_t *)calloc(o->L, 1);
        if (!memo || !have) return VRGO_ERR;
    }
    for (int pass = 0; pass < 2; pass++) {
        olist *l = pass ? &o->outer : &o->inner;
        for (int64_t i = 0; i < l->n; i++) {
            int32_t p = l->items[i];
            if (p < 0) continue;
            double ic, oc, ac;
            int64_t lv = o->density_mode ? o->levidx[p] : -1;
            if (lv >= 0 && have[lv]) { ic = memo[3 * lv]; oc = memo[3 * lv + 1]; ac = memo[3 * lv + 2]; }
            else {
                double v = o->data[p];
                ic = kern_sum(o, innerAdded, nia, v);          /* :240 */
                oc = kern_sum(o, outerAdded, noa, v);          /* :241 */
                ac = kern_sum(o, addedPoints, o->nincl, v);    /* :242 */
                if (lv >= 0) { memo[3 * lv] = ic; memo[3 * lv + 1] = oc; memo[3 * lv + 2] = ac; have[lv] = 1; }
            }
            o->ip[p] += ic;      /* :243 */
            o->ip[p] -= oc;      /* :244 */
            o->op[p] -= ic;      /* :245 */
            o->op[p] += oc;      /* :246 */
            o->op[p] += ac;      /* :247 */
        }
    }
    free(memo); free(have);
    free(innerAdded); free(outerAdded); free(addedPoints);
    /* :249-255 exact recompute for the voxels promoted this sweep */
    double *iv, *ov; int64_t nin, nout;
    if (gather_region_values(o, &iv, &nin, &ov, &nout)) return VRGO_ERR;
    for (int64_t i = 0; i < o->nnewin; i++) exact_probs(o, o->newin[i], iv, nin, ov, nout);
    for (int64_t i = 0; i < o->nnewout; i++) exact_probs(o, o->newout[i], iv, nin, ov, nout);
    free(iv); free(ov);
    ol_compact(&o->inner); ol_compact(&o->outer); ol_compact(&o->segl);   /* :257-259 */
    return VRGO_OK;
}

/* ------------------------------------------------------------------ public API */
vrgo *vrgo_create(int nx, int ny, int nz, const double *data, const uint8_t *labels, double H,
                  int density_mode) {
    vrgo *o = (vrgo *)calloc(1, sizeof(vrgo));
    if (!o) return NULL;
    o->nx = nx; o->ny = ny; o->nz = nz; o->V = (int64_t)nx * ny * nz;
    o->H = H; o->density_mode = density_mode;
    o->A = pow(2 * M_PI, -0.5);                               /* :7 */
    o->data = (double *)malloc(sizeof(double) * o->V);
    o->label = (uint8_t *)malloc(o->V);
    o->seg = (uint8_t *)calloc(o->V, 1);
    o->ip = (double *)calloc(o->V, sizeof(double));           /* :132 */
    o->op = (double *)calloc(o->V, sizeof(double));           /* :133 */
    if (!o->data || !o->label || !o->seg || !o->ip || !o->op) return NULL;
    memcpy(o->data, data, sizeof(double) * o->V);
    memcpy(o->label, labels, o->V);
    if (ol_init(&o->inner, o->V) || ol_init(&o->outer, o->V) || ol_init(&o->segl, o->V)) return NULL;
    if (density_mode && build_levels(o)) return NULL;
    o->iterNum = 0;
    return o;
}

void vrgo_destroy(vrgo *o) {
    if (!o) return;
    free(o->data); free(o->label); free(o->seg); free(o->ip); free(o->op);
    ol_free(&o->inner); ol_free(&o->outer); ol_free(&o->segl);
    free(o->lev); free(o->levidx); free(o->hin); free(o->hout);
    free(o->scr); free(o->trace); free(o->flips); free(o->newin); free(o->newout); free(o->incl);
    free(o);
}

/* variationalRegionGrowing :38-52 - seeds, segmentedMap, init update, region sizes */
int vrgo_init(vrgo *o) {
    o->t0 = now_s();                                           /* :38 */
    for (int64_t i = 0; i < o->V; i++)                         /* :44-46 */
        if (o->label[i] == 0) { if (ol_append(&o->segl, (int32_t)i)) return VRGO_ERR; o->seg[i] = 1; }
    if (o->segl.n == 0) return VRGO_ERR;                       /* reference raises at :48 (concatenate of empty lists) */
    int rc = update_init(o);                                   /* :47 */
    if (rc) return rc;
    o->nflips = 0;
    recount_sizes(o);                                          /* :49-52 */
    o->iterNum = 1;                                            /* :57 */
    o->stop = 0;
    return VRGO_OK;
}

/* One trip through the while-loop body :58-117.  Returns 0 if an update was applied,
 * or the stop reason (>0) without applying the pending flips. */
int vrgo_step(vrgo *o, int64_t iterMax, int64_t maxSegmentSize, double maxSeconds) {
    if (o->stop) return o->stop;
    if (o->iterNum > iterMax) { o->stop = VRGO_STOP_ITERMAX; return o->stop; }   /* :58 */
    /* :79-88 - decide flips over allBnd = concat(innerBnd, outerBnd) */
    o->nflips = 0;
    for (int pass = 0; pass < 2; pass++) {
        olist *l = pass ? &o->outer : &o->inner;
        for (int64_t i = 0; i < l->n; i++) {
            int32_t p = l->items[i];
            double inN = o->ip[p] / (double)o->innerSize;      /* :81 */
            double outN = o->op[p] / (double)o->outerSize;     /* :82 */
            int ge = inN >= outN;
            if ((o->seg[p] != 0) != (ge != 0))                 /* :87 logical_xor */
                if (push32(&o->flips, &o->nflips, &o->capflips, p)) return VRGO_ERR;
        }
    }
    if (o->nflips == 0) { o->stop = VRGO_STOP_CONVERGED; return o->stop; }      /* :91 */
    if (maxSeconds >= 0 && now_s() - o->t0 >= maxSeconds) { o->stop = VRGO_STOP_TIME; return o->stop; } /* :97 */
    if (o->segl.n >= maxSegmentSize) { o->stop = VRGO_STOP_SIZE; return o->stop; }   /* :101 */
    int rc = update_incremental(o);                            /* :109 */
    if (rc) return rc;
    recount_sizes(o);                                          /* :113-116 */
    o->iterNum++;                                              /* :117 */
    return VRGO_OK;
}

/* whole driver :56-121; returns the stop reason */
int vrgo_run(vrgo *o, int64_t iterMax, int64_t maxSegmentSize, double maxSeconds) {
    if (o->iterNum == 0) { int rc = vrgo_init(o); if (rc) return rc; }
    for (;;) { int rc = vrgo_step(o, iterMax, maxSegmentSize, maxSeconds); if (rc) return rc; }
}

/* all-cores build only: threads used by the dense loops (returns what is in effect; 1 in the serial build) */
int vrgo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}
int64_t vrgo_iter_num(const vrgo *o) { return o->iterNum; }
int64_t vrgo_nseg(const vrgo *o) { return o->segl.n; }
int64_t vrgo_ninner(const vrgo *o) { return o->inner.n; }
int64_t vrgo_nouter(const vrgo *o) { return o->outer.n; }
int64_t vrgo_inner_size(const vrgo *o) { return o->innerSize; }
int64_t vrgo_outer_size(const vrgo *o) { return o->outerSize; }
int64_t vrgo_nlevels(const vrgo *o) { return o->L; }
void vrgo_get_labels(const vrgo *o, uint8_t *out) { memcpy(out, o->label, o->V); }
void vrgo_get_segmap(const vrgo *o, uint8_t *out) { memcpy(out, o->seg, o->V); }
void vrgo_get_segmented(const vrgo *o, int64_t *out) { for (int64_t i = 0; i < o->segl.n; i++) out[i] = o->segl.items[i]; }
/* which: 0 inner list, 1 outer list; lex indices and the band densities in list order */
void vrgo_get_band(const vrgo *o, int which, int64_t *idx, double *ip, double *op) {
    const olist *l = which ? &o->outer : &o->inner;
    for (int64_t i = 0; i < l->n; i++) { int32_t p = l->items[i]; idx[i] = p; ip[i] = o->ip[p]; op[i] = o->op[p]; }
}
int64_t vrgo_ntrace(const vrgo *o) { return o->ntrace; }
void vrgo_get_trace(const vrgo *o, vrgo_trace *out) { memcpy(out, o->trace, sizeof(vrgo_trace) * o->ntrace); }
