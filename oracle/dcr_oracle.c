/*
 * dcr_oracle.c — CPU restatement (plain C) of Decombinator's per-read V/J
 * tag-matching hot path.  TEST INFRASTRUCTURE ONLY: nothing under
 * decombinator_amd/ may call, link or import this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker (never as the thing measured as "the product").
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against the JSON fixtures under tests/golden/, vectors produced by oracle/gen_golden.py
 * from the reference's own unmodified src/decombinator/decombine.py (imported
 * in the build container with stand-ins for the three absent wheels), and
 * against the rows of the reference's tests/resources/dcr_TINY_1_{alpha,beta}.n12
 * reproducible from the fixture-derived tag set.  What stays unpinned (no
 * acora / Biopython / Levenshtein wheel is installable here) is listed in
 * DESIGN.md "Oracle": acora's tie order for keywords of unequal length that
 * end at the same position, and Biopython's complement of non-ACGTN bytes.
 *
 * Each function cites the reference lines it follows
 * (reference = /root/reference/src/decombinator/decombine.py).
 *
 * Third-party algorithms restated (not under /root/reference):
 *   acora==2.4 (pyproject.toml:10)        Aho-Corasick build + findall
 *   Levenshtein==0.25.1 (pyproject.toml:20) hamming
 *   biopython==1.84 (pyproject.toml:11)   Seq.reverse_complement
 */
#include "dcr_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Python slice semantics: s[a:b] on a sequence of length n.                 */
/* ------------------------------------------------------------------------ */
static void pyslice(long n, long a, long b, long *lo, long *hi) {
  if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
  if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
  if (b < a) b = a;
  *lo = a; *hi = b;
}

/* G[ga:gb] == R[ra:rb] as Python strings */
static int slice_eq(const char *G, long Lg, long ga, long gb,
                    const char *R, long n, long ra, long rb) {
  long glo, ghi, rlo, rhi;
  pyslice(Lg, ga, gb, &glo, &ghi);
  pyslice(n, ra, rb, &rlo, &rhi);
  if (ghi - glo != rhi - rlo) return 0;
  return memcmp(G + glo, R + rlo, (size_t)(ghi - glo)) == 0;
}

/* ------------------------------------------------------------------------ */
/* acora stand-in: Aho-Corasick over a set of distinct keywords.             */
/* findall(): every occurrence (overlaps included) as (keyword, start),      */
/* emitted in order of END position while the text is scanned left to right; */
/* among keywords ending at the same position the longest comes first (the   */
/* order of the output-link chain).  A byte that occurs in no keyword sends  */
/* the machine back to the root.  Duplicate keywords are stored once         */
/* (AcoraBuilder keeps a set).  Call sites: decombine.py:722-746 (build),    */
/* :275,:294,:339,:399,:422,:473 (findall).                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
  int n_kw;
  char **kw;      /* distinct keyword strings */
  int *kw_len;
  int n_sym;      /* distinct bytes in keywords + 1 ("other" = n_sym-1) */
  unsigned char sym_of[256];
  int n_nodes;
  int *delta;     /* [n_nodes][n_sym] full transition function */
  int *node_kw;   /* keyword ending exactly at node, or -1 */
  int *out_link;  /* nearest proper suffix node that ends a keyword, or -1 */
} ac_t;

static void ac_free(ac_t *a) {
  if (!a) return;
  for (int i = 0; i < a->n_kw; i++) free(a->kw[i]);
  free(a->kw); free(a->kw_len); free(a->delta); free(a->node_kw); free(a->out_link);
  free(a);
}

static ac_t *ac_build(int n, const char *const *words) {
  ac_t *a = (ac_t *)calloc(1, sizeof(ac_t));
  a->kw = (char **)calloc((size_t)(n > 0 ? n : 1), sizeof(char *));
  a->kw_len = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  /* de-duplicate; the empty string is not a keyword */
  for (int i = 0; i < n; i++) {
    int dup = 0;
    if (words[i][0] == 0) continue;
    for (int j = 0; j < a->n_kw; j++) if (strcmp(a->kw[j], words[i]) == 0) { dup = 1; break; }
    if (dup) continue;
    a->kw[a->n_kw] = strdup(words[i]);
    a->kw_len[a->n_kw] = (int)strlen(words[i]);
    a->n_kw++;
  }
  /* alphabet */
  int seen[256]; memset(seen, 0, sizeof seen);
  long total = 1;
  for (int i = 0; i < a->n_kw; i++) {
    total += a->kw_len[i];
    for (int k = 0; k < a->kw_len[i]; k++) seen[(unsigned char)a->kw[i][k]] = 1;
  }
  int ns = 0;
  for (int c = 0; c < 256; c++) if (seen[c]) ns++;
  a->n_sym = ns + 1;
  { int s = 0; for (int c = 0; c < 256; c++) a->sym_of[c] = (unsigned char)(seen[c] ? s++ : ns); }
  int S = a->n_sym;
  int *go = (int *)malloc(sizeof(int) * (size_t)total * (size_t)S);
  for (long i = 0; i < total * S; i++) go[i] = -1;
  a->node_kw = (int *)malloc(sizeof(int) * (size_t)total);
  a->out_link = (int *)malloc(sizeof(int) * (size_t)total);
  int *fail = (int *)calloc((size_t)total, sizeof(int));
  for (long i = 0; i < total; i++) { a->node_kw[i] = -1; a->out_link[i] = -1; }
  int nn = 1;
  for (int i = 0; i < a->n_kw; i++) {
    int s = 0;
    for (int k = 0; k < a->kw_len[i]; k++) {
      int c = a->sym_of[(unsigned char)a->kw[i][k]];
      if (go[s * S + c] < 0) go[s * S + c] = nn++;
      s = go[s * S + c];
    }
    a->node_kw[s] = i;
  }
  a->n_nodes = nn;
  /* BFS: failure links, output links, full delta */
  a->delta = (int *)malloc(sizeof(int) * (size_t)nn * (size_t)S);
  int *queue = (int *)malloc(sizeof(int) * (size_t)nn);
  int qh = 0, qt = 0;
  for (int c = 0; c < S; c++) {
    int t = go[c];
    if (t >= 0 && c != S - 1) { a->delta[c] = t; fail[t] = 0; queue[qt++] = t; }
    else a->delta[c] = 0;
  }
  while (qh < qt) {
    int s = queue[qh++];
    int f = fail[s];
    a->out_link[s] = (a->node_kw[f] >= 0) ? f : a->out_link[f];
    for (int c = 0; c < S; c++) {
      int t = go[s * S + c];
      if (c == S - 1) { a->delta[s * S + c] = 0; continue; } /* unknown byte: reset */
      if (t >= 0) { a->delta[s * S + c] = t; fail[t] = a->delta[f * S + c]; queue[qt++] = t; }
      else a->delta[s * S + c] = a->delta[f * S + c];
    }
  }
  free(go); free(fail); free(queue);
  return a;
}

typedef struct { int kw; int start; } ac_hit_t;

/* Returns the number of hits; writes at most cap of them. */
static int ac_findall(const ac_t *a, const char *text, int n, ac_hit_t *hits, int cap) {
  int cnt = 0, s = 0, S = a->n_sym;
  for (int i = 0; i < n; i++) {
    s = a->delta[s * S + a->sym_of[(unsigned char)text[i]]];
    int t = (a->node_kw[s] >= 0) ? s : a->out_link[s];
    while (t >= 0) {
      if (cnt < cap) { hits[cnt].kw = a->node_kw[t]; hits[cnt].start = i + 1 - a->kw_len[a->node_kw[t]]; }
      cnt++;
      t = a->out_link[t];
    }
  }
  return cnt;
}

/* ------------------------------------------------------------------------ */
/* Levenshtein.hamming stand-in (call sites decombine.py:309,359,436,493):   */
/* mismatching positions; a length difference counts as that many more       */
/* (rapidfuzz Hamming pads by default).                                      */
/* ------------------------------------------------------------------------ */
static int hamming(const char *a, int la, const char *b, int lb) {
  int m = la < lb ? la : lb, d = (la > lb ? la : lb) - m;
  for (int i = 0; i < m; i++) d += a[i] != b[i];
  return d;
}

/* ------------------------------------------------------------------------ */
/* Biopython Seq.reverse_complement stand-in (decombine.py:182-184): the     */
/* ambiguous-DNA complement table, both cases, U -> A; other bytes kept.     */
/* ------------------------------------------------------------------------ */
static unsigned char comp_tab[256];
static int comp_ready = 0;
static void comp_init(void) {
  if (comp_ready) return;
  for (int c = 0; c < 256; c++) comp_tab[c] = (unsigned char)c;
  const char *k = "ACGTMRWSYKVHDBXNU";
  const char *v = "TGCAKYWSRMBDHVXNA";
  for (int i = 0; k[i]; i++) {
    comp_tab[(unsigned char)k[i]] = (unsigned char)v[i];
    comp_tab[(unsigned char)(k[i] + 32)] = (unsigned char)(v[i] + 32);
  }
  comp_ready = 1;
}

void dcro_revcomp(const char *in, int n, char *out) {
  comp_init();
  for (int i = 0; i < n; i++) out[i] = (char)comp_tab[(unsigned char)in[n - 1 - i]];
  out[n] = 0;
}

/* ------------------------------------------------------------------------ */
/* Tables: the module globals that import_tcr_info() creates                 */
/* (decombine.py:593-746) for one chain.                                     */
/* ------------------------------------------------------------------------ */
typedef struct {
  int n;
  char **seqs;   int *len;       /* v_seqs / j_seqs            :711-718 */
  char **half1;  char **half2;   /* half1_*_seqs, half2_*_seqs :837-842, :859-864 */
  int *jump;                     /* jump_to_end_v / jump_to_start_j */
  char **region; int *region_len;/* v_regions / j_regions (upper-cased) :692-696 */
  int split;                     /* v_half_split / j_half_split :657-661 */
  ac_t *key, *half1_key, *half2_key; /* :722-746 */
} gene_t;

struct dcro_tables { gene_t v, j; };

static char *substr_dup(const char *s, long lo, long hi) {
  char *r = (char *)malloc((size_t)(hi - lo) + 1);
  memcpy(r, s + lo, (size_t)(hi - lo)); r[hi - lo] = 0; return r;
}

static void gene_init(gene_t *g, int n, const char *const *tags, const int *jumps,
                      const char *const *regions, int split) {
  g->n = n; g->split = split;
  g->seqs = (char **)calloc((size_t)n + 1, sizeof(char *));
  g->half1 = (char **)calloc((size_t)n + 1, sizeof(char *));
  g->half2 = (char **)calloc((size_t)n + 1, sizeof(char *));
  g->region = (char **)calloc((size_t)n + 1, sizeof(char *));
  g->len = (int *)calloc((size_t)n + 1, sizeof(int));
  g->jump = (int *)calloc((size_t)n + 1, sizeof(int));
  g->region_len = (int *)calloc((size_t)n + 1, sizeof(int));
  for (int i = 0; i < n; i++) {
    long L = (long)strlen(tags[i]), lo, hi;
    g->seqs[i] = strdup(tags[i]); g->len[i] = (int)L; g->jump[i] = jumps[i];
    pyslice(L, 0, split, &lo, &hi); g->half1[i] = substr_dup(tags[i], lo, hi);   /* tag[0:split] */
    pyslice(L, split, L, &lo, &hi); g->half2[i] = substr_dup(tags[i], lo, hi);   /* tag[split:]  */
    g->region[i] = strdup(regions[i]); g->region_len[i] = (int)strlen(regions[i]);
    for (char *p = g->region[i]; *p; p++) if (*p >= 'a' && *p <= 'z') *p = (char)(*p - 32); /* .seq.upper() :695 */
  }
  g->key = ac_build(n, (const char *const *)g->seqs);
  g->half1_key = ac_build(n, (const char *const *)g->half1);
  g->half2_key = ac_build(n, (const char *const *)g->half2);
}

static void gene_free(gene_t *g) {
  for (int i = 0; i < g->n; i++) { free(g->seqs[i]); free(g->half1[i]); free(g->half2[i]); free(g->region[i]); }
  free(g->seqs); free(g->half1); free(g->half2); free(g->region);
  free(g->len); free(g->jump); free(g->region_len);
  ac_free(g->key); ac_free(g->half1_key); ac_free(g->half2_key);
}

dcro_tables *dcro_tables_new(int nv, const char *const *v_tags, const int *v_jumps,
                             const char *const *v_regions, int nj, const char *const *j_tags,
                             const int *j_jumps, const char *const *j_regions,
                             int v_half_split, int j_half_split) {
  dcro_tables *t = (dcro_tables *)calloc(1, sizeof(dcro_tables));
  gene_init(&t->v, nv, v_tags, v_jumps, v_regions, v_half_split);
  gene_init(&t->j, nj, j_tags, j_jumps, j_regions, j_half_split);
  return t;
}

void dcro_tables_free(dcro_tables *t) {
  if (!t) return;
  gene_free(&t->v); gene_free(&t->j); free(t);
}

/* list.index(): first index holding that string */
static int list_index(char **list, int n, const char *s) {
  for (int i = 0; i < n; i++) if (strcmp(list[i], s) == 0) return i;
  return -1;
}

/* findall() wrapper: hits go to a small inline buffer, a heap buffer only when a read has more
 * than 32 of them (keeps the timed CPU baseline free of allocator traffic) */
typedef struct { ac_hit_t *h; int n; ac_hit_t inl[32]; } hits_t;
static void findall_into(hits_t *r, const ac_t *a, const char *read, int n) {
  r->h = r->inl;
  r->n = ac_findall(a, read, n, r->h, 32);
  if (r->n > 32) {
    r->h = (ac_hit_t *)malloc(sizeof(ac_hit_t) * (size_t)r->n);
    r->n = ac_findall(a, read, n, r->h, r->n);
  }
}
static void hits_free(hits_t *r) { if (r->h != r->inl) free(r->h); }

/* ------------------------------------------------------------------------ */
/* get_v_deletions — decombine.py:749-785                                    */
/* returns 1 and [end_v, deletions_v] on success                             */
/* ------------------------------------------------------------------------ */
static int get_v_deletions(const gene_t *g, const char *read, long n, int v_match,
                           long temp_end_v, long *end_v, long *deletions_v, uint64_t *counts) {
  long f = temp_end_v;                              /* :753 */
  const char *G = g->region[v_match]; long Lg = g->region_len[v_match];
  long pos = Lg - 10;                               /* :754-756 */
  if (f >= n) { counts[DCRX_C_V_DEL_FAILED_TAG_AT_END]++; return 0; } /* :760-762 */
  f += 1;                                           /* :764 */
  long num_del = 0;                                 /* :765 */
  while (0 <= f && f < n) {                         /* :767 */
    if (slice_eq(G, Lg, pos, pos + 10, read, n, f - 10, f)) { /* :769-772 */
      *deletions_v = num_del;                       /* :774 */
      *end_v = temp_end_v - num_del;                /* :775 */
      return 1;
    }
    pos -= 1; num_del += 1; f -= 1;                 /* :777-779 */
  }
  counts[DCRX_C_V_DEL_FAILED]++;                    /* :784 */
  return 0;
}

/* ------------------------------------------------------------------------ */
/* get_j_deletions — decombine.py:788-817                                    */
/* ------------------------------------------------------------------------ */
static int get_j_deletions(const gene_t *g, const char *read, long n, int j_match,
                           long temp_start_j, long end_of_v, long *start_j, long *deletions_j,
                           uint64_t *counts) {
  long f = temp_start_j;                            /* :792 */
  const char *G = g->region[j_match]; long Lg = g->region_len[j_match];
  long pos = 0;                                     /* :793 */
  while (0 <= f + 2 && f + 2 < n) {                 /* :795 */
    if (f < end_of_v) { pos += 1; f += 1; }         /* :798-800 */
    else if (slice_eq(G, Lg, pos, pos + 10, read, n, f, f + 10)) { /* :802-805 */
      *deletions_j = pos; *start_j = f;             /* :807-808 */
      return 1;
    } else { pos += 1; f += 1; }                    /* :810-811 */
  }
  counts[DCRX_C_J_DEL_FAILED]++;                    /* :816 */
  return 0;
}

typedef struct { long match, pos, dels, tagpos; } xdat_t; /* (v_match,end_v,v_dels,v_seq_start) / (j_match,start_j,j_dels,j_seq_end) */

/* ------------------------------------------------------------------------ */
/* vanalysis — decombine.py:273-394.  *status gets the exit path.            */
/* ------------------------------------------------------------------------ */
static int vanalysis(const gene_t *g, const char *read, long n, xdat_t *out, int *status,
                     uint64_t *counts) {
  hits_t hold_v; findall_into(&hold_v, g->key, read, (int)n);               /* :275 */
  if (hold_v.n) {                                              /* :277 */
    if (hold_v.n > 1) {                                        /* :278-280 */
      counts[DCRX_C_MULTIPLE_V_MATCHES]++; *status = DCRX_S_V_MULTI; hits_free(&hold_v); return 0;
    }
    int v_match = list_index(g->seqs, g->n, g->key->kw[hold_v.h[0].kw]); /* :282 */
    long p = hold_v.h[0].start;
    long temp_end_v = p + g->jump[v_match] - 1;                /* :283-285 */
    hits_free(&hold_v);
    long end_v, dels;
    uint64_t before = counts[DCRX_C_V_DEL_FAILED_TAG_AT_END];
    if (get_v_deletions(g, read, n, v_match, temp_end_v, &end_v, &dels, counts)) { /* :288-290 */
      out->match = v_match; out->pos = end_v; out->dels = dels; out->tagpos = p; return 1;
    }
    *status = (counts[DCRX_C_V_DEL_FAILED_TAG_AT_END] != before) ? DCRX_S_V_WALK_FAIL_AT_END : DCRX_S_V_WALK_FAIL;
    return 0;                                                  /* falls off the if: None */
  }
  hits_free(&hold_v);
  hits_t hold_v1; findall_into(&hold_v1, g->half1_key, read, (int)n);        /* :294 */
  if (hold_v1.n) {                                             /* :296 */
    for (int i = 0; i < hold_v1.n; i++) {                      /* :297 */
      const char *h = g->half1_key->kw[hold_v1.h[i].kw]; long p = hold_v1.h[i].start;
      int k0 = list_index(g->half1, g->n, h);                  /* half1_v_seqs.index(...) :305 */
      for (int k = 0; k < g->n; k++) {                         /* indices :298-301 */
        if (strcmp(g->half1[k], h) != 0) continue;
        long lo, hi;
        pyslice(n, p, p + g->len[k0], &lo, &hi);               /* :302-307 */
        if (g->len[k] != hi - lo) continue;
        pyslice(n, p, p + g->len[k], &lo, &hi);                /* :311-314 */
        if (hamming(g->seqs[k], g->len[k], read + lo, (int)(hi - lo)) <= 1) { /* :308-317 */
          counts[DCRX_C_VERR2]++;                              /* :318 */
          long temp_end_v = p + g->jump[k] - 1;                /* :320-322 */
          long end_v, dels;
          if (get_v_deletions(g, read, n, k, temp_end_v, &end_v, &dels, counts)) { /* :323-326 */
            out->match = k; out->pos = end_v; out->dels = dels; out->tagpos = p; /* :327-333 */
            hits_free(&hold_v1); return 1;
          }
        }
      }
    }
    counts[DCRX_C_FOUNDV1NOTV2]++; *status = DCRX_S_V_HALF1_EXHAUSTED; /* :334-335 */
    hits_free(&hold_v1); return 0;
  }
  hits_free(&hold_v1);
  hits_t hold_v2; findall_into(&hold_v2, g->half2_key, read, (int)n);        /* :339 */
  if (hold_v2.n) {                                             /* :340 */
    for (int i = 0; i < hold_v2.n; i++) {                      /* :341 */
      const char *h = g->half2_key->kw[hold_v2.h[i].kw]; long p = hold_v2.h[i].start;
      int k0 = list_index(g->half2, g->n, h);                  /* :354 */
      long q = p - g->split;
      for (int k = 0; k < g->n; k++) {                         /* :342-347 */
        if (strcmp(g->half2[k], h) != 0) continue;
        long lo, hi;
        pyslice(n, q, q + g->len[k0], &lo, &hi);               /* :348-357 */
        if (g->len[k] != hi - lo) continue;
        pyslice(n, q, p + g->len[k] - g->split, &lo, &hi);     /* :361-366 */
        if (hamming(g->seqs[k], g->len[k], read + lo, (int)(hi - lo)) <= 1) { /* :358-369 */
          counts[DCRX_C_VERR1]++;                              /* :370 */
          long temp_end_v = p + g->jump[k] - g->split - 1;     /* :372-377 */
          long end_v, dels;
          if (get_v_deletions(g, read, n, k, temp_end_v, &end_v, &dels, counts)) { /* :378-381 */
            out->match = k; out->pos = end_v; out->dels = dels; out->tagpos = q; /* :382-388 */
            hits_free(&hold_v2); return 1;
          }
        }
      }
    }
    counts[DCRX_C_FOUNDV2NOTV1]++; *status = DCRX_S_V_HALF2_EXHAUSTED; /* :389-390 */
    hits_free(&hold_v2); return 0;
  }
  hits_free(&hold_v2);
  counts[DCRX_C_NO_VTAGS_FOUND]++; *status = DCRX_S_V_NONE;    /* :393-394 */
  return 0;
}

/* ------------------------------------------------------------------------ */
/* janalysis — decombine.py:397-531                                          */
/* ------------------------------------------------------------------------ */
static int janalysis(const gene_t *g, const char *read, long n, long end_of_v, xdat_t *out,
                     int *status, uint64_t *counts) {
  hits_t hold_j; findall_into(&hold_j, g->key, read, (int)n);               /* :399 */
  if (hold_j.n) {                                              /* :401 */
    if (hold_j.n > 1) {                                        /* :402-404 */
      counts[DCRX_C_MULTIPLE_J_MATCHES]++; *status = DCRX_S_J_MULTI; hits_free(&hold_j); return 0;
    }
    const char *kw = g->key->kw[hold_j.h[0].kw];
    int j_match = list_index(g->seqs, g->n, kw);               /* :406 */
    long p = hold_j.h[0].start;
    long temp_start_j = p - g->jump[j_match];                  /* :407-409 */
    long j_seq_end = p + (long)strlen(kw);                     /* :411 */
    hits_free(&hold_j);
    long start_j, dels;
    if (get_j_deletions(g, read, n, j_match, temp_start_j, end_of_v, &start_j, &dels, counts)) { /* :413-418 */
      out->match = j_match; out->pos = start_j; out->dels = dels; out->tagpos = j_seq_end; return 1;
    }
    *status = DCRX_S_J_WALK_FAIL; return 0;
  }
  hits_free(&hold_j);
  hits_t hold_j1; findall_into(&hold_j1, g->half1_key, read, (int)n);        /* :422 */
  if (hold_j1.n) {                                             /* :423 */
    for (int i = 0; i < hold_j1.n; i++) {                      /* :424 */
      const char *h = g->half1_key->kw[hold_j1.h[i].kw]; long p = hold_j1.h[i].start;
      int k0 = list_index(g->half1, g->n, h);                  /* :432 */
      for (int k = 0; k < g->n; k++) {                         /* :425-428 */
        if (strcmp(g->half1[k], h) != 0) continue;
        long lo, hi;
        pyslice(n, p, p + g->len[k0], &lo, &hi);               /* :429-434 */
        if (g->len[k] != hi - lo) continue;
        pyslice(n, p, p + g->len[k], &lo, &hi);                /* :438-441 */
        if (hamming(g->seqs[k], g->len[k], read + lo, (int)(hi - lo)) <= 1) { /* :435-444 */
          counts[DCRX_C_JERR2]++;                              /* :445 */
          long temp_start_j = p - g->jump[k];                  /* :447-449 */
          long j_seq_end = p + (long)strlen(h) + g->split;     /* :450-454 */
          long start_j, dels;
          if (get_j_deletions(g, read, n, k, temp_start_j, end_of_v, &start_j, &dels, counts)) { /* :455-462 */
            out->match = k; out->pos = start_j; out->dels = dels; out->tagpos = j_seq_end; /* :463-468 */
            hits_free(&hold_j1); return 1;
          }
        }
      }
    }
    counts[DCRX_C_FOUNDJ1NOTJ2]++; *status = DCRX_S_J_HALF1_EXHAUSTED; /* :469-470 */
    hits_free(&hold_j1); return 0;
  }
  hits_free(&hold_j1);
  hits_t hold_j2; findall_into(&hold_j2, g->half2_key, read, (int)n);        /* :473 */
  if (hold_j2.n) {                                             /* :474 */
    for (int i = 0; i < hold_j2.n; i++) {                      /* :475 */
      const char *h = g->half2_key->kw[hold_j2.h[i].kw]; long p = hold_j2.h[i].start;
      int k0 = list_index(g->half2, g->n, h);                  /* :488 */
      long q = p - g->split;
      for (int k = 0; k < g->n; k++) {                         /* :476-481 */
        if (strcmp(g->half2[k], h) != 0) continue;
        long lo, hi;
        pyslice(n, q, q + g->len[k0], &lo, &hi);               /* :482-491 */
        if (g->len[k] != hi - lo) continue;
        pyslice(n, q, p + g->len[k] - g->split, &lo, &hi);     /* :495-500 */
        if (hamming(g->seqs[k], g->len[k], read + lo, (int)(hi - lo)) <= 1) { /* :492-503 */
          counts[DCRX_C_JERR1]++;                              /* :504 */
          long temp_start_j = p - g->jump[k] - g->split;       /* :506-510 */
          long j_seq_end = p + (long)strlen(h);                /* :511 */
          long start_j, dels;
          if (get_j_deletions(g, read, n, k, temp_start_j, end_of_v, &start_j, &dels, counts)) { /* :512-519 */
            out->match = k; out->pos = start_j; out->dels = dels; out->tagpos = j_seq_end; /* :520-525 */
            hits_free(&hold_j2); return 1;
          }
        }
      }
    }
    counts[DCRX_C_FOUNDV2NOTV1]++; *status = DCRX_S_J_HALF2_EXHAUSTED; /* :526-527 (the reference bumps the V key) */
    hits_free(&hold_j2); return 0;
  }
  hits_free(&hold_j2);
  counts[DCRX_C_NO_J_ASSIGNED]++; *status = DCRX_S_J_NONE;      /* :530-531 */
  return 0;
}

/* ------------------------------------------------------------------------ */
/* dcr — decombine.py:534-585.  Returns 1 when the reference returns the     */
/* 7-list, 0 when it returns None.                                           */
/* ------------------------------------------------------------------------ */
int dcro_dcr(const dcro_tables *t, const char *read, int n_, int allow_ns, int lenthreshold,
             dcro_result *res, uint64_t *counts) {
  long n = n_;
  memset(res, 0, sizeof *res);
  xdat_t vdat, jdat; int status = 0;
  if (!vanalysis(&t->v, read, n, &vdat, &status, counts)) { res->status = status; return 0; } /* :542-545 */
  long end_of_v = vdat.pos + 1;                                 /* :547 */
  if (!janalysis(&t->j, read, n, end_of_v, &jdat, &status, counts)) { /* :548 */
    counts[DCRX_C_VJ_ASSIGNMENT_FAILED]++;                      /* :583-585 */
    res->status = status; return 0;
  }
  long lo, hi; int hasN = 0;
  pyslice(n, vdat.tagpos, jdat.tagpos, &lo, &hi);               /* read[vdat[3]:jdat[3]] :554 */
  for (long i = lo; i < hi; i++) if (read[i] == 'N') { hasN = 1; break; }
  if (hasN && !allow_ns) { counts[DCRX_C_DCRFILTER_INTERTAGN]++; res->status = DCRX_S_F_INTERTAG_N; return 0; } /* :553-556 */
  if ((vdat.tagpos - jdat.tagpos) >= lenthreshold) {            /* :557-560 */
    counts[DCRX_C_DCRFILTER_TOOLONG_INTERTAG]++; res->status = DCRX_S_F_TOOLONG; return 0;
  }
  if (vdat.dels > (t->v.jump[vdat.match] - t->v.len[vdat.match]) ||
      jdat.dels > t->j.jump[jdat.match]) {                      /* :561-565 */
    counts[DCRX_C_DCRFILTER_IMPOSS_DELETION]++; res->status = DCRX_S_F_IMPOSS_DEL; return 0;
  }
  if ((vdat.tagpos + t->v.len[vdat.match]) > (jdat.tagpos + t->j.len[jdat.match])) { /* :566-569 */
    counts[DCRX_C_DCRFILTER_TAG_OVERLAP]++; res->status = DCRX_S_F_OVERLAP; return 0;
  }
  pyslice(n, vdat.pos + 1, jdat.pos, &lo, &hi);                 /* insert = read[end_v+1 : start_j] :577 */
  res->status = DCRX_S_OK;
  res->v = (int)vdat.match; res->j = (int)jdat.match;           /* :572-581 */
  res->vdel = (int)vdat.dels; res->jdel = (int)jdat.dels;
  res->ins_start = (int)lo; res->ins_len = (int)(hi - lo);
  res->v_start = (int)vdat.tagpos; res->j_end = (int)jdat.tagpos;
  return 1;
}

/* ------------------------------------------------------------------------ */
/* One read through the orientation dispatch of the driver loop,             */
/* decombine.py:991 and :998-1013.  `vdj` is the read as it sits in the      */
/* FASTQ (after any barcode slicing).                                        */
/* ------------------------------------------------------------------------ */
int dcro_decombine_read(const dcro_tables *t, const char *vdj, int n, int orientation,
                        int allow_ns, int lenthreshold, dcro_result *res, uint64_t *counts) {
  int ok = 0;
  char stackbuf[1024];
  char *rc = (n < (int)sizeof stackbuf) ? stackbuf : (char *)malloc((size_t)n + 1);
  counts[DCRX_C_READ_COUNT]++;                                  /* :991 */
  if (orientation == DCRX_ORIENT_REVERSE) {                     /* :999-1001 */
    dcro_revcomp(vdj, n, rc);
    ok = dcro_dcr(t, rc, n, allow_ns, lenthreshold, res, counts); res->frame = 0;
  } else if (orientation == DCRX_ORIENT_FORWARD) {              /* :1002-1004 */
    ok = dcro_dcr(t, vdj, n, allow_ns, lenthreshold, res, counts); res->frame = 1;
  } else {                                                      /* :1005-1010 */
    dcro_revcomp(vdj, n, rc);
    ok = dcro_dcr(t, rc, n, allow_ns, lenthreshold, res, counts); res->frame = 0;
    if (!ok) { ok = dcro_dcr(t, vdj, n, allow_ns, lenthreshold, res, counts); res->frame = 1; }
  }
  if (ok) {
    counts[DCRX_C_VJ_COUNT]++;                                  /* :1012-1013 */
    if (res->frame) counts[DCRX_C_FRAME_FORWARD]++;
  }
  if (rc != stackbuf) free(rc);
  return ok;
}

/* Batch form over a concatenated ASCII buffer; offsets has n_reads+1 entries. */
void dcro_decombine_batch(const dcro_tables *t, const char *ascii, const uint64_t *offsets,
                          uint64_t n_reads, int orientation, int allow_ns, int lenthreshold,
                          dcro_result *res, uint64_t *counts) {
  for (uint64_t r = 0; r < n_reads; r++)
    dcro_decombine_read(t, ascii + offsets[r], (int)(offsets[r + 1] - offsets[r]), orientation,
                        allow_ns, lenthreshold, &res[r], counts);
}

/* The same over n_threads POSIX threads (contiguous slices of the reads; per-thread counters
 * summed at the end): the checker for full-size batches on the GPU box and bench.py's CPU
 * baseline.  passes > 1 repeats each slice (timing only: results and counters are those of one
 * pass). */
#include <pthread.h>
typedef struct {
  const dcro_tables *t; const char *ascii; const uint64_t *offsets; uint64_t lo, hi;
  int orientation, allow_ns, lenthreshold, passes; dcro_result *res; uint64_t counts[DCRX_N_COUNTERS];
} __attribute__((aligned(128))) dcro_job;      /* a job per thread on cache lines of its own */
static void *dcro_worker(void *arg) {
  dcro_job *j = (dcro_job *)arg;
  uint64_t local[DCRX_N_COUNTERS];             /* tallied on the thread's stack, handed back once */
  for (int p = 0; p < j->passes; p++) {
    memset(local, 0, sizeof local);
    for (uint64_t r = j->lo; r < j->hi; r++)
      dcro_decombine_read(j->t, j->ascii + j->offsets[r], (int)(j->offsets[r + 1] - j->offsets[r]), j->orientation,
                          j->allow_ns, j->lenthreshold, &j->res[r], local);
  }
  memcpy(j->counts, local, sizeof local);
  return NULL;
}
int dcro_decombine_batch_mt(const dcro_tables *t, const char *ascii, const uint64_t *offsets,
                            uint64_t n_reads, int orientation, int allow_ns, int lenthreshold,
                            dcro_result *res, uint64_t *counts, int n_threads, int passes) {
  if (n_threads < 1) n_threads = 1;
  if (passes < 1) passes = 1;
  dcro_job *jobs = NULL;
  if (posix_memalign((void **)&jobs, 128, (size_t)n_threads * sizeof *jobs) != 0) jobs = NULL;
  else memset(jobs, 0, (size_t)n_threads * sizeof *jobs);
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof *th);
  if (!jobs || !th) { free(jobs); free(th); return -1; }
  int started = 0;
  for (int k = 0; k < n_threads; k++) {
    jobs[k].t = t; jobs[k].ascii = ascii; jobs[k].offsets = offsets; jobs[k].res = res;
    jobs[k].lo = n_reads * (uint64_t)k / (uint64_t)n_threads; jobs[k].hi = n_reads * (uint64_t)(k + 1) / (uint64_t)n_threads;
    jobs[k].orientation = orientation; jobs[k].allow_ns = allow_ns; jobs[k].lenthreshold = lenthreshold; jobs[k].passes = passes;
    if (pthread_create(&th[k], NULL, dcro_worker, &jobs[k]) != 0) break;
    started++;
  }
  for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
  for (int k = started; k < n_threads; k++) dcro_worker(&jobs[k]);      /* threads that could not start: here */
  for (int c = 0; c < DCRX_N_COUNTERS; c++) {
    uint64_t s = 0;
    for (int k = 0; k < n_threads; k++) s += jobs[k].counts[c];
    counts[c] += s;
  }
  free(jobs); free(th);
  return 0;
}

/* findall exposed for the acora-contract tests: which = 0 key, 1 half1, 2 half2; gene 0 V, 1 J.
 * Writes (index of the first tag holding the keyword, start) pairs; returns the hit count. */
int dcro_findall(const dcro_tables *t, int gene, int which, const char *text, int n,
                 int *first_idx, int *start, int cap) {
  const gene_t *g = gene ? &t->j : &t->v;
  const ac_t *a = which == 0 ? g->key : which == 1 ? g->half1_key : g->half2_key;
  char **list = which == 0 ? g->seqs : which == 1 ? g->half1 : g->half2;
  hits_t h; findall_into(&h, a, text, n);
  for (int i = 0; i < h.n && i < cap; i++) { first_idx[i] = list_index(list, g->n, a->kw[h.h[i].kw]); start[i] = h.h[i].start; }
  int cnt = h.n; hits_free(&h); return cnt;
}
