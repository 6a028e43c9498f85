/* TEST ORACLE - NOT PRODUCT CODE.  See hao_oracle.h for the rules.
 *
 * CPU restatement of the reference hot path; each function cites the reference
 * file:line whose observable behaviour it reproduces.  Written for clarity, not
 * speed (single thread; sorting instead of hash tables where the result is
 * order-independent).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "hao_oracle.h"

#define N_COUNTS 4096          /* YAK_N_COUNTS  htab.cpp:14 */
#define MAX_COUNT 4095         /* YAK_MAX_COUNT htab.cpp:15 */

/* ------------------------------------------------------------------ */
/* small utilities                                                     */
/* ------------------------------------------------------------------ */

static void *xrealloc(void *p, size_t n) { void *q = realloc(p, n ? n : 1); if (!q) abort(); return q; }

/* LSD radix sort of u64 keys (ascending); tmp must hold n entries */
static void sort_u64(uint64_t *a, uint64_t n, uint64_t *tmp)
{
	uint64_t *src = a, *dst = tmp; int pass;
	for (pass = 0; pass < 8; ++pass) {
		uint64_t cnt[256], i, s = 0; int sh = pass * 8;
		memset(cnt, 0, sizeof(cnt));
		for (i = 0; i < n; ++i) ++cnt[src[i] >> sh & 255];
		if (cnt[src[0] >> sh & 255] == n) continue; /* all equal in this digit */
		for (i = 0; i < 256; ++i) { uint64_t c = cnt[i]; cnt[i] = s; s += c; }
		for (i = 0; i < n; ++i) dst[cnt[src[i] >> sh & 255]++] = src[i];
		{ uint64_t *t = src; src = dst; dst = t; }
	}
	if (src != a) memcpy(a, src, n * sizeof(uint64_t));
}

/* stable LSD radix sort of 16-byte records by .x */
static void sort_mz_by_x(hao_or_mz_t *a, uint64_t n, hao_or_mz_t *tmp)
{
	hao_or_mz_t *src = a, *dst = tmp; int pass;
	if (n == 0) return;
	for (pass = 0; pass < 8; ++pass) {
		uint64_t cnt[256], i, s = 0; int sh = pass * 8;
		memset(cnt, 0, sizeof(cnt));
		for (i = 0; i < n; ++i) ++cnt[src[i].x >> sh & 255];
		for (i = 0; i < 256; ++i) { uint64_t c = cnt[i]; cnt[i] = s; s += c; }
		for (i = 0; i < n; ++i) dst[cnt[src[i].x >> sh & 255]++] = src[i];
		{ hao_or_mz_t *t = src; src = dst; dst = t; }
	}
	if (src != a) memcpy(a, src, n * sizeof(hao_or_mz_t));
}

/* yak_hash64_64, htab.h:149-159 : invertible 64-bit mix */
uint64_t hao_or_hash64(uint64_t key)
{
	key = ~key + (key << 21);
	key ^= key >> 24;
	key = key + (key << 3) + (key << 8);
	key ^= key >> 14;
	key = key + (key << 2) + (key << 4);
	key ^= key >> 28;
	key += key << 31;
	return key;
}

void hao_or_opt_default(hao_or_opt_t *o)
{
	o->k = 51; o->w = 51; o->hpc = 1; o->sample_dist = 500; o->rewin = 1000; o->min_hist_cnt = 5;
	o->max_kmer_cnt = 2000; o->high_factor = 5.0; o->max_n_chain = 100; o->is_ont = 0; o->bf_shift = 0; o->bw_thres = 0; o->hg_size = -1;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */

struct hao_or_ctx {
	hao_or_opt_t opt;
	const uint8_t *codes; const uint64_t *off; uint64_t n_reads;
	/* ft */
	int has_ft; int64_t ft_hist[N_COUNTS]; int ft_peak_hom, ft_peak_het, ft_cutoff;
	uint64_t ft_n; uint64_t *ft_keys; int32_t *ft_vals; /* vals as ha_ft_cnt returns them */
	int max_n_chain;
	/* pt */
	int64_t pt_hist[N_COUNTS]; int hom_cov, het_cov;
	uint64_t pt_nk, pt_np; uint64_t *pt_keys, *pt_off, *pt_pos;
	/* scratch */
	hao_or_mz_t *mz; int64_t mz_m; uint64_t *mt; int64_t mt_m;
	int pre_on; int64_t pre_n, pre_m, pre_totl; uint64_t *pre_x, *pre_ord; uint32_t *pre_cnt, *pre_pos;      /* hao_or_sketch_pre: the list mz1_select_mz_h receives */
	hao_or_hit_t *hits; int64_t hits_m;
	hao_or_ovlp_t *ol; int64_t ol_n, ol_m; uint64_t *fc; int64_t fc_n, fc_m; uint64_t *fc_off; int64_t fco_m;
	hao_or_hit_t *cl; int64_t cl_m;
	/* chain dp arrays */
	int32_t *f, *ii; int64_t *p, *t; int64_t dp_m;
	uint64_t *cc; int64_t cc_m;
};

hao_or_ctx *hao_or_create(const uint8_t *codes, const uint64_t *off, uint64_t n_reads, const hao_or_opt_t *opt)
{
	hao_or_ctx *c = (hao_or_ctx*)calloc(1, sizeof(*c));
	c->opt = *opt; c->codes = codes; c->off = off; c->n_reads = n_reads;
	c->max_n_chain = opt->max_n_chain; c->hom_cov = c->het_cov = -1; c->ft_peak_hom = c->ft_peak_het = -1;
	return c;
}

void hao_or_destroy(hao_or_ctx *c)
{
	if (!c) return;
	free(c->ft_keys); free(c->ft_vals); free(c->pt_keys); free(c->pt_off); free(c->pt_pos);
	free(c->mz); free(c->mt); free(c->hits); free(c->ol); free(c->fc); free(c->fc_off); free(c->cl);
	free(c->f); free(c->ii); free(c->p); free(c->t); free(c->cc);
	free(c->pre_x); free(c->pre_ord); free(c->pre_cnt); free(c->pre_pos);
	free(c);
}

const int64_t *hao_or_ft_hist(const hao_or_ctx *c) { return c->ft_hist; }
const int64_t *hao_or_pt_hist(const hao_or_ctx *c) { return c->pt_hist; }
uint64_t hao_or_ft_table(const hao_or_ctx *c, const uint64_t **keys, const int32_t **vals) { *keys = c->ft_keys; *vals = c->ft_vals; return c->ft_n; }
uint64_t hao_or_pt_table(const hao_or_ctx *c, const uint64_t **keys, const uint64_t **off, const uint64_t **pos, uint64_t *n_pos)
{ *keys = c->pt_keys; *off = c->pt_off; *pos = c->pt_pos; *n_pos = c->pt_np; return c->pt_nk; }

static void occ_thresholds(const hao_or_ctx *c, uint32_t *high_occ, uint32_t *low_occ)
{	/* ecovlp.cpp:3237-3238 with HA_KMER_GOOD_RATIO 0.333 (double arithmetic, truncation to uint32) */
	*high_occ = (uint32_t)(c->hom_cov * (2.0 - 0.333));
	*low_occ = (uint32_t)(c->hom_cov * 0.333);
}

void hao_or_stats(const hao_or_ctx *c, int64_t out[8])
{
	uint32_t h, l; occ_thresholds(c, &h, &l);
	out[0] = c->ft_peak_hom; out[1] = c->ft_peak_het; out[2] = c->ft_cutoff; out[3] = c->max_n_chain;
	out[4] = c->hom_cov; out[5] = c->het_cov; out[6] = h; out[7] = l;
}

/* ------------------------------------------------------------------ */
/* a3: all-k-mer hashing   (htab.cpp:608-645; hash htab.h:161-166)      */
/* ------------------------------------------------------------------ */

int64_t hao_or_kmer_hashes(const uint8_t *s, int64_t len, int k, int hpc, uint64_t *out)
{
	uint64_t pl[4] = {0, 0, 0, 0}, mask = (1ULL << k) - 1; int sh = k - 1, last = -1; int64_t i, l = 0, n = 0;
	for (i = 0; i < len; ++i) {
		int c = s[i];
		if (c >= 4) { l = 0; last = -1; pl[0] = pl[1] = pl[2] = pl[3] = 0; continue; } /* N restarts, planes cleared */
		if (hpc && c == last) continue;                                                   /* inside a homopolymer run */
		pl[0] = (pl[0] << 1 | (uint64_t)(c & 1)) & mask;          /* forward strand, low bit plane  */
		pl[1] = (pl[1] << 1 | (uint64_t)(c >> 1)) & mask;         /* forward strand, high bit plane */
		pl[2] = pl[2] >> 1 | (uint64_t)(1 - (c & 1)) << sh;       /* reverse complement             */
		pl[3] = pl[3] >> 1 | (uint64_t)(1 - (c >> 1)) << sh;
		last = c;
		if (++l >= k) {
			int j = pl[1] < pl[3] ? 0 : 1;                       /* yak_hash_long strand choice */
			out[n++] = hao_or_hash64(pl[j << 1]) + hao_or_hash64(pl[j << 1 | 1]);
		}
	}
	return n;
}

/* ------------------------------------------------------------------ */
/* a6: histogram -> peaks   (hist.cpp:74-157, m_peak_hom <= 0)          */
/* ------------------------------------------------------------------ */

/* hist.cpp:46-72: with a prior homozygous peak m (total bases / --hg-size) choose among (left, top, right) the candidate nearest
 * to m (top wins ties); if it lies below m by >= 51 % of itself it is the heterozygous peak and m stands; else the next candidate to
 * its left (if any) is the heterozygous peak */
static int adj_peak_with_prior(int m, int top, int left, int right, int *peak_het)
{
	int64_t mm[3], d, min_d = -1; int i, min_i = -1;
	mm[0] = left; mm[1] = top; mm[2] = right;
	for (i = 0; i < 3; ++i) {
		if (mm[i] <= 0) continue;
		d = mm[i] >= m ? mm[i] - m : m - mm[i];
		if (min_d == -1 || min_d > d || (min_d == d && i == 1)) { min_d = d; min_i = i; }
	}
	if (min_i < 0) return m;
	if (mm[min_i] < m) { d = m - mm[min_i]; if (d >= mm[min_i] * 0.51) { *peak_het = (int)mm[min_i]; return m; } }
	for (i = min_i - 1; i >= 0; --i) { if (mm[i] <= 0) continue; *peak_het = (int)mm[i]; break; }
	return (int)mm[min_i];
}

int hao_or_analyze_count(int n_cnt, int start_cnt, const int64_t *cnt, int *peak_het) { return hao_or_analyze_count_m(n_cnt, start_cnt, -1, cnt, peak_het); }

int hao_or_analyze_count_m(int n_cnt, int start_cnt, int m_peak_hom, const int64_t *cnt, int *peak_het)
{
	int i, start, low_i, max_i, max2_i, max3_i; int64_t max, max2, max3, mn;
	*peak_het = -1;
	start = cnt[1] > 0 ? 1 : 2;
	/* first rise after the left edge */
	low_i = start > start_cnt ? start : start_cnt;
	for (i = low_i + 1; i < n_cnt; ++i) if (cnt[i] > cnt[i - 1]) break;
	low_i = i - 1;
	if (low_i == n_cnt - 1) return -1;                                  /* monotone: low coverage */
	/* global maximum to the right of the valley (first one wins ties) */
	max_i = low_i + 1; max = cnt[max_i];
	for (i = low_i + 1; i < n_cnt; ++i) if (cnt[i] > max) max = cnt[i], max_i = i;
	/* best local peak between valley and maximum (scan right-to-left, strict improvement) */
	max2 = -1; max2_i = -1;
	for (i = max_i - 1; i > low_i; --i)
		if (cnt[i] >= cnt[i - 1] && cnt[i] >= cnt[i + 1] && cnt[i] > max2) max2 = cnt[i], max2_i = i;
	if (max2_i > low_i && max2_i < max_i) {
		for (i = max2_i + 1, mn = max; i < max_i; ++i) if (cnt[i] < mn) mn = cnt[i];
		if (max2 < max * 0.05 || mn > max2 * 0.95) max2 = -1, max2_i = -1;
	}
	/* best local peak to the right of the maximum */
	max3 = -1; max3_i = -1;
	for (i = max_i + 1; i < n_cnt - 1; ++i)
		if (cnt[i] >= cnt[i - 1] && cnt[i] >= cnt[i + 1] && cnt[i] > max3) max3 = cnt[i], max3_i = i;
	if (max3_i > max_i) {
		for (i = max_i + 1, mn = max; i < max3_i; ++i) if (cnt[i] < mn) mn = cnt[i];
		if (max3 < max * 0.05 || mn > max3 * 0.95 || max3_i > max_i * 2.5) max3 = -1, max3_i = -1;
	}
	if (m_peak_hom > 0) return adj_peak_with_prior(m_peak_hom, max_i, max2_i, max3_i, peak_het);
	if (max3_i > 0) { *peak_het = max_i; return max3_i; }
	if (max2_i > 0) *peak_het = max2_i;
	return max_i;
}

/* ------------------------------------------------------------------ */
/* a5/a7: exact k-mer counts -> histogram -> high-count filter table    */
/* (htab.cpp:181-214 counting saturates at 4095; :240-254 histogram;    */
/*  :1136-1169 ha_ft_gen; :1038-1070 gen_hh / ha_ft_cnt)                */
/* ------------------------------------------------------------------ */

int hao_or_ft_gen(hao_or_ctx *c)
{
	uint64_t tot = c->off[c->n_reads], n = 0, i, j, r, bias; int max_cnt;
	uint64_t *h = (uint64_t*)xrealloc(0, (tot + 1) * 8), *tmp;
	for (r = 0; r < c->n_reads; ++r)
		n += hao_or_kmer_hashes(c->codes + c->off[r], (int64_t)(c->off[r + 1] - c->off[r]), c->opt.k, c->opt.hpc, h + n);
	/* Bloom filter in front of the count table (ha_ct_init htab.cpp:140-160, yak_bf_insert :99-116, ha_ct_insert_list :181-214):
	 * one filter of 2^(bf_shift-12) bits per sub-table (sub-table = low 12 hash bits), 512-bit blocks, 4 probes inside one block.
	 * A k-mer occurrence reaches the count table only when all its probe bits were already set; the table entry then starts at
	 * 1 and is incremented, so count = 1 + (occurrences that found all bits set).  k-mers arrive per sub-table in global
	 * (read, position) order: step 1 of the counting pipeline fills the 4096 buffers read by read, step 2 drains each buffer in
	 * order, and kt_pipeline keeps blocks ordered (htab.cpp:826-843, 860-880). */
	bias = 0;
	if (c->opt.bf_shift >= 21) {                                          /* below 2^9 bits per sub-table yak_bf_init returns NULL: exact counting */
		const int nsh = c->opt.bf_shift - 12, xb = nsh - 9;               /* yak_bf_init(n_shift - pre), YAK_BLK_SHIFT = 9 */
		const uint64_t blocks = 4096ULL << xb; uint64_t m = 0;
		uint8_t *bf = (uint8_t*)calloc(blocks, 64);
		if (xb < 0 || !bf) { fprintf(stderr, "oracle: unsupported bf_shift %d\n", c->opt.bf_shift); abort(); }
		for (i = 0; i < n; ++i) {
			const uint64_t x = h[i] >> 12, y = x & ((1ULL << xb) - 1);
			int h1 = (int)(x >> xb & 511), h2 = (int)(x >> nsh & 511), z = h1, q, cnt = 0;
			uint8_t *p = bf + (((h[i] & 4095) << xb | y) << 6);
			if ((h2 & 31) == 0) h2 = (h2 + 1) & 511;
			for (q = 0; q < 4; ++q, z = (z + h2) & 511) { const uint8_t u = (uint8_t)(1 << (z & 7)); cnt += !!(p[z >> 3] & u); p[z >> 3] |= u; }
			if (cnt == 4) h[m++] = h[i];
		}
		free(bf); n = m; bias = 1;
	}
	tmp = (uint64_t*)xrealloc(0, (n + 1) * 8);
	if (n) sort_u64(h, n, tmp);
	memset(c->ft_hist, 0, sizeof(c->ft_hist));
	for (i = 0; i < n; i = j) {
		for (j = i + 1; j < n && h[j] == h[i]; ++j) {}
		++c->ft_hist[j - i + bias > MAX_COUNT ? MAX_COUNT : j - i + bias];
	}
	c->ft_peak_hom = hao_or_analyze_count_m(N_COUNTS, c->opt.min_hist_cnt, c->opt.hg_size > 0 ? (int)(c->off[c->n_reads] / (uint64_t)c->opt.hg_size) : -1, c->ft_hist, &c->ft_peak_het);
	c->ft_cutoff = (int)(c->ft_peak_hom * c->opt.high_factor);            /* htab.cpp:1160 */
	if (c->ft_cutoff > MAX_COUNT - 1) c->ft_cutoff = MAX_COUNT - 1;
	max_cnt = c->opt.max_kmer_cnt;                                         /* gen_hh clamps, htab.cpp:1042-1043 */
	if (max_cnt > MAX_COUNT - 1) max_cnt = MAX_COUNT - 1;
	if (max_cnt > INT16_MAX - 1) max_cnt = INT16_MAX - 1;
	/* keep count in [cutoff, 4095] (ha_ct_shrink, htab.cpp:1163) */
	free(c->ft_keys); free(c->ft_vals); c->ft_keys = 0; c->ft_vals = 0; c->ft_n = 0;
	{
		uint64_t m = 0;
		for (i = 0; i < n; i = j) {
			int64_t cnt;
			for (j = i + 1; j < n && h[j] == h[i]; ++j) {}
			cnt = j - i + bias > MAX_COUNT ? MAX_COUNT : (int64_t)(j - i + bias);
			if (cnt >= c->ft_cutoff) ++m;
		}
		c->ft_keys = (uint64_t*)xrealloc(0, (m + 1) * 8); c->ft_vals = (int32_t*)xrealloc(0, (m + 1) * 4);
		for (i = 0, m = 0; i < n; i = j) {
			int64_t cnt;
			for (j = i + 1; j < n && h[j] == h[i]; ++j) {}
			cnt = j - i + bias > MAX_COUNT ? MAX_COUNT : (int64_t)(j - i + bias);
			if (cnt >= c->ft_cutoff) {
				c->ft_keys[m] = h[i];
				c->ft_vals[m] = cnt > max_cnt ? INT32_MAX : (int32_t)cnt;    /* INT16_MAX in the map -> INT32_MAX from ha_ft_cnt */
				++m;
			}
		}
		c->ft_n = m;
	}
	c->has_ft = 1;
	free(h); free(tmp);
	/* ha_opt_update_cov (CommandLines.cpp:411-418) */
	{
		int mx = (int)(c->ft_peak_hom * c->opt.high_factor + .499);
		c->hom_cov = c->ft_peak_hom;
		if (c->max_n_chain < mx) c->max_n_chain = mx;
	}
	return c->ft_peak_hom;
}

int32_t hao_or_ft_cnt(const hao_or_ctx *c, uint64_t y)
{
	uint64_t lo = 0, hi = c->ft_n;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (c->ft_keys[m] < y) lo = m + 1; else hi = m; }
	return (lo < c->ft_n && c->ft_keys[lo] == y) ? c->ft_vals[lo] : 0;
}

/* ------------------------------------------------------------------ */
/* a8: minimizer sketch   (sketch.cpp:454-579)                          */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t x; uint32_t cnt, pos; uint8_t rev, span; } cand_t;  /* cnt plays ha_mz1_t::rid during sketching */
#define CNT_DUMMY ((1u << 28) - 1)

static inline int cand_cmp(const cand_t *a, const cand_t *b)     /* mz1_mzcmp, sketch.cpp:184 : (count, hash) */
{
	if (a->cnt != b->cnt) return a->cnt < b->cnt ? -1 : 1;
	return (a->x > b->x) - (a->x < b->x);
}

typedef struct { cand_t *a; uint64_t *mt; int64_t n, m; } cvec_t;

static inline void cv_push(cvec_t *v, const cand_t *e, uint32_t ord)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m * 2 : 256;
		v->a = (cand_t*)xrealloc(v->a, v->m * sizeof(cand_t)); v->mt = (uint64_t*)xrealloc(v->mt, v->m * 8);
	}
	v->a[v->n] = *e; v->mt[v->n] = ord; ++v->n;
}

/* heap primitives with klib semantics (ksort.h:43-66), `lt` = (count,hash) order: a max-heap */
static void heap_down(size_t i, size_t n, cand_t *l)
{
	size_t k = i; cand_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && cand_cmp(&l[k], &l[k + 1]) < 0) ++k;
		if (cand_cmp(&l[k], &tmp) < 0) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

/* mz1_hf_select, sketch.cpp:194-216: in the run (si, ei) keep the <= min(16, q) smallest */
static void hf_select(cvec_t *v, int32_t si, int32_t ei, int32_t n, int32_t len, int32_t sample_dist)
{
	cand_t b[16]; int32_t ps, pe, j, k, q;
	if (ei - si <= 1) return;
	ps = si < 0 ? 0 : (int32_t)v->a[si].pos; pe = ei == n ? len : (int32_t)v->a[ei].pos;
	q = (int32_t)((double)(pe - ps) / sample_dist + .499);
	if (q > 16) q = 16;
	for (j = si + 1, k = 0; j < ei && k < q; ++j, ++k) { b[k] = v->a[j]; b[k].pos = (uint32_t)j; }   /* pos field carries the index */
	{ size_t i; for (i = ((size_t)k >> 1) - 1; i != (size_t)-1; --i) heap_down(i, k, b); }
	for (; j < ei; ++j)
		if (cand_cmp(&v->a[j], &b[0]) < 0) { b[0] = v->a[j]; b[0].pos = (uint32_t)j; heap_down(0, k, b); }
	for (j = 0; j < k; ++j)
		if ((int64_t)b[j].cnt < (int64_t)(pe - ps)) v->a[b[j].pos].cnt = 0;
}

#define ORD(v, i) ((int64_t)(uint32_t)(v)->mt[i])          /* GL(), sketch.cpp:14 */
#define MARK 0x100000000ULL
#define HIGH(v, i) ((i) >= 0 && (v)->a[i].cnt > 0)         /* A_M(), sketch.cpp:15 */

static int cmp_l(const cvec_t *v, int32_t ai, int32_t bi)   /* mz1_mzcmp_l, sketch.cpp:217-225 */
{
	if (ai >= 0 && bi >= 0) {
		const cand_t *a = &v->a[ai], *b = &v->a[bi];
		if (a->cnt > 0 && b->cnt > 0) return cand_cmp(a, b);
		return (a->cnt == 0) - (b->cnt == 0);
	}
	return (ai < 0) - (bi < 0);
}

static void rescan_mark(cvec_t *v, int32_t si, int32_t i, int32_t *mi)
{	/* newest minimum of [si, i] under cmp_l, then mark every high-count entry equal to it */
	int32_t m;
	for (m = si, *mi = -1; m <= i; ++m) if (cmp_l(v, *mi, m) >= 0) *mi = m;
	if (HIGH(v, *mi))
		for (m = si; m <= i; ++m) if (HIGH(v, m) && cmp_l(v, *mi, m) == 0) v->mt[m] |= MARK;
}

/* mz1_select_mz_h, sketch.cpp:247-330.  w = mz_rewin, tot_l = number of valid k-mer iterations */
static void select_high(cvec_t *v, int len, int sample_dist, int32_t w, int32_t k, int32_t tot_l)
{
	int32_t n = (int32_t)v->n, i, m, mi = -1, si, last0, any = 0, ws = w + k - 1;
	if (n == 0) return;
	for (i = 0, last0 = -1; i <= n; ++i) {                           /* any run long enough to sample from? */
		if (i == n || v->a[i].cnt == 0) {
			if (i - last0 > 1) {
				int32_t ps = last0 < 0 ? 0 : (int32_t)v->a[last0].pos, pe = i == n ? len : (int32_t)v->a[i].pos;
				if ((int32_t)((double)(pe - ps) / sample_dist + .499) > 0) { any = 1; break; }
			}
			last0 = i;
		}
	}
	if (!any) return;                                                /* sketch.cpp:266: keep everything */
	/* first window (mz1_qfw, sketch.cpp:226-246) */
	for (i = 0; i < n; ++i) {
		if (ORD(v, i) >= ws || (i + 1 < n && ORD(v, i) < ws && ORD(v, i + 1) > ws) || (i + 1 == n && tot_l >= ws && ORD(v, i) < ws)) {
			for (m = 0; m <= i; ++m) if (HIGH(v, m) && cmp_l(v, mi, m) >= 0) mi = m;
			if (mi >= 0 && HIGH(v, mi))
				for (m = 0; m <= i; ++m) if (HIGH(v, m) && cmp_l(v, mi, m) == 0) v->mt[m] |= MARK;
			break;
		}
	}
	if (i < n) {
		for (si = 0, ++i; i < n; ++i) {                              /* sliding second-level window, sketch.cpp:271-289 */
			for (; si < i; ++si) if (ORD(v, si) + w > ORD(v, i)) break;
			if (cmp_l(v, i, mi) <= 0) { if (HIGH(v, mi)) v->mt[mi] |= MARK; mi = i; }
			else if (si > mi) { if (HIGH(v, mi)) v->mt[mi] |= MARK; rescan_mark(v, si, i, &mi); }
		}
		if (HIGH(v, mi)) v->mt[mi] |= MARK;
		for (i = n - 1; si < n && ORD(v, si) + w <= tot_l + 1; ++si)   /* tail windows, sketch.cpp:291-304 */
			if (si > mi) { if (HIGH(v, mi)) v->mt[mi] |= MARK; rescan_mark(v, si, i, &mi); }
		for (i = 0, last0 = -1; i <= n; ++i) {                       /* per run: marked ones or heap pick, sketch.cpp:307-322 */
			if (i == n || v->a[i].cnt == 0) {
				if (i - last0 > 1) {
					int32_t ps = last0 < 0 ? 0 : (int32_t)v->a[last0].pos, pe = i == n ? len : (int32_t)v->a[i].pos;
					if ((int32_t)((double)(pe - ps) / sample_dist + .499) > 0) {
						int32_t nm = 0;
						for (m = last0 + 1; m < i; ++m) if (v->mt[m] & MARK) { v->a[m].cnt = 0; ++nm; }
						if (nm == 0) hf_select(v, last0, i, n, len, sample_dist);
					}
				}
				last0 = i;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i) if (v->a[i].cnt == 0) { v->a[m] = v->a[i]; v->mt[m] = v->mt[i]; ++m; }   /* :326-329 */
	v->n = m;
}

static int64_t sketch_core(hao_or_ctx *c, const uint8_t *s, int64_t len, uint32_t rid, int use_ft, int sample_dist)
{
	const int w = c->opt.w, k = c->opt.k, hpc = c->opt.hpc;
	const cand_t dummy = { UINT64_MAX, CNT_DUMMY, 0, 0, 0 };
	uint64_t mask = (1ULL << k) - 1, pl[4] = {0, 0, 0, 0}; int sh = k - 1;
	cand_t ring[256], min = dummy; uint32_t ring_ord[256], min_ord = (uint32_t)-1;
	int q_run[64], q_front = 0, q_cnt = 0, span = 0;
	int64_t i; int l = 0, tl = 0, bp = 0, min_bp = 0, j;
	cvec_t v = { 0, 0, 0, 0 };
	assert(len > 0 && w > 0 && w < 256 && k > 0 && k <= 63);
	for (j = 0; j < w; ++j) { ring[j].x = UINT64_MAX; ring[j].cnt = CNT_DUMMY; ring[j].pos = (1u << 27) - 1; ring[j].rev = 1; ring[j].span = 255; }
	for (i = 0; i < len; ++i) {
		int b = s[i]; cand_t info = dummy;
		if (b < 4) {
			int z;
			if (hpc) {                                         /* jump to the end of the homopolymer run, sketch.cpp:480-492 */
				int run = 1;
				while (i + run < len && s[i + run] == b) ++run;
				i += run - 1;
				q_run[(q_cnt++ + q_front) & 63] = run; span += run;
				if (q_cnt > k) { span -= q_run[q_front]; q_front = (q_front + 1) & 63; --q_cnt; }
			} else span = l + 1 < k ? l + 1 : k;
			pl[0] = (pl[0] << 1 | (uint64_t)(b & 1)) & mask; pl[1] = (pl[1] << 1 | (uint64_t)(b >> 1)) & mask;
			pl[2] = pl[2] >> 1 | (uint64_t)(1 - (b & 1)) << sh; pl[3] = pl[3] >> 1 | (uint64_t)(1 - (b >> 1)) << sh;
			if (pl[1] == pl[3]) continue;                      /* strand-symmetric k-mer: skipped without touching ring/l, :502 */
			z = pl[1] < pl[3] ? 0 : 1;
			++l; ++tl;
			if (l >= k && span < 256) {
				uint64_t y = hao_or_hash64(pl[z << 1]) + hao_or_hash64(pl[z << 1 | 1]);
				int32_t cnt = use_ft ? hao_or_ft_cnt(c, y) : 0;
				if (cnt < (1 << 28)) { info.x = y; info.cnt = (uint32_t)cnt; info.pos = (uint32_t)i; info.rev = (uint8_t)z; info.span = (uint8_t)span; }
			}
		} else { l = 0; q_cnt = q_front = 0; span = 0; }        /* N: restart (planes and ring are NOT cleared), :520 */
		ring[bp] = info; ring_ord[bp] = (uint32_t)l;
		if (l == w + k - 1 && min.x != UINT64_MAX) {           /* first full window: emit ties of min, :523-534 */
			for (j = bp + 1; j < w; ++j) if (cand_cmp(&min, &ring[j]) == 0 && ring[j].pos != min.pos) cv_push(&v, &ring[j], ring_ord[j]);
			for (j = 0; j < bp; ++j) if (cand_cmp(&min, &ring[j]) == 0 && ring[j].pos != min.pos) cv_push(&v, &ring[j], ring_ord[j]);
		}
		if (cand_cmp(&min, &info) >= 0) {                      /* new minimum (ties -> newest), :543-547 */
			if (l >= w + k && min.x != UINT64_MAX) cv_push(&v, &min, min_ord);
			min = info; min_bp = bp; min_ord = ring_ord[bp];
		} else if (bp == min_bp) {                             /* minimum left the window, :548-567 */
			if (l >= w + k - 1 && min.x != UINT64_MAX) cv_push(&v, &min, min_ord);
			min = dummy;
			for (j = bp + 1; j < w; ++j) if (cand_cmp(&min, &ring[j]) >= 0) { min = ring[j]; min_bp = j; min_ord = ring_ord[j]; }
			for (j = 0; j <= bp; ++j) if (cand_cmp(&min, &ring[j]) >= 0) { min = ring[j]; min_bp = j; min_ord = ring_ord[j]; }
			if (l >= w + k - 1 && min.x != UINT64_MAX) {
				for (j = bp + 1; j < w; ++j) if (cand_cmp(&min, &ring[j]) == 0 && min.pos != ring[j].pos) cv_push(&v, &ring[j], ring_ord[j]);
				for (j = 0; j <= bp; ++j) if (cand_cmp(&min, &ring[j]) == 0 && min.pos != ring[j].pos) cv_push(&v, &ring[j], ring_ord[j]);
			}
		}
		if (++bp == w) bp = 0;
	}
	if (min.x != UINT64_MAX) cv_push(&v, &min, min_ord);       /* :571-573 */
	if (c->pre_on) {                                           /* (tests) the list as mz1_select_mz_h receives it */
		if (v.n > c->pre_m) { c->pre_m = v.n + 64; c->pre_x = (uint64_t*)xrealloc(c->pre_x, c->pre_m * 8); c->pre_ord = (uint64_t*)xrealloc(c->pre_ord, c->pre_m * 8);
			c->pre_cnt = (uint32_t*)xrealloc(c->pre_cnt, c->pre_m * 4); c->pre_pos = (uint32_t*)xrealloc(c->pre_pos, c->pre_m * 4); }
		for (i = 0; i < v.n; ++i) { c->pre_x[i] = v.a[i].x; c->pre_cnt[i] = v.a[i].cnt; c->pre_pos[i] = v.a[i].pos; c->pre_ord[i] = (uint64_t)(uint32_t)v.mt[i]; }
		c->pre_n = v.n; c->pre_totl = tl;
	}
	if (sample_dist > w) select_high(&v, (int)len, sample_dist, c->opt.rewin, k, tl);   /* :575 */
	if (v.n > c->mz_m) { c->mz_m = v.n + 64; c->mz = (hao_or_mz_t*)xrealloc(c->mz, c->mz_m * sizeof(hao_or_mz_t)); }
	for (i = 0; i < v.n; ++i) {                                /* rid overwritten by the caller's id, :577-578 */
		c->mz[i].x = v.a[i].x;
		c->mz[i].info = (uint64_t)(rid & 0xfffffffu) | (uint64_t)(v.a[i].pos & 0x7ffffffu) << 28 | (uint64_t)(v.a[i].rev & 1) << 55 | (uint64_t)v.a[i].span << 56;
	}
	i = v.n; free(v.a); free(v.mt);
	return i;
}

int64_t hao_or_sketch_seq(hao_or_ctx *c, const uint8_t *codes, int64_t len, uint32_t rid, int use_ft, int sample_dist, const hao_or_mz_t **out)
{
	int64_t n = sketch_core(c, codes, len, rid, use_ft && c->has_ft, sample_dist);
	*out = c->mz; return n;
}

int64_t hao_or_sketch(hao_or_ctx *c, uint64_t rid, int use_ft, int sample_dist, const hao_or_mz_t **out)
{
	return hao_or_sketch_seq(c, c->codes + c->off[rid], (int64_t)(c->off[rid + 1] - c->off[rid]), (uint32_t)rid, use_ft, sample_dist, out);
}

/* test support: the sketch of read rid as usual, plus the candidate list (hash, count, position, k-mer ordinal) mz1_select_mz_h was given and tot_l */
int64_t hao_or_sketch_pre(hao_or_ctx *c, uint64_t rid, int sample_dist, const hao_or_mz_t **out, int64_t *pre_n, const uint64_t **x, const uint32_t **cnt, const uint32_t **pos,
		const uint64_t **ord, int64_t *tot_l)
{
	int64_t n;
	c->pre_on = 1; n = hao_or_sketch(c, rid, 1, sample_dist, out); c->pre_on = 0;
	*pre_n = c->pre_n; *x = c->pre_x; *cnt = c->pre_cnt; *pos = c->pre_pos; *ord = c->pre_ord; *tot_l = c->pre_totl;
	return n;
}

/* ------------------------------------------------------------------ */
/* a9: count + position index   (htab.cpp:1232-1287, :380-397, :437-460) */
/* ------------------------------------------------------------------ */

int hao_or_pt_gen(hao_or_ctx *c)
{
	uint64_t r, n = 0, m = 0, i, j, nk = 0, np = 0; hao_or_mz_t *a = 0, *tmp; int lo = 2, hi;
	for (r = 0; r < c->n_reads; ++r) {                           /* minimizers of every read, in read order */
		const hao_or_mz_t *z; int64_t nz = hao_or_sketch(c, r, 1, c->opt.sample_dist, &z);
		if (n + nz > m) { m = (n + nz) * 2 + 1024; a = (hao_or_mz_t*)xrealloc(a, m * sizeof(*a)); }
		memcpy(a + n, z, nz * sizeof(*a)); n += nz;
	}
	tmp = (hao_or_mz_t*)xrealloc(0, (n + 1) * sizeof(*a));
	sort_mz_by_x(a, n, tmp);                                     /* stable: per key, (rid,pos) order is kept */
	free(tmp);
	memset(c->pt_hist, 0, sizeof(c->pt_hist));
	for (i = 0; i < n; i = j) {
		for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
		++c->pt_hist[j - i > MAX_COUNT ? MAX_COUNT : j - i];
	}
	c->hom_cov = hao_or_analyze_count_m(N_COUNTS, c->opt.min_hist_cnt, c->opt.hg_size > 0 ? (int)(c->off[c->n_reads] / (uint64_t)c->opt.hg_size) : -1, c->pt_hist, &c->het_cov);
	if (c->has_ft) hi = MAX_COUNT - 1;                           /* htab.cpp:1266-1269 */
	else { hi = (int)(c->hom_cov * c->opt.high_factor); if (hi > MAX_COUNT - 1) hi = MAX_COUNT - 1; }   /* :1258-1262 */
	for (i = 0; i < n; i = j) {
		for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
		if ((int64_t)(j - i) >= lo && (int64_t)(j - i) <= hi) ++nk, np += j - i;
	}
	free(c->pt_keys); free(c->pt_off); free(c->pt_pos);
	c->pt_keys = (uint64_t*)xrealloc(0, (nk + 1) * 8); c->pt_off = (uint64_t*)xrealloc(0, (nk + 1) * 8); c->pt_pos = (uint64_t*)xrealloc(0, (np + 1) * 8);
	nk = np = 0;
	for (i = 0; i < n; i = j) {
		for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
		if ((int64_t)(j - i) >= lo && (int64_t)(j - i) <= hi) {
			uint64_t t;
			c->pt_keys[nk] = a[i].x; c->pt_off[nk] = np; ++nk;
			for (t = i; t < j; ++t) c->pt_pos[np++] = a[t].info;  /* ha_idxpos_t has the ha_mz1_t bit layout (htab.h:20-22) */
		}
	}
	c->pt_off[nk] = np; c->pt_nk = nk; c->pt_np = np;
	free(a);
	if (!c->has_ft) {                                            /* Assembly.cpp:1011-1012: only when ha_flt_tab == 0 */
		int mx = (int)(c->hom_cov * c->opt.high_factor + .499);
		if (c->max_n_chain < mx) c->max_n_chain = mx;
	}
	return c->hom_cov;
}

static int64_t pt_get(const hao_or_ctx *c, uint64_t x, const uint64_t **pos)    /* ha_pt_get, htab.cpp:518-527 */
{
	uint64_t lo = 0, hi = c->pt_nk;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (c->pt_keys[m] < x) lo = m + 1; else hi = m; }
	if (lo < c->pt_nk && c->pt_keys[lo] == x) { *pos = c->pt_pos + c->pt_off[lo]; return (int64_t)(c->pt_off[lo + 1] - c->pt_off[lo]); }
	*pos = 0; return 0;
}

/* ------------------------------------------------------------------ */
/* a11: seed hits   (minimizers_qgen0, anchor.cpp:987-1081)             */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t srt; uint32_t self_off, other_off, cnt; } anchor_t;   /* anchor1_t, anchor.cpp:16-21 */

static int anchor_cmp(const void *pa, const void *pb)
{
	const anchor_t *a = (const anchor_t*)pa, *b = (const anchor_t*)pb;
	if (a->srt != b->srt) return a->srt < b->srt ? -1 : 1;
	return (a->other_off > b->other_off) - (a->other_off < b->other_off);
}

#define INFO_RID(v)  ((uint32_t)((v) & 0xfffffffu))
#define INFO_POS(v)  ((uint32_t)((v) >> 28 & 0x7ffffffu))
#define INFO_REV(v)  ((uint32_t)((v) >> 55 & 1))
#define INFO_SPAN(v) ((uint32_t)((v) >> 56))

static int64_t seed_hits(hao_or_ctx *c, uint64_t rid, uint32_t high_occ, uint32_t low_occ)
{
	const hao_or_mz_t *mz; int64_t nmz = hao_or_sketch_seq(c, c->codes + c->off[rid], (int64_t)(c->off[rid + 1] - c->off[rid]), 0, 1, c->opt.sample_dist, &mz);
	uint64_t max_cnt = high_occ < 2 ? 2 : high_occ, min_cnt = low_occ < 2 ? 2 : low_occ;
	int64_t i, j, na = 0, k; anchor_t *an;
	for (i = 0; i < nmz; ++i) { const uint64_t *pos; na += pt_get(c, mz[i].x, &pos); }
	an = (anchor_t*)xrealloc(0, (na + 1) * sizeof(anchor_t));
	for (i = 0, k = 0; i < nmz; ++i) {
		const uint64_t *pos; int64_t n = pt_get(c, mz[i].x, &pos);
		uint32_t zrev = INFO_REV(mz[i].info), zpos = INFO_POS(mz[i].info), zspan = INFO_SPAN(mz[i].info);
		for (j = 0; j < n; ++j) {
			uint64_t y = pos[j]; anchor_t *p = &an[k++];
			uint32_t rev = zrev != INFO_REV(y);
			p->other_off = rev ? (uint32_t)-1 - 1 - (INFO_POS(y) + 1 - INFO_SPAN(y)) : INFO_POS(y);
			p->self_off = zpos;
			p->cnt = (uint32_t)(n > 0xffffff ? 0xffffff : n) << 8 | (zspan <= 0xff ? zspan : 0xff);
			p->srt = (uint64_t)INFO_RID(y) << 33 | (uint64_t)rev << 32 | p->self_off;
		}
	}
	qsort(an, na, sizeof(anchor_t), anchor_cmp);   /* radix by srt then by other_off inside ties (anchor.cpp:1046-1049): a total order */
	if (na > c->hits_m) { c->hits_m = na + 1024; c->hits = (hao_or_hit_t*)xrealloc(c->hits, c->hits_m * sizeof(hao_or_hit_t)); }
	for (i = 0; i < na; ++i) {                      /* anchor.cpp:1055-1076 */
		hao_or_hit_t *p = &c->hits[i]; uint32_t tid = (uint32_t)(an[i].srt >> 33), strand = an[i].srt >> 32 & 1, occ = an[i].cnt >> 8, wgt;
		uint64_t tl = c->off[tid + 1] - c->off[tid];
		p->w0 = tid | strand << 31;
		p->offset = strand ? (uint32_t)(tl - ((uint32_t)-1 - an[i].other_off)) : an[i].other_off;
		p->self_offset = an[i].self_off;
		if (occ < max_cnt && occ > min_cnt) wgt = 1;
		else if (occ <= min_cnt) wgt = 2;
		else { wgt = (uint32_t)(1 + ((occ + (max_cnt << 1) - 1) / (max_cnt << 1))); wgt = (uint32_t)pow((double)wgt, 1.1); }
		if (wgt > 0xffffff) wgt = 0xffffff;
		p->cnt = wgt << 8 | (an[i].cnt & 0xff);
	}
	free(an);
	return na;
}

int64_t hao_or_seed_hits(hao_or_ctx *c, uint64_t rid, const hao_or_hit_t **out)
{
	uint32_t h, l; int64_t n; occ_thresholds(c, &h, &l);
	n = seed_hits(c, rid, h, l); *out = c->hits; return n;
}

/* ------------------------------------------------------------------ */
/* a12: per-target chaining  (Hash_Table.cpp:2007-2284 and helpers)     */
/* ------------------------------------------------------------------ */

#define H_ID(h)     ((h).w0 & 0x7fffffffu)
#define H_STRAND(h) ((h).w0 >> 31)
#define H_SPAN(h)   ((int32_t)((h).cnt & 0xffu))
#define H_WGT(h)    ((int32_t)((h).cnt >> 8))

typedef struct { double pen_gap, pen_skip, bw; int64_t max_skip, max_iter, max_dis; int64_t xl, yl; } chn_par_t;

static int64_t ext_len(int64_t x_beg, int64_t x_end, int64_t xl, int64_t y_beg, int64_t y_end, int64_t yl)
{	/* get_chainLen, Hash_Table.cpp:779-809: query span after extending both ends to a read boundary */
	int64_t xr, yr;
	if (x_beg <= y_beg) x_beg = 0; else x_beg -= y_beg;
	xr = xl - x_end - 1; yr = yl - y_end - 1;
	if (xr <= yr) x_end = xl - 1; else x_end += yr;
	return x_end - x_beg + 1;
}

static int32_t band_of(const hao_or_hit_t *ai, const hao_or_hit_t *aj, const chn_par_t *P)
{	/* cal_bw, Hash_Table.cpp:1475-1488 */
	int64_t sf_s = aj->self_offset, sf_e = (int64_t)ai->self_offset + 1, ot_s = aj->offset, ot_e = (int64_t)ai->offset + 1;
	int64_t sf_r = P->xl - sf_e, ot_r = P->yl - ot_e;
	if (sf_s <= ot_s) sf_s = 0; else sf_s -= ot_s;
	if (sf_r <= ot_r) sf_e = P->xl; else sf_e += ot_r;
	return (int32_t)((sf_e - sf_s) * P->bw);
}

/* comput_sc_ch_ec, Hash_Table.cpp:1515-1541.  *dd_out (if given) receives the diagonal gap. */
static int32_t pair_score(const hao_or_hit_t *ai, const hao_or_hit_t *aj, const chn_par_t *P, int64_t *dd_out)
{
	int32_t dq, dr, dd, dg, span, sc;
	dq = (int32_t)((int64_t)ai->self_offset - (int64_t)aj->self_offset); if (dq <= 0) return INT32_MIN;
	dr = (int32_t)((int64_t)ai->offset - (int64_t)aj->offset); if (dr <= 0) return INT32_MIN;
	dd = dr > dq ? dr - dq : dq - dr;
	if (dd > 16 && dd > band_of(ai, aj, P)) return INT32_MIN;
	dg = dr < dq ? dr : dq; span = H_SPAN(*ai);
	sc = span < dg ? span : dg;
	{ int32_t wgt = H_WGT(*ai); sc = sc >= wgt ? sc / wgt : 1; }     /* normal_w, Hash_Table.cpp:20 */
	if (dd || (dg > span && dg > 0)) {
		double lin = P->pen_gap * (double)dd, ap = (double)sc * (((double)dd / (double)dg) / P->bw);
		if (dd < 4) lin = lin > ap ? ap : lin; else lin = lin < ap ? ap : lin;
		lin += P->pen_skip * (double)dg;
		sc -= (int32_t)lin;
	}
	if (dd_out) *dd_out = dd;
	return sc;
}

static void dp_reserve(hao_or_ctx *c, int64_t n)
{
	if (n + 1 > c->dp_m) {
		c->dp_m = n + 1024;
		c->f = (int32_t*)xrealloc(c->f, c->dp_m * 4); c->ii = (int32_t*)xrealloc(c->ii, c->dp_m * 4);
		c->p = (int64_t*)xrealloc(c->p, c->dp_m * 8); c->t = (int64_t*)xrealloc(c->t, c->dp_m * 8);
	}
}

static hao_or_ovlp_t *ol_push(hao_or_ctx *c)
{
	if (c->ol_n == c->ol_m) { c->ol_m = c->ol_m ? c->ol_m * 2 : 256; c->ol = (hao_or_ovlp_t*)xrealloc(c->ol, c->ol_m * sizeof(hao_or_ovlp_t)); }
	memset(&c->ol[c->ol_n], 0, sizeof(hao_or_ovlp_t));
	return &c->ol[c->ol_n++];
}

/* per-overlap fake cigars are kept in one pool; fc_off[ordinal .. ordinal+1) */
typedef struct { uint64_t *a; int64_t n, m; } u64v_t;

static void region_from_chain(hao_or_ovlp_t *o, uint32_t xid, int64_t xl, int64_t yl, int64_t sc, const hao_or_hit_t *beg, const hao_or_hit_t *end)
{	/* push_ovlp_chain_qgen, Hash_Table.cpp:1752-1780 */
	int64_t xr, yr;
	o->x_id = xid; o->y_id = H_ID(*beg); o->x_pos_strand = 0; o->y_pos_strand = H_STRAND(*beg);
	o->x_pos_s = beg->self_offset; o->y_pos_s = beg->offset; o->x_pos_e = end->self_offset; o->y_pos_e = end->offset;
	if (o->x_pos_s <= o->y_pos_s) { o->y_pos_s -= o->x_pos_s; o->x_pos_s = 0; } else { o->x_pos_s -= o->y_pos_s; o->y_pos_s = 0; }
	xr = xl - o->x_pos_e - 1; yr = yl - o->y_pos_e - 1;
	if (xr <= yr) { o->x_pos_e = (uint32_t)(xl - 1); o->y_pos_e += (uint32_t)xr; } else { o->y_pos_e = (uint32_t)(yl - 1); o->x_pos_e += (uint32_t)yr; }
	o->shared_seed = (uint32_t)(int32_t)sc; o->align_length = 0; o->non_homopolymer_errors = 0;
}

static inline uint64_t fc_entry(uint32_t site, int32_t shift)
{	/* add_fake_cigar, Hash_Table.cpp:1295-1328 */
	uint32_t lo = shift < 0 ? ((uint32_t)(-shift) << 1 | 1u) : (uint32_t)shift << 1;
	return (uint64_t)site << 32 | lo;
}

/* gen_fake_cigar (apend_be = 1), Hash_Table.cpp:88-109; appends to the pool, returns length */
static uint32_t fake_cigar(u64v_t *pool, const hao_or_ovlp_t *o, const hao_or_hit_t *hit, int64_t n_hit)
{
	int64_t k, pdd = INT32_MAX, n0 = pool->n;
	if (pool->n + n_hit + 2 > pool->m) { pool->m = (pool->n + n_hit + 2) * 2; pool->a = (uint64_t*)xrealloc(pool->a, pool->m * 8); }
	pool->a[pool->n++] = fc_entry(o->x_pos_s, 0);
	for (k = 0; k < n_hit; ++k) {
		int64_t dq = (int64_t)hit[k].self_offset - o->x_pos_s, dr = (int64_t)hit[k].offset - o->y_pos_s, dd = dr - dq;
		if (dd != pdd) { pdd = dd; pool->a[pool->n++] = fc_entry(hit[k].self_offset, (int32_t)pdd); }
	}
	{	/* closing entry unless the last one already sits on x_pos_e */
		uint64_t last = pool->a[pool->n - 1]; int32_t lsh = (int32_t)((uint32_t)last >> 1); if (last & 1) lsh = -lsh;
		if ((int64_t)(int32_t)(last >> 32) != (int64_t)o->x_pos_e) pool->a[pool->n++] = fc_entry(o->x_pos_e, lsh);
	}
	return (uint32_t)(pool->n - n0);
}

/* quick_ck_lchain, Hash_Table.cpp:2007-2094 */
static void quick_check(const hao_or_hit_t *a, int64_t a_n, const chn_par_t *P, int64_t *p, int64_t *t, int32_t *f, int32_t *ii,
						int64_t *plus, int64_t *msc, int64_t *msc_i, int64_t *movl, int64_t *si, int64_t *ei)
{
	int64_t l, k, z, sorted = 1;
	*plus = 0; *msc = *msc_i = INT32_MIN; *movl = INT32_MAX; *si = 0; *ei = a_n;
	for (k = 1, l = 0; k <= a_n; ++k) {
		t[k - 1] = 0; ii[k - 1] = 0;
		if (k < a_n && H_STRAND(a[k]) == H_STRAND(a[l])) {
			if (a[k].self_offset <= a[k - 1].self_offset || a[k].offset <= a[k - 1].offset) sorted = 0;
			continue;
		}
		if (sorted) {                                  /* strand block [l,k) strictly increasing in both coordinates */
			int64_t plus0 = 0, msc0 = INT32_MIN, msc_i0 = INT32_MIN, ddt = 0, sc, dd;
			p[l] = -1; f[l] = H_SPAN(a[l]);
			if (f[l] >= msc0) { msc0 = f[l]; msc_i0 = l; }
			if (f[l] < plus0) plus0 = f[l];
			for (z = l + 1; z < k; ++z) {
				int32_t s = pair_score(&a[z], &a[z - 1], P, &dd);
				if (s == INT32_MIN) break;
				sc = (int64_t)s + f[z - 1];
				if (sc < H_SPAN(a[z])) break;
				p[z] = z - 1; f[z] = (int32_t)sc; ddt += dd;
				if (f[z] >= msc0) { msc0 = f[z]; msc_i0 = z; }
				if (f[z] < plus0) plus0 = f[z];
			}
			if (z >= k && msc_i0 == k - 1) {
				if (k - l >= 2 && ddt > 16 && ddt > band_of(&a[k - 1], &a[l], P)) msc_i0 = INT32_MIN;
				if (msc_i0 == k - 1) {
					if (msc0 >= *msc) {
						int64_t ov = ext_len(a[msc_i0].self_offset, a[msc_i0].self_offset, P->xl, a[msc_i0].offset, a[msc_i0].offset, P->yl);
						if (msc0 > *msc || ov < *movl) { *msc = msc0; *msc_i = msc_i0; *movl = ov; }
					}
					if (plus0 < *plus) *plus = plus0;
					if (*ei > k) *si = k; else *ei = l;   /* exclude the accepted block from the DP range */
				}
			}
		}
		l = k; sorted = 1;
	}
}

/* lchain_qdp_mcopy_fast, Hash_Table.cpp:2097-2284 with quick_check=1, gen_cigar=1, apend_be=1, khit_n=1.
 * a[0..a_n) = hits of one target; chained hits are appended to c->cl at *cl_n. Returns #hits appended. */
static int64_t chain_target(hao_or_ctx *c, const hao_or_hit_t *a, int64_t a_n, int64_t *cl_n, u64v_t *pool, const chn_par_t *P,
							uint32_t xid, int64_t mcopy_num, double mcopy_rate, int64_t mcopy_khit_cut)
{
	int64_t *p, *t, max_f, n_skip, st, max_j, end_j, sc, msc, msc_i, max_ii, ovl, movl, plus = 0, min_sc, ch_n, si, ei, i, j, k, cL = 0;
	int32_t *f, *ii, mx, tmp; hao_or_hit_t *des; hao_or_ovlp_t *z;
	if (a_n <= 0) return 0;
	dp_reserve(c, a_n);
	if (*cl_n + a_n > c->cl_m) { c->cl_m = (*cl_n + a_n) * 2 + 1024; c->cl = (hao_or_hit_t*)xrealloc(c->cl, c->cl_m * sizeof(hao_or_hit_t)); }
	des = c->cl + *cl_n;
	t = c->t; f = c->f; p = c->p; ii = c->ii;
	quick_check(a, a_n, P, p, t, f, ii, &plus, &msc, &msc_i, &movl, &si, &ei);
	for (i = st = si, max_ii = -1; i < ei; ++i) {               /* DP over the blocks the quick check did not settle, :2124-2176 */
		max_f = H_SPAN(a[i]); n_skip = 0; max_j = end_j = -1;
		if (i - st > P->max_iter) st = i - P->max_iter;
		while (H_STRAND(a[i]) != H_STRAND(a[st])) ++st;
		for (j = i - 1; j >= st; --j) {
			int32_t s = pair_score(&a[i], &a[j], P, 0);
			if (s == INT32_MIN) continue;
			sc = (int64_t)s + f[j];
			if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
			else if (t[j] == (int32_t)i) { if (++n_skip > P->max_skip) break; }
			if (p[j] >= 0) t[p[j]] = i;
		}
		end_j = j;
		if (max_ii < 0 || a[i].self_offset > a[max_ii].self_offset + P->max_dis || H_STRAND(a[i]) != H_STRAND(a[max_ii])) {
			mx = INT32_MIN; max_ii = -1;
			for (j = i - 1; j >= st && (int64_t)a[i].self_offset <= P->max_dis + (int64_t)a[j].self_offset && H_STRAND(a[i]) == H_STRAND(a[j]); --j)
				if (mx < f[j]) { mx = f[j]; max_ii = j; }
		}
		if (max_ii >= 0 && max_ii < end_j && H_STRAND(a[i]) == H_STRAND(a[max_ii])) {
			tmp = pair_score(&a[i], &a[max_ii], P, 0);
			if (tmp != INT32_MIN && max_f < (int64_t)tmp + f[max_ii]) { max_f = (int64_t)tmp + f[max_ii]; max_j = max_ii; }
		}
		f[i] = (int32_t)max_f; p[i] = max_j;
		if (max_ii < 0 || ((int64_t)a[i].self_offset <= P->max_dis + (int64_t)a[max_ii].self_offset && H_STRAND(a[i]) == H_STRAND(a[max_ii]) && f[max_ii] < f[i])) max_ii = i;
		if (f[i] >= msc) {
			ovl = ext_len(a[i].self_offset, a[i].self_offset, P->xl, a[i].offset, a[i].offset, P->yl);
			if (f[i] > msc || ovl < movl) { msc = f[i]; msc_i = i; movl = ovl; }
		}
		if (f[i] < plus) plus = f[i];
		ii[i] = 0;
	}
	for (i = msc_i, cL = 0; i >= 0; i = p[i]) { ii[i] = 1; t[cL++] = i; }     /* best chain, end -> start */

	if (mcopy_num > 1 && cL >= mcopy_khit_cut) {                             /* multi-copy chains, :2180-2270 */
		msc -= plus; min_sc = (int64_t)(msc * mcopy_rate); ii[msc_i] = 0;
		for (i = ch_n = 0; i < a_n; ++i) {
			f[i] -= (int32_t)plus; if (i >= ch_n) t[i] = 0;
			if (!ii[i] && f[i] >= min_sc) { t[ch_n] = (int64_t)((uint64_t)f[i] << 32); t[ch_n] += i << 1; ++ch_n; }
		}
		if (ch_n > 1) {
			int64_t n_v, n_v0, ni, n_u, n_u0 = c->ol_n;
			{ uint64_t *tt = (uint64_t*)xrealloc(0, (ch_n + 1) * 8); sort_u64((uint64_t*)t, ch_n, tt); free(tt); }   /* keys are non-negative and distinct */
			for (k = ch_n - 1, n_v = n_u = 0; k >= 0 && n_u < mcopy_num; --k) {
				n_v0 = n_v;
				for (i = (int64_t)((uint32_t)t[k] >> 1); i >= 0 && (t[i] & 1) == 0; ) { ii[n_v++] = (int32_t)i; t[i] |= 1; i = p[i]; }
				if (n_v0 == n_v) continue;
				sc = i < 0 ? (t[k] >> 32) : ((t[k] >> 32) - f[i]);
				if (sc >= min_sc) {
					z = ol_push(c);
					region_from_chain(z, xid, P->xl, P->yl, sc + plus, &a[ii[n_v - 1]], &a[ii[n_v0]]);
					if (!n_u || n_v - n_v0 > 1) { z->align_length = (uint32_t)(n_v - n_v0); z->x_id = (uint32_t)n_v0; ++n_u; }
					else { --c->ol_n; n_v = n_v0; }
				} else n_v = n_v0;
			}
			n_u = c->ol_n;
			{	/* write the kept chains' hits, each in increasing order, tagged with the overlap ordinal.
				 * (the reference stages through a swap area when >1 chain is kept, :2229-2249; a separate
				 * output array makes that unnecessary: the observable result is the same) */
				hao_or_hit_t *stage = (hao_or_hit_t*)xrealloc(0, (a_n + 1) * sizeof(hao_or_hit_t));
				for (k = n_u0, i = 0; k < n_u; ++k) {
					z = &c->ol[k];
					z->non_homopolymer_errors = (uint32_t)(*cl_n + i);
					n_v0 = z->x_id; ni = z->align_length;
					for (j = 0; j < ni; ++j, ++i) { stage[i] = a[ii[n_v0 + (ni - j - 1)]]; stage[i].w0 = (stage[i].w0 & 0x80000000u) | ((uint32_t)k & 0x7fffffffu); }
					z->x_id = xid;
					z->fc_len = fake_cigar(pool, z, stage + i - ni, ni);
				}
				memcpy(des, stage, i * sizeof(hao_or_hit_t)); free(stage);
			}
			*cl_n += i;
			return i;
		} else {
			msc += plus; i = msc_i; cL = 0;
			while (i >= 0) { t[cL++] = i; i = p[i]; }
		}
	}
	z = ol_push(c);
	region_from_chain(z, xid, P->xl, P->yl, msc, &a[t[cL - 1]], &a[t[0]]);
	{
		hao_or_hit_t *stage = (hao_or_hit_t*)xrealloc(0, (cL + 1) * sizeof(hao_or_hit_t));
		for (i = 0; i < cL; ++i) { stage[i] = a[t[cL - i - 1]]; stage[i].w0 = (stage[i].w0 & 0x80000000u) | ((uint32_t)(c->ol_n - 1) & 0x7fffffffu); }
		memcpy(des, stage, cL * sizeof(hao_or_hit_t)); free(stage);
	}
	z->non_homopolymer_errors = (uint32_t)*cl_n;
	z->fc_len = fake_cigar(pool, z, des, cL);
	z->align_length = (uint32_t)cL;
	*cl_n += cL;
	return cL;
}

/* ------------------------------------------------------------------ */
/* klib introsort restated on an index permutation                      */
/* (ksort.h:110-160; result order for equal keys is observable)         */
/* ------------------------------------------------------------------ */

typedef int (*lt_fn)(const hao_or_ovlp_t *a, const hao_or_ovlp_t *b);
static int lt_score_desc(const hao_or_ovlp_t *a, const hao_or_ovlp_t *b) { return (int32_t)a->shared_seed > (int32_t)b->shared_seed; }   /* oreg_ss_lt, anchor.cpp:35 */
static int lt_xs(const hao_or_ovlp_t *a, const hao_or_ovlp_t *b)
{ return ((uint64_t)a->x_pos_s << 32 | a->x_pos_e) < ((uint64_t)b->x_pos_s << 32 | b->x_pos_e); }                                          /* oreg_xs_lt, anchor.cpp:32 */

/* NOTE: fc_off must travel with the record: we carry the pool offset in a side array swapped in lockstep. */
typedef struct { hao_or_ovlp_t *r; uint64_t *fo; } srt_t;
static inline void sw(srt_t *s, int64_t i, int64_t j)
{ hao_or_ovlp_t t = s->r[i]; uint64_t o = s->fo[i]; s->r[i] = s->r[j]; s->r[j] = t; s->fo[i] = s->fo[j]; s->fo[j] = o; }

static void ins_sort(srt_t *s, int64_t lo, int64_t hi, lt_fn lt)     /* [lo, hi) */
{ int64_t i, j; for (i = lo + 1; i < hi; ++i) for (j = i; j > lo && lt(&s->r[j], &s->r[j - 1]); --j) sw(s, j, j - 1); }

static void comb_sort(srt_t *s, int64_t lo, int64_t n, lt_fn lt)
{
	const double shrink = 1.2473309501039786540366528676643; int64_t gap = n, i; int swapped;
	do {
		if (gap > 2) { gap = (int64_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = 0;
		for (i = lo; i < lo + n - gap; ++i) if (lt(&s->r[i + gap], &s->r[i])) { sw(s, i, i + gap); swapped = 1; }
	} while (swapped || gap > 2);
	if (gap != 1) ins_sort(s, lo, lo + n, lt);
}

static void intro_sort(srt_t *S, int64_t n, lt_fn lt)
{
	int64_t stack[3 * 130], top = 0, s, t, i, j, k; int d;
	if (n < 1) return;
	if (n == 2) { if (lt(&S->r[1], &S->r[0])) sw(S, 0, 1); return; }
	for (d = 2; (1ul << d) < (uint64_t)n; ++d) {}
	s = 0; t = n - 1; d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { comb_sort(S, s, t - s + 1, lt); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(&S->r[k], &S->r[i])) { if (lt(&S->r[k], &S->r[j])) k = j; }
			else k = lt(&S->r[j], &S->r[i]) ? i : j;
			if (k != t) sw(S, k, t);                      /* pivot now lives at t until the final swap */
			for (;;) {
				do ++i; while (lt(&S->r[i], &S->r[t]));
				do --j; while (i <= j && lt(&S->r[t], &S->r[j]));
				if (j <= i) break;
				sw(S, i, j);
			}
			sw(S, i, t);
			if (i - s > t - i) {
				if (i - s > 16) { stack[top++] = s; stack[top++] = i - 1; stack[top++] = d; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top++] = i + 1; stack[top++] = t; stack[top++] = d; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { ins_sort(S, 0, n, lt); return; }
			d = (int)stack[--top]; t = stack[--top]; s = stack[--top];
		}
	}
}

/* ------------------------------------------------------------------ */
/* a13/a14: per-read chain selection + glue                             */
/* (lchain_qgen_mcopy_fast anchor.cpp:1920-2100; h_ec_lchain :2302-2315; */
/*  set_lchain_dp_op :2272-2285; ha_ov_type :86-91)                     */
/* ------------------------------------------------------------------ */

static int ov_type(const hao_or_ovlp_t *r, uint32_t len)
{
	if (r->x_pos_s == 0 && r->x_pos_e == len - 1) return 2;
	if (r->x_pos_s > 0 && r->x_pos_e < len - 1) return 3;
	return r->x_pos_s == 0 ? 0 : 1;
}

static void cov_add(uint64_t *cc, uint64_t cwn, uint64_t ocv_w, uint64_t rl, uint64_t rs, uint64_t re)
{	/* add [rs,re) to the per-window covered-base counters (low 32 bits, saturating), anchor.cpp:1985-1998 */
	uint64_t m = rs / ocv_w, cws = m * ocv_w, cwe, os, oe;
	for (; m < cwn; ++m, cws += ocv_w) {
		cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
		os = rs >= cws ? rs : cws; oe = re <= cwe ? re : cwe;
		if (oe <= os) break;
		if ((uint32_t)cc[m] + (oe - os) < UINT32_MAX) cc[m] += oe - os;
		else { cc[m] >>= 32; cc[m] <<= 32; cc[m] |= UINT32_MAX; }
	}
}

int64_t hao_or_lchain(hao_or_ctx *c, uint64_t rid, const hao_or_ovlp_t **ol_out, const uint64_t **fc_out, const uint64_t **fc_off_out,
					  const hao_or_hit_t **cl_out, int64_t *cl_n_out)
{
	uint32_t high_occ, low_occ; int64_t cn, l, k, m = 0, i, lch = 0; uint64_t rl = c->off[rid + 1] - c->off[rid];
	const uint64_t max_n_chain = (uint64_t)c->max_n_chain, ocv_w = 3072; const uint32_t chain_cutoff = 2;
	chn_par_t P; u64v_t pool = { 0, 0, 0 }; srt_t S; uint64_t *fo = 0;
	occ_thresholds(c, &high_occ, &low_occ);
	{	/* set_lchain_dp_op(is_accurate=1): float expf, float constants, double products */
		double tmp = expf(-0.01 * (double)c->opt.k);
		P.pen_gap = 0.5f * tmp; P.pen_skip = 0.0005f * tmp; P.max_skip = 25; P.max_iter = 5000; P.max_dis = 5000;
		P.bw = c->opt.bw_thres > 0 ? c->opt.bw_thres : (c->opt.is_ont ? 0.05 : 0.02); P.xl = (int64_t)rl;
	}
	cn = seed_hits(c, rid, high_occ, low_occ);
	c->ol_n = 0;
	for (l = 0, k = 1; k <= cn; ++k) {                          /* one chaining call per target read, anchor.cpp:1929-1944 */
		if (k == cn || H_ID(c->hits[k]) != H_ID(c->hits[l])) {
			if (H_ID(c->hits[l]) != (uint32_t)rid) {
				uint32_t yid = H_ID(c->hits[l]); int64_t ol0 = c->ol_n;
				P.yl = (int64_t)(c->off[yid + 1] - c->off[yid]);
				chain_target(c, c->hits + l, k - l, &m, &pool, &P, (uint32_t)rid, 3, 0.7, 32);
				if (!lch) for (i = ol0; i < c->ol_n && !lch; ++i) if (c->ol[i].align_length < chain_cutoff) lch = 1;
			}
			l = k;
		}
	}
	/* fake-cigar pool offsets, one per region in creation order */
	fo = (uint64_t*)xrealloc(0, (c->ol_n + 1) * 8);
	{ uint64_t o = 0; for (i = 0; i < c->ol_n; ++i) { fo[i] = o; o += c->ol[i].fc_len; } }
	S.r = c->ol; S.fo = fo;

	if ((uint64_t)c->ol_n > max_n_chain) {                      /* too many chains: per overlap type keep the max_n_chain best, :1957-2056 */
		int32_t w, n[4] = {0, 0, 0, 0}, s[4] = {0, 0, 0, 0}; uint64_t cwn = 0, *cc = 0, kk, mm;
		intro_sort(&S, c->ol_n, lt_score_desc);
		for (i = 0; i < c->ol_n; ++i) { w = ov_type(&c->ol[i], (uint32_t)rl); if ((uint64_t)++n[w] == max_n_chain) s[w] = (int32_t)c->ol[i].shared_seed; }
		if (s[0] > 0 || s[1] > 0 || s[2] > 0 || s[3] > 0) {
			if ((uint64_t)n[3] >= max_n_chain && rl >= ocv_w) { /* coverage windows to rescue contained chains */
				uint64_t cws = 0, cwe;
				cwn = rl / ocv_w + (rl % ocv_w ? 1 : 0);
				if ((int64_t)cwn > c->cc_m) { c->cc_m = cwn + 16; c->cc = (uint64_t*)xrealloc(c->cc, c->cc_m * 8); }
				cc = c->cc;
				for (mm = 0; mm < cwn; ++mm, cws += ocv_w) {
					cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
					cc[mm] = (cwe - cws) * (max_n_chain >> 1); if (cc[mm] > UINT32_MAX) cc[mm] = UINT32_MAX; cc[mm] <<= 32;
				}
			}
			for (i = 0, kk = 0, lch = 0; i < c->ol_n; ++i) {
				hao_or_ovlp_t *r = &c->ol[i]; int keep = 0;
				w = ov_type(r, (uint32_t)rl);
				if ((int32_t)r->shared_seed >= s[w]) { if (cwn) cov_add(cc, cwn, ocv_w, rl, r->x_pos_s, (uint64_t)r->x_pos_e + 1); keep = 1; }
				else if (w == 3 && cwn > 0) {
					uint64_t rs = r->x_pos_s, re = (uint64_t)r->x_pos_e + 1, cw0 = 0, cw1 = 0, cws, cwe, os, oe;
					for (mm = rs / ocv_w, cws = mm * ocv_w; mm < cwn; ++mm, cws += ocv_w) {
						cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
						os = rs >= cws ? rs : cws; oe = re <= cwe ? re : cwe;
						if (oe <= os) break;
						if ((oe - os) + (uint64_t)(uint32_t)cc[mm] >= (cc[mm] >> 32)) cw1 += oe - os; else cw0 += oe - os;
					}
					if (cw0 >= (cw0 + cw1) * 0.7) { cov_add(cc, cwn, ocv_w, rl, rs, re); keep = 1; }
				}
				if (keep) {
					if (kk != (uint64_t)i) sw(&S, (int64_t)kk, i);
					if (c->ol[kk].align_length < chain_cutoff) lch = 1;
					++kk;
				}
			}
			c->ol_n = (int64_t)kk;
		}
	}
	intro_sort(&S, c->ol_n, lt_xs);
	if (lch) {                                                   /* drop weak chains shadowed by a much stronger one, :2061-2096 */
		int64_t kk, ll;
		for (i = ll = 0; i < c->ol_n; ++i) {
			if (c->ol[i].align_length < chain_cutoff) {
				uint64_t zs = c->ol[i].x_pos_s, ze = (uint64_t)c->ol[i].x_pos_e + 1, ob = (uint64_t)((ze - zs) * 0.95), ocn = (uint64_t)c->ol[i].align_length << 4;
				int64_t osc = (int64_t)(int32_t)c->ol[i].shared_seed * 16;
				if (ob < 16) ob = 16;
				for (kk = 0; kk < c->ol_n && ze > c->ol[kk].x_pos_s; ++kk) {
					uint64_t rs, re, os, oe;
					if (c->ol[kk].align_length < chain_cutoff || c->ol[kk].align_length < ocn || (int64_t)(int32_t)c->ol[kk].shared_seed < osc) continue;
					rs = c->ol[kk].x_pos_s; re = (uint64_t)c->ol[kk].x_pos_e + 1; os = rs >= zs ? rs : zs; oe = re <= ze ? re : ze;
					if (oe > os && oe - os >= ob) {
						uint64_t mm = c->ol[kk].non_homopolymer_errors, pp = H_ID(c->cl[mm]), kn = 0;
						for (; mm < (uint64_t)m && H_ID(c->cl[mm]) == pp && kn < ocn; ++mm) {
							uint64_t me = c->cl[mm].self_offset, ms = me - (c->cl[mm].cnt & 0xffu);
							if (ms >= os && me <= oe) ++kn;
						}
						if (kn >= ocn) break;
					}
				}
				if (kk < c->ol_n && ze > c->ol[kk].x_pos_s) continue;
			}
			if (ll != i) sw(&S, ll, i);
			++ll;
		}
		c->ol_n = ll;
	}
	for (i = 0; i < c->ol_n; ++i) c->ol[i].align_length = 0;    /* :2098 */
	/* gather fake cigars in final order */
	{
		uint64_t o = 0;
		if (c->ol_n + 1 > c->fco_m) { c->fco_m = c->ol_n + 64; c->fc_off = (uint64_t*)xrealloc(c->fc_off, c->fco_m * 8); }
		for (i = 0; i < c->ol_n; ++i) o += c->ol[i].fc_len;
		if ((int64_t)o + 1 > c->fc_m) { c->fc_m = o + 1024; c->fc = (uint64_t*)xrealloc(c->fc, c->fc_m * 8); }
		for (i = 0, o = 0; i < c->ol_n; ++i) { c->fc_off[i] = o; memcpy(c->fc + o, pool.a + fo[i], c->ol[i].fc_len * 8); o += c->ol[i].fc_len; }
		c->fc_off[c->ol_n] = o; c->fc_n = (int64_t)o;
	}
	free(fo); free(pool.a);
	*ol_out = c->ol; *fc_out = c->fc; *fc_off_out = c->fc_off; *cl_out = c->cl; *cl_n_out = m;
	return c->ol_n;
}


/* ------------------------------------------------------------------ */
/* f2: exact-overlap check after chaining (ecovlp.cpp:2803-2808 via :5103-5131) */
/* ------------------------------------------------------------------ */
void hao_or_exact(const hao_or_ctx *c, const hao_or_ovlp_t *ol, int64_t n, uint8_t *out)
{
	int64_t j, i;
	for (j = 0; j < n; ++j) {
		const hao_or_ovlp_t *z = &ol[j];
		const uint8_t *q = c->codes + c->off[z->x_id], *t = c->codes + c->off[z->y_id];
		int64_t tl = (int64_t)(c->off[z->y_id + 1] - c->off[z->y_id]);
		int64_t qs = z->x_pos_s, qe = (int64_t)z->x_pos_e + 1, ts = z->y_pos_s, te = (int64_t)z->y_pos_e + 1;
		uint8_t ok = qe - qs == te - ts;                       /* exact_ec_check: different lengths are never exact */
		for (i = 0; ok && i < qe - qs; ++i) {
			uint8_t a = q[qs + i] > 3 ? 4 : q[qs + i], b;      /* the read as characters: A C G T, anything else N */
			if (!z->y_pos_strand) b = t[ts + i] > 3 ? 4 : t[ts + i];
			else { uint8_t f = t[tl - 1 - (ts + i)]; b = f > 3 ? 4 : (uint8_t)(3 - f); }      /* strand 1: reverse complement, N stays N (Process_Read.cpp:564-614) */
			if (a != b) ok = 0;
		}
		out[j] = ok;
	}
}


/* ------------------------------------------------------------------ */
/* f3: windowed bit-vector edit distance (Levenshtein_distance.h:3727-3776) */
/* ------------------------------------------------------------------ */
static inline uint8_t ed_chr(const hao_or_ctx *c, uint32_t rid, int64_t pos, int rev)
{	/* character of the read on a strand, as a seq_nt4_table code: 0..3, 4 = N (Process_Read.cpp:524-614 builds the same strings) */
	const uint8_t *r = c->codes + c->off[rid]; int64_t L = (int64_t)(c->off[rid + 1] - c->off[rid]);
	uint8_t b = r[rev ? L - 1 - pos : pos];
	return b > 3 ? 4 : (rev ? (uint8_t)(3 - b) : b);
}

/* ------------------------------------------------------------------ */
/* f3, second variant: global alignment inside the band WITH traceback:
 * ed_band_cal_global_64_w_trace (Levenshtein_distance.h:3370-3442) on a cleared bit_extz_t, then gen_trace(ez, thre, 1) (:903-985).
 * Pattern and text are both consumed entirely (|pn - tn| <= thre); every column's D0 / VP / VN / HP / HN words are kept and the cigar is
 * read back from them (indels preferred, :924-936).  Cigar entries are push_trace's (:522-531): op << 14 | len, ops 0 match, 1 mismatch,
 * 2 more pattern, 3 more text.  out[q] = {err, ps, pe, ts, te, cigar entries}; err = INT32_MAX (pe = te = -1, no cigar) when the pair
 * has no alignment within thre.  Cigar q goes to cig + q * cap (entries beyond cap are counted, not written).
 * mode 3 = the semi-global variant with traceback, ed_band_cal_semi_64_w_absent_diag_trace (:3778-3848): the text is consumed, the pattern may start and end
 * inside the band; ps comes out of the walk (gen_trace with ptrim = abs_diag), te = tn - 1 whether or not an alignment exists. */
/* ------------------------------------------------------------------ */
static void tr_push(uint16_t *cg, int64_t cap, int32_t *n, int32_t op, int32_t len)
{	/* push_trace */
	while (len >= 0x3fff) { if (*n < cap) cg[*n] = (uint16_t)((op << 14) + 0x3fff); ++*n; len -= 0x3fff; }
	if (len) { if (*n < cap) cg[*n] = (uint16_t)((op << 14) + len); ++*n; }
}

#define HAO_CAT_(a, b) a##b
#define HAO_CAT(a, b) HAO_CAT_(a, b)
#define WT uint64_t
#define FN(x) HAO_CAT(ed64_, x)
#include "hao_oracle_ed.inc"
#undef WT
#undef FN
#define WT unsigned __int128
#define FN(x) HAO_CAT(ed128_, x)
#include "hao_oracle_ed.inc"
#undef WT
#undef FN

/* one band width per call, chosen like cal_exz_global does (Correct.cpp:15482-15494): 2 thre + 1 diagonals in one 64-bit word, else in two */
void hao_or_window_ed(const hao_or_ctx *c, const uint32_t *task, int64_t n, int32_t *out)
{
	int64_t q;
	for (q = 0; q < n; ++q) { if (2 * task[10 * q + 8] + 1 <= 64) ed64_window_ed(c, task + 10 * q, 1, out + 2 * q); else ed128_window_ed(c, task + 10 * q, 1, out + 2 * q); }
}

void hao_or_window_trace(const hao_or_ctx *c, const uint32_t *task, int64_t n, int mode, int32_t *out, uint16_t *cig, int64_t cap)
{
	int64_t q;
	for (q = 0; q < n; ++q) {
		if (2 * task[10 * q + 8] + 1 <= 64) ed64_window_trace(c, task + 10 * q, 1, mode, out + 6 * q, cig + q * cap, cap);
		else ed128_window_trace(c, task + 10 * q, 1, mode, out + 6 * q, cig + q * cap, cap);
	}
}
