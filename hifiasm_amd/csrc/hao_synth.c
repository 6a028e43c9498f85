/* Seeded synthetic genome / read generator (bench + test tooling, not on the hot path).
 *
 * genome = i.i.d. uniform ACGT (optionally with planted repeat
 * families + a tandem satellite array: the "repeat-rich" variant of BASELINE.md 2b);
 * reads: uniform start, strand Bernoulli(1/2), errors split 40/30/30
 * substitution/deletion/insertion.  Every read has its own counter-based RNG
 * stream (seed, read index), so any rank can generate any shard of the same set
 * and the bytes are identical on every machine.
 *
 * Base codes 0..3 = A,C,G,T (the reference's seq_nt4_table order, htab.cpp:17-34).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>

/* reads are independent RNG streams: both passes split [0, n) over host threads (HAO_SYNTH_THREADS, default = online cores, at most 64) */
static int synth_threads(uint64_t n)
{
	long t = sysconf(_SC_NPROCESSORS_ONLN); const char *e = getenv("HAO_SYNTH_THREADS");
	if (e) t = atol(e);
	if (t > 64) t = 64;
	if (t < 1) t = 1;
	if ((uint64_t)t > n / 64 + 1) t = (long)(n / 64 + 1);
	return (int)t;
}
typedef struct { void (*fn)(void *, uint64_t, uint64_t); void *arg; uint64_t lo, hi; } synth_job;
static void *synth_tramp(void *p) { synth_job *j = (synth_job*)p; j->fn(j->arg, j->lo, j->hi); return 0; }
static void synth_parallel(uint64_t n, void (*fn)(void *, uint64_t, uint64_t), void *arg)
{
	int t = synth_threads(n), i; pthread_t th[64]; synth_job jb[64];
	if (t <= 1) { fn(arg, 0, n); return; }
	for (i = 0; i < t; ++i) { jb[i].fn = fn; jb[i].arg = arg; jb[i].lo = n * i / t; jb[i].hi = n * (i + 1) / t; pthread_create(&th[i], 0, synth_tramp, &jb[i]); }
	for (i = 0; i < t; ++i) pthread_join(th[i], 0);
}

static inline uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

static inline uint64_t rnd_below(uint64_t *s, uint64_t n) /* n > 0; tiny modulo bias is irrelevant here */
{
	return splitmix64(s) % n;
}

/* out[size] <- codes 0..3 */
void hao_synth_genome(uint8_t *out, uint64_t size, uint64_t seed, int repeat_rich)
{
	uint64_t s = seed * 0xD1342543DE82EF95ULL + 1, i;
	for (i = 0; i + 32 <= size; i += 32) {
		uint64_t r = splitmix64(&s); int j;
		for (j = 0; j < 32; ++j) out[i + j] = (r >> (2 * j)) & 3;
	}
	for (; i < size; ++i) out[i] = splitmix64(&s) & 3;
	if (repeat_rich) {
		/* level 1: per 2 Mb, 10 families x 25 copies x 4 kb + a 300 x 171 bp tandem array;
		 * level 2 (miniature, for small fixtures): per 100 kb, 3 families x 8 copies x 1.5 kb + 40 x 171 bp */
		const uint64_t unit_sz = repeat_rich == 1 ? 2000000 : 100000;
		const int n_fam = repeat_rich == 1 ? 10 : 3, n_cp = repeat_rich == 1 ? 25 : 8, e_len = repeat_rich == 1 ? 4000 : 1500, n_tan = repeat_rich == 1 ? 300 : 40;
		uint64_t n_units = size / unit_sz ? size / unit_sz : 1, u;
		uint8_t elem[4000], unit[171];
		for (u = 0; u < n_units; ++u) {
			int fam, c, j;
			for (fam = 0; fam < n_fam; ++fam) {
				for (j = 0; j < e_len; ++j) elem[j] = splitmix64(&s) & 3;
				for (c = 0; c < n_cp; ++c) {
					uint64_t p = size > (uint64_t)e_len ? rnd_below(&s, size - e_len) : 0;
					for (j = 0; j < e_len && p + j < size; ++j) {
						uint8_t b = elem[j];
						if (rnd_below(&s, 1000) < 15) b = (b + 1 + rnd_below(&s, 3)) & 3; /* 1.5 % divergence */
						out[p + j] = b;
					}
				}
			}
			for (j = 0; j < 171; ++j) unit[j] = splitmix64(&s) & 3;
			{
				uint64_t alen = (uint64_t)n_tan * 171, p = size > alen ? rnd_below(&s, size - alen) : 0, q;
				for (q = 0; q < alen && p + q < size; ++q) out[p + q] = unit[q % 171];
			}
		}
	}
}

/* Generate read `rid` into buf (capacity cap codes); returns its length.
 * err_ppm = error rate in parts per million; len_jit = +- uniform jitter on read_len;
 * n_ppm = rate of N bases (code 4). */
static uint32_t synth_one(const uint8_t *g, uint64_t G, uint64_t rid, uint32_t read_len, uint32_t len_jit, uint32_t err_ppm, uint32_t n_ppm, uint64_t seed, uint8_t *buf, uint32_t cap)
{
	uint64_t s = (seed + 0x51ED270B7ULL) * 0x9E3779B97F4A7C15ULL + rid * 0xD6E8FEB86659FD93ULL;
	uint32_t L = read_len, i, n = 0;
	uint64_t start; int rev;
	splitmix64(&s);
	if (len_jit) L = read_len - len_jit + (uint32_t)rnd_below(&s, 2 * (uint64_t)len_jit + 1);
	if (L > G) L = (uint32_t)G;
	start = rnd_below(&s, G - L + 1);
	rev = splitmix64(&s) & 1;
	for (i = 0; i < L && n < cap; ++i) {
		uint8_t b = rev ? 3 - g[start + L - 1 - i] : g[start + i];
		uint64_t r = splitmix64(&s);
		uint32_t e = (uint32_t)(r % 1000000u);
		if (e < err_ppm) {
			uint32_t t = (uint32_t)((r >> 32) % 10);
			if (t < 4) buf[n++] = (b + 1 + (uint32_t)((r >> 40) % 3)) & 3;      /* substitution */
			else if (t < 7) { /* deletion */ }
			else { buf[n++] = b; if (n < cap) buf[n++] = (r >> 44) & 3; }          /* insertion after */
		} else buf[n++] = b;
		if (n_ppm && n > 0 && (uint32_t)((r >> 20) % 1000000u) < n_ppm) buf[n - 1] = 4;
	}
	return n;
}

/* pass 1: lengths of reads [rid0, rid0+n). */
typedef struct {
	const uint8_t *g; uint64_t G, rid0; uint32_t read_len, len_jit, err_ppm, n_ppm; uint64_t seed;
	uint32_t *len_out; uint8_t *codes; const uint64_t *code_off; uint8_t *packed; const uint64_t *pk_off;
} synth_args;

static void synth_len_range(void *p, uint64_t lo, uint64_t hi)
{
	const synth_args *a = (const synth_args*)p; uint32_t cap = 2 * (a->read_len + a->len_jit) + 16; uint64_t i;
	uint8_t *buf = (uint8_t*)malloc(cap);
	for (i = lo; i < hi; ++i) a->len_out[i] = synth_one(a->g, a->G, a->rid0 + i, a->read_len, a->len_jit, a->err_ppm, a->n_ppm, a->seed, buf, cap);
	free(buf);
}

void hao_synth_read_lengths(const uint8_t *g, uint64_t G, uint64_t rid0, uint64_t n, uint32_t read_len, uint32_t len_jit, uint32_t err_ppm, uint32_t n_ppm, uint64_t seed, uint32_t *len_out)
{
	synth_args a; memset(&a, 0, sizeof(a));
	a.g = g; a.G = G; a.rid0 = rid0; a.read_len = read_len; a.len_jit = len_jit; a.err_ppm = err_ppm; a.n_ppm = n_ppm; a.seed = seed; a.len_out = len_out;
	synth_parallel(n, synth_len_range, &a);
}

/* pass 2: fill. Any of the outputs may be NULL.
 *   codes   : concatenated codes (0..4), offsets code_off[i] (n+1 entries, caller computed from lengths)
 *   packed  : reference read-store layout, len/4+1 bytes per read, 4 bases/byte MSB first, N -> A
 *             (ha_compress_base, Process_Read.cpp:792-850); byte offsets pk_off[i] */
static void synth_fill_range(void *p, uint64_t lo, uint64_t hi)
{
	const synth_args *a = (const synth_args*)p; uint32_t cap = 2 * (a->read_len + a->len_jit) + 16; uint64_t i;
	uint8_t *buf = (uint8_t*)malloc(cap);
	for (i = lo; i < hi; ++i) {
		uint32_t L = synth_one(a->g, a->G, a->rid0 + i, a->read_len, a->len_jit, a->err_ppm, a->n_ppm, a->seed, buf, cap), j;
		if (a->codes) memcpy(a->codes + a->code_off[i], buf, L);
		if (a->packed) {
			uint8_t *d = a->packed + a->pk_off[i];
			memset(d, 0, L / 4 + 1);
			for (j = 0; j < L; ++j) d[j >> 2] |= (uint8_t)((buf[j] & 3 & -(buf[j] < 4)) << (6 - 2 * (j & 3)));
		}
	}
	free(buf);
}

void hao_synth_reads(const uint8_t *g, uint64_t G, uint64_t rid0, uint64_t n, uint32_t read_len, uint32_t len_jit, uint32_t err_ppm, uint32_t n_ppm, uint64_t seed,
					 uint8_t *codes, const uint64_t *code_off, uint8_t *packed, const uint64_t *pk_off)
{
	synth_args a; memset(&a, 0, sizeof(a));
	a.g = g; a.G = G; a.rid0 = rid0; a.read_len = read_len; a.len_jit = len_jit; a.err_ppm = err_ppm; a.n_ppm = n_ppm; a.seed = seed;
	a.codes = codes; a.code_off = code_off; a.packed = packed; a.pk_off = pk_off;
	synth_parallel(n, synth_fill_range, &a);
}

/* FASTA / FASTQ text of reads [rid0, rid0+n) straight to a file (no codes array: a 7.5 Gbase set does not have to sit in memory twice).
 * Returns the number of bytes written, 0 on an I/O error. */
#include <stdio.h>
uint64_t hao_synth_fasta_file(const uint8_t *g, uint64_t G, uint64_t rid0, uint64_t n, uint32_t read_len, uint32_t len_jit, uint32_t err_ppm, uint32_t n_ppm, uint64_t seed,
							  const char *path, int fastq, int qual)
{
	static const char tab[] = "ACGTN";
	uint32_t cap = 2 * (read_len + len_jit) + 16; uint64_t i, tot = 0; uint32_t j;
	uint8_t *buf = (uint8_t*)malloc(cap); char *line = (char*)malloc((size_t)cap * 2 + 64);
	FILE *fp = fopen(path, "wb");
	if (!fp) { free(buf); free(line); return 0; }
	setvbuf(fp, 0, _IOFBF, 1 << 22);
	for (i = 0; i < n; ++i) {
		uint32_t L = synth_one(g, G, rid0 + i, read_len, len_jit, err_ppm, n_ppm, seed, buf, cap); size_t o;
		o = (size_t)sprintf(line, "%cr%llu\n", fastq ? '@' : '>', (unsigned long long)(rid0 + i));
		for (j = 0; j < L; ++j) line[o + j] = tab[buf[j]];
		o += L; line[o++] = '\n';
		if (fastq) { line[o++] = '+'; line[o++] = '\n'; memset(line + o, 33 + qual, L); o += L; line[o++] = '\n'; }
		if (fwrite(line, 1, o, fp) != o) { tot = 0; break; }
		tot += o;
	}
	fclose(fp); free(buf); free(line);
	return tot;
}

/* codes -> FASTA text (ACGTN). returns bytes written. out must hold sum(len)+n*(name+3). */
uint64_t hao_synth_fasta(const uint8_t *codes, const uint64_t *code_off, uint64_t rid0, uint64_t n, char *out, int fastq, int qual)
{
	static const char tab[] = "ACGTN"; uint64_t i, o = 0, j;
	for (i = 0; i < n; ++i) {
		char name[32]; int nl = 0; uint64_t v = rid0 + i, L = code_off[i + 1] - code_off[i]; char tmp[24]; int tl = 0;
		do { tmp[tl++] = '0' + v % 10; v /= 10; } while (v);
		name[nl++] = fastq ? '@' : '>'; name[nl++] = 'r';
		while (tl) name[nl++] = tmp[--tl];
		name[nl++] = '\n';
		memcpy(out + o, name, nl); o += nl;
		for (j = 0; j < L; ++j) out[o + j] = tab[codes[code_off[i] + j]];
		o += L; out[o++] = '\n';
		if (fastq) {
			out[o++] = '+'; out[o++] = '\n';
			memset(out + o, 33 + qual, L); o += L; out[o++] = '\n';
		}
	}
	return o;
}
