// Window alignment on the device: the banded bit-vector edit distance the reference runs per (query window, candidate) pair after chaining -
// ed_band_cal_semi_64_w_absent_diag (Levenshtein_distance.h:3727-3776; callers Correct.cpp:3897,4092,4156) - and its four traced siblings:
// ed_band_cal_global_64_w_trace (:3370-3442), ed_band_cal_extension_64_0/1_w_trace (:3512-3735), ed_band_cal_semi_64_w_absent_diag_trace (:3778-3848), each
// followed by gen_trace (:903-985); one-word bands (thre <= 31) and two-word bands (thre 32 .. 63: the reference's HA_ED_INIT(128) text, :1287-2129).
//
// What is forced by bit-exactness is Myers' recurrence on the band (five logic operations per text column) and the reference's tie rules in the final scans
// and in the traceback.  Everything around it is laid out for the device:
//   * Work is a TILE of 64 tasks of the batch sorted by text window: a read has ~20 windows and each window ~60 candidates, so a wave's lanes mostly share ONE
//     text.  The wave decodes the text once - strand folded, N sites of the read folded in by one pass over the read's (short) N list - into LDS as a byte
//     per base, 1024 columns at a time, and every lane reads the column's character from the same LDS address (a broadcast).  A tile that spans several
//     windows is processed window by window (the lanes of the other windows wait).
//   * A lane owns one candidate: its pattern streams out of the packed 2-bit read store 32 bases per 8-byte load, forwards or backwards, complemented
//     on the reverse strand, with a cursor in the read's sorted N list instead of a search per character.
//   * The sweep is a resumable per-column step (hao_al_column) over a lane state (hao_al_state), so the staged text can be refilled between
//     column ranges and lanes whose alignment died simply stop stepping.
//   * Alignments with traceback run TWICE: a first sweep without any column storage decides which pairs end within the threshold; only those get the 40
//     (80) bytes per text column of the second sweep and the traceback.  (Round 2 kept the columns of every pair: 11 GB for 383 k pairs.)
// The lane-level functions are plain C++ (HAO_ALIGN_HOST_MODEL: tests/ed_model.cpp compiles this file with g++, replaces a wave by a loop over 64 lanes and is
// checked against the oracle on the CPU); only the kernels at the end use wave intrinsics.
#pragma once
#include <stdint.h>
#include "hao.h"
#ifdef HAO_ALIGN_HOST_MODEL
#include <string.h>
#define HAO_AL_FN static inline
#define HAO_AL_MFN inline
#define HAO_AL_MEMCPY memcpy
#else
#include "hao_common.cuh"
#define HAO_AL_FN __host__ __device__ __forceinline__
#define HAO_AL_MFN __host__ __device__ __forceinline__
#define HAO_AL_MEMCPY __builtin_memcpy
#endif

typedef unsigned __int128 hao_u128;
#define HAO_AL_NONE 0x7fffffff           // "no alignment" (INT32_MAX, the reference's clear_align state)
#define HAO_AL_CH 1024                   // text columns staged per refill
enum { HAO_AL_GLOBAL = 0, HAO_AL_EXT_FWD = 1, HAO_AL_EXT_BWD = 2, HAO_AL_SEMI = 3, HAO_AL_ED = 4 };      // 0 .. 3: the numbering of Correct.cpp:14536-14545; 4: semi-global without traceback

// Bands of more than two words (thre 64 .. 127: the reference's *_infi_* functions, Levenshtein_distance.h:2134-3100, picked by cal_exz_infi, Correct.cpp:14508-14564, with
// nword = ceil((2 thre + 1) / 64)): the same lane functions over an N-word integer.  The word count is part of the semantics - what shifts in at the top of the last word
// reaches the band after enough columns - so N is exactly the reference's nword (3 or 4), not "wide enough".  tests/ed_model.cpp instantiates these on the CPU and
// tests/test_ed_model_cpu.py compares them with the reference's own functions (tests/golden/ed_wide.npz) on the CPU, tests/test_gpu_zz_new.py on the device
// (hao_capi_rest.hpp launches hao_al_kernel<hao_wide<3|4>, ...> for the tasks whose band needs three / four words).
template<int N> struct hao_wide {
	uint64_t a[N];
	HAO_AL_MFN hao_wide() {}
	HAO_AL_MFN hao_wide(int v) { a[0] = (uint64_t)(int64_t)v; for (int k = 1; k < N; ++k) a[k] = v < 0 ? ~0ULL : 0ULL; }
	HAO_AL_MFN explicit operator int32_t() const { return (int32_t)a[0]; }
	HAO_AL_MFN explicit operator uint64_t() const { return a[0]; }
	HAO_AL_MFN bool operator!() const { uint64_t o = 0; for (int k = 0; k < N; ++k) o |= a[k]; return o == 0; }
	HAO_AL_MFN hao_wide operator~() const { hao_wide r; for (int k = 0; k < N; ++k) r.a[k] = ~a[k]; return r; }
	HAO_AL_MFN hao_wide operator|(const hao_wide &o) const { hao_wide r; for (int k = 0; k < N; ++k) r.a[k] = a[k] | o.a[k]; return r; }
	HAO_AL_MFN hao_wide operator&(const hao_wide &o) const { hao_wide r; for (int k = 0; k < N; ++k) r.a[k] = a[k] & o.a[k]; return r; }
	HAO_AL_MFN hao_wide operator^(const hao_wide &o) const { hao_wide r; for (int k = 0; k < N; ++k) r.a[k] = a[k] ^ o.a[k]; return r; }
	HAO_AL_MFN hao_wide operator+(const hao_wide &o) const { hao_wide r; uint64_t c = 0; for (int k = 0; k < N; ++k) { const uint64_t x = a[k] + c; c = x < c; r.a[k] = x + o.a[k]; c |= r.a[k] < o.a[k]; } return r; }
	HAO_AL_MFN hao_wide operator-(int v) const { hao_wide r; uint64_t b = (uint64_t)v; for (int k = 0; k < N; ++k) { r.a[k] = a[k] - b; b = a[k] < b; } return r; }      // (v >= 0)
	HAO_AL_MFN hao_wide operator<<(int s) const {
		hao_wide r; const int ws = s >> 6, bs = s & 63;
		for (int k = N - 1; k >= 0; --k) { const int j = k - ws; uint64_t v = j >= 0 ? a[j] << bs : 0; if (bs && j - 1 >= 0) v |= a[j - 1] >> (64 - bs); r.a[k] = v; }
		return r;
	}
	HAO_AL_MFN hao_wide operator>>(int s) const {
		hao_wide r; const int ws = s >> 6, bs = s & 63;
		for (int k = 0; k < N; ++k) { const int j = k + ws; uint64_t v = j < N ? a[j] >> bs : 0; if (bs && j + 1 < N) v |= a[j + 1] << (64 - bs); r.a[k] = v; }
		return r;
	}
	HAO_AL_MFN hao_wide &operator|=(const hao_wide &o) { for (int k = 0; k < N; ++k) a[k] |= o.a[k]; return *this; }
	HAO_AL_MFN hao_wide &operator<<=(int s) { *this = *this << s; return *this; }
	HAO_AL_MFN hao_wide &operator>>=(int s) { *this = *this >> s; return *this; }
};
// 64-bit word h of a band vector
HAO_AL_FN uint64_t hao_al_word64(uint64_t v, int h) { (void)h; return v; }
HAO_AL_FN uint64_t hao_al_word64(hao_u128 v, int h) { return (uint64_t)(v >> (64 * h)); }
template<int N> HAO_AL_FN uint64_t hao_al_word64(const hao_wide<N> &v, int h) { return v.a[h]; }

// words a band of 2 thre + 1 diagonals needs (the reference's nword); one launch serves one band word type and skips the tasks of the others
// (cal_exz_global / cal_exz_infi pick by band width, Correct.cpp:15482-15494, 14508-14564)
HAO_AL_FN uint32_t hao_al_nword(uint32_t thre) { return (2 * thre + 1 + 63) >> 6; }
template<typename WT> HAO_AL_FN bool hao_al_mine(uint32_t thre) { return hao_al_nword(thre) == (uint32_t)(sizeof(WT) / 8); }
// w_<sf>_set_bit_lsub (:1029-1033): the low l bits set.  The 128-bit macro shifts a 64-bit word by 64 when l == 64: undefined in C, 0 on x86-64 (shift count
// modulo 64) - the reference as built starts abs_diag = 64 from VN = 0, and so does this.
template<typename WT> HAO_AL_FN WT hao_al_lsub(int32_t l) { return (sizeof(WT) == 16 && l == 64) ? (WT)0 : (WT)((((WT)1) << l) - 1); }

struct hao_ed_reads { const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len; const uint64_t *nsite_off; const uint32_t *nsite; };

// A strand-oriented walk over a stretch of one read: element k is the base at forward coordinate f0 + dir * k, complemented when comp.
struct hao_al_walk { const uint8_t *rd; const uint32_t *ns; int64_t nb, ne; int64_t f0; int32_t dir, comp; };
// stretch [pos, pos + len) of strand `rev` of read `rid`, read from its first element (from_end: from its last one backwards)
HAO_AL_FN hao_al_walk hao_al_walk_of(const hao_ed_reads &R, uint32_t rid, uint32_t pos, uint32_t len, uint32_t rev, bool from_end)
{
	hao_al_walk w; const int64_t L = R.len[rid];
	w.rd = R.packed + R.pk_off[rid];
	const int64_t s0 = from_end ? (int64_t)pos + len - 1 : (int64_t)pos; const int32_t sdir = from_end ? -1 : 1;
	w.f0 = rev ? L - 1 - s0 : s0; w.dir = rev ? -sdir : sdir; w.comp = rev ? 1 : 0;
	w.ns = R.nsite; w.nb = R.nsite_off ? (int64_t)R.nsite_off[rid] : 0; w.ne = R.nsite_off ? (int64_t)R.nsite_off[rid + 1] : 0;
	return w;
}
// the 32 bases [32 wi, 32 wi + 32) of a packed read, base 32 wi in bits 63..62 (reads are packed 4 bases per byte, first base in bits 7..6; the store has
// 16 bytes of slack behind its last read, so the load may run past a read's last byte)
HAO_AL_FN uint64_t hao_al_word32(const uint8_t *rd, int64_t wi)
{
	uint64_t v; HAO_AL_MEMCPY(&v, rd + 8 * wi, 8);
	return __builtin_bswap64(v);
}
// first index in [b, e) of the ascending list ns with ns[i] >= f
HAO_AL_FN int64_t hao_al_nlower(const uint32_t *ns, int64_t b, int64_t e, int64_t f)
{ while (b < e) { const int64_t m = (b + e) >> 1; if ((int64_t)ns[m] < f) b = m + 1; else e = m; } return b; }

// ---- text: staged once per wave ----
// elements [k0, k0 + n) of the walk as codes 0..3 into codes[0 .. n): lane `lane` of `nl` takes every nl-th element ...
HAO_AL_FN void hao_al_stage_bases(const hao_al_walk &w, int64_t k0, int32_t n, uint8_t *codes, int lane, int nl)
{
	for (int32_t k = lane; k < n; k += nl) {
		const int64_t f = w.f0 + (int64_t)w.dir * (k0 + k);
		const uint32_t b = (w.rd[f >> 2] >> (6 - 2 * (f & 3))) & 3;
		codes[k] = (uint8_t)(w.comp ? 3 - b : b);
	}
}
// ... then (after the bases are in place) the N sites of the read inside the staged range become code 4: a pass over the read's N list, not over the bases
HAO_AL_FN void hao_al_stage_nsites(const hao_al_walk &w, int64_t k0, int32_t n, uint8_t *codes, int lane, int nl)
{
	if (w.nb >= w.ne || n <= 0) return;
	const int64_t fa = w.f0 + (int64_t)w.dir * k0, fb = w.f0 + (int64_t)w.dir * (k0 + n - 1), flo = fa < fb ? fa : fb, fhi = fa < fb ? fb : fa;
	for (int64_t j = hao_al_nlower(w.ns, w.nb, w.ne, flo) + lane; j < w.ne && (int64_t)w.ns[j] <= fhi; j += nl) codes[((int64_t)w.ns[j] - w.f0) * w.dir - k0] = 4;
}

// ---- pattern: streamed per lane ----
struct hao_al_pstream { hao_al_walk w; int64_t k, cur; uint64_t word; int64_t ni; };
HAO_AL_FN void hao_al_pstream_init(hao_al_pstream &P, const hao_al_walk &w)
{
	P.w = w; P.k = 0; P.cur = -1; P.word = 0;
	// N cursor: the next site the walk can meet - at or after f0 when walking up, at or before f0 when walking down
	const int64_t lb = hao_al_nlower(w.ns, w.nb, w.ne, w.f0);
	P.ni = w.dir > 0 ? lb : ((lb < w.ne && (int64_t)w.ns[lb] == w.f0) ? lb : lb - 1);
}
// next element of the walk: 0..3, 4 = N
HAO_AL_FN uint32_t hao_al_pstream_next(hao_al_pstream &P)
{
	const int64_t f = P.w.f0 + (int64_t)P.w.dir * P.k; ++P.k;
	const int64_t wi = f >> 5;
	if (wi != P.cur) { P.word = hao_al_word32(P.w.rd, wi); P.cur = wi; }
	uint32_t c = (uint32_t)(P.word >> (62 - 2 * (f & 31))) & 3;
	if (P.w.comp) c = 3 - c;
	if (P.w.dir > 0) { if (P.ni < P.w.ne && (int64_t)P.w.ns[P.ni] == f) { c = 4; ++P.ni; } }
	else if (P.ni >= P.w.nb && (int64_t)P.w.ns[P.ni] == f) { c = 4; --P.ni; }
	return c;
}

// ---- lane state ----
template<typename WT> struct hao_al_state {
	WT eq[4], VP, VN, HP, HN, D0;        // Myers' vectors over the band: match masks per character (N matches nothing: no mask), vertical / horizontal deltas, diagonal zeros
	WT top;                              // the band's highest diagonal: where the next pattern character enters
	int32_t pn, tn, thre, adiag, cut;    // (clipped) lengths, threshold, missing leading diagonals, error bound of the sweep (3 thre)
	int32_t err, i_bd;                   // running error on the band's lowest diagonal; pattern index of the character that entered last
	int32_t best, a_p, a_t, tmp_e;       // extension modes: best end so far (error, pattern / text index), running error along the pattern's last row
	int32_t alive, dead;                 // alive: the task takes part in the sweep; dead: the sweep was abandoned (error bound passed)
	hao_al_pstream ps;
};

// the match masks are only ever indexed by compile-time constants (a run-time index would push the whole lane state into scratch memory: measured 15 x slower)
template<typename WT> HAO_AL_FN WT hao_al_eq_of(const hao_al_state<WT> &S, uint32_t c)
{ return c == 0 ? S.eq[0] : c == 1 ? S.eq[1] : c == 2 ? S.eq[2] : c == 3 ? S.eq[3] : (WT)0; }
template<typename WT> HAO_AL_FN void hao_al_eq_or(hao_al_state<WT> &S, uint32_t c, WT m)
{ S.eq[0] |= c == 0 ? m : (WT)0; S.eq[1] |= c == 1 ? m : (WT)0; S.eq[2] |= c == 2 ? m : (WT)0; S.eq[3] |= c == 3 ? m : (WT)0; }

// set-up of a task (the checks and initial vectors of the reference's functions): false = the answer is already known (no alignment)
template<typename WT, int MODE> HAO_AL_FN bool hao_al_init(hao_al_state<WT> &S, const hao_ed_reads &R, const hao_ed_task_t &T)
{
	const bool back = MODE == HAO_AL_EXT_BWD;
	S.pn = (int32_t)T.p_len; S.tn = (int32_t)T.t_len; S.thre = (int32_t)T.thre; S.adiag = (MODE == HAO_AL_SEMI || MODE == HAO_AL_ED) ? (int32_t)T.abs_diag : 0;
	S.cut = S.thre + (S.thre << 1); S.best = HAO_AL_NONE; S.a_p = -1; S.a_t = -1; S.tmp_e = HAO_AL_NONE; S.dead = 0; S.alive = 0;
	const int32_t thre = S.thre;
	if (MODE == HAO_AL_ED) { if (S.pn > S.tn + S.cut || S.tn > S.pn + S.cut || S.tn <= 0) return false; }
	else if (MODE == HAO_AL_SEMI) { if (S.pn <= 0 || S.tn <= 0 || S.pn > S.tn + S.cut || S.tn > S.pn + S.cut) return false; }
	else if (MODE == HAO_AL_GLOBAL) { if (S.pn <= 0 || S.tn <= 0 || S.pn > S.tn + thre || S.tn > S.pn + thre) return false; }
	else { if (S.pn <= 0 || S.tn <= 0) return false; if (S.pn > S.tn + thre) S.pn = S.tn + thre; else if (S.tn > S.pn + thre) S.tn = S.pn + thre; }
	hao_al_pstream_init(S.ps, hao_al_walk_of(R, T.p_rid, T.p_pos, T.p_len, T.p_rev, back));
	S.eq[0] = 0; S.eq[1] = 0; S.eq[2] = 0; S.eq[3] = 0;
	int32_t first, bd;
	if (MODE == HAO_AL_SEMI || MODE == HAO_AL_ED) {      // the band starts abs_diag diagonals in: the pattern was clipped at the start of its read
		first = S.adiag; bd = ((thre << 1) + 1) - S.adiag; S.i_bd = (thre << 1) - S.adiag; S.err = S.adiag;
		S.VP = 0; S.VN = hao_al_lsub<WT>(S.adiag);
	} else {                                             // both strings start together: the band is centred on the main diagonal
		first = thre; bd = thre + 1; S.i_bd = thre; S.err = thre;
		S.VN = (((WT)1) << thre) - 1; S.VP = ((((WT)1) << ((thre << 1) + 1)) - 1) ^ S.VN;
	}
	if (bd > S.pn) bd = S.pn;
	WT mm = ((WT)1) << first;
	for (int32_t i = 0; i < bd; ++i) { hao_al_eq_or(S, hao_al_pstream_next(S.ps), mm); mm <<= 1; }      // (an N in the pattern sets no mask: the reference's Peq[4] is cleared after its loop)
	S.top = ((WT)1) << (thre << 1);
	S.HP = 0; S.HN = 0; S.D0 = 0;
	S.alive = 1;
	return true;
}

// the five words of text column i of a traced sweep: [column][word][pair], so that the lanes of a wave write consecutive addresses
template<typename WT> HAO_AL_FN void hao_al_keep(const hao_al_state<WT> &S, uint64_t *col, uint64_t stride, int32_t i)
{
	constexpr int NW = sizeof(WT) / 8;
	uint64_t *w_ = col + 5 * NW * (uint64_t)i * stride;
#define HAO_AL_PUT(k_, v_) { for (int h_ = 0; h_ < NW; ++h_) w_[((uint64_t)(k_) * NW + h_) * stride] = hao_al_word64(v_, h_); }
	HAO_AL_PUT(0, S.D0) HAO_AL_PUT(1, S.VP) HAO_AL_PUT(2, S.VN) HAO_AL_PUT(3, S.HP) HAO_AL_PUT(4, S.HN)
#undef HAO_AL_PUT
}

// text column i (character tc) of the sweep.  KEEP: the column's vectors go to the scratch array.
template<typename WT, int MODE, bool KEEP> HAO_AL_FN void hao_al_column(hao_al_state<WT> &S, uint32_t tc, int32_t i, uint64_t *col, uint64_t stride)
{
	WT X = hao_al_eq_of(S, tc) | S.VN;
	S.D0 = ((S.VP + (X & S.VP)) ^ S.VP) | X; S.HN = S.VP & S.D0; S.HP = S.VN | ~(S.VP | S.D0);
	X = S.D0 >> 1; S.VN = X & S.HP; S.VP = S.HN | ~(X | S.HP);
	if (!(S.D0 & (WT)1)) { ++S.err; if (S.err > S.cut) { S.dead = 1; return; } }
	const bool last = i == S.tn - 1;
	if ((MODE == HAO_AL_EXT_FWD || MODE == HAO_AL_EXT_BWD) && !last) {      // error along the pattern's last row, once the band has reached it
		const int32_t thre = S.thre, pe_l = S.pn - 1; int32_t poff = i - thre, k = i + thre - pe_l;
		if (k >= 0) {
			if (S.tmp_e == HAO_AL_NONE) { S.tmp_e = S.err; for (k = 0; poff < pe_l; ++poff, ++k) { S.tmp_e += (int32_t)((S.VP >> k) & (WT)1); S.tmp_e -= (int32_t)((S.VN >> k) & (WT)1); } }
			else { k = (thre << 1) - k; if (k >= 0) { S.tmp_e += (int32_t)((S.HP >> k) & (WT)1); S.tmp_e -= (int32_t)((S.HN >> k) & (WT)1); } }
			if (S.tmp_e <= thre && S.tmp_e < S.best) { S.best = S.tmp_e; S.a_p = pe_l; S.a_t = i; }
		}
	}
	if (!last) { S.eq[0] >>= 1; S.eq[1] >>= 1; S.eq[2] >>= 1; S.eq[3] >>= 1; }
	if (KEEP) hao_al_keep(S, col, stride, i);
	if (!last) {
		++S.i_bd;
		if (S.i_bd < S.pn) hao_al_eq_or(S, hao_al_pstream_next(S.ps), S.top);
	}
}

// push_trace (:522-531); entries past the capacity are counted only
HAO_AL_FN void hao_al_push(uint16_t *cg, uint32_t cap, int32_t &n, int32_t op, int32_t len)
{
	while (len >= 0x3fff) { if ((uint32_t)n < cap) cg[n] = (uint16_t)((op << 14) + 0x3fff); ++n; len -= 0x3fff; }
	if (len) { if ((uint32_t)n < cap) cg[n] = (uint16_t)((op << 14) + len); ++n; }
}
// gen_trace (:903-985) over the kept columns: from text column i (exclusive), band bit sft, pattern offset poff and error cur back to the start; indels are
// preferred over (mis)matches on ties (:924-936).  Emits push_trace's entries in walking order (op << 14 | len; 0 match, 1 mismatch, 2 more pattern,
// 3 more text); returns the pattern offset where the walk ended, + 1.
template<typename WT> HAO_AL_FN int32_t hao_al_walk_back(const uint64_t *col, uint64_t stride, int32_t thre, int32_t i, int32_t sft, int32_t poff, int32_t cur,
		uint16_t *cg, uint32_t cap, int32_t &ncg, int32_t &pdir, int32_t &pdn)
{
	constexpr int NW = sizeof(WT) / 8;
	const int32_t low = thre << 1; int32_t d = 0;
	while (i > 0 && cur > 0) {
		const uint64_t *w_ = col + 5 * NW * (uint64_t)(i - 1) * stride;
#define HAO_AL_BIT(k_, b_) ((int32_t)((w_[((uint64_t)(k_) * NW + ((b_) >> 6)) * stride] >> ((b_) & 63)) & 1ULL))      /* bit b_ of kept value k_ (D0, VP, VN, HP, HN) */
		const int32_t D = cur - (HAO_AL_BIT(0, sft) ^ 1); int32_t mn = D; d = 0;
		if (sft != low) { const int32_t H = cur + HAO_AL_BIT(4, sft) - HAO_AL_BIT(3, sft); if (H + 1 == cur && H <= mn) { mn = H; d = 3; } }
		if (sft != 0) { const int32_t V = cur + HAO_AL_BIT(2, sft - 1) - HAO_AL_BIT(1, sft - 1); if (V + 1 == cur && V <= mn) { mn = V; d = 2; } }
#undef HAO_AL_BIT
		if (d == 0) { if (D != cur) d = 1; --i; --poff; }
		else if (d == 2) { --sft; --poff; }
		else { --i; ++sft; }
		if (d == pdir) ++pdn; else { if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = 1; }
		cur = mn;
	}
	if (i > 0) { d = 0; poff -= i; if (d == pdir) pdn += i; else { if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = i; } }
	return poff + 1;
}

// after the last column: the final scans along the last text column, the result, and - TRACE - the traceback into cg.
// Returns true iff the task has (TRACE: got) a cigar: the first, column-free sweep uses it to select the tasks of the second.
template<typename WT, int MODE, bool TRACE> HAO_AL_FN bool hao_al_finish(hao_al_state<WT> &S, const hao_ed_task_t &T, hao_trace_result_t &res, const uint64_t *col, uint64_t stride,
		uint16_t *cg, uint32_t cap)
{
	const bool back = MODE == HAO_AL_EXT_BWD, ext = MODE == HAO_AL_EXT_FWD || MODE == HAO_AL_EXT_BWD;
	const int32_t pidx = (int32_t)T.p_len - 1, tidx = (int32_t)T.t_len - 1, thre = S.thre, pn = S.pn, tn = S.tn;
	res.err = HAO_AL_NONE; res.n_cigar = 0;
	if (ext) { res.ps = back ? HAO_AL_NONE : 0; res.pe = back ? pidx : -1; res.ts = back ? HAO_AL_NONE : 0; res.te = back ? tidx : -1; }
	else { res.ps = (MODE == HAO_AL_SEMI || MODE == HAO_AL_ED) ? -1 : 0; res.pe = -1; res.ts = 0; res.te = (MODE == HAO_AL_SEMI || MODE == HAO_AL_ED) ? tn - 1 : -1; }
	if (MODE == HAO_AL_SEMI || MODE == HAO_AL_ED) res.te = (int32_t)T.t_len - 1;
	if (!S.alive) return false;
	int32_t ez = HAO_AL_NONE, pe = -1;
	if (ext) {
		if (!S.dead) {      // along the last column, from the band's lowest diagonal down the pattern
			int32_t site = tn - 1 - thre, err = S.err; WT VP = S.VP, VN = S.VN;
			while (site < pn - 1) {
				err += (int32_t)(VP & (WT)1); VP >>= 1; err -= (int32_t)(VN & (WT)1); VN >>= 1; ++site;
				if (err <= thre && err < S.best) { S.best = err; S.a_p = site; S.a_t = tn - 1; }
			}
			if (err <= thre && err < S.best) { S.best = err; S.a_p = site; S.a_t = tn - 1; }
		}
		res.err = S.best;
		if (back) { res.ps = S.best <= thre ? pidx - S.a_p : HAO_AL_NONE; res.ts = S.best <= thre ? tidx - S.a_t : HAO_AL_NONE; }
		else { res.pe = S.a_p; res.te = S.a_t; }
		if (S.dead || S.best > thre) return false;      // (an end found before the sweep was abandoned keeps its coordinates but gets no cigar, as in the reference)
		if (TRACE) {
			int32_t ncg = 0, pdir = -1, pdn = 0;
			const int32_t poff = hao_al_walk_back<WT>(col, stride, thre, S.a_t + 1, thre + S.a_p - S.a_t, S.a_p, S.best, cg, cap, ncg, pdir, pdn);
			if (poff > 0) { if (pdir == 2) pdn += poff; else { if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn); pdir = 2; pdn = poff; } }
			if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn);
			if (!back && (uint32_t)ncg <= cap) for (int32_t q = 0; q < ncg / 2; ++q) { const uint16_t x_ = cg[q]; cg[q] = cg[ncg - 1 - q]; cg[ncg - 1 - q] = x_; }
			res.n_cigar = ncg;
		}
		return true;
	}
	if (S.dead) return false;
	int32_t err = S.err;
	if (MODE == HAO_AL_GLOBAL) {
		int32_t site = tn - 1 - thre; WT VP = S.VP, VN = S.VN;
		for (; site < pn - 1; ++site) { err += (int32_t)(VP & (WT)1); VP >>= 1; err -= (int32_t)(VN & (WT)1); VN >>= 1; }
		if (site == pn - 1 && err <= thre) { ez = err; pe = pn - 1; }
	} else {      // semi-global: the pattern may end anywhere inside the band; the last equally good end wins, except that the end on the main diagonal wins ties
		int32_t site = tn - 1 - S.adiag, i = 0; const int32_t ai = pn - tn + S.adiag; int32_t uge = HAO_AL_NONE;
		for (; site < 0 && i < ai; ++i, ++site) { err += (int32_t)((S.VP >> i) & (WT)1); err -= (int32_t)((S.VN >> i) & (WT)1); }
		if (err <= thre && err <= ez) { ez = err; pe = site; }
		site -= i;
		while (i < ai) {
			err += (int32_t)((S.VP >> i) & (WT)1); err -= (int32_t)((S.VN >> i) & (WT)1); ++i;
			if (err <= thre && err <= ez) { ez = err; pe = site + i; }
			if (i == thre) uge = err;
		}
		if (uge <= thre && uge == ez) pe = site + thre;
	}
	if (ez > thre) return false;
	res.err = ez; res.pe = pe; res.te = tn - 1;
	if (MODE == HAO_AL_ED) return false;
	if (TRACE) {
		const int32_t low = thre << 1, ptrim = MODE == HAO_AL_GLOBAL ? thre : S.adiag;
		int32_t ncg = 0, pdir = -1, pdn = 0;
		const int32_t poff = hao_al_walk_back<WT>(col, stride, thre, tn, (low + 1) - (tn + low - pe - ptrim), pe, ez, cg, cap, ncg, pdir, pdn);
		if (MODE == HAO_AL_SEMI) res.ps = poff;
		else if (poff > 0) { if (pdir == 2) pdn += poff; else { if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn); pdir = 2; pdn = poff; } }
		if (pdn > 0) hao_al_push(cg, cap, ncg, pdir, pdn);
		if ((uint32_t)ncg <= cap) for (int32_t k = 0; k < ncg / 2; ++k) { const uint16_t x_ = cg[k]; cg[k] = cg[ncg - 1 - k]; cg[ncg - 1 - k] = x_; }
		res.n_cigar = ncg;
	}
	return true;
}

// do two tasks align against the same text (same stretch, same strand, same band word)?
HAO_AL_FN bool hao_al_same_text(const hao_ed_task_t &a, const hao_ed_task_t &b)
{ return a.t_rid == b.t_rid && a.t_pos == b.t_pos && a.t_len == b.t_len && a.t_rev == b.t_rev && hao_al_nword(a.thre) == hao_al_nword(b.thre); }
// sort key of a task: band word, text read, text position, text strand (tasks of one text window become neighbours; the text length is compared on the spot)
HAO_AL_FN uint64_t hao_al_sort_key(const hao_ed_task_t &t)
{ return (uint64_t)((hao_al_nword(t.thre) - 1) & 3u) << 62 | (uint64_t)(t.t_rid & 0xfffffffu) << 34 | (uint64_t)(t.t_pos & 0x7ffffffu) << 7 | (uint64_t)(t.t_rev & 1u) << 6; }

#ifndef HAO_ALIGN_HOST_MODEL
// ---------------------------------------------------------------------------------------
// The kernels: one wave per tile of 64 tasks of the sorted order (order[] = task indices; TRACE launches run over the selected tasks only and keep their
// columns in path[], slot = position in the launch, `stride` slots per row).
// ---------------------------------------------------------------------------------------
__global__ void hao_al_key_kernel(const hao_ed_task_t *task, uint64_t n, uint64_t *key, uint32_t *idx)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { key[i] = hao_al_sort_key(task[i]); idx[i] = (uint32_t)i; }
}
__global__ void hao_al_iota_kernel(uint32_t *idx, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) idx[i] = (uint32_t)i; }
struct hao_al_flagged { const uint8_t *f; __host__ __device__ bool operator()(const uint32_t &i) const { return f[i] != 0; } };

template<typename WT, int MODE, bool TRACE>
__global__ __launch_bounds__(256) void hao_al_kernel(hao_ed_reads R, const hao_ed_task_t *task, const uint32_t *order, uint64_t n_order, uint64_t *path, uint64_t stride,
		hao_ed_result_t *out_ed, hao_trace_result_t *out_tr, uint8_t *want_trace, uint16_t *cig, uint32_t cap)
{
	__shared__ uint8_t s_text[4][HAO_AL_CH];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint8_t *codes = s_text[wv];
	const uint64_t slot = ((uint64_t)blockIdx.x * 4 + wv) * 64 + lane;
	const bool have = slot < n_order;
	const uint32_t ti = have ? order[slot] : 0;
	hao_ed_task_t T; if (have) T = task[ti]; else { T.p_rid = T.p_pos = T.p_len = T.p_rev = T.t_rid = T.t_pos = T.t_len = T.t_rev = T.thre = T.abs_diag = 0; }
	const bool mine = have && hao_al_mine<WT>(T.thre);
	// the tile's text windows: a lane starts a segment when its text differs from its left neighbour's (neighbour fields through DPP moves)
	hao_ed_task_t L = T;
	L.t_rid = hao_wave_shr1(T.t_rid, 0xffffffffu); L.t_pos = hao_wave_shr1(T.t_pos, 0u); L.t_len = hao_wave_shr1(T.t_len, 0u); L.t_rev = hao_wave_shr1(T.t_rev, 0u); L.thre = hao_wave_shr1(T.thre, 0u);
	const bool head = mine && (lane == 0 || !hao_al_same_text(T, L) || !hao_al_mine<WT>(L.thre));
	const unsigned long long heads = __ballot(head), live = __ballot(mine);
	hao_al_state<WT> S; S.alive = 0; S.dead = 0;
	if (mine) hao_al_init<WT, MODE>(S, R, T);
	uint64_t *col = path + slot;      // (TRACE only)
	// All segments of the tile sweep TOGETHER: the wave's HAO_AL_CH staged bytes are cut into one strip of `chs` columns per segment (one segment: all of
	// them; 64 different texts: 16 columns each), every lane reads the column's character from its own segment's strip, and the wave refills the strips every chs
	// columns - 16 base decodes per lane and refill whatever the number of segments, i.e. at worst (no two tasks share a text) what decoding one's own text costs.
	// (Round 3 swept the segments one after the other with the lanes of the others idle: 6 % of the lanes busy on task lists whose pairs rarely share a text.)
	const int nseg = __popcll(heads), my_seg = __popcll(heads & ((2ULL << lane) - 1)) - 1;
	const int32_t chs = nseg > 0 ? (int32_t)(HAO_AL_CH / nseg) & ~3 : HAO_AL_CH;
	int32_t tn_mx = (mine && S.alive && !S.dead) ? S.tn : 0;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const int32_t o_ = __shfl_xor(tn_mx, d); tn_mx = o_ > tn_mx ? o_ : tn_mx; }
	const uint8_t *strip = codes + (my_seg > 0 ? my_seg : 0) * chs;
	for (int32_t k0 = 0; k0 < tn_mx; k0 += chs) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (the previous range has been read)
		int sg = 0;
		for (unsigned long long hm = heads; hm; hm &= hm - 1, ++sg) {      // the segments' texts (fields of a segment's first lane: wave-uniform)
			const int h0 = __ffsll((long long)hm) - 1;
			const uint32_t t_rid = (uint32_t)__builtin_amdgcn_readlane((int)T.t_rid, h0), t_pos = (uint32_t)__builtin_amdgcn_readlane((int)T.t_pos, h0),
						   t_len = (uint32_t)__builtin_amdgcn_readlane((int)T.t_len, h0), t_rev = (uint32_t)__builtin_amdgcn_readlane((int)T.t_rev, h0);
			if ((int64_t)t_len <= (int64_t)k0) continue;
			const hao_al_walk tw = hao_al_walk_of(R, t_rid, t_pos, t_len, t_rev, MODE == HAO_AL_EXT_BWD);
			const int32_t n = (int32_t)((int64_t)t_len - k0 < chs ? (int64_t)t_len - k0 : chs);
			hao_al_stage_bases(tw, k0, n, codes + sg * chs, lane, 64);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		if (R.nsite_off) {      // N sites (after the bases are in place)
			sg = 0;
			for (unsigned long long hm = heads; hm; hm &= hm - 1, ++sg) {
				const int h0 = __ffsll((long long)hm) - 1;
				const uint32_t t_rid = (uint32_t)__builtin_amdgcn_readlane((int)T.t_rid, h0), t_pos = (uint32_t)__builtin_amdgcn_readlane((int)T.t_pos, h0),
							   t_len = (uint32_t)__builtin_amdgcn_readlane((int)T.t_len, h0), t_rev = (uint32_t)__builtin_amdgcn_readlane((int)T.t_rev, h0);
				if ((int64_t)t_len <= (int64_t)k0) continue;
				const hao_al_walk tw = hao_al_walk_of(R, t_rid, t_pos, t_len, t_rev, MODE == HAO_AL_EXT_BWD);
				const int32_t n = (int32_t)((int64_t)t_len - k0 < chs ? (int64_t)t_len - k0 : chs);
				hao_al_stage_nsites(tw, k0, n, codes + sg * chs, lane, 64);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		if (mine && S.alive && !S.dead)
			for (int32_t i = k0; i < k0 + chs && i < S.tn && !S.dead; ++i) hao_al_column<WT, MODE, TRACE>(S, strip[i - k0], i, col, stride);
	}
	(void)live;
	if (mine) {
		hao_trace_result_t res;
		const bool tr = hao_al_finish<WT, MODE, TRACE>(S, T, res, col, stride, TRACE ? cig + (uint64_t)ti * cap : nullptr, cap);
		if (MODE == HAO_AL_ED) { hao_ed_result_t r2; r2.err = res.err; r2.pe = res.pe; out_ed[ti] = r2; }
		else { out_tr[ti] = res; if (!TRACE) want_trace[ti] = tr ? 1 : 0; }
	}
}
#endif
