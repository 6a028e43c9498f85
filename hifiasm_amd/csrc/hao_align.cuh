// Windowed bit-vector edit distance: ed_band_cal_semi_64_w_absent_diag (Levenshtein_distance.h:3727-3776), the workhorse of the
// window alignment that follows chaining (Correct.cpp:3897,4092,4156: one call per 775-base query window and candidate).  Myers' bit-parallel
// recurrence (ed_core_64, :3116-3125) over a band of 2 thre + 1 diagonals held in ONE 64-bit word: semi-global (the text window must be consumed,
// the pattern = padded target region may start / end anywhere inside the band), `abs_diag` leading diagonals missing when the pattern was clipped
// at the start of its read.
// The recurrence is sequential in the text position but every (window, candidate) pair is independent and a read has ~20 windows x ~60 candidates:
// one LANE per pair, both strings taken straight from the packed 2-bit reads resident in HBM (the pattern possibly on the reverse strand).
// Returns err (INT32_MAX = no alignment within thre, like clear_align) and pe (end on the pattern, -1 = none); ps / ts / te are constants of the
// call (-1, 0, tn - 1).  Characters: 0..3, 4 = N (never matches: Peq[4] = 0).
#pragma once
#include "hao_common.cuh"

typedef unsigned __int128 hao_u128;
// The band (2 thre + 1 diagonals) lives in one 64-bit word, or in two (hao_u128) for thre = 32 .. 63: the reference generates its 128-bit functions from the
// same text (HA_ED_INIT(128), Levenshtein_distance.h:1287-2129) and picks by band width (cal_exz_global, Correct.cpp:15482-15494); every kernel here is
// instantiated for both word types and a launch of one type skips the tasks of the other.
template<typename WT> __device__ __forceinline__ bool hao_ed_mine(uint32_t thre) { return (2 * thre + 1 <= 64) == (sizeof(WT) == 8); }
// w_<sf>_set_bit_lsub (:1029-1033): the low l bits set.  The 128-bit macro shifts a 64-bit word by 64 when l == 64: undefined in C, 0 on x86-64 (shift count
// modulo 64) - the reference as built starts abs_diag = 64 from VN = 0, and so does this.
template<typename WT> __device__ __forceinline__ WT hao_ed_lsub(int32_t l) { return (sizeof(WT) == 16 && l == 64) ? (WT)0 : (WT)((((WT)1) << l) - 1); }

struct hao_ed_reads { const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len; const uint64_t *nsite_off; const uint32_t *nsite; };

// base `pos` of read `rid` on strand `rev` as a code 0..3, 4 = N
__device__ __forceinline__ uint32_t hao_ed_base(const hao_ed_reads &R, uint64_t rid, const uint8_t *rd, uint32_t L, int64_t pos, int rev)
{
	const int64_t f = rev ? (int64_t)L - 1 - pos : pos;
	uint32_t b = hao_base_at(rd, (uint32_t)f);
	if (R.nsite_off) { for (uint64_t k = R.nsite_off[rid]; k < R.nsite_off[rid + 1]; ++k) { const uint32_t p = R.nsite[k]; if ((int64_t)p == f) return 4; if ((int64_t)p > f) break; } }
	return rev ? 3 - b : b;
}

template<typename WT>
__global__ __launch_bounds__(256) void hao_window_ed_kernel(hao_ed_reads R, const hao_ed_task_t *task, uint64_t n_task, hao_ed_result_t *out)
{
	const uint64_t i_ = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i_ >= n_task) return;
	const hao_ed_task_t T = task[i_];
	if (!hao_ed_mine<WT>(T.thre)) return;
	const uint8_t *pd = R.packed + R.pk_off[T.p_rid], *td = R.packed + R.pk_off[T.t_rid]; const uint32_t pL = R.len[T.p_rid], tL = R.len[T.t_rid];
	const int32_t pn = (int32_t)T.p_len, tn = (int32_t)T.t_len, thre = (int32_t)T.thre, abs_diag = (int32_t)T.abs_diag;
	hao_ed_result_t res; res.err = INT32_MAX; res.pe = -1;
	auto P = [&](int32_t k) { return hao_ed_base(R, T.p_rid, pd, pL, (int64_t)T.p_pos + k, T.p_rev); };
	auto Tx = [&](int32_t k) { return hao_ed_base(R, T.t_rid, td, tL, (int64_t)T.t_pos + k, T.t_rev); };
	const int32_t last_high = thre << 1, tn0 = tn - 1, cut = thre + last_high;
	int32_t err = abs_diag;
	if (pn > tn + cut || tn > pn + cut || tn <= 0) { out[i_] = res; return; }
	WT Peq[5] = {0, 0, 0, 0, 0}, VP = 0, VN, X, D0, HN, HP, mm;
	int32_t bd = ((thre << 1) + 1) - abs_diag; if (bd > pn) bd = pn;
	int32_t i, i_bd = abs_diag;
	for (i = 0, mm = (WT)1 << i_bd; i < bd; ++i) { Peq[P(i)] |= mm; mm <<= 1; }
	i_bd = (thre << 1) - abs_diag; VN = hao_ed_lsub<WT>(abs_diag);
	Peq[4] = 0; mm = (WT)1 << (thre << 1);
#define HAO_ED_CORE(z) { X = Peq[(z)] | VN; D0 = ((VP + (X & VP)) ^ VP) | X; HN = VP & D0; HP = VN | ~(VP | D0); X = D0 >> 1; VN = X & HP; VP = HN | ~(X | HP); }
	i = 0;
	bool dead = false;
	while (i < tn0) {
		HAO_ED_CORE(Tx(i));
		if (!(D0 & (WT)1)) { ++err; if (err > cut) { dead = true; break; } }
		Peq[0] >>= 1; Peq[1] >>= 1; Peq[2] >>= 1; Peq[3] >>= 1;
		++i; ++i_bd;
		uint32_t cc = 4;
		if (i_bd < pn) cc = P(i_bd);
		if (cc < 4) Peq[cc] |= mm;
	}
	if (!dead) {
		HAO_ED_CORE(Tx(i));
		if (!(D0 & (WT)1)) { ++err; if (err > cut) dead = true; }
	}
#undef HAO_ED_CORE
	if (!dead) {
		int32_t site = tn - 1 - abs_diag;
		const int32_t ai = pn - tn + abs_diag; int32_t uge = INT32_MAX;
		for (i = 0; site < 0 && i < ai; ++i, ++site) { err += (int32_t)((VP >> i) & (WT)1); err -= (int32_t)((VN >> i) & (WT)1); }
		if (err <= thre && err <= res.err) { res.err = err; res.pe = site; }
		site -= i;
		while (i < ai) {
			err += (int32_t)((VP >> i) & (WT)1); err -= (int32_t)((VN >> i) & (WT)1); ++i;
			if (err <= thre && err <= res.err) { res.err = err; res.pe = site + i; }
			if (i == thre) uge = err;
		}
		if (uge <= thre && uge == res.err) res.pe = site + thre;
	}
	out[i_] = res;
}

// ---------------------------------------------------------------------------------------
// f3, second variant: global alignment inside the band with traceback - ed_band_cal_global_64_w_trace (Levenshtein_distance.h:3370-3442) on a cleared
// bit_extz_t followed by gen_trace(ez, thre, 1) (:903-985).  Pattern and text are consumed entirely (|pn - tn| <= thre).  The forward sweep keeps the
// five words of every text column (D0, VP, VN, HP, HN: 40 bytes per base and pair) in a scratch array laid out [column][word][pair], so that the lanes of
// a wave - one pair each - write and read consecutive addresses; the traceback walks the columns backwards (indels preferred, :924-936) and emits
// push_trace's entries (op << 14 | len; 0 match, 1 mismatch, 2 more pattern, 3 more text), reversed at the end like the reference does.
// `path` holds `stride` pairs per row; pair i of the launch uses row slot i.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void hao_tr_push(uint16_t *cg, uint32_t cap, int32_t &n, int32_t op, int32_t len)
{	// push_trace (:522-531); entries past the capacity are counted only
	while (len >= 0x3fff) { if ((uint32_t)n < cap) cg[n] = (uint16_t)((op << 14) + 0x3fff); ++n; len -= 0x3fff; }
	if (len) { if ((uint32_t)n < cap) cg[n] = (uint16_t)((op << 14) + len); ++n; }
}

template<int MODE, typename WT>      // 0: ed_band_cal_global_64_w_trace (:3370), 3: ed_band_cal_semi_64_w_absent_diag_trace (:3778) - the numbering of Correct.cpp:14536-14545
__device__ __forceinline__ void hao_window_trace_body(hao_ed_reads R, const hao_ed_task_t *task, uint64_t n_task, uint64_t *path, uint64_t stride,
		hao_trace_result_t *out, uint16_t *cig, uint32_t cap)
{
	const uint64_t i_ = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i_ >= n_task) return;
	const hao_ed_task_t T = task[i_];
	if (!hao_ed_mine<WT>(T.thre)) return;
	const uint8_t *pd = R.packed + R.pk_off[T.p_rid], *td = R.packed + R.pk_off[T.t_rid]; const uint32_t pL = R.len[T.p_rid], tL = R.len[T.t_rid];
	const int32_t pn = (int32_t)T.p_len, tn = (int32_t)T.t_len, thre = (int32_t)T.thre, abs_diag = MODE == 3 ? (int32_t)T.abs_diag : 0;
	hao_trace_result_t res; res.err = INT32_MAX; res.ps = MODE == 3 ? -1 : 0; res.pe = -1; res.ts = 0; res.te = MODE == 3 ? tn - 1 : -1; res.n_cigar = 0;
	auto P = [&](int32_t k) { return hao_ed_base(R, T.p_rid, pd, pL, (int64_t)T.p_pos + k, T.p_rev); };
	auto Tx = [&](int32_t k) { return hao_ed_base(R, T.t_rid, td, tL, (int64_t)T.t_pos + k, T.t_rev); };
	const int32_t tn0 = tn - 1, cut = thre + (thre << 1);
	if (pn <= 0 || tn <= 0 || (MODE == 0 ? (pn > tn + thre || tn > pn + thre) : (pn > tn + cut || tn > pn + cut))) { out[i_] = res; return; }
	constexpr int NW = sizeof(WT) / 8;
	uint64_t *col = path + i_;      // column i, value k (D0, VP, VN, HP, HN), 64-bit half h: col[((5 i + k) * NW + h) * stride]
	WT Peq[5] = {0, 0, 0, 0, 0}, VP, VN, X, D0, HN, HP, mm;
	int32_t i, i_bd, err, bd;
	if (MODE == 0) {
		bd = thre + 1; if (bd > pn) bd = pn;
		for (i = 0, mm = (WT)1 << thre; i < bd; ++i) { Peq[P(i)] |= mm; mm <<= 1; }
		i_bd = thre; err = thre;
		VN = ((WT)1 << thre) - 1; VP = (((WT)1 << ((thre << 1) + 1)) - 1) ^ VN;
	} else {
		bd = ((thre << 1) + 1) - abs_diag; if (bd > pn) bd = pn;
		for (i = 0, mm = (WT)1 << abs_diag; i < bd; ++i) { Peq[P(i)] |= mm; mm <<= 1; }
		i_bd = (thre << 1) - abs_diag; err = abs_diag;
		VP = 0; VN = hao_ed_lsub<WT>(abs_diag);
	}
	Peq[4] = 0; mm = (WT)1 << (thre << 1);
#define HAO_ED_CORE(z) { X = Peq[(z)] | VN; D0 = ((VP + (X & VP)) ^ VP) | X; HN = VP & D0; HP = VN | ~(VP | D0); X = D0 >> 1; VN = X & HP; VP = HN | ~(X | HP); }
#define HAO_ED_PUT(k_, v_) { w_[(k_) * NW * stride] = (uint64_t)(v_); if (NW == 2) w_[((k_) * NW + 1) * stride] = (uint64_t)((hao_u128)(v_) >> 64); }
#define HAO_ED_KEEP() { uint64_t *w_ = col + 5 * NW * (uint64_t)i * stride; HAO_ED_PUT(0, D0) HAO_ED_PUT(1, VP) HAO_ED_PUT(2, VN) HAO_ED_PUT(3, HP) HAO_ED_PUT(4, HN) }
	bool dead = false;
	for (i = 0; i < tn0; ) {
		HAO_ED_CORE(Tx(i));
		if (!(D0 & (WT)1)) { ++err; if (err > cut) { dead = true; break; } }
		Peq[0] >>= 1; Peq[1] >>= 1; Peq[2] >>= 1; Peq[3] >>= 1;
		HAO_ED_KEEP();
		++i; ++i_bd;
		uint32_t cc = 4;
		if (i_bd < pn) cc = P(i_bd);
		if (cc < 4) Peq[cc] |= mm;
	}
	if (!dead) {
		HAO_ED_CORE(Tx(i));
		if (!(D0 & (WT)1)) { ++err; if (err > cut) dead = true; }
	}
	if (dead) { out[i_] = res; return; }
	HAO_ED_KEEP();
#undef HAO_ED_CORE
#undef HAO_ED_KEEP
#undef HAO_ED_PUT
	int32_t ez_err = INT32_MAX, pe = -1;
	if (MODE == 0) {
		int32_t site = tn - 1 - thre;
		for (; site < pn - 1; ++site) { err += (int32_t)(VP & (WT)1); VP >>= 1; err -= (int32_t)(VN & (WT)1); VN >>= 1; }
		if (site == pn - 1 && err <= thre) { ez_err = err; pe = pn - 1; }
	} else {
		int32_t site = tn - 1 - abs_diag; const int32_t ai = pn - tn + abs_diag; int32_t uge = INT32_MAX;
		for (i = 0; site < 0 && i < ai; ++i, ++site) { err += (int32_t)((VP >> i) & (WT)1); err -= (int32_t)((VN >> i) & (WT)1); }
		if (err <= thre && err <= ez_err) { ez_err = err; pe = site; }
		site -= i;
		while (i < ai) {
			err += (int32_t)((VP >> i) & (WT)1); err -= (int32_t)((VN >> i) & (WT)1); ++i;
			if (err <= thre && err <= ez_err) { ez_err = err; pe = site + i; }
			if (i == thre) uge = err;
		}
		if (uge <= thre && uge == ez_err) pe = site + thre;
	}
	if (ez_err > thre) { out[i_] = res; return; }
	res.err = ez_err; res.pe = pe; res.te = tn - 1;
	// gen_trace(ez, ptrim, reverse = 1) with ts = 0, te = tn - 1; ptrim = thre (global: ps = 0 is known) / abs_diag (semi: ps comes out of the walk)
	uint16_t *cg = cig + i_ * cap; int32_t ncg = 0;
	const int32_t low = thre << 1, ptrim = MODE == 0 ? thre : abs_diag; int32_t sft = (low + 1) - (tn + low - pe - ptrim), poff = pe, cur = ez_err, d = 0, pdir = -1, pdn = 0;
	i = tn;
	while (i > 0 && cur > 0) {
		const uint64_t *w_ = col + 5 * NW * (uint64_t)(i - 1) * stride;
		auto bit = [&](int k_, int32_t b_) -> int32_t { return (int32_t)((w_[((uint64_t)k_ * NW + (b_ >> 6)) * stride] >> (b_ & 63)) & 1ULL); };      // bit b_ of value k_
		const int32_t D = cur - (bit(0, sft) ^ 1); int32_t mn = D; d = 0;
		if (sft != low) { const int32_t H = cur + bit(4, sft) - bit(3, sft); if (H + 1 == cur && H <= mn) { mn = H; d = 3; } }
		if (sft != 0) { const int32_t V = cur + bit(2, sft - 1) - bit(1, sft - 1); if (V + 1 == cur && V <= mn) { mn = V; d = 2; } }
		if (d == 0) { if (D != cur) d = 1; --i; --poff; }
		else if (d == 2) { --sft; --poff; }
		else { --i; ++sft; }
		if (d == pdir) ++pdn; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = 1; }
		cur = mn;
	}
	if (i > 0) { d = 0; poff -= i; if (d == pdir) pdn += i; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = i; } }
	++poff;
	if (MODE == 3) res.ps = poff;
	else if (poff > 0) { d = 2; if (d == pdir) pdn += poff; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = poff; } }
	if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn);
	if ((uint32_t)ncg <= cap) for (int32_t k = 0; k < ncg / 2; ++k) { const uint16_t x_ = cg[k]; cg[k] = cg[ncg - 1 - k]; cg[ncg - 1 - k] = x_; }
	res.n_cigar = ncg;
	out[i_] = res;
}

// Modes 1 / 2: ed_band_cal_extension_64_0_w_trace (Levenshtein_distance.h:3512-3618) / ed_band_cal_extension_64_1_w_trace (:3620-3735) on a cleared
// bit_extz_t.  Forward extension: both strings start together and the alignment ends wherever the pattern or the text runs out (the longer string is first
// cut to the other's length + thre); the best end is tracked along the pattern's last row while the text is swept (tmp_e), then along the last column.
// Backward extension is the same sweep over both strings read from their ends: ps / ts move instead of pe / te and the cigar is not reversed.  A sweep
// abandoned because the running error passed 3 thre returns before gen_trace: an end found earlier keeps its err / coordinates but gets no cigar.
template<bool BACK, typename WT>
__device__ __forceinline__ void hao_window_ext_trace_body(hao_ed_reads R, const hao_ed_task_t *task, uint64_t n_task, uint64_t *path, uint64_t stride,
		hao_trace_result_t *out, uint16_t *cig, uint32_t cap)
{
	const uint64_t i_ = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i_ >= n_task) return;
	const hao_ed_task_t T = task[i_];
	if (!hao_ed_mine<WT>(T.thre)) return;
	const uint8_t *pd = R.packed + R.pk_off[T.p_rid], *td = R.packed + R.pk_off[T.t_rid]; const uint32_t pL = R.len[T.p_rid], tL = R.len[T.t_rid];
	const int32_t pn0 = (int32_t)T.p_len, tn0_ = (int32_t)T.t_len, thre = (int32_t)T.thre, pidx = pn0 - 1, tidx = tn0_ - 1;
	int32_t pn = pn0, tn = tn0_, ez_err = INT32_MAX, a_p = -1, a_t = -1, ncg = 0;      // a_p / a_t: the moving end (pe / te forward; pidx - ps / tidx - ts backward)
	auto P = [&](int32_t k) { return hao_ed_base(R, T.p_rid, pd, pL, (int64_t)T.p_pos + (BACK ? pidx - k : k), T.p_rev); };
	auto Tx = [&](int32_t k) { return hao_ed_base(R, T.t_rid, td, tL, (int64_t)T.t_pos + (BACK ? tidx - k : k), T.t_rev); };
	auto put = [&]() {
		hao_trace_result_t res; res.err = ez_err; res.n_cigar = ncg;
		if (BACK) { res.ps = ez_err <= thre ? pidx - a_p : INT32_MAX; res.pe = pidx; res.ts = ez_err <= thre ? tidx - a_t : INT32_MAX; res.te = tidx; }
		else { res.ps = 0; res.pe = a_p; res.ts = 0; res.te = a_t; }
		out[i_] = res;
	};
	if (pn0 <= 0 || tn0_ <= 0) { put(); return; }
	if (pn > tn + thre) pn = tn + thre; else if (tn > pn + thre) tn = pn + thre;
	constexpr int NW = sizeof(WT) / 8;
	uint64_t *col = path + i_;
	const int32_t cut = thre + (thre << 1), pe_l = pn - 1;
	WT Peq[5] = {0, 0, 0, 0, 0}, VP, VN, X, D0, HN, HP, mm;
	int32_t i, i_bd = thre, err = thre, tmp_e = INT32_MAX, k, poff, bd = thre + 1; if (bd > pn) bd = pn;
	for (i = 0, mm = (WT)1 << thre; i < bd; ++i) { Peq[P(i)] |= mm; mm <<= 1; }
	Peq[4] = 0;
	VN = ((WT)1 << thre) - 1; VP = (((WT)1 << ((thre << 1) + 1)) - 1) ^ VN;
	mm = (WT)1 << (thre << 1);
#define HAO_ED_CORE(z) { X = Peq[(z)] | VN; D0 = ((VP + (X & VP)) ^ VP) | X; HN = VP & D0; HP = VN | ~(VP | D0); X = D0 >> 1; VN = X & HP; VP = HN | ~(X | HP); }
#define HAO_ED_PUT(k_, v_) { w_[(k_) * NW * stride] = (uint64_t)(v_); if (NW == 2) w_[((k_) * NW + 1) * stride] = (uint64_t)((hao_u128)(v_) >> 64); }
#define HAO_ED_KEEP() { uint64_t *w_ = col + 5 * NW * (uint64_t)i * stride; HAO_ED_PUT(0, D0) HAO_ED_PUT(1, VP) HAO_ED_PUT(2, VN) HAO_ED_PUT(3, HP) HAO_ED_PUT(4, HN) }
	for (i = 0; i < tn - 1; ) {
		HAO_ED_CORE(Tx(i));
		if (!(D0 & (WT)1)) { ++err; if (err > cut) { put(); return; } }
		poff = i - thre; k = i + thre - pe_l;
		if (k >= 0) {
			if (tmp_e == INT32_MAX) { tmp_e = err; for (k = 0; poff < pe_l; ++poff, ++k) { tmp_e += (int32_t)((VP >> k) & (WT)1); tmp_e -= (int32_t)((VN >> k) & (WT)1); } }
			else { k = (thre << 1) - k; if (k >= 0) { tmp_e += (int32_t)((HP >> k) & (WT)1); tmp_e -= (int32_t)((HN >> k) & (WT)1); } }
			if (tmp_e <= thre && tmp_e < ez_err) { ez_err = tmp_e; a_p = pe_l; a_t = i; }
		}
		Peq[0] >>= 1; Peq[1] >>= 1; Peq[2] >>= 1; Peq[3] >>= 1;
		HAO_ED_KEEP();
		++i; ++i_bd;
		if (i_bd < pn) { const uint32_t cc = P(i_bd); if (cc < 4) Peq[cc] |= mm; }
	}
	HAO_ED_CORE(Tx(i));
	if (!(D0 & (WT)1)) { ++err; if (err > cut) { put(); return; } }
	HAO_ED_KEEP();
#undef HAO_ED_CORE
#undef HAO_ED_KEEP
#undef HAO_ED_PUT
	int32_t site = tn - 1 - thre;
	while (site < pn - 1) {
		err += (int32_t)(VP & (WT)1); VP >>= 1; err -= (int32_t)(VN & (WT)1); VN >>= 1; ++site;
		if (err <= thre && err < ez_err) { ez_err = err; a_p = site; a_t = tn - 1; }
	}
	if (err <= thre && err < ez_err) { ez_err = err; a_p = site; a_t = tn - 1; }
	if (ez_err <= thre) {      // gen_trace(ez, thre, !BACK) on the columns 0 .. a_t; ps (mirrored for the backward sweep) = 0 is known
		uint16_t *cg = cig + i_ * cap;
		const int32_t low = thre << 1; int32_t sft = thre + a_p - a_t, cur = ez_err, d = 0, pdir = -1, pdn = 0;
		poff = a_p; i = a_t + 1;
		while (i > 0 && cur > 0) {
			const uint64_t *w_ = col + 5 * NW * (uint64_t)(i - 1) * stride;
			auto bit = [&](int k_, int32_t b_) -> int32_t { return (int32_t)((w_[((uint64_t)k_ * NW + (b_ >> 6)) * stride] >> (b_ & 63)) & 1ULL); };      // bit b_ of value k_
			const int32_t D = cur - (bit(0, sft) ^ 1); int32_t mn = D; d = 0;
			if (sft != low) { const int32_t H = cur + bit(4, sft) - bit(3, sft); if (H + 1 == cur && H <= mn) { mn = H; d = 3; } }
			if (sft != 0) { const int32_t V = cur + bit(2, sft - 1) - bit(1, sft - 1); if (V + 1 == cur && V <= mn) { mn = V; d = 2; } }
			if (d == 0) { if (D != cur) d = 1; --i; --poff; }
			else if (d == 2) { --sft; --poff; }
			else { --i; ++sft; }
			if (d == pdir) ++pdn; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = 1; }
			cur = mn;
		}
		if (i > 0) { d = 0; poff -= i; if (d == pdir) pdn += i; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = i; } }
		++poff;
		if (poff > 0) { d = 2; if (d == pdir) pdn += poff; else { if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn); pdir = d; pdn = poff; } }
		if (pdn > 0) hao_tr_push(cg, cap, ncg, pdir, pdn);
		if (!BACK && (uint32_t)ncg <= cap) for (int32_t q = 0; q < ncg / 2; ++q) { const uint16_t x_ = cg[q]; cg[q] = cg[ncg - 1 - q]; cg[ncg - 1 - q] = x_; }
	}
	put();
}

// the four alignments as kernels over one word type (the host launches <uint64_t> and, when some band needs two words, <hao_u128>)
#define HAO_TR_KERNEL(NAME, BODY) template<typename WT> __global__ __launch_bounds__(256) void NAME(hao_ed_reads R, const hao_ed_task_t *task, uint64_t n_task, \
		uint64_t *path, uint64_t stride, hao_trace_result_t *out, uint16_t *cig, uint32_t cap) { BODY(R, task, n_task, path, stride, out, cig, cap); }
HAO_TR_KERNEL(hao_tr_global, (hao_window_trace_body<0, WT>))
HAO_TR_KERNEL(hao_tr_semi, (hao_window_trace_body<3, WT>))
HAO_TR_KERNEL(hao_tr_ext_fwd, (hao_window_ext_trace_body<false, WT>))
HAO_TR_KERNEL(hao_tr_ext_bwd, (hao_window_ext_trace_body<true, WT>))
#undef HAO_TR_KERNEL
