// K8/K9: per-target linear chaining and per-read chain selection for gfx950.
//
// K8 = lchain_qdp_mcopy_fast (Hash_Table.cpp:2097-2284) with quick_ck_lchain (:2007-2094),
// comput_sc_ch_ec (:1515-1541), cal_bw (:1475-1488), get_chainLen (:779-809),
// push_ovlp_chain_qgen (:1752-1780), gen_fake_cigar (:88-109).
// K9 = lchain_qgen_mcopy_fast (anchor.cpp:1920-2100): max_n_chain pruning with
// coverage-window rescue, klib introsort tie order (ksort.h:110-160), weak-chain filter.
//
// Scores mix int32 with FP64 penalties; the translation unit is compiled with
// -ffp-contract=off and every double expression keeps the reference's operation order,
// so results are bit-identical to the x86-64 SSE2 build of the reference.
//
// Work decomposition (groups arrive as size-class work lists, hao_query.cuh):
//   chain_tiny_kernel   groups of <= 8 hits, one LANE per group, the whole sequential algorithm
//   chain_group_kernel  one wave per group: quick_ck_lchain as a segmented scan; settles > 99.9 % of the groups of a repeat-free genome
//   chain_dp*_kernel    the groups the quick check leaves: speculative 64-hit tiles + wave-parallel sequential rounds, multi-copy tail
//   chain_assemble_kernel, chain_select_kernel<CAP>, chain_final_kernel: records / chained hits in read order, per-read selection, output
#pragma once
#include "hao_common.cuh"
#include "hao_host.hpp"
#include "hao_query.cuh"

#define HAO_MCOPY_MAX 3

struct hao_chain_rec {          // one kept chain of a group
	uint32_t x_pos_s, x_pos_e, y_pos_s, y_pos_e; int32_t score; uint32_t n_hits, hit_rel, fc_rel, fc_len, strand;
	uint32_t src_rel, in_place;     // in_place: the chain is the contiguous run hits[g_start + src_rel ..) of the sorted seed hits (fast path, nothing was copied)
};

#define HH_ID(h)     ((h).w0 & 0x7fffffffu)
#define HH_STRAND(h) ((h).w0 >> 31)
#define HH_SPAN(h)   ((int32_t)((h).cnt & 0xffu))
#define HH_WGT(h)    ((int32_t)((h).cnt >> 8))

struct hao_cpar { double pen_gap, pen_skip, bw; int64_t max_skip, max_iter, max_dis, xl, yl; };

__device__ __forceinline__ int64_t hao_ext_len(int64_t x_beg, int64_t x_end, int64_t xl, int64_t y_beg, int64_t y_end, int64_t yl)
{	// get_chainLen
	if (x_beg <= y_beg) x_beg = 0; else x_beg -= y_beg;
	int64_t xr = xl - x_end - 1, yr = yl - y_end - 1;
	if (xr <= yr) x_end = xl - 1; else x_end += yr;
	return x_end - x_beg + 1;
}

__device__ __forceinline__ int32_t hao_band(const hao_hit_t &ai, const hao_hit_t &aj, const hao_cpar &P)
{	// cal_bw
	int64_t sf_s = aj.self_offset, sf_e = (int64_t)ai.self_offset + 1, ot_s = aj.offset, ot_e = (int64_t)ai.offset + 1;
	int64_t sf_r = P.xl - sf_e, ot_r = P.yl - ot_e;
	if (sf_s <= ot_s) sf_s = 0; else sf_s -= ot_s;
	if (sf_r <= ot_r) sf_e = P.xl; else sf_e += ot_r;
	return (int32_t)((double)(sf_e - sf_s) * P.bw);
}

__device__ __forceinline__ int32_t hao_pair_score(const hao_hit_t &ai, const hao_hit_t &aj, const hao_cpar &P, int64_t *dd_out)
{	// comput_sc_ch_ec
	int32_t dq = (int32_t)((int64_t)ai.self_offset - (int64_t)aj.self_offset); if (dq <= 0) return INT32_MIN;
	int32_t dr = (int32_t)((int64_t)ai.offset - (int64_t)aj.offset); if (dr <= 0) return INT32_MIN;
	int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (dd > 16 && dd > hao_band(ai, aj, P)) return INT32_MIN;
	int32_t dg = dr < dq ? dr : dq, span = HH_SPAN(ai), sc = span < dg ? span : dg, wgt = HH_WGT(ai);
	if (wgt != 1) sc = sc >= wgt ? (wgt == 2 ? sc >> 1 : sc / wgt) : 1;      // weight 1 (and 2) are the rule: the integer division (~40 instructions) only runs for rarer seeds
	if (dd) {
		double lin = P.pen_gap * (double)dd; const double skip = P.pen_skip * (double)dg;
		// dd < 4: the penalty is min(lin, ap) + skip with ap >= 0, and FP addition is monotonic, so it lies in [0, lin + skip]; when that
		// bound is below 1 the penalty truncates to 0 whatever ap is - the two FP64 divisions are skipped (the common 1-3 base indel)
		if (!(dd < 4 && lin + skip < 1.0)) {
			const double ap = (double)sc * (((double)dd / (double)dg) / P.bw);
			if (dd < 4) lin = lin > ap ? ap : lin; else lin = lin < ap ? ap : lin;
			lin += skip;
			sc -= (int32_t)lin;
		}
	} else if (dg > span) {
		// dd == 0: both penalty terms are exactly +0.0 (pen_gap * 0, sc * ((0 / dg) / bw)), min(0,0) = 0, so only the
		// skip term remains - same IEEE result as the general expression without the two divisions
		sc -= (int32_t)(0.0 + P.pen_skip * (double)dg);
	}
	if (dd_out) *dd_out = dd;
	return sc;
}

__device__ __forceinline__ void hao_region(hao_chain_rec &o, int64_t xl, int64_t yl, int64_t sc, const hao_hit_t &beg, const hao_hit_t &end)
{	// push_ovlp_chain_qgen
	o.strand = HH_STRAND(beg);
	o.x_pos_s = beg.self_offset; o.y_pos_s = beg.offset; o.x_pos_e = end.self_offset; o.y_pos_e = end.offset;
	if (o.x_pos_s <= o.y_pos_s) { o.y_pos_s -= o.x_pos_s; o.x_pos_s = 0; } else { o.x_pos_s -= o.y_pos_s; o.y_pos_s = 0; }
	int64_t xr = xl - o.x_pos_e - 1, yr = yl - o.y_pos_e - 1;
	if (xr <= yr) { o.x_pos_e = (uint32_t)(xl - 1); o.y_pos_e += (uint32_t)xr; } else { o.y_pos_e = (uint32_t)(yl - 1); o.x_pos_e += (uint32_t)yr; }
	o.score = (int32_t)sc;
}

__device__ __forceinline__ uint64_t hao_fc_entry(uint32_t site, int32_t shift)
{ uint32_t lo = shift < 0 ? ((uint32_t)(-shift) << 1 | 1u) : (uint32_t)shift << 1; return (uint64_t)site << 32 | lo; }

template<class HitAt>      // hit(k): hit k of the chain (the callers read the chain where it already sits - LDS or the sorted seed hits -, not the copy they have just stored)
__device__ uint32_t hao_fake_cigar(uint64_t *fc, const hao_chain_rec &o, HitAt hit, int64_t n_hit)
{	// gen_fake_cigar, apend_be = 1
	int64_t pdd = INT32_MAX; uint32_t n = 0;
	fc[n++] = hao_fc_entry(o.x_pos_s, 0);
	for (int64_t k = 0; k < n_hit; ++k) {
		const hao_hit_t h = hit(k);
		int64_t dq = (int64_t)h.self_offset - o.x_pos_s, dr = (int64_t)h.offset - o.y_pos_s, dd = dr - dq;
		if (dd != pdd) { pdd = dd; fc[n++] = hao_fc_entry(h.self_offset, (int32_t)pdd); }
	}
	uint64_t last = fc[n - 1]; int32_t lsh = (int32_t)((uint32_t)last >> 1); if (last & 1) lsh = -lsh;
	if ((int64_t)(int32_t)(last >> 32) != (int64_t)o.x_pos_e) fc[n++] = hao_fc_entry(o.x_pos_e, lsh);
	return n;
}

// broadcast of lane src (wave-uniform src): v_readlane, no LDS crossbar

// gen_fake_cigar (apend_be = 1) by one wave: the k-th hit of the chain is hit(k); entries are a flagged compaction
// (an entry wherever the diagonal changes).  Returns the entry count (uniform).
template<class HitAt>
__device__ __forceinline__ uint32_t hao_fake_cigar_wave(uint64_t *fcs, uint32_t x_pos_s, uint32_t y_pos_s, uint32_t x_pos_e, int64_t cL, HitAt hit)
{
	const int lane = hao_lane();
	uint32_t cnt = 1; int64_t carry_dd = INT32_MAX; uint32_t last_site = 0; int64_t last_dd = 0;
	if (lane == 0) fcs[0] = hao_fc_entry(x_pos_s, 0);
	for (int64_t t0 = 0; t0 < cL; t0 += 64) {
		const int64_t k = t0 + lane; const bool act = k < cL;
		const hao_hit_t h = hit(act ? k : (int64_t)0);
		int64_t dd = ((int64_t)h.offset - y_pos_s) - ((int64_t)h.self_offset - x_pos_s);
		const int64_t pd = (int64_t)((uint64_t)hao_wave_shr1((uint32_t)dd, (uint32_t)carry_dd) | (uint64_t)hao_wave_shr1((uint32_t)((uint64_t)dd >> 32), (uint32_t)((uint64_t)carry_dd >> 32)) << 32);
		const bool flag = act && dd != pd;
		unsigned long long bal = __ballot(flag);
		if (flag) fcs[cnt + __popcll(bal & ((1ULL << lane) - 1))] = hao_fc_entry(h.self_offset, (int32_t)dd);
		if (bal) { int src = 63 - __clzll((long long)bal); last_site = hao_bcast(h.self_offset, src); last_dd = hao_readlane_i64(dd, src); }
		cnt += __popcll(bal); carry_dd = hao_readlane_i64(dd, 63);
	}
	if (last_site != x_pos_e) { if (lane == 0) fcs[cnt] = hao_fc_entry(x_pos_e, (int32_t)last_dd); ++cnt; }
	return cnt;
}

#define HAO_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

template<class I64P>
__device__ void hao_heapsort_i64(I64P a, int64_t n)
{
	auto down = [&](int64_t i, int64_t m) { int64_t v = a[i]; for (;;) { int64_t c = 2 * i + 1; if (c >= m) break; if (c + 1 < m && a[c + 1] > a[c]) ++c; if (a[c] <= v) break; a[i] = a[c]; i = c; } a[i] = v; };
	for (int64_t i = n / 2 - 1; i >= 0; --i) down(i, n);
	for (int64_t m = n - 1; m > 0; --m) { int64_t tmp = a[0]; a[0] = a[m]; a[m] = tmp; down(0, m); }
}

struct hao_chain_args {
	const hao_hit_t *hits; const uint64_t *g_start; const uint32_t *g_read; const uint64_t *g_off; const uint64_t *seg; uint64_t n_groups;
	uint64_t rid_lo; const uint32_t *len;
	hao_chain_par par;
	int32_t *f, *ii, *p; int64_t *t;             // per-hit scratch
	hao_hit_t *ohits; uint64_t *fcs; hao_chain_rec *rec; uint32_t *nch, *nout;
	unsigned long long *stats;                   // [0 .. HAO_NCLS) groups of each size class needing the DP kernel, [HAO_NCLS] their hits
	int32_t *tm;                                 // per-hit mark scratch for oversize groups
	int dbg_seq, dbg_stats;          // dbg_seq: 1 one-lane sequential chaining, 3 one-lane DP tail, 4 no speculative tiles (all give identical results)
	unsigned long long *dbg_qc;   // optional phase timers of chain_group_kernel (HAO_DBG_PRINT=qc)
	uint32_t exc_every;      // (tests) every n-th hit of a group gets the code 0xff: exercises the verbatim list
	const uint16_t *hq; uint8_t *hcode;      // delivery path: query minimizer index of every seed hit (seed kernel) -> wire code byte of every seed hit relative to its
	                                         // predecessor in the sorted order (hao_deliver.cuh): the quick check has both hits in registers anyway
	uint16_t *ohq;                           // delivery path: minimizer index of every hit the DP compacts into ohits (same index as ohits): the DP tails code those chains themselves
};

// Wire code (hao_deliver.cuh) of chain hit h with query minimizer q after hit ph with minimizer pq: minimizers skipped << 4 | diagonal shift + 8; what that byte
// cannot say travels verbatim: esc = 0xff for a hit the packer finds in hits[] at its position, 0xfd for one it finds in ohits[] / ohq[] (HAO_CODE_EXC_OHITS)
#define HAO_CODE_EXC_OHITS 0xfd
__device__ __forceinline__ uint8_t hao_wire_code(uint32_t q, uint32_t pq, const hao_hit_t &h, const hao_hit_t &ph, uint32_t exc_every, uint32_t idx, uint8_t esc)
{
	const int32_t dq = (int32_t)(q - pq), sh = (int32_t)((h.offset - ph.offset) - (h.self_offset - ph.self_offset));
	return (dq < 1 || dq > 15 || sh < -8 || sh > 7 || q == 65535u || (exc_every && idx % exc_every == exc_every - 1)) ? esc : (uint8_t)((dq - 1) << 4 | (sh + 8));
}
// the codes of one chain the DP compacted into ohits[o, o + n) of group gs (hit j of the chain = seed hit a[src(j)]), by ONE lane
template<class HitP, class SrcAt>
__device__ __forceinline__ void hao_code_chain_lane(const hao_chain_args &A, const uint64_t gs, HitP a, int64_t o, int64_t n, SrcAt src)
{
	const uint16_t *hq = A.hq + gs; uint8_t *hc = A.hcode + gs; uint16_t *oq = A.ohq + gs;
	uint32_t pq = 0; hao_hit_t ph; ph.w0 = ph.offset = ph.self_offset = ph.cnt = 0;
	for (int64_t j = 0; j < n; ++j) {
		const int64_t x = src(j); const uint32_t q = hq[x]; const hao_hit_t h = a[x];
		oq[o + j] = (uint16_t)q;
		hc[o + j] = j ? hao_wire_code(q, pq, h, ph, A.exc_every, (uint32_t)j, (uint8_t)HAO_CODE_EXC_OHITS) : (uint8_t)0x08;
		pq = q; ph = h;
	}
}

// the same by a wave: hit j against hit j - 1 through two reads of L2-resident data (a call would make the DP kernels save ~100 registers around it: inlined)
template<class SrcAt>
__device__ __forceinline__ void hao_code_chain_wave(const hao_chain_args &A, const uint64_t gs, const hao_hit_t *a, int64_t o, int64_t n, SrcAt src)
{
	for (int64_t j = hao_lane(); j < n; j += 64) {
		const int64_t x = src(j); const uint32_t q = A.hq[gs + x];
		A.ohq[gs + o + j] = (uint16_t)q;
		uint8_t code = 0x08;
		if (j) { const int64_t xp = src(j - 1); code = hao_wire_code(q, A.hq[gs + xp], a[x], a[xp], A.exc_every, (uint32_t)j, (uint8_t)HAO_CODE_EXC_OHITS); }
		A.hcode[gs + o + j] = code;
	}
}

// Sequential tail shared by both paths (ONE lane): backtrack the best chain, multi-copy chains
// (Hash_Table.cpp:2178-2270), regions, chained hits, fake cigars.  f/p may live in LDS or global memory.
template<class HitP, class I32P, class I64P>      // HitP: the group's hits, a[i] (a pointer into the sorted seed hits, or the copy chain_tiny_kernel keeps in LDS)
__device__ void hao_chain_tail(const hao_chain_args &A, const uint64_t g, HitP a, const int64_t a_n, const hao_cpar &P,
		I32P f, I32P p, I64P t, I32P ii, int64_t msc, int64_t msc_i, int64_t plus)
{
	const uint64_t gs = A.g_start[g];
	hao_hit_t *des = A.ohits + gs; uint64_t *fcs = A.fcs + gs + 6 * g; hao_chain_rec *rec = A.rec + g * HAO_MCOPY_MAX;
	int64_t cL = 0, i;
	for (i = msc_i; i >= 0; i = p[i]) { ii[i] = 1; t[cL++] = i; }
	// ---- multi-copy chains ----
	if (A.par.mcopy_num > 1 && cL >= A.par.mcopy_khit_cut) {
		int64_t ch_n, min_sc;
		msc -= plus; min_sc = (int64_t)((double)msc * A.par.mcopy_rate); ii[msc_i] = 0;
		for (i = ch_n = 0; i < a_n; ++i) {
			f[i] -= (int32_t)plus; if (i >= ch_n) t[i] = 0;
			if (!ii[i] && f[i] >= min_sc) { t[ch_n] = (int64_t)((uint64_t)f[i] << 32); t[ch_n] += i << 1; ++ch_n; }
		}
		if (ch_n > 1) {
			int64_t n_v = 0, n_v0, n_u = 0, k, sc, j, ni; uint32_t fcn = 0;
			hao_heapsort_i64(t, ch_n);
			uint32_t c_nv0[HAO_MCOPY_MAX], c_ni[HAO_MCOPY_MAX];
			for (k = ch_n - 1; k >= 0 && n_u < A.par.mcopy_num; --k) {
				n_v0 = n_v;
				for (i = (int64_t)((uint32_t)t[k] >> 1); i >= 0 && (t[i] & 1) == 0; ) { ii[n_v++] = (int32_t)i; t[i] |= 1; i = p[i]; }
				if (n_v0 == n_v) continue;
				sc = i < 0 ? (t[k] >> 32) : ((t[k] >> 32) - f[i]);
				if (sc >= min_sc) {
					if (!n_u || n_v - n_v0 > 1) {
						hao_region(rec[n_u], P.xl, P.yl, sc + plus, a[ii[n_v - 1]], a[ii[n_v0]]);
						c_nv0[n_u] = (uint32_t)n_v0; c_ni[n_u] = (uint32_t)(n_v - n_v0); ++n_u;
					} else n_v = n_v0;
				} else n_v = n_v0;
			}
			// chain member lists live in ii[]; stage the hits through the tail of t[] is not possible (t holds marks), so
			// write each chain to des[] from a private walk: des and a may alias only if des == a (they do not: separate buffers)
			for (k = 0, i = 0; k < n_u; ++k) {
				n_v0 = c_nv0[k]; ni = c_ni[k];
				rec[k].hit_rel = (uint32_t)i; rec[k].n_hits = (uint32_t)ni; rec[k].src_rel = (uint32_t)i; rec[k].in_place = 0;
				if (A.hcode) hao_code_chain_lane(A, gs, a, i, ni, [&](int64_t q_) { return (int64_t)ii[n_v0 + (ni - q_ - 1)]; });
				for (j = 0; j < ni; ++j, ++i) des[i] = a[ii[n_v0 + (ni - j - 1)]];
				rec[k].fc_rel = fcn; rec[k].fc_len = hao_fake_cigar(fcs + fcn, rec[k], [&](int64_t q_) { return a[ii[n_v0 + (ni - q_ - 1)]]; }, ni); fcn += rec[k].fc_len;
			}
			A.nch[g] = (uint32_t)n_u; A.nout[g] = (uint32_t)i;
			return;
		} else {
			msc += plus; i = msc_i; cL = 0;
			while (i >= 0) { t[cL++] = i; i = p[i]; }
		}
	}
	hao_region(rec[0], P.xl, P.yl, msc, a[t[cL - 1]], a[t[0]]);
	if (A.hcode) hao_code_chain_lane(A, gs, a, 0, cL, [&](int64_t q_) { return (int64_t)t[cL - q_ - 1]; });
	for (i = 0; i < cL; ++i) des[i] = a[t[cL - i - 1]];
	rec[0].hit_rel = 0; rec[0].n_hits = (uint32_t)cL; rec[0].src_rel = 0; rec[0].in_place = 0; rec[0].fc_rel = 0; rec[0].fc_len = hao_fake_cigar(fcs, rec[0], [&](int64_t q_) { return a[t[cL - q_ - 1]]; }, cL);
	A.nch[g] = 1; A.nout[g] = (uint32_t)cL;
}


// Generic path, executed by ONE lane: the full sequential algorithm (quick check, DP, multi-copy).
// core: group g = hits a[0, a_n) of query xid against target yid; f/ii/p/t = per-hit scratch of the caller
template<class HitP, class I32P, class I64P>
__device__ void hao_chain_generic_core(const hao_chain_args &A, const uint64_t g, const uint64_t gs, HitP a, const int64_t a_n, const uint32_t xid, const uint32_t yid,
		const uint32_t xl, const uint32_t yl, I32P f, I32P ii, I32P p, I64P t)
{
	A.nch[g] = 0; A.nout[g] = 0;
	if (yid == xid || a_n <= 0) return;                      // hits to the query itself are skipped (anchor.cpp:1931)
	hao_cpar P; P.pen_gap = A.par.pen_gap; P.pen_skip = A.par.pen_skip; P.bw = A.par.bw; P.max_skip = A.par.max_skip; P.max_iter = A.par.max_iter; P.max_dis = A.par.max_dis;
	P.xl = xl; P.yl = yl;
	int64_t plus = 0, msc = INT32_MIN, msc_i = INT32_MIN, movl = INT32_MAX, si = 0, ei = a_n;
	// ---- quick_ck_lchain ----
	{
		int64_t l = 0, k, z; bool sorted = true;
		for (k = 1; k <= a_n; ++k) {
			t[k - 1] = 0; ii[k - 1] = 0;
			if (k < a_n && HH_STRAND(a[k]) == HH_STRAND(a[l])) {
				if (a[k].self_offset <= a[k - 1].self_offset || a[k].offset <= a[k - 1].offset) sorted = false;
				continue;
			}
			if (sorted) {
				int64_t plus0 = 0, msc0 = INT32_MIN, msc_i0 = INT32_MIN, ddt = 0, sc, dd = 0;
				p[l] = -1; f[l] = HH_SPAN(a[l]);
				if (f[l] >= msc0) { msc0 = f[l]; msc_i0 = l; }
				if (f[l] < plus0) plus0 = f[l];
				for (z = l + 1; z < k; ++z) {
					int32_t s = hao_pair_score(a[z], a[z - 1], P, &dd);
					if (s == INT32_MIN) break;
					sc = (int64_t)s + f[z - 1];
					if (sc < HH_SPAN(a[z])) break;
					p[z] = (int32_t)(z - 1); f[z] = (int32_t)sc; ddt += dd;
					if (f[z] >= msc0) { msc0 = f[z]; msc_i0 = z; }
					if (f[z] < plus0) plus0 = f[z];
				}
				if (z >= k && msc_i0 == k - 1) {
					if (k - l >= 2 && ddt > 16 && ddt > hao_band(a[k - 1], a[l], P)) msc_i0 = INT32_MIN;
					if (msc_i0 == k - 1) {
						if (msc0 >= msc) {
							int64_t ov = hao_ext_len(a[msc_i0].self_offset, a[msc_i0].self_offset, P.xl, a[msc_i0].offset, a[msc_i0].offset, P.yl);
							if (msc0 > msc || ov < movl) { msc = msc0; msc_i = msc_i0; movl = ov; }
						}
						if (plus0 < plus) plus = plus0;
						if (ei > k) si = k; else ei = l;
					}
				}
			}
			l = k; sorted = true;
		}
	}
	// ---- DP over what the quick check left ----
	{
		int64_t i, j, st, max_ii = -1;
		for (i = st = si; i < ei; ++i) {
			int64_t max_f = HH_SPAN(a[i]), n_skip = 0, max_j = -1, end_j, sc;
			if (i - st > P.max_iter) st = i - P.max_iter;
			while (HH_STRAND(a[i]) != HH_STRAND(a[st])) ++st;
			for (j = i - 1; j >= st; --j) {
				int32_t s = hao_pair_score(a[i], a[j], P, nullptr);
				if (s == INT32_MIN) continue;
				sc = (int64_t)s + f[j];
				if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
				else if (t[j] == (int32_t)i) { if (++n_skip > P.max_skip) break; }
				if (p[j] >= 0) t[p[j]] = i;
			}
			end_j = j;
			if (max_ii < 0 || (int64_t)a[i].self_offset > (int64_t)a[max_ii].self_offset + P.max_dis || HH_STRAND(a[i]) != HH_STRAND(a[max_ii])) {
				int32_t mx = INT32_MIN; max_ii = -1;
				for (j = i - 1; j >= st && (int64_t)a[i].self_offset <= P.max_dis + (int64_t)a[j].self_offset && HH_STRAND(a[i]) == HH_STRAND(a[j]); --j)
					if (mx < f[j]) { mx = f[j]; max_ii = j; }
			}
			if (max_ii >= 0 && max_ii < end_j && HH_STRAND(a[i]) == HH_STRAND(a[max_ii])) {
				int32_t tmp = hao_pair_score(a[i], a[max_ii], P, nullptr);
				if (tmp != INT32_MIN && max_f < (int64_t)tmp + f[max_ii]) { max_f = (int64_t)tmp + f[max_ii]; max_j = max_ii; }
			}
			f[i] = (int32_t)max_f; p[i] = (int32_t)max_j;
			if (max_ii < 0 || ((int64_t)a[i].self_offset <= P.max_dis + (int64_t)a[max_ii].self_offset && HH_STRAND(a[i]) == HH_STRAND(a[max_ii]) && f[max_ii] < f[i])) max_ii = i;
			if (f[i] >= msc) {
				int64_t ovl = hao_ext_len(a[i].self_offset, a[i].self_offset, P.xl, a[i].offset, a[i].offset, P.yl);
				if (f[i] > msc || ovl < movl) { msc = f[i]; msc_i = i; movl = ovl; }
			}
			if (f[i] < plus) plus = f[i];
			ii[i] = 0;
		}
	}
	hao_chain_tail(A, g, a, a_n, P, f, p, t, ii, msc, msc_i, plus);
}

// the same through the group arrays and the global per-hit scratch (HAO_DBG_FORCE=seq_chain)
__device__ void hao_chain_generic(const hao_chain_args &A, const uint64_t g)
{
	const uint32_t r = A.g_read[g]; const uint64_t gs = A.g_start[g], ge = (g + 1 < A.g_off[r + 1]) ? A.g_start[g + 1] : A.seg[r + 1];
	const hao_hit_t *a = A.hits + gs; const int64_t a_n = (int64_t)(ge - gs);
	const uint32_t xid = (uint32_t)(A.rid_lo + r), yid = a_n > 0 ? HH_ID(a[0]) : xid;
	hao_chain_generic_core(A, g, gs, a, a_n, xid, yid, A.len[xid], a_n > 0 ? A.len[yid] : 0, A.f + gs, A.ii + gs, A.p + gs, A.t + gs);
}

// Groups of a handful of hits (spurious repeat matches: most groups of a repeat-rich read): one LANE per group runs the whole sequential
// algorithm - quick check, DP, output - with its per-hit scratch lane-interleaved in LDS.  A wave settles 64 groups at once instead of one.
// element i of a lane-interleaved LDS array (conflict-free: lane L owns words L, L + 64, ...)
template<class T> struct hao_lane_arr { T *p; __device__ __forceinline__ T &operator[](int64_t i) const { return p[i * 64]; } };

// The lane's hits are read ONCE, with independent loads, into LDS: the sequential algorithm indexes them through its scratch arrays (a[t[k]], a[ii[k]]), and
// every such access was a dependent global load of the same 16 - 128 bytes - ~10 memory round trips per group, the whole cost of the kernel on the repeat-rich
// sets (28 M groups of one or two hits per batch: 8.1 ms).
struct hao_lane_hits { const hao_hit_t *p; __device__ __forceinline__ hao_hit_t operator[](int64_t i) const { return p[i * 64]; } };
// slow == nullptr: every group of the list (HAO_DBG_FORCE=seq_chain); else the groups slow[0 .. *slow_cnt) that chain_pack8_kernel's quick check did not settle
__global__ __launch_bounds__(64) void chain_tiny_kernel(hao_chain_args A, const hao_gent *list, uint64_t n_list, const uint32_t *slow, const unsigned long long *slow_cnt)
{
	__shared__ int32_t l_f[HAO_TINY_MAX * 64], l_ii[HAO_TINY_MAX * 64], l_p[HAO_TINY_MAX * 64]; __shared__ int64_t l_t[HAO_TINY_MAX * 64]; __shared__ hao_hit_t l_a[HAO_TINY_MAX * 64];
	const uint64_t n = slow ? (uint64_t)*slow_cnt : n_list;
	for (uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64) {
		const hao_gent e = list[slow ? (uint64_t)slow[i] : i];
		{
			const hao_hit_t *src = A.hits + e.start;
#pragma unroll
			for (int k = 0; k < HAO_TINY_MAX; ++k) if ((uint32_t)k < e.n) l_a[k * 64 + threadIdx.x] = src[k];
		}
		const hao_lane_arr<int32_t> f{l_f + threadIdx.x}, ii{l_ii + threadIdx.x}, p{l_p + threadIdx.x}; const hao_lane_arr<int64_t> t{l_t + threadIdx.x};
		hao_chain_generic_core(A, e.g, e.start, hao_lane_hits{l_a + threadIdx.x}, (int64_t)e.n, (uint32_t)(A.rid_lo + e.r), e.yid, e.xl, e.yl, f, ii, p, t);
	}
}

__device__ __forceinline__ hao_hit_t hao_shfl_hit(const hao_hit_t &h, int src)
{ hao_hit_t o; o.w0 = hao_bcast(h.w0, src); o.offset = hao_bcast(h.offset, src); o.self_offset = hao_bcast(h.self_offset, src); o.cnt = hao_bcast(h.cnt, src); return o; }
__device__ __forceinline__ hao_hit_t hao_shfl_up_hit(const hao_hit_t &h)      // lane 0 keeps its own value (callers overwrite it)
{ hao_hit_t o; o.w0 = hao_wave_shr1(h.w0, h.w0); o.offset = hao_wave_shr1(h.offset, h.offset); o.self_offset = hao_wave_shr1(h.self_offset, h.self_offset); o.cnt = hao_wave_shr1(h.cnt, h.cnt); return o; }

// One wave per (query,target) group.
// Fast path (data-parallel): every strand block passes quick_ck_lchain - a segmented prefix sum of pair
// scores with per-pair validity flags - and no second chain qualifies for multi-copy output; then the best
// block IS the chain: hits are copied through, the fake cigar is a flagged compaction.  >99.9 % of groups on
// repeat-free genomes.  Everything else runs hao_chain_generic on lane 0 (exact sequential algorithm).
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void chain_group_kernel(hao_chain_args A, const hao_gent *list, uint64_t n_list, uint32_t *slow, int cls)
{
	const uint64_t li = blockIdx.x;      // (one 64-thread workgroup per group)
	if (li >= n_list) return;
	const int lane = hao_lane();
	// fake-cigar entries are collected during the scan (an entry wherever offset - self_offset changes: the region's constant only shifts the
	// values), up to 64 per strand block in LDS, so the hits are read ONCE; longer cigars fall back to a second sweep
	__shared__ uint64_t l_eb[2][64];
	uint64_t (*eb)[64] = l_eb; uint32_t ce0 = 0, ce1 = 0;
	const unsigned long long tq0 = A.dbg_qc ? wall_clock64() : 0;
	const hao_gent e = list[li];                                  // wave-uniform: scalar loads
	const uint64_t g = e.g, gs = e.start; const int32_t a_n = (int32_t)e.n;      // (a group has < 2^31 hits: a batch has < 2^32 seed hits)
	const hao_hit_t *a = A.hits + gs;
	const uint32_t xid = (uint32_t)(A.rid_lo + e.r), yid = e.yid;
	if (yid == xid || a_n <= 0) {      // hits to the query itself are skipped (anchor.cpp:1931); their positions of the code array say "nothing" (every position of the array is written by the kernel that sees its group: no fill pass)
		if (A.hcode) for (int32_t i = lane; i < a_n; i += 64) A.hcode[gs + i] = 0x08;
		if (lane == 0) { A.nch[g] = 0; A.nout[g] = 0; }
		return;
	}
	hao_hit_t hn = a[lane < a_n ? lane : 0];                     // tile 0; every later tile is requested one iteration ahead
	const uint16_t *hqg = A.hcode ? A.hq + gs : nullptr; uint8_t *hcg = A.hcode ? A.hcode + gs : nullptr;
	uint32_t qn = hcg ? hqg[lane < a_n ? lane : 0] : 0u, carry_q = 0;
	const hao_hit_t first0 = hao_shfl_hit(hn, 0);
	hao_cpar P; P.pen_gap = A.par.pen_gap; P.pen_skip = A.par.pen_skip; P.bw = A.par.bw; P.max_skip = A.par.max_skip; P.max_iter = A.par.max_iter; P.max_dis = A.par.max_dis;
	P.xl = e.xl; P.yl = e.yl;
	const unsigned long long tq1 = A.dbg_qc ? wall_clock64() + (first0.w0 & 0) : 0;      // (+ 0 through the loaded hit: the stamp waits for tile 0)
	// ---- parallel quick check ----
	// A group's hits are sorted by (strand, query minimizer, list order): at most TWO strand blocks, block 0 in front.  So the per-block running score is a plain
	// prefix sum (six v_add_dpp) minus its value in front of the boundary - no segmented scan -, every vote per block is ONE wave-wide vote split by the mask of
	// block 1's lanes in scalar registers, and a block's last hit is the lane in front of the boundary (or the group's last lane): nothing of the NEXT tile is looked
	// at, its load has a whole iteration to arrive.  (A third block - it cannot happen - sends the group to the exact path.)
	int32_t carry_f = 0; hao_hit_t carry_h = first0; carry_h.w0 ^= 0x80000000u;      // (hit 0 sees a predecessor of the other strand: it starts a block like block 1's first hit does)
	bool fail0 = false, fail1 = false, seen = false; int32_t maxf0 = INT32_MIN, maxf1 = INT32_MIN, flast0 = 0, flast1 = 0, k1 = 0, n_bnd = 0;
	uint32_t ddt0 = 0, ddt1 = 0;      // a block's sum of dd, SATURATING at 2^32 - 1: all that is asked of it is whether it exceeds 16 and a band of less than 2^31 (a 64-bit sum over the wave: ~35 instructions per block)
	uint32_t last0_so = first0.self_offset, last0_of = first0.offset, last1_so = last0_so, last1_of = last0_of;      // (self_offset, offset) of each block's last hit: all that is used of it
	unsigned long long first = 1ULL;      // bit 0 in the group's first tile
	const uint32_t ulane = (uint32_t)lane;
	for (int32_t t0 = 0; t0 < a_n; t0 += 64) {
		const int32_t left = a_n - t0; const bool act = lane < left;
		const unsigned long long actm = left >= 64 ? ~0ULL : (1ULL << left) - 1ULL;
		const hao_hit_t h = hn;      // (lanes behind the group's end hold a stale hit: every use below is under `act` or a vote masked by actm)
		const uint32_t q = qn;
		if (left > 64) {      // the next tile, from a uniform base; lanes behind its end repeat its last hit
			const uint32_t j = min(ulane, (uint32_t)(left - 65)); const hao_hit_t *an = a + (t0 + 64);
			hn = an[j]; if (hcg) qn = (hqg + (t0 + 64))[j];
		}
		hao_hit_t ph; ph.w0 = hao_wave_shr1(h.w0, carry_h.w0); ph.offset = hao_wave_shr1(h.offset, carry_h.offset); ph.self_offset = hao_wave_shr1(h.self_offset, carry_h.self_offset); ph.cnt = 0;      // lane 0: the previous tile's last hit
		const bool bnd = (int32_t)(h.w0 ^ ph.w0) < 0, st = act && bnd;
		const unsigned long long Mst = __ballot(bnd) & actm, Mbnd = Mst & ~first;      // a block starts at this lane; block 1 does
		// lanes of block 1: behind the boundary (inactive lanes fall on either side: they add nothing)
		unsigned long long Mb = 0; uint32_t bthr = 64; int B = 0;
		if (seen) { Mb = actm; bthr = 0; }
		else if (Mbnd) { B = __ffsll((long long)Mbnd) - 1; Mb = (~0ULL << B) & actm; bthr = (uint32_t)B; }
		const int b = ulane >= bthr;
		if (hcg) {      // wire code of this hit relative to the previous one of its strand block: minimizers skipped << 4 | diagonal shift + 8; 0xff = not expressible
			const int32_t idx = t0 + lane;
			const uint32_t pq = hao_wave_shr1(q, carry_q);
			const int32_t dq = (int32_t)(q - pq), sh = (int32_t)((h.offset - ph.offset) - (h.self_offset - ph.self_offset));
			bool exc = false;
			if (A.exc_every) { uint32_t ii = (uint32_t)idx, ee = A.exc_every; HAO_OPAQUE_U32(ii); HAO_OPAQUE_U32(ee); exc = ii % ee == ee - 1; }      // (tests only; opaque: the division's set-up - 12 instructions - otherwise moves in front of the loop of every group)
			const uint8_t code = st ? (uint8_t)0x08 : ((dq < 1 || dq > 15 || sh < -8 || sh > 7 || q == 65535u || exc) ? (uint8_t)0xff : (uint8_t)((dq - 1) << 4 | (sh + 8)));
			if (act) hcg[idx] = code;
			carry_q = hao_bcast(q, 63);
		}
		// comput_sc_ch_ec (hao_pair_score) of every lane with its predecessor, WITHOUT its early returns: `bad` collects them, the arithmetic of a bad pair runs on
		// and is thrown away (three nested exec regions with their defaults cost more than the few lanes they spare)
		const int32_t span = HH_SPAN(h);
		const int32_t dq = (int32_t)(h.self_offset - ph.self_offset), dr = (int32_t)(h.offset - ph.offset);
		bool bad = dq <= 0 || dr <= 0;
		const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
		if (!bad && dd > 16) { if (dd > hao_band(h, ph, P)) bad = true; }
		int32_t sc = span < dg ? span : dg;
		{	const int32_t wgt = HH_WGT(h);
			if (wgt != 1) sc = sc >= wgt ? (wgt == 2 ? sc >> 1 : sc / wgt) : 1; }      // weight 1 (and 2) are the rule: the integer division (~40 instructions) only runs for rarer seeds
		if (dd) {
			double lin = P.pen_gap * (double)dd; const double skip = P.pen_skip * (double)dg;
			if (!(dd < 4 && lin + skip < 1.0)) {      // (see hao_pair_score)
				const double ap = (double)sc * (((double)dd / (double)dg) / P.bw);
				if (dd < 4) lin = lin > ap ? ap : lin; else lin = lin < ap ? ap : lin;
				lin += skip;
				sc -= (int32_t)lin;
			}
		} else if (dg > span) sc -= (int32_t)(0.0 + P.pen_skip * (double)dg);
		const bool scored = act && !bnd && !bad;                       // the pair counts
		const int32_t s = bad ? INT32_MIN : sc;                        // (for the break test; block starts are masked there)
		const int32_t x = st ? span : (scored ? sc : 0);
		const uint32_t dde = scored ? (uint32_t)dd : 0u;              // (dd < 2^28 per pair)
		{	const bool cf = act && (bnd || dr != dq);                  // the diagonal offset - self_offset changes here: a fake-cigar entry
			const unsigned long long Mcf = (__ballot(dr != dq) | Mst) & actm;      // = the vote on cf
			const uint32_t n0 = (uint32_t)__popcll(Mcf & ~Mb), n1 = (uint32_t)__popcll(Mcf & Mb);
			if (cf) { const uint32_t at = hao_mbcnt(Mcf) + (b ? ce1 - n0 : ce0); if (at < 64) eb[b][at] = (uint64_t)h.self_offset << 32 | (h.offset - h.self_offset); }      // (block 0's entries are all below block 1's lanes)
			ce0 += n0; ce1 += n1; }
		const int32_t F = (int32_t)hao_wave_incl_scan_u32((uint32_t)x);
		int32_t f;
		if (Mbnd) {
			n_bnd += __popcll(Mbnd);
			const int32_t sub = B ? hao_bcast(F, B - 1) : 0;
			f = lane >= B ? F - sub : F + carry_f;
			// the hit in front of the boundary is block 0's last one
			if (B) { flast0 = hao_bcast(f, B - 1); last0_so = hao_bcast(h.self_offset, B - 1); last0_of = hao_bcast(h.offset, B - 1); }
			else { flast0 = carry_f; last0_so = carry_h.self_offset; last0_of = carry_h.offset; }
		} else f = F + carry_f;
		const int32_t fp = hao_wave_shr1(f, carry_f);
		// the chain breaks at a hit that cannot follow its predecessor, or whose score with it falls below its own span: s + fp < span in 64 bits = the same with a
		// SATURATING 32-bit sum (span >= 0; a sum beyond INT32_MAX is not below it, one below INT32_MIN - or s = INT32_MIN with any fp - is): v_add_i32 clamp + one compare
		const unsigned long long Mbrk = __ballot(hao_add_sat_i32(s, fp) < span) & actm & ~Mst;
		if (Mbrk & ~Mb) fail0 = true;
		if (Mbrk & Mb) fail1 = true;
		// a lane behind the group's end repeats the last hit's f and adds dd = 0: no guard
		if (bthr == 64) { maxf0 = max(maxf0, f); ddt0 = hao_add_sat_u32(ddt0, dde); }
		else if (bthr == 0) { maxf1 = max(maxf1, f); ddt1 = hao_add_sat_u32(ddt1, dde); }
		else if (b == 0) { maxf0 = max(maxf0, f); ddt0 = hao_add_sat_u32(ddt0, dde); } else { maxf1 = max(maxf1, f); ddt1 = hao_add_sat_u32(ddt1, dde); }
		k1 += __popcll(actm & ~Mb);
		if (left <= 64) {      // the group's last hit ends its block
			const int Ls = left - 1; const int32_t vf = hao_bcast(f, Ls); const uint32_t vso = hao_bcast(h.self_offset, Ls), vof = hao_bcast(h.offset, Ls);
			if ((Mb >> Ls) & 1ULL) { flast1 = vf; last1_so = vso; last1_of = vof; } else { flast0 = vf; last0_so = vso; last0_of = vof; }
		}
		carry_f = hao_bcast(f, 63); carry_h = hao_shfl_hit(h, 63);
		if (Mbnd) seen = true;
		first = 0;
	}
	if (n_bnd > 1) fail0 = fail1 = true;
	const bool two = k1 < a_n;
	maxf0 = hao_wave_max_i32(maxf0); ddt0 = hao_wave_sum_sat_u32(ddt0);
	if (two) { maxf1 = hao_wave_max_i32(maxf1); ddt1 = hao_wave_sum_sat_u32(ddt1); }      // (most groups have one strand block)
	const unsigned long long tq2 = A.dbg_qc ? wall_clock64() + (unsigned long long)(maxf0 & 0) : 0;
	const hao_hit_t first1 = two ? a[k1] : first0;
	hao_hit_t last0, last1; last0.w0 = first0.w0; last0.cnt = 0; last0.self_offset = last0_so; last0.offset = last0_of; last1.w0 = first1.w0; last1.cnt = 0; last1.self_offset = last1_so; last1.offset = last1_of;
	bool acc0 = !fail0 && flast0 == maxf0 && !(k1 >= 2 && ddt0 > 16u && (int64_t)ddt0 > (int64_t)hao_band(last0, first0, P));
	bool acc1 = two && !fail1 && flast1 == maxf1 && !(a_n - k1 >= 2 && ddt1 > 16u && (int64_t)ddt1 > (int64_t)hao_band(last1, first1, P));
	bool fast = acc0 && (!two || acc1);
	int best = 0; int64_t msc = flast0;
	if (fast && two) {
		int64_t ov0 = hao_ext_len(last0.self_offset, last0.self_offset, P.xl, last0.offset, last0.offset, P.yl);
		int64_t ov1 = hao_ext_len(last1.self_offset, last1.self_offset, P.xl, last1.offset, last1.offset, P.yl);
		if (flast1 >= flast0 && (flast1 > flast0 || ov1 < ov0)) { best = 1; msc = flast1; }
	}
	const int32_t bl = best ? k1 : 0, cL = best ? a_n - k1 : k1;
	if (fast && A.par.mcopy_num > 1 && cL >= A.par.mcopy_khit_cut && two) {
		int64_t min_sc = (int64_t)((double)msc * A.par.mcopy_rate);          // plus == 0 here: every f >= span > 0
		if ((int64_t)(best ? maxf0 : maxf1) >= min_sc) fast = false;         // a second chain may qualify: exact sequential path
	}
	if (!fast) {
		// the DP decides this group's chains: the codes written above describe none of them (a void 0xff would only become a useless verbatim-list entry)
		if (hcg) for (int32_t i = lane; i < a_n; i += 64) hcg[i] = 0x08;
		if (lane == 0) { unsigned long long si_ = atomicAdd(A.stats + cls, 1ULL); atomicAdd(A.stats + HAO_NCLS, (unsigned long long)a_n); slow[si_] = (uint32_t)li; A.nch[g] = 0; A.nout[g] = 0; }
		return;
	}
	// ---- single chain = the whole best block ----
	if (hcg && two) for (int32_t i = (best ? 0 : k1) + lane; i < (best ? k1 : a_n); i += 64) hcg[i] = 0x08;      // the other strand block is in no chain: a 0xff there would only become a verbatim-list entry nobody reads
	uint64_t *fcs = A.fcs + gs + 6 * g; hao_chain_rec rc;
	hao_region(rc, P.xl, P.yl, msc, best ? first1 : first0, best ? last1 : last0);
	uint32_t cnt; const uint32_t ce = best ? ce1 : ce0;
	if (ce <= 64) {
		HAO_WAVE_FENCE();
		const int64_t cdiag = (int64_t)rc.y_pos_s - (int64_t)rc.x_pos_s;      // dd of a hit = (offset - self_offset) - cdiag
		if (lane == 0) fcs[0] = hao_fc_entry(rc.x_pos_s, 0);
		if ((uint32_t)lane < ce) { const uint64_t raw = eb[best][lane]; fcs[1 + lane] = hao_fc_entry((uint32_t)(raw >> 32), (int32_t)((int64_t)(int32_t)(uint32_t)raw - cdiag)); }
		const uint64_t rlast = eb[best][ce - 1];
		cnt = 1 + ce;
		if ((uint32_t)(rlast >> 32) != rc.x_pos_e) { if (lane == 0) fcs[cnt] = hao_fc_entry(rc.x_pos_e, (int32_t)((int64_t)(int32_t)(uint32_t)rlast - cdiag)); ++cnt; }
	} else cnt = hao_fake_cigar_wave(fcs, rc.x_pos_s, rc.y_pos_s, rc.x_pos_e, cL, [&](int64_t k) { return a[bl + k]; });
	if (lane == 0) {
		rc.hit_rel = 0; rc.n_hits = (uint32_t)cL; rc.src_rel = (uint32_t)bl; rc.in_place = 1; rc.fc_rel = 0; rc.fc_len = cnt;
		A.rec[g * HAO_MCOPY_MAX] = rc; A.nch[g] = 1; A.nout[g] = (uint32_t)cL;
		if (A.dbg_qc && (li & 63) == 0) { /* sampled: one group in 64 */ const unsigned long long tq3 = wall_clock64(); atomicAdd(A.dbg_qc, tq1 - tq0); atomicAdd(A.dbg_qc + 1, tq2 - tq1); atomicAdd(A.dbg_qc + 2, tq3 - tq2); atomicAdd(A.dbg_qc + 3, 1ULL); atomicAdd(A.dbg_qc + 4, (unsigned long long)a_n); }
	}
}


// Groups of <= 8 hits, EIGHT per wave: lanes 8 s .. 8 s + 7 hold the hits of the wave's s-th group, and chain_group_kernel's data-parallel quick check
// (quick_ck_lchain, Hash_Table.cpp:2007-2094, as a segmented prefix sum over strand blocks with per-pair validity flags) runs on all eight groups at once:
// the scan's segments end at group boundaries, every wave-wide vote / reduction becomes its 8-lane counterpart (a byte of a ballot, three xor-shuffles).  A group
// the check settles gets its one chain in place (the best strand block of the sorted seed hits), its region, fake cigar and wire codes from its own lanes; the
// group it does not settle goes to the class's slow list for chain_tiny_kernel (the exact sequential routine, one lane per group).  (chain_tiny_kernel ran that routine on
// one lane per group for ALL of them: 64 divergent walks per wave - 8.1 ms per batch of the repeat-rich 250 Mb set, whose 28 M tiny groups per batch are mostly
// single hits of repeat copies.)
__global__ __launch_bounds__(256) void chain_pack8_kernel(hao_chain_args A, const hao_gent *list, uint64_t n_list, uint32_t *slow)
{
	const int lane = hao_lane(), wv = threadIdx.x >> 6, seg = lane >> 3, idx = lane & 7, sb = lane & ~7;
	const uint64_t li = ((uint64_t)blockIdx.x * 4 + wv) * 8 + seg;      // index in the class list
	const bool have = li < n_list;
	hao_gent e; e.g = 0; e.r = 0; e.start = 0; e.n = 0; e.yid = 0; e.xl = 0; e.yl = 0;
	if (have) e = list[li];
	const int32_t a_n = (int32_t)e.n; const uint64_t g = e.g, gs = e.start;
	const uint32_t xid = (uint32_t)(A.rid_lo + e.r), yid = e.yid;
	const bool skip = !have || yid == xid || a_n <= 0;                       // hits to the query itself are skipped (anchor.cpp:1931)
	const bool act = !skip && idx < a_n;
	hao_hit_t h; h.w0 = 0; h.offset = 0; h.self_offset = 0; h.cnt = 0;
	if (act) h = A.hits[gs + idx];
	const uint32_t q = (A.hcode && act) ? A.hq[gs + idx] : 0u;
	hao_cpar P; P.pen_gap = A.par.pen_gap; P.pen_skip = A.par.pen_skip; P.bw = A.par.bw; P.max_skip = A.par.max_skip; P.max_iter = A.par.max_iter; P.max_dis = A.par.max_dis;
	P.xl = e.xl; P.yl = e.yl;
	const hao_hit_t ph = hao_shfl_up_hit(h);
	const uint32_t strand0 = (uint32_t)__shfl((int)HH_STRAND(h), sb);
	const bool st = act && (idx == 0 || HH_STRAND(h) != HH_STRAND(ph));      // first hit of a strand block
	const int b = act && HH_STRAND(h) != strand0;
	int32_t s = HH_SPAN(h); bool ok = true; int64_t dd = 0;
	if (act && !st) { s = hao_pair_score(h, ph, P, &dd); ok = s != INT32_MIN; if (!ok) { s = 0; dd = 0; } }
	uint8_t code = 0x08;
	if (A.hcode) {      // wire code of this hit relative to the previous one of its strand block (hao_deliver.cuh)
		const uint32_t pq = hao_wave_shr1(q, 0u);
		const int32_t dq = (int32_t)(q - pq), sh = (int32_t)((h.offset - ph.offset) - (h.self_offset - ph.self_offset));
		if (!st) code = (dq < 1 || dq > 15 || sh < -8 || sh > 7 || q == 65535u || (A.exc_every && (uint32_t)idx % A.exc_every == A.exc_every - 1)) ? (uint8_t)0xff : (uint8_t)((dq - 1) << 4 | (sh + 8));
	}
	const uint32_t diag = h.offset - h.self_offset, pdiag = ph.offset - ph.self_offset;
	const bool cf = act && (st || diag != pdiag);                            // a fake-cigar entry starts here
	int32_t x = act ? s : 0; int fl = (st || idx == 0) ? 1 : 0;
	hao_seg_scan_add(x, fl);
	const int32_t f = x, fp = hao_wave_shr1(f, 0);
	const bool brk = act && !st && (!ok || (int64_t)s + fp < (int64_t)HH_SPAN(h));
	auto seg_byte = [&](unsigned long long m) { return (uint32_t)(m >> sb) & 0xffu; };      // the group's eight lanes of a wave-wide vote
	const uint32_t in0 = seg_byte(__ballot(act && b == 0)), brk0 = seg_byte(__ballot(brk && b == 0)), brk1 = seg_byte(__ballot(brk && b == 1)),
				   cf0 = seg_byte(__ballot(cf && b == 0)), cf1 = seg_byte(__ballot(cf && b == 1));
	const int32_t k1 = __popc(in0);                                          // hits of the first strand block
	const bool two = k1 < a_n;
	int32_t maxf0 = (act && b == 0) ? f : INT32_MIN, maxf1 = (act && b == 1) ? f : INT32_MIN, ddt0 = (act && b == 0) ? (int32_t)dd : 0, ddt1 = (act && b == 1) ? (int32_t)dd : 0;      // (dd < 2^27 per pair)
#pragma unroll
	for (int d = 1; d < 8; d <<= 1) { maxf0 = max(maxf0, __shfl_xor(maxf0, d)); maxf1 = max(maxf1, __shfl_xor(maxf1, d)); ddt0 += __shfl_xor(ddt0, d); ddt1 += __shfl_xor(ddt1, d); }
	const int l0 = sb + (k1 > 0 ? k1 - 1 : 0), f1 = sb + (two ? k1 : 0), l1 = sb + (a_n > 0 ? a_n - 1 : 0);      // lanes of: last hit of block 0, first / last hit of block 1
	const int32_t flast0 = __shfl(f, l0), flast1 = __shfl(f, l1);
	hao_hit_t first0, last0, first1, last1;
	first0.w0 = (uint32_t)__shfl((int)h.w0, sb); first0.self_offset = (uint32_t)__shfl((int)h.self_offset, sb); first0.offset = (uint32_t)__shfl((int)h.offset, sb); first0.cnt = 0;
	last0.w0 = first0.w0; last0.self_offset = (uint32_t)__shfl((int)h.self_offset, l0); last0.offset = (uint32_t)__shfl((int)h.offset, l0); last0.cnt = 0;
	first1.w0 = (uint32_t)__shfl((int)h.w0, f1); first1.self_offset = (uint32_t)__shfl((int)h.self_offset, f1); first1.offset = (uint32_t)__shfl((int)h.offset, f1); first1.cnt = 0;
	last1.w0 = first1.w0; last1.self_offset = (uint32_t)__shfl((int)h.self_offset, l1); last1.offset = (uint32_t)__shfl((int)h.offset, l1); last1.cnt = 0;
	const bool acc0 = !brk0 && flast0 == maxf0 && !(k1 >= 2 && ddt0 > 16 && ddt0 > hao_band(last0, first0, P));
	const bool acc1 = two && !brk1 && flast1 == maxf1 && !(a_n - k1 >= 2 && ddt1 > 16 && ddt1 > hao_band(last1, first1, P));
	bool fast = acc0 && (!two || acc1);
	int best = 0; int64_t msc = flast0;
	if (fast && two) {
		const int64_t ov0 = hao_ext_len(last0.self_offset, last0.self_offset, P.xl, last0.offset, last0.offset, P.yl);
		const int64_t ov1 = hao_ext_len(last1.self_offset, last1.self_offset, P.xl, last1.offset, last1.offset, P.yl);
		if (flast1 >= flast0 && (flast1 > flast0 || ov1 < ov0)) { best = 1; msc = flast1; }
	}
	const int32_t bl = best ? k1 : 0, cL = best ? a_n - k1 : k1;
	if (fast && A.par.mcopy_num > 1 && cL >= A.par.mcopy_khit_cut && two) {
		const int64_t min_sc = (int64_t)((double)msc * A.par.mcopy_rate);        // plus == 0 here: every f >= span > 0
		if ((int64_t)(best ? maxf0 : maxf1) >= min_sc) fast = false;             // a second chain may qualify: exact sequential path
	}
	{	// The groups the check does not settle go to the class's slow list (one atomic per wave; A.stats[0] counts them like chain_group_kernel's classes) and
		// are chained by chain_tiny_kernel, 64 to a wave: run here - one lane of a group's eight, in a quarter of the waves of a repeat-rich set (3.7 % of its
		// tiny groups) - the sequential routine cost as much as everything else in this kernel.  Their codes keep the "nothing to say" prefill until then.
		const bool sl = !skip && !fast && idx == 0; const unsigned long long sq = __ballot(sl);
		if (sq) {
			unsigned long long base = 0;
			if (lane == 0) base = atomicAdd(A.stats, (unsigned long long)__popcll(sq));
			base = (unsigned long long)hao_readlane_i64((int64_t)base, 0);
			if (sl) { slow[base + __popcll(sq & ((1ULL << lane) - 1))] = (uint32_t)(li & 0xffffffffu); A.nch[g] = 0; A.nout[g] = 0; }
		}
	}
	if (A.hcode && have && idx < a_n) A.hcode[gs + idx] = (!skip && fast && b == best) ? code : (uint8_t)0x08;      // every position of the group gets its byte here (a chain's hits their code)
	if (skip) { if (have && idx == 0) { A.nch[g] = 0; A.nout[g] = 0; } return; }
	if (!fast) return;
	// ---- single chain = the whole best strand block, in place ----
	uint64_t *fcs = A.fcs + gs + 6 * g; hao_chain_rec rc;
	hao_region(rc, P.xl, P.yl, msc, best ? first1 : first0, best ? last1 : last0);
	const int64_t cdiag = (int64_t)rc.y_pos_s - (int64_t)rc.x_pos_s;            // dd of a hit = (offset - self_offset) - cdiag
	const uint32_t cfm = best ? cf1 : cf0, ce = (uint32_t)__popc(cfm);
	if (cf && b == best) fcs[1 + __popc(cfm & ((1u << idx) - 1u))] = hao_fc_entry(h.self_offset, (int32_t)((int64_t)(int32_t)diag - cdiag));
	const int lc = sb + (cfm ? 31 - __clz((int)cfm) : 0);                     // lane of the block's last cigar entry
	const uint32_t last_site = (uint32_t)__shfl((int)h.self_offset, lc), last_diag = (uint32_t)__shfl((int)diag, lc);
	uint32_t cnt = 1 + ce;
	if (last_site != rc.x_pos_e) { if (idx == 0) fcs[cnt] = hao_fc_entry(rc.x_pos_e, (int32_t)((int64_t)(int32_t)last_diag - cdiag)); ++cnt; }
	if (idx == 0) {
		fcs[0] = hao_fc_entry(rc.x_pos_s, 0);
		rc.hit_rel = 0; rc.n_hits = (uint32_t)cL; rc.src_rel = (uint32_t)bl; rc.in_place = 1; rc.fc_rel = 0; rc.fc_len = cnt;
		A.rec[g * HAO_MCOPY_MAX] = rc; A.nch[g] = 1; A.nout[g] = (uint32_t)cL;
	}
}

// Wave-cooperative hao_chain_tail for chain_dp_kernel: the order-dependent walks (backtrack, chain extraction) stay on lane 0
// over (normally LDS-resident) arrays; candidate collection, the sort of the (distinct) candidate keys, hit copies and fake
// cigars run on all lanes.  cn[8] / l_rec[3]: LDS scratch of the wave.
__device__ __forceinline__ void hao_chain_tail_wave(const hao_chain_args &A, const uint64_t g, const uint64_t gs, const hao_hit_t *a, const int64_t a_n, const hao_cpar &P,
		int32_t *f, int32_t *p, int64_t *t, int32_t *ii, const int64_t t_cap, uint32_t *cn, hao_chain_rec *l_rec, int64_t msc, int64_t msc_i, int64_t plus)
{
	const int lane = hao_lane();
	hao_hit_t *des = A.ohits + gs; uint64_t *fcs = A.fcs + gs + 6 * g; hao_chain_rec *rec = A.rec + g * HAO_MCOPY_MAX;
	int64_t cL = 0;
	if (lane == 0) for (int64_t i = msc_i; i >= 0; i = p[i]) { ii[i] = 1; t[cL++] = i; }
	cL = __shfl(cL, 0);
	HAO_WAVE_FENCE();
	if (A.par.mcopy_num > 1 && cL >= A.par.mcopy_khit_cut) {      // multi-copy chains (Hash_Table.cpp:2178-2270)
		msc -= plus; const int64_t min_sc = (int64_t)((double)msc * A.par.mcopy_rate);
		if (lane == 0) ii[msc_i] = 0;
		HAO_WAVE_FENCE();
		int64_t ch_n = 0;
		for (int64_t b = 0; b < a_n; b += 64) {     // candidates: chain ends off the best chain scoring >= min_sc, in index order
			const int64_t i = b + lane; int32_t fv = 0; bool cand = false;
			if (i < a_n) { fv = f[i] - (int32_t)plus; f[i] = fv; cand = !ii[i] && fv >= min_sc; }
			const unsigned long long bal = __ballot(cand);
			if (cand) t[ch_n + __popcll(bal & ((1ULL << lane) - 1))] = (int64_t)((uint64_t)fv << 32) + (i << 1);
			ch_n += __popcll(bal);
		}
		HAO_WAVE_FENCE();
		for (int64_t i = ch_n + lane; i < a_n; i += 64) t[i] = 0;
		HAO_WAVE_FENCE();
		if (ch_n > 1) {
			// ascending sort of t[0, ch_n): keys are distinct (they carry the hit index), so any correct sort reproduces the reference's order
			if (ch_n <= 64) {
				const int64_t key = lane < ch_n ? t[lane] : INT64_MAX; int rank = 0;
				for (int l = 0; l < (int)ch_n; ++l) rank += hao_readlane_i64(key, l) < key;
				HAO_WAVE_FENCE();
				if (lane < ch_n) t[rank] = key;
			} else {
				int64_t P2 = 128; while (P2 < ch_n) P2 <<= 1;
				if (P2 <= t_cap) {
					for (int64_t i = ch_n + lane; i < P2; i += 64) t[i] = INT64_MAX;
					HAO_WAVE_FENCE();
					for (int64_t k = 2; k <= P2; k <<= 1)
						for (int64_t j = k >> 1; j > 0; j >>= 1) {
							for (int64_t i = lane; i < P2; i += 64) {
								const int64_t x = i ^ j;
								if (x > i) { const int64_t u = t[i], v = t[x]; if ((u > v) == ((i & k) == 0)) { t[i] = v; t[x] = u; } }
							}
							HAO_WAVE_FENCE();
						}
					for (int64_t i = ch_n + lane; i < P2; i += 64) t[i] = 0;
				} else if (lane == 0) hao_heapsort_i64(t, ch_n);
			}
			HAO_WAVE_FENCE();
			if (lane == 0) {     // walk the candidates from the best score down; a chain stops where it meets an already used hit
				int64_t n_v = 0, n_v0, k, sc, i; uint32_t nu = 0;
				for (k = ch_n - 1; k >= 0 && nu < (uint32_t)A.par.mcopy_num; --k) {
					n_v0 = n_v;
					for (i = (int64_t)((uint32_t)t[k] >> 1); i >= 0 && (t[i] & 1) == 0; ) { ii[n_v++] = (int32_t)i; t[i] |= 1; i = p[i]; }
					if (n_v0 == n_v) continue;
					sc = i < 0 ? (t[k] >> 32) : ((t[k] >> 32) - f[i]);
					if (sc >= min_sc && (!nu || n_v - n_v0 > 1)) {
						hao_region(l_rec[nu], P.xl, P.yl, sc + plus, a[ii[n_v - 1]], a[ii[n_v0]]);
						cn[nu] = (uint32_t)n_v0; cn[3 + nu] = (uint32_t)(n_v - n_v0); ++nu;
					} else n_v = n_v0;
				}
				cn[6] = nu;
			}
			HAO_WAVE_FENCE();
			const uint32_t n_u = cn[6]; uint32_t o = 0, fcn = 0;
			for (uint32_t k = 0; k < n_u; ++k) {
				const int64_t n_v0 = cn[k], ni = cn[3 + k]; hao_chain_rec rc = l_rec[k];
				for (int64_t j = lane; j < ni; j += 64) des[o + j] = a[ii[n_v0 + (ni - j - 1)]];
				if (A.hcode) hao_code_chain_wave(A, gs, a, (int64_t)o, ni, [&](int64_t q_) { return (int64_t)ii[n_v0 + (ni - q_ - 1)]; });
				rc.hit_rel = o; rc.n_hits = (uint32_t)ni; rc.src_rel = o; rc.in_place = 0; rc.fc_rel = fcn;
				rc.fc_len = hao_fake_cigar_wave(fcs + fcn, rc.x_pos_s, rc.y_pos_s, rc.x_pos_e, ni, [&](int64_t q) { return a[ii[n_v0 + (ni - q - 1)]]; });
				fcn += rc.fc_len; o += (uint32_t)ni;
				if (lane == 0) rec[k] = rc;
			}
			if (lane == 0) { A.nch[g] = n_u; A.nout[g] = o; }
			return;
		}
		msc += plus; cL = 0;
		if (lane == 0) for (int64_t i = msc_i; i >= 0; i = p[i]) t[cL++] = i;
		cL = __shfl(cL, 0);
		HAO_WAVE_FENCE();
	}
	hao_chain_rec rc;
	hao_region(rc, P.xl, P.yl, msc, a[t[cL - 1]], a[t[0]]);
	for (int64_t i = lane; i < cL; i += 64) des[i] = a[t[cL - i - 1]];
	if (A.hcode) hao_code_chain_wave(A, gs, a, (int64_t)0, cL, [&](int64_t q_) { return (int64_t)t[cL - q_ - 1]; });
	rc.hit_rel = 0; rc.n_hits = (uint32_t)cL; rc.src_rel = 0; rc.in_place = 0; rc.fc_rel = 0;
	rc.fc_len = hao_fake_cigar_wave(fcs, rc.x_pos_s, rc.y_pos_s, rc.x_pos_e, cL, [&](int64_t q) { return a[t[cL - q - 1]]; });
	if (lane == 0) { rec[0] = rc; A.nch[g] = 1; A.nout[g] = (uint32_t)cL; }
}

// ---------------------------------------------------------------------------------------
// Groups the quick check does not settle: one wave per group.
//   * quick check again, data-parallel, this time storing f/p of every block (accepted blocks keep them);
//   * the chain DP (Hash_Table.cpp:2124-2176) stays sequential in i, but the inner predecessor loop is
//     evaluated 64 candidates at a time: pair scores in parallel, "t[p[j]] = i" marks through LDS (a mark
//     can only land on a smaller j, so publishing a whole tile before reading is order-safe), an exclusive
//     prefix maximum decides which candidates improve, and the n_skip / max_skip early exit is replayed over
//     two ballot masks by scalar code - bit-identical to the sequential scan;
//   * backtrack / multi-copy / output: hao_chain_tail on lane 0.
// f, p and the marks live in LDS for groups up to HAO_DP_CAP hits, else in global scratch.
// ---------------------------------------------------------------------------------------
#define HAO_DP_CAP 2048

// INLDS is a template parameter (not a run-time select) so that every access to f/p/tm/ii/t and the staged hits is a DS
// instruction instead of a flat one.
template<int CAP, bool STAGE, bool INLDS>
__device__ __forceinline__ void hao_dp_body(const hao_chain_args &A, const hao_gent &e, const hao_hit_t *ag,
		int32_t *l_f, int32_t *l_p, int32_t *l_tm, int32_t *l_ii, int64_t *l_t, hao_hit_t *l_a, uint32_t *l_cn, hao_chain_rec *l_rec)
{
	const int lane = hao_lane();
	const uint64_t g = e.g, gs = e.start; const int64_t a_n = e.n;
	if (STAGE && INLDS) { for (int64_t i = lane; i < a_n; i += 64) l_a[i] = ag[i]; HAO_WAVE_FENCE(); }
	const hao_hit_t *a = (STAGE && INLDS) ? l_a : ag;          // the DP re-reads predecessors many times: small groups keep their hits in LDS
	hao_cpar P; P.pen_gap = A.par.pen_gap; P.pen_skip = A.par.pen_skip; P.bw = A.par.bw; P.max_skip = A.par.max_skip; P.max_iter = A.par.max_iter; P.max_dis = A.par.max_dis;
	P.xl = e.xl; P.yl = e.yl;
	int32_t *f = INLDS ? l_f : A.f + gs, *p = INLDS ? l_p : A.p + gs, *tm = INLDS ? l_tm : A.tm + gs;
	int32_t *ii = INLDS ? l_ii : A.ii + gs; int64_t *t = INLDS ? l_t : A.t + gs;
	const uint32_t strand0 = HH_STRAND(a[0]);
	// ---- parallel quick check, storing f/p ----
	int32_t carry_f = 0; hao_hit_t carry_h = a[0];
	bool fail0 = false, fail1 = false; int32_t maxf0 = INT32_MIN, maxf1 = INT32_MIN, flast0 = 0, flast1 = 0; int64_t ddt0 = 0, ddt1 = 0, k1 = 0;
	hao_hit_t last0 = a[0], last1 = a[0];
	for (int64_t t0 = 0; t0 < a_n; t0 += 64) {
		const int64_t idx = t0 + lane; const bool act = idx < a_n;
		hao_hit_t h = act ? a[idx] : carry_h;
		hao_hit_t ph = hao_shfl_up_hit(h); if (lane == 0) ph = carry_h;
		const bool st = act && (idx == 0 || HH_STRAND(h) != HH_STRAND(ph));
		const int b = act && HH_STRAND(h) != strand0;
		int32_t s = HH_SPAN(h); bool ok = true; int64_t dd = 0;
		if (act && !st) { s = hao_pair_score(h, ph, P, &dd); ok = s != INT32_MIN; if (!ok) { s = 0; dd = 0; } }
		int32_t x = act ? s : 0; int fl = st;
		hao_seg_scan_add(x, fl);
		if (!fl) x += carry_f;
		const int32_t fv = x; const int32_t fp = hao_wave_shr1(fv, carry_f);
		const bool brk = act && !st && (!ok || (int64_t)s + fp < (int64_t)HH_SPAN(h));
		if (__ballot(brk && b == 0)) fail0 = true;
		if (__ballot(brk && b == 1)) fail1 = true;
		if (act) { if (b == 0) { maxf0 = max(maxf0, fv); ddt0 += dd; } else { maxf1 = max(maxf1, fv); ddt1 += dd; } f[idx] = fv; p[idx] = st ? -1 : (int32_t)(idx - 1); tm[idx] = -1; ii[idx] = 0; t[idx] = 0; }
		k1 += __popcll(__ballot(act && b == 0));
		const bool isend = act && (idx == a_n - 1 || HH_STRAND(a[idx + 1]) != HH_STRAND(h));
		unsigned long long m0 = __ballot(isend && b == 0), m1 = __ballot(isend && b == 1);
		if (m0) { int src = __ffsll((long long)m0) - 1; flast0 = hao_bcast(fv, src); last0 = hao_shfl_hit(h, src); }
		if (m1) { int src = __ffsll((long long)m1) - 1; flast1 = hao_bcast(fv, src); last1 = hao_shfl_hit(h, src); }
		carry_f = hao_bcast(fv, 63); carry_h = hao_shfl_hit(h, 63);
	}
	maxf0 = hao_wave_max_i32(maxf0); maxf1 = hao_wave_max_i32(maxf1); ddt0 = hao_wave_sum_i64(ddt0); ddt1 = hao_wave_sum_i64(ddt1);
	const bool two = k1 < a_n;
	const hao_hit_t first0 = a[0], first1 = two ? a[k1] : a[0];
	const bool acc0 = !fail0 && flast0 == maxf0 && !(k1 >= 2 && ddt0 > 16 && ddt0 > hao_band(last0, first0, P));
	const bool acc1 = two && !fail1 && flast1 == maxf1 && !(a_n - k1 >= 2 && ddt1 > 16 && ddt1 > hao_band(last1, first1, P));
	int64_t plus = 0, msc = INT32_MIN, msc_i = INT32_MIN, movl = INT32_MAX, si = 0, ei = a_n;
	if (acc0) {     // quick_ck_lchain bookkeeping for an accepted block (Hash_Table.cpp:2075-2088)
		msc = flast0; msc_i = k1 - 1; movl = hao_ext_len(last0.self_offset, last0.self_offset, P.xl, last0.offset, last0.offset, P.yl);
		if (ei > k1) si = k1; else ei = 0;
	}
	if (acc1) {
		if ((int64_t)flast1 >= msc) {
			int64_t ov = hao_ext_len(last1.self_offset, last1.self_offset, P.xl, last1.offset, last1.offset, P.yl);
			if ((int64_t)flast1 > msc || ov < movl) { msc = flast1; msc_i = a_n - 1; movl = ov; }
		}
		if (ei > a_n) si = a_n; else ei = k1;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	uint32_t n_spec_ok_out = 0, n_spec_fail_out = 0;
	// ---- DP over [si, ei) ----
	// The chain DP is sequential in i (about one LDS/FP64-latency-bound round per hit).  Most hits of a group that failed the quick
	// check still simply extend the chain through their predecessor, so the loop first SPECULATES on a tile of 64 hits: guess
	// f[i] = f[i-1] + score(i, i-1), p[i] = i-1, then let lane k replay the complete predecessor scan of hit i+k (scores, running
	// maximum, marks via p[], n_skip / max_skip break) against the guessed f/p of the tile.  If the scan of every hit before k
	// reproduces its guess, the state hit k sees is exact, so the longest prefix of lanes that reproduce their guess is exact
	// by induction and is committed at once; the first hit that does not is handled by the sequential round below.
	{
		int64_t i, st, max_ii = -1; int spec_wait = 0, spec_fail = 0; uint32_t n_spec_ok = 0, n_spec_fail = 0;
		const bool spec_on = A.dbg_seq != 4;
		for (i = st = si; i < ei; ++i) {
			if (spec_wait > 0) --spec_wait;
			else if (spec_on) {
				int64_t st2 = st; if (i - st2 > P.max_iter) st2 = i - P.max_iter;
				const hao_hit_t h0 = a[i]; const uint32_t str_i = HH_STRAND(h0);
				while (str_i != HH_STRAND(a[st2])) ++st2;
				// the "best recent predecessor" bookkeeping (max_ii) is a no-op along a rising chain: require that state at the tile's first hit
				if (i == st2 || max_ii == i - 1) {
					const int64_t ti = i + lane; const bool inb = ti < ei;
					const hao_hit_t hi = inb ? a[ti] : h0;
					bool cand = inb && HH_STRAND(hi) == str_i && ti - st2 <= P.max_iter;
					int32_t s1 = 0;
					if (cand && ti > st2) {
						const hao_hit_t hp = a[ti - 1];
						s1 = hao_pair_score(hi, hp, P, nullptr);
						cand = s1 != INT32_MIN && s1 > 0 && (int64_t)hi.self_offset <= P.max_dis + (int64_t)hp.self_offset;   // rising, within max_dis: max_ii follows the chain
					}
					const unsigned long long badm = __ballot(!cand);
					const int nc = badm ? __ffsll((long long)badm) - 1 : 64;       // lanes [0, nc) carry a guess
					int64_t fg = ti > st2 ? (int64_t)s1 : (int64_t)HH_SPAN(hi);
					if (lane == 0 && i > st2) fg += f[i - 1];
					if (lane >= nc) fg = 0;
#pragma unroll
					for (int d = 1; d < 64; d <<= 1) { const int64_t y = __shfl_up(fg, d); if (lane >= d) fg += y; }
					const int32_t fg32 = (int32_t)fg; const int64_t pg = ti > st2 ? ti - 1 : -1;
					if (lane < nc) { f[ti] = fg32; p[ti] = (int32_t)pg; }
					HAO_WAVE_FENCE();
					// replay of the predecessor scan of hit ti (Hash_Table.cpp:2131-2150) by lane k; marks of "t[p[j]] = i" kept as a bit per distance
					int64_t mf = HH_SPAN(hi), mj = -1; int nsk = 0; unsigned long long marks = 0; bool done = lane >= nc;
					for (int d = 1; d <= 64; ++d) {
						const int64_t j = ti - d; const bool in = !done && j >= st2;
						if (!__ballot(in)) break;
						if (in) {
							const hao_hit_t hj = a[j]; const int32_t sj = hao_pair_score(hi, hj, P, nullptr);
							if (sj != INT32_MIN) {
								const int64_t sc = (int64_t)sj + f[j]; const int32_t pj = p[j];
								if (sc > mf) { mf = sc; mj = j; if (nsk > 0) --nsk; }
								else if (marks >> (d - 1) & 1) { if (++nsk > (int)P.max_skip) done = true; }
								if (pj >= 0) { const int64_t bb = ti - 1 - pj; if (bb < 64) marks |= 1ULL << bb; }
							}
						}
					}
					const bool ok = lane < nc && (done || ti - 65 < st2) && mf == (int64_t)fg32 && mj == pg;     // not ok: scan unfinished after 64 predecessors, or a different result
					const unsigned long long nokm = __ballot(!ok);
					const int m = nokm ? __ffsll((long long)nokm) - 1 : 64;
					if (m > 0) {
						// commit hits [i, i+m): running best chain end (first maximum, then smallest extension length, then first) and the minimum score
						int32_t fmx = lane < m ? fg32 : INT32_MIN, fmn = lane < m ? fg32 : INT32_MAX;
#pragma unroll
						for (int d = 32; d >= 1; d >>= 1) { fmx = max(fmx, __shfl_xor(fmx, d)); fmn = min(fmn, __shfl_xor(fmn, d)); }
						if ((int64_t)fmx >= msc) {
							const int64_t ovl = hao_ext_len(hi.self_offset, hi.self_offset, P.xl, hi.offset, hi.offset, P.yl);
							int64_t key = (lane < m && fg32 == fmx) ? ovl * 64 + lane : INT64_MAX;
#pragma unroll
							for (int d = 32; d >= 1; d >>= 1) { const int64_t y = __shfl_xor(key, d); if (y < key) key = y; }
							if ((int64_t)fmx > msc || (key >> 6) < movl) { msc = fmx; msc_i = i + (key & 63); movl = key >> 6; }
						}
						if ((int64_t)fmn < plus) plus = fmn;
						max_ii = i + m - 1; st = st2; i += m - 1;
						if (m >= 8) spec_fail = 0;
						if (m < 64) spec_wait = 1;          // hit i+m did not reproduce its guess: it takes the sequential round
						n_spec_ok += m;
						continue;
					}
					if (spec_fail < 5) ++spec_fail;
					spec_wait = 1 << spec_fail; ++n_spec_fail;
				}
			}
			const hao_hit_t hi = a[i];
			int64_t max_f = HH_SPAN(hi), n_skip = 0, max_j = -1, end_j;
			if (i - st > P.max_iter) st = i - P.max_iter;
			while (HH_STRAND(hi) != HH_STRAND(a[st])) ++st;
			end_j = st - 1;
			for (int64_t jb = i - 1; jb >= st; jb -= 64) {
				const int64_t j = jb - lane; const bool actv = j >= st;
				int32_t s = INT32_MIN; hao_hit_t hj; int32_t pj = -1; int64_t sc = INT64_MIN;
				if (actv) { hj = a[j]; s = hao_pair_score(hi, hj, P, nullptr); }
				const bool valid = actv && s != INT32_MIN;
				if (valid) { sc = (int64_t)s + f[j]; pj = p[j]; if (pj >= 0) tm[pj] = (int32_t)i; }
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				const bool mark = valid && tm[j] == (int32_t)i;
				// candidates that improve on everything scanned before them = running records of sc, found from the lowest lane up
				const unsigned long long vm = __ballot(valid);
				unsigned long long imask = 0, rem = vm; int64_t cur = max_f;
				while (rem) {
					const unsigned long long gm = __ballot(valid && sc > cur) & rem;
					if (!gm) break;
					const int l = __ffsll((long long)gm) - 1;
					imask |= 1ULL << l; cur = hao_readlane_i64(sc, l); rem = l == 63 ? 0 : (vm & (~0ULL << (l + 1)));
				}
				// n_skip bookkeeping of the sequential scan (improve: n = max(n-1, 0); marked non-improving: ++n, stop when n > max_skip), on scalar masks
				const unsigned long long cmask = __ballot(mark) & ~imask;
				int brk_lane = -1; int64_t nsk = n_skip;
				if (cmask == 0 || imask == 0 || (63 - __clzll((long long)imask)) < (__ffsll((long long)cmask) - 1)) {   // all improvements precede all marks (the usual case)
					nsk -= __popcll(imask); if (nsk < 0) nsk = 0;
					const int64_t need = P.max_skip + 1 - nsk;
					if ((int64_t)__popcll(cmask) >= need) brk_lane = __ffsll((long long)__ballot((cmask >> lane & 1) && (int64_t)__popcll(cmask & ((1ULL << lane) - 1)) == need - 1)) - 1;   // the need-th mark
					else nsk += __popcll(cmask);
				} else {
					unsigned long long ev = imask | cmask;
					while (ev) {
						const int l = __ffsll((long long)ev) - 1; ev &= ev - 1;
						if (imask >> l & 1) { if (nsk > 0) --nsk; } else if (++nsk > P.max_skip) { brk_lane = l; break; }
					}
				}
				n_skip = nsk;
				if (brk_lane >= 0) imask &= (1ULL << brk_lane) - 1;
				if (imask) { const int src = 63 - __clzll((long long)imask); max_f = hao_readlane_i64(sc, src); max_j = jb - src; }
				if (brk_lane >= 0) { end_j = jb - brk_lane; break; }
			}
			if (max_ii < 0 || (int64_t)hi.self_offset > (int64_t)a[max_ii].self_offset + P.max_dis || HH_STRAND(hi) != HH_STRAND(a[max_ii])) {
				int32_t mx = INT32_MIN; max_ii = -1;     // rare: rebuild the "best recent predecessor" (uniform sequential scan)
				for (int64_t j = i - 1; j >= st && (int64_t)hi.self_offset <= P.max_dis + (int64_t)a[j].self_offset && HH_STRAND(hi) == HH_STRAND(a[j]); --j)
					if (mx < f[j]) { mx = f[j]; max_ii = j; }
			}
			if (max_ii >= 0 && max_ii < end_j && HH_STRAND(hi) == HH_STRAND(a[max_ii])) {
				int32_t tmp = hao_pair_score(hi, a[max_ii], P, nullptr);
				if (tmp != INT32_MIN && max_f < (int64_t)tmp + f[max_ii]) { max_f = (int64_t)tmp + f[max_ii]; max_j = max_ii; }
			}
			if (lane == 0) { f[i] = (int32_t)max_f; p[i] = (int32_t)max_j; }
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			if (max_ii < 0 || ((int64_t)hi.self_offset <= P.max_dis + (int64_t)a[max_ii].self_offset && HH_STRAND(hi) == HH_STRAND(a[max_ii]) && f[max_ii] < (int32_t)max_f)) max_ii = i;
			if ((int32_t)max_f >= msc) {
				int64_t ovl = hao_ext_len(hi.self_offset, hi.self_offset, P.xl, hi.offset, hi.offset, P.yl);
				if ((int32_t)max_f > msc || ovl < movl) { msc = (int32_t)max_f; msc_i = i; movl = ovl; }
			}
			if ((int32_t)max_f < plus) plus = (int32_t)max_f;
		}
		n_spec_ok_out = n_spec_ok; n_spec_fail_out = n_spec_fail;
	}
	if (A.dbg_stats && lane == 0) { atomicAdd(A.stats + HAO_NCLS + 1, (unsigned long long)n_spec_ok_out); atomicAdd(A.stats + HAO_NCLS + 2, (unsigned long long)n_spec_fail_out); atomicAdd(A.stats + HAO_NCLS + 3, (unsigned long long)(ei - si)); }
	if (A.dbg_seq == 3) { if (lane == 0) hao_chain_tail(A, g, a, a_n, P, f, p, t, ii, msc, msc_i, plus); return; }
	hao_chain_tail_wave(A, g, gs, a, a_n, P, f, p, t, ii, INLDS ? (int64_t)CAP : (int64_t)0, l_cn, l_rec, msc, msc_i, plus);
}

template<int CAP, bool STAGE>
__device__ __forceinline__ void hao_dp_kernel_body(const hao_chain_args &A, const hao_gent *list, const uint32_t *slow, const unsigned long long *slow_cnt)
{
	__shared__ int32_t l_f[CAP], l_p[CAP], l_tm[CAP], l_ii[CAP]; __shared__ int64_t l_t[CAP]; __shared__ hao_hit_t l_a[STAGE ? CAP : 1];
	__shared__ uint32_t l_cn[8]; __shared__ hao_chain_rec l_rec[HAO_MCOPY_MAX];
	const uint64_t n_slow = *slow_cnt;
	for (uint64_t b = blockIdx.x; b < n_slow; b += gridDim.x) {      // persistent waves: the launch does not know how many groups failed the quick check
		const hao_gent e = list[slow[b]];
		if (A.dbg_seq == 1) { if (hao_lane() == 0) hao_chain_generic(A, e.g); continue; }
		const hao_hit_t *ag = A.hits + e.start;
		if ((int64_t)e.n <= CAP) hao_dp_body<CAP, STAGE, true>(A, e, ag, l_f, l_p, l_tm, l_ii, l_t, l_a, l_cn, l_rec);
		else hao_dp_body<CAP, STAGE, false>(A, e, ag, l_f, l_p, l_tm, l_ii, l_t, l_a, l_cn, l_rec);
		HAO_WAVE_FENCE();
	}
}

template<int CAP, bool STAGE>
__global__ __launch_bounds__(64) void chain_dp_kernel(hao_chain_args A, const hao_gent *list, const uint32_t *slow, const unsigned long long *slow_cnt)
{ hao_dp_kernel_body<CAP, STAGE>(A, list, slow, slow_cnt); }
// groups of up to 128 hits: 5 KB of LDS per wave, so registers bound the occupancy - trade a few spills for twice the waves per SIMD
// (many small slow groups on repeat-rich reads; the DP is latency-bound)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void chain_dp128_kernel(hao_chain_args A, const hao_gent *list, const uint32_t *slow, const unsigned long long *slow_cnt)
{ hao_dp_kernel_body<128, true>(A, list, slow, slow_cnt); }

// ---------------------------------------------------------------------------------------
// Assembly: per group, materialise overlap records (creation order), chained hits (tagged with
// the overlap ordinal inside the read) and fake cigars.  One wave per group.
// ---------------------------------------------------------------------------------------
// Where a chain's hits live.  Chained hits are NOT copied when the records are assembled: > 99.9 % of the chains of clean data are contiguous runs of
// the sorted seed hits (in_place), the rest were compacted by the DP kernel into ohits.  cl->list (the hits tagged with the chain's ordinal, in chain
// order) exists only as this descriptor until somebody needs the bytes: the wire packer of the delivery path (hao_deliver.cuh) or
// chain_materialize_kernel (blocking fetch API, digests).  The ordinal tag of lchain_qdp_mcopy_fast's output (Hash_Table.cpp:2272-2281) travels as w0.
struct hao_cdesc {
	uint64_t src;      // index of the first hit; bit 63 set: in ohits, else in hits
	uint64_t dst;      // index of the chain's first hit in the batch's cl->list concatenation
	uint32_t n, w0;    // hits; readID field of every hit of the chain = ordinal << 0 | strand << 31
	uint32_t r, pad;   // query read of the chain (index inside the batch)
};
#define HAO_CD_OHITS (1ULL << 63)
__device__ __forceinline__ const hao_hit_t *hao_cd_src(const hao_cdesc &d, const hao_hit_t *hits, const hao_hit_t *ohits)
{ return ((d.src & HAO_CD_OHITS) ? ohits : hits) + (d.src & ~HAO_CD_OHITS); }

struct hao_asm_args {
	const uint64_t *g_start; const uint32_t *g_read; const uint8_t *g_cls; const uint64_t *g_off; uint64_t n_groups; uint64_t rid_lo;
	const hao_hit_t *ohits, *hits; const uint64_t *fcs; const hao_chain_rec *rec; const uint32_t *nch;
	const uint64_t *ch_base, *cl_base, *fc_base;   // exclusive scans over groups (chains, hits) and over chain slots (fake-cigar entries)
	hao_ovlp_t *ol; uint64_t *ol_fc_off; hao_cdesc *cd; uint64_t *fc;
};

// One LANE per group (every size class): the records, the chain descriptors and the short fake cigars of its <= 3 chains; cigars of more than 8
// entries (noisy reads) are copied by the whole wave, one chain after the other.
__global__ __launch_bounds__(256) void chain_assemble_kernel(hao_asm_args A)
{
	const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; const int lane = hao_lane();
	const uint32_t n = g < A.n_groups ? A.nch[g] : 0;
	uint32_t r = 0; uint64_t ord0 = 0, cl0 = 0, gs = 0, chb = 0, clb = 0; uint32_t yid = 0;
	if (n) {
		r = A.g_read[g]; const uint64_t g0 = A.g_off[r];
		chb = A.ch_base[g]; clb = A.cl_base[g]; ord0 = chb - A.ch_base[g0]; cl0 = A.cl_base[g0]; gs = A.g_start[g];
		yid = HH_ID(A.hits[gs]);                     // every seed hit of the group has the group's target id
	}
#pragma unroll
	for (uint32_t c = 0; c < HAO_MCOPY_MAX; ++c) {
		const bool has = c < n; uint32_t fl = 0; uint64_t fd = 0; const uint64_t *fs = nullptr;
		if (has) {
			const hao_chain_rec rc = A.rec[g * HAO_MCOPY_MAX + c];
			const uint64_t oi = chb + c, hd = clb + rc.hit_rel; const uint32_t ord = (uint32_t)(ord0 + c);
			fd = A.fc_base[g * HAO_MCOPY_MAX + c]; fl = rc.fc_len; fs = A.fcs + gs + 6 * g + rc.fc_rel;
			hao_ovlp_t o;
			o.x_id = (uint32_t)(A.rid_lo + r); o.x_pos_s = rc.x_pos_s; o.x_pos_e = rc.x_pos_e; o.x_pos_strand = 0;
			o.y_id = yid; o.y_pos_s = rc.y_pos_s; o.y_pos_e = rc.y_pos_e; o.y_pos_strand = rc.strand;
			o.shared_seed = rc.score; o.align_length = rc.n_hits; o.non_homopolymer_errors = (uint32_t)(hd - cl0); o.fc_len = rc.fc_len;
			A.ol[oi] = o; A.ol_fc_off[oi] = fd;
			hao_cdesc d; d.src = (gs + rc.src_rel) | (rc.in_place ? 0 : HAO_CD_OHITS); d.dst = hd; d.n = rc.n_hits; d.w0 = (rc.strand << 31) | (ord & 0x7fffffffu); d.r = r; d.pad = (rc.in_place && A.g_cls[g] >= 1) ? 1u : 0u;      // pad bit 0: the hits' wire codes exist (chain_group_kernel saw the group)
			A.cd[oi] = d;
			if (fl <= 8) for (uint32_t i = 0; i < fl; ++i) A.fc[fd + i] = fs[i];      // (all entries requested before the first store, as chain_final_kernel does: 124 registers, 4 waves per SIMD, 0.3 ms per pass slower - the kernel is bound by its 48-byte records)
		}
		for (unsigned long long big = __ballot(has && fl > 8); big; big &= big - 1) {
			const int l = __ffsll((long long)big) - 1;
			const uint64_t *cfs = (const uint64_t*)hao_readlane_i64((int64_t)fs, l); const uint64_t cfd = (uint64_t)hao_readlane_i64((int64_t)fd, l); const uint32_t cfl = hao_bcast(fl, l);
			for (uint32_t j = lane; j < cfl; j += 64) A.fc[cfd + j] = cfs[j];
		}
	}
}

// cl->list of the batch as plain tagged k_mer_hits (blocking fetch API, digests): one wave per chain, streamed once in, once out (non-temporal)
__global__ __launch_bounds__(256) void chain_materialize_kernel(const hao_cdesc *cd, uint64_t n_chains, const hao_hit_t *hits, const hao_hit_t *ohits, hao_hit_t *cl)
{
	const uint64_t ci = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (ci >= n_chains) return;
	const hao_cdesc d = cd[ci];
	typedef uint32_t hao_u32x4 __attribute__((ext_vector_type(4)));
	const hao_u32x4 *src4 = (const hao_u32x4*)hao_cd_src(d, hits, ohits); hao_u32x4 *dst4 = (hao_u32x4*)(cl + d.dst);
	for (uint32_t i = hao_lane(); i < d.n; i += 256) {       // four 16-byte loads in flight per lane
		hao_u32x4 h4[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) if (i + u * 64 < d.n) h4[u] = __builtin_nontemporal_load(src4 + i + u * 64);
#pragma unroll
		for (int u = 0; u < 4; ++u) if (i + u * 64 < d.n) { h4[u].x = d.w0; __builtin_nontemporal_store(h4[u], dst4 + i + u * 64); }
	}
}

// ---------------------------------------------------------------------------------------
// K9: per-read chain selection on an index permutation (records never move; swaps of whole
// structs in the reference = swaps of permutation entries here).  One wave per read: all
// lanes stage the sort keys (query interval, score, hit count) into LDS, lane 0 replays the
// order-dependent sequential algorithm (klib introsort tie order is observable) on LDS-resident
// keys, all lanes write the permutation back.  Reads with more chains than the LDS slice holds
// use global scratch for the keys.
// ---------------------------------------------------------------------------------------
#define HAO_SEL_CAP 512
struct hao_sel_ctx { const uint64_t *xs; const int32_t *sc; const uint32_t *al; uint32_t *pm; int32_t *stack; uint32_t *pm2, *lpos, *rasc; };
__device__ __forceinline__ void hao_sw(const hao_sel_ctx &S, int64_t i, int64_t j) { uint32_t t = S.pm[i]; S.pm[i] = S.pm[j]; S.pm[j] = t; }
template<int MODE> __device__ __forceinline__ bool hao_lt(const hao_sel_ctx &S, int64_t i, int64_t j)
{	// MODE 0: oreg_ss_lt (score, descending; anchor.cpp:35)   MODE 1: oreg_xs_lt ((x_pos_s, x_pos_e) ascending; anchor.cpp:32)
	return MODE == 0 ? S.sc[S.pm[i]] > S.sc[S.pm[j]] : S.xs[S.pm[i]] < S.xs[S.pm[j]];
}

template<int MODE> __device__ void hao_ins_sort(const hao_sel_ctx &S, int64_t lo, int64_t hi)
{ for (int64_t i = lo + 1; i < hi; ++i) for (int64_t j = i; j > lo && hao_lt<MODE>(S, j, j - 1); --j) hao_sw(S, j, j - 1); }

template<int MODE> __device__ void hao_comb_sort(const hao_sel_ctx &S, int64_t lo, int64_t n)
{
	const double shrink = 1.2473309501039786540366528676643; int64_t gap = n; bool swapped;
	do {
		if (gap > 2) { gap = (int64_t)((double)gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (int64_t i = lo; i < lo + n - gap; ++i) if (hao_lt<MODE>(S, i + gap, i)) { hao_sw(S, i, i + gap); swapped = true; }
	} while (swapped || gap > 2);
	if (gap != 1) hao_ins_sort<MODE>(S, lo, lo + n);
}

// klib introsort (ksort.h:110-160) restated over positions; the pivot sits at position t during partitioning
template<int MODE> __device__ void hao_intro_sort(const hao_sel_ctx &S, int64_t n)
{
	int32_t *stack = S.stack; int64_t top = 0, s, t, i, j, k; int d;
	if (n < 1) return;
	if (n == 2) { if (hao_lt<MODE>(S, 1, 0)) hao_sw(S, 0, 1); return; }
	for (d = 2; (1ull << d) < (uint64_t)n; ++d) {}
	s = 0; t = n - 1; d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { hao_comb_sort<MODE>(S, s, t - s + 1); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (hao_lt<MODE>(S, k, i)) { if (hao_lt<MODE>(S, k, j)) k = j; }
			else k = hao_lt<MODE>(S, j, i) ? i : j;
			if (k != t) hao_sw(S, k, t);
			for (;;) {
				do ++i; while (hao_lt<MODE>(S, i, t));
				do --j; while (i <= j && hao_lt<MODE>(S, t, j));
				if (j <= i) break;
				hao_sw(S, i, j);
			}
			hao_sw(S, i, t);
			if (i - s > t - i) {
				if (i - s > 16) { stack[top++] = (int32_t)s; stack[top++] = (int32_t)(i - 1); stack[top++] = d; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top++] = (int32_t)(i + 1); stack[top++] = (int32_t)t; stack[top++] = d; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { hao_ins_sort<MODE>(S, 0, n); return; }
			d = (int)stack[--top]; t = stack[--top]; s = stack[--top];
		}
	}
}


// ---------------------------------------------------------------------------------------
// The same klib introsort, executed by a whole wave with identical results.
//  * control (range stack, depth budget, median of three) is uniform scalar work;
//  * the Hoare partition is data-parallel: the left pointer stops at positions whose key is not
//    below the pivot, the right pointer at positions whose key is not above it; stop number k from
//    the left swaps with stop number k from the right while they have not crossed, so both stop
//    lists come from two ballots per 64-position tile and all swaps of one partition happen at
//    once; the final pivot slot is min(left stop K+1, right stop K) (K = number of swaps);
//  * ranges of <= 16 elements are left alone by klib and finished by ONE insertion sort over the
//    whole array; insertion sort is stable, i.e. the finish is the stable sort of the array as the
//    partitions left it = a bitonic sort of (key, position) pairs;
//  * the combsort fallback (depth budget exhausted) stays sequential on lane 0.
// ---------------------------------------------------------------------------------------
template<int MODE> __device__ __forceinline__ uint64_t hao_skey(const hao_sel_ctx &S, uint32_t idx)
{	// order-preserving 64-bit key: MODE 0 = score descending, MODE 1 = (x_pos_s, x_pos_e) ascending
	return MODE == 0 ? (uint64_t)((int64_t)INT32_MAX - (int64_t)S.sc[idx]) : S.xs[idx];
}

#define HAO_WFENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

template<int MODE> __device__ void hao_wave_intro_sort(const hao_sel_ctx &S, int64_t n)
{
	const int lane = hao_lane(); int32_t *stack = S.stack; int64_t top = 0, s, t, i, j, k; int d;
	if (n < 1) return;
	if (n == 2) { if (lane == 0 && hao_lt<MODE>(S, 1, 0)) hao_sw(S, 0, 1); HAO_WFENCE(); return; }
	if (n <= 64) {
		// One key per lane.  Whatever klib's partitions do, its closing insertion sort leaves the array sorted by key, and the order of EQUAL keys is all that depends on the
		// path it took: when no two keys are equal the answer is the sorted order, and a lane's slot is the number of smaller keys - n rounds of two v_readlane and a compare
		// instead of the replay's dozens of fenced LDS round trips (a HiFi read's ~60 chains: the common case of the position sort).  A tie sends the read through the replay.
		const bool act = lane < (int)n; const uint32_t me = act ? S.pm[lane] : 0u; const uint64_t key = act ? hao_skey<MODE>(S, me) : 0ULL;
		uint32_t rank = 0; bool tie = false;
		for (int jj = 0; jj < (int)n; ++jj) { const uint64_t kj = (uint64_t)hao_readlane_i64((int64_t)key, jj); rank += kj < key ? 1u : 0u; tie = tie || (kj == key && jj != lane); }
		if (!__any(act && tie)) { HAO_WFENCE(); if (act) S.pm[rank] = me; HAO_WFENCE(); return; }
	}
	for (d = 2; (1ull << d) < (uint64_t)n; ++d) {}
	s = 0; t = n - 1; d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { if (lane == 0) hao_comb_sort<MODE>(S, s, t - s + 1); HAO_WFENCE(); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (hao_lt<MODE>(S, k, i)) { if (hao_lt<MODE>(S, k, j)) k = j; }
			else k = hao_lt<MODE>(S, j, i) ? i : j;
			HAO_LOCKSTEP();      // every lane has compared the three pivot candidates
			if (k != t) { if (lane == 0) hao_sw(S, k, t); HAO_WFENCE(); }
			const uint64_t rp = hao_skey<MODE>(S, S.pm[t]);
			uint32_t nL = 0, nR = 0;
			for (int64_t p0 = s + 1; p0 <= t; p0 += 64) {
				const int64_t p = p0 + lane; const bool act = p <= t;
				const uint64_t key = act ? hao_skey<MODE>(S, S.pm[p]) : 0;
				const bool Lf = act && !(key < rp), Rf = act && p < t && !(rp < key);
				const unsigned long long bl = __ballot(Lf), br = __ballot(Rf), lt_ = (1ULL << lane) - 1;
				if (Lf) S.lpos[nL + __popcll(bl & lt_)] = (uint32_t)p;
				if (Rf) S.rasc[nR + __popcll(br & lt_)] = (uint32_t)p;
				nL += __popcll(bl); nR += __popcll(br);
			}
			HAO_WFENCE();
			const uint32_t m = nL < nR ? nL : nR; uint32_t K = 0;       // swaps: left stop k with right stop k while left < right
			for (uint32_t k0 = 0; k0 < m; k0 += 64) {
				const uint32_t kk = k0 + lane; const bool pr = kk < m && S.lpos[kk] < S.rasc[nR - 1 - kk];
				const unsigned long long b = __ballot(pr); const int c = __popcll(b);
				K += c; if (c < 64) break;
			}
			for (uint32_t k0 = 0; k0 < K; k0 += 64) { const uint32_t kk = k0 + lane; if (kk < K) hao_sw(S, S.lpos[kk], S.rasc[nR - 1 - kk]); }
			HAO_WFENCE();
			i = K == 0 ? S.lpos[0] : (S.lpos[K] < S.rasc[nR - K] ? S.lpos[K] : S.rasc[nR - K]);
			if (lane == 0) hao_sw(S, i, t);
			HAO_WFENCE();
			if (i - s > t - i) {
				if (i - s > 16) { if (lane == 0) { stack[top] = (int32_t)s; stack[top + 1] = (int32_t)(i - 1); stack[top + 2] = d; } top += 3; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (lane == 0) { stack[top] = (int32_t)(i + 1); stack[top + 1] = (int32_t)t; stack[top + 2] = d; } top += 3; }
				t = i - s > 16 ? i - 1 : s;
			}
			HAO_WFENCE();
		} else {
			if (top == 0) break;
			d = stack[--top]; t = stack[--top]; s = stack[--top];
		}
	}
	// klib finishes with ONE insertion sort over the whole array.  Insertion sort is stable, so its result is THE stable sort
	// of the current array (elements can travel far: klib's partition never examines a[s], which may be out of place).
	// A stable sort is any sort of (key, current position): bitonic network over the next power of two, padding = +inf.
	uint32_t *A_ = S.pm2, *P_ = S.lpos;                      // element = (record index, position before the sort)
	{	// cheap attempt first: if nothing has to move further than 16 slots, final slot = position - (#larger among the 16 before)
		// + (#smaller among the 16 after).  Accepted only when the result is a permutation that is sorted by (key, old position):
		// that IS the stable sort.  Otherwise (an out-of-place a[s]) fall through to the general network.
		for (int64_t p = lane; p < n; p += 64) A_[p] = 0xffffffffu;
		HAO_WFENCE();
		for (int64_t p = lane; p < n; p += 64) {
			const uint32_t me = S.pm[p]; const uint64_t key = hao_skey<MODE>(S, me); int64_t dst = p;
			for (int64_t q = p - 16 < 0 ? 0 : p - 16; q < p; ++q) if (hao_skey<MODE>(S, S.pm[q]) > key) --dst;
			for (int64_t q = p + 1; q <= p + 16 && q < n; ++q) if (hao_skey<MODE>(S, S.pm[q]) < key) ++dst;
			if (dst >= 0 && dst < n) { A_[dst] = me; P_[dst] = (uint32_t)p; }
		}
		HAO_WFENCE();
		int bad = 0;
		for (int64_t p = lane; p < n; p += 64) {
			if (A_[p] == 0xffffffffu) bad = 1;
			else if (p > 0 && A_[p - 1] != 0xffffffffu) {
				const uint64_t ka = hao_skey<MODE>(S, A_[p - 1]), kb = hao_skey<MODE>(S, A_[p]);
				if (ka > kb || (ka == kb && P_[p - 1] > P_[p])) bad = 1;
			}
		}
		if (!__any(bad)) {
			HAO_WFENCE();
			for (int64_t p = lane; p < n; p += 64) S.pm[p] = A_[p];
			HAO_WFENCE();
			return;
		}
		HAO_WFENCE();
	}
	uint32_t n2 = 1; while (n2 < (uint32_t)n) n2 <<= 1;
	for (uint32_t p = lane; p < n2; p += 64) { A_[p] = p < (uint32_t)n ? S.pm[p] : 0xffffffffu; P_[p] = p; }
	HAO_WFENCE();
	for (uint32_t kk = 2; kk <= n2; kk <<= 1) {
		for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
			for (uint32_t i0 = 0; i0 < n2; i0 += 64) {
				const uint32_t a = i0 + lane, b = a ^ jj;
				if (a < n2 && b > a) {
					const uint32_t ia = A_[a], ib = A_[b], pa = P_[a], pb = P_[b];
					const bool infa = ia == 0xffffffffu, infb = ib == 0xffffffffu;
					const uint64_t ka = infa ? 0 : hao_skey<MODE>(S, ia), kb = infb ? 0 : hao_skey<MODE>(S, ib);
					const bool a_gt_b = infa ? (!infb || pa > pb) : (infb ? false : (ka > kb || (ka == kb && pa > pb)));
					const bool up = (a & kk) == 0;
					if (a_gt_b == up) { A_[a] = ib; A_[b] = ia; P_[a] = pb; P_[b] = pa; }
				}
			}
			HAO_WFENCE();
		}
	}
	for (int64_t p = lane; p < n; p += 64) S.pm[p] = A_[p];
	HAO_WFENCE();
}

// The same sort by a workgroup of NW waves.  Sub-ranges of the quicksort phase are independent (klib's explicit stack only fixes the order
// in which they are visited, and the depth budget d travels with each sub-range), so the phase runs level by level: every wave partitions
// the sub-ranges of the current level assigned to it (stop lists of sub-range [s, t] live at lpos/rasc[s ..]) and appends the children
// that klib would still partition (> 16 elements) to the next level's list.  The stable finish is data-parallel over all threads.
#define HAO_BSORT_MAXSEG 64       // > CAP / 18 sub-ranges can never be alive in one level (CAP <= 1024)
template<int MODE, int NW> __device__ void hao_block_intro_sort(const hao_sel_ctx &S, int64_t n, int32_t *segs /*[2][3 * MAXSEG]*/, uint32_t *segn /*[2]*/, int *flag)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NT = NW * 64;
	if (n < 1) return;
	if (n == 2) { if (tid == 0 && hao_lt<MODE>(S, 1, 0)) hao_sw(S, 0, 1); __syncthreads(); return; }
	if (tid == 0) { int d0; for (d0 = 2; (1ull << d0) < (uint64_t)n; ++d0) {} segs[0] = 0; segs[1] = (int32_t)(n - 1); segs[2] = d0 << 1; segn[0] = 1; segn[1] = 0; }
	__syncthreads();
	for (int ci = 0; ; ci ^= 1) {
		const uint32_t nseg = segn[ci];
		if (nseg == 0) break;
		const int32_t *cur = segs + ci * 3 * HAO_BSORT_MAXSEG; int32_t *nxt = segs + (ci ^ 1) * 3 * HAO_BSORT_MAXSEG;
		for (uint32_t e = wv; e < nseg; e += NW) {
			const int64_t s = cur[3 * e], t = cur[3 * e + 1]; int d = cur[3 * e + 2];
			int64_t i, j, k;
			if (--d == 0) { if (lane == 0) hao_comb_sort<MODE>(S, s, t - s + 1); HAO_WFENCE(); continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (hao_lt<MODE>(S, k, i)) { if (hao_lt<MODE>(S, k, j)) k = j; }
			else k = hao_lt<MODE>(S, j, i) ? i : j;
			HAO_LOCKSTEP();      // every lane has compared the three pivot candidates
			if (k != t) { if (lane == 0) hao_sw(S, k, t); HAO_WFENCE(); }
			const uint64_t rp = hao_skey<MODE>(S, S.pm[t]);
			uint32_t *lpos = S.lpos + s, *rasc = S.rasc + s;
			uint32_t nL = 0, nR = 0;
			for (int64_t p0 = s + 1; p0 <= t; p0 += 64) {
				const int64_t p = p0 + lane; const bool act = p <= t;
				const uint64_t key = act ? hao_skey<MODE>(S, S.pm[p]) : 0;
				const bool Lf = act && !(key < rp), Rf = act && p < t && !(rp < key);
				const unsigned long long bl = __ballot(Lf), br = __ballot(Rf), lt_ = (1ULL << lane) - 1;
				if (Lf) lpos[nL + __popcll(bl & lt_)] = (uint32_t)p;
				if (Rf) rasc[nR + __popcll(br & lt_)] = (uint32_t)p;
				nL += __popcll(bl); nR += __popcll(br);
			}
			HAO_WFENCE();
			const uint32_t m = nL < nR ? nL : nR; uint32_t K = 0;
			for (uint32_t k0 = 0; k0 < m; k0 += 64) {
				const uint32_t kk = k0 + lane; const bool pr = kk < m && lpos[kk] < rasc[nR - 1 - kk];
				const unsigned long long b = __ballot(pr); const int c = __popcll(b);
				K += c; if (c < 64) break;
			}
			for (uint32_t k0 = 0; k0 < K; k0 += 64) { const uint32_t kk = k0 + lane; if (kk < K) hao_sw(S, lpos[kk], rasc[nR - 1 - kk]); }
			HAO_WFENCE();
			i = K == 0 ? lpos[0] : (lpos[K] < rasc[nR - K] ? lpos[K] : rasc[nR - K]);
			if (lane == 0) {
				hao_sw(S, i, t);
				if (i - s > 16) { const uint32_t q = atomicAdd(&segn[ci ^ 1], 1u); nxt[3 * q] = (int32_t)s; nxt[3 * q + 1] = (int32_t)(i - 1); nxt[3 * q + 2] = d; }
				if (t - i > 16) { const uint32_t q = atomicAdd(&segn[ci ^ 1], 1u); nxt[3 * q] = (int32_t)(i + 1); nxt[3 * q + 1] = (int32_t)t; nxt[3 * q + 2] = d; }
			}
			HAO_WFENCE();
		}
		__syncthreads();
		if (tid == 0) segn[ci] = 0;
		__syncthreads();
	}
	// stable finish = klib's final insertion sort (see hao_wave_intro_sort), all threads
	uint32_t *A_ = S.pm2, *P_ = S.lpos;
	for (int64_t p = tid; p < n; p += NT) A_[p] = 0xffffffffu;
	if (tid == 0) *flag = 0;
	__syncthreads();
	for (int64_t p = tid; p < n; p += NT) {
		const uint32_t me = S.pm[p]; const uint64_t key = hao_skey<MODE>(S, me); int64_t dst = p;
		for (int64_t q = p - 16 < 0 ? 0 : p - 16; q < p; ++q) if (hao_skey<MODE>(S, S.pm[q]) > key) --dst;
		for (int64_t q = p + 1; q <= p + 16 && q < n; ++q) if (hao_skey<MODE>(S, S.pm[q]) < key) ++dst;
		if (dst >= 0 && dst < n) { A_[dst] = me; P_[dst] = (uint32_t)p; }      // (colliding writes leave a hole somewhere else: caught below)
	}
	__syncthreads();
	{
		int bad = 0;
		for (int64_t p = tid; p < n; p += NT) {
			if (A_[p] == 0xffffffffu) bad = 1;
			else if (p > 0 && A_[p - 1] != 0xffffffffu) {
				const uint64_t ka = hao_skey<MODE>(S, A_[p - 1]), kb = hao_skey<MODE>(S, A_[p]);
				if (ka > kb || (ka == kb && P_[p - 1] > P_[p])) bad = 1;
			}
		}
		if (bad) *flag = 1;
	}
	__syncthreads();
	if (!*flag) {
		for (int64_t p = tid; p < n; p += NT) S.pm[p] = A_[p];
		__syncthreads();
		return;
	}
	__syncthreads();
	uint32_t n2 = 1; while (n2 < (uint32_t)n) n2 <<= 1;
	for (uint32_t p = tid; p < n2; p += NT) { A_[p] = p < (uint32_t)n ? S.pm[p] : 0xffffffffu; P_[p] = p; }
	__syncthreads();
	for (uint32_t kk = 2; kk <= n2; kk <<= 1) {
		for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
			for (uint32_t a = tid; a < n2; a += NT) {
				const uint32_t b = a ^ jj;
				if (b > a) {
					const uint32_t ia = A_[a], ib = A_[b], pa = P_[a], pb = P_[b];
					const bool infa = ia == 0xffffffffu, infb = ib == 0xffffffffu;
					const uint64_t ka = infa ? 0 : hao_skey<MODE>(S, ia), kb = infb ? 0 : hao_skey<MODE>(S, ib);
					const bool a_gt_b = infa ? (!infb || pa > pb) : (infb ? false : (ka > kb || (ka == kb && pa > pb)));
					const bool up = (a & kk) == 0;
					if (a_gt_b == up) { A_[a] = ib; A_[b] = ia; P_[a] = pb; P_[b] = pa; }
				}
			}
			__syncthreads();
		}
	}
	for (int64_t p = tid; p < n; p += NT) S.pm[p] = A_[p];
	__syncthreads();
}

__device__ __forceinline__ int hao_ov_type(uint64_t xs, uint32_t len)       // ha_ov_type, anchor.cpp:86-91
{
	const uint32_t x_pos_s = (uint32_t)(xs >> 32), x_pos_e = (uint32_t)xs;
	if (x_pos_s == 0 && x_pos_e == len - 1) return 2;
	if (x_pos_s > 0 && x_pos_e < len - 1) return 3;
	return x_pos_s == 0 ? 0 : 1;
}

__device__ void hao_cov_add(uint64_t *cc, uint64_t cwn, uint64_t ocv_w, uint64_t rl, uint64_t rs, uint64_t re)
{
	uint64_t m = rs / ocv_w, cws = m * ocv_w;
	for (; m < cwn; ++m, cws += ocv_w) {
		uint64_t cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
		uint64_t os = rs >= cws ? rs : cws, oe = re <= cwe ? re : cwe;
		if (oe <= os) break;
		if ((uint32_t)cc[m] + (oe - os) < UINT32_MAX) cc[m] += oe - os;
		else { cc[m] >>= 32; cc[m] <<= 32; cc[m] |= UINT32_MAX; }
	}
}

struct hao_sel_args {
	const hao_ovlp_t *ol; const uint64_t *g_off; const uint64_t *ch_base; const uint64_t *cl_base; const hao_cdesc *cd; const hao_hit_t *hits, *ohits;
	uint64_t n_sel, rid_lo; const uint32_t *len; const uint64_t *cc_off; uint64_t *cc;
	uint64_t *key_xs; int32_t *key_sc; uint32_t *key_al, *key_tmp;   // global key scratch (reads with more chains than the LDS slice holds); key_tmp: 5 words per chain
	uint32_t *perm; uint32_t *n_final; uint64_t *fc_final;        // outputs: permutation (per read slice), kept count, kept fake-cigar entries
	uint64_t max_n_chain, ocv_w; uint32_t chain_cutoff;
	int dbg_seq_prune;            // HAO_DBG_FORCE=seq_prune: the one-lane pruning scan
	unsigned long long *dbg;      // optional phase timers (HAO_DBG_PRINT=sel): wall-clock ticks summed over reads: score sort, prune, position sort, weak filter, reads
};

// the sequential part (lane 0). returns the kept count
// max_n_chain pruning (anchor.cpp:1957-2056) on the score-sorted permutation: sequential, lane 0
#define HAO_WEAK_LANEWISE_N 128   // weak-chain filter: reads with more chains than this keep the wave's sweep for every batch
#define HAO_WEAK_SWEEP_MAX 6   // weak-chain filter: up to this many candidate chains of a 64-chain batch are swept by the whole wave, more are searched one per lane
#define HAO_SEL_CCAP 128       // coverage windows (read length / ocv_w) kept in LDS during the pruning scan; longer reads use the global array
template<bool CCLDS>
__device__ int64_t hao_select_prune(const hao_sel_args &A, const hao_sel_ctx &S, int64_t n, int lch, uint64_t r, int *lch_out, uint64_t *l_cc)
{
	const uint64_t rl = A.len[A.rid_lo + r], max_n_chain = A.max_n_chain, ocv_w = A.ocv_w; const uint32_t chain_cutoff = A.chain_cutoff;
	int64_t i;
#define XS(i) S.xs[S.pm[i]]
#define SC(i) S.sc[S.pm[i]]
#define AL(i) S.al[S.pm[i]]
	if ((uint64_t)n > max_n_chain) {
		int32_t w, nn[4] = {0, 0, 0, 0}, s[4] = {0, 0, 0, 0}; uint64_t cwn = 0, *cc = CCLDS ? l_cc : A.cc + A.cc_off[r], kk, mm;      // the scan is sequential (one lane): window counters in LDS, not a global round trip each
		for (i = 0; i < n; ++i) { w = hao_ov_type(XS(i), (uint32_t)rl); if ((uint64_t)++nn[w] == max_n_chain) s[w] = SC(i); }
		if (s[0] > 0 || s[1] > 0 || s[2] > 0 || s[3] > 0) {
			if ((uint64_t)nn[3] >= max_n_chain && rl >= ocv_w) {
				uint64_t cws = 0, cwe;
				cwn = rl / ocv_w + (rl % ocv_w ? 1 : 0);
				for (mm = 0; mm < cwn; ++mm, cws += ocv_w) {
					cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
					cc[mm] = (cwe - cws) * (max_n_chain >> 1); if (cc[mm] > UINT32_MAX) cc[mm] = UINT32_MAX; cc[mm] <<= 32;
				}
			}
			for (i = 0, kk = 0, lch = 0; i < n; ++i) {
				const uint64_t xs = XS(i); bool keep = false;
				const uint64_t rs = xs >> 32, re = (uint64_t)(uint32_t)xs + 1;
				w = hao_ov_type(xs, (uint32_t)rl);
				if (SC(i) >= s[w]) { if (cwn) hao_cov_add(cc, cwn, ocv_w, rl, rs, re); keep = true; }
				else if (w == 3 && cwn > 0) {
					uint64_t cw0 = 0, cw1 = 0, cws, cwe, os, oe;
					for (mm = rs / ocv_w, cws = mm * ocv_w; mm < cwn; ++mm, cws += ocv_w) {
						cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
						os = rs >= cws ? rs : cws; oe = re <= cwe ? re : cwe;
						if (oe <= os) break;
						if ((oe - os) + (uint64_t)(uint32_t)cc[mm] >= (cc[mm] >> 32)) cw1 += oe - os; else cw0 += oe - os;
					}
					if ((double)cw0 >= (double)(cw0 + cw1) * 0.7) { hao_cov_add(cc, cwn, ocv_w, rl, rs, re); keep = true; }
				}
				if (keep) {
					if (kk != (uint64_t)i) hao_sw(S, (int64_t)kk, i);
					if (AL(kk) < chain_cutoff) lch = 1;
					++kk;
				}
			}
			n = (int64_t)kk;
		}
	}
	*lch_out = lch;
#undef XS
#undef SC
#undef AL
	return n;
}


// The same pruning by the whole wave, bit-exact:
//   * the per-type thresholds are a counting scan (ballots);
//   * a chain at or above its threshold is kept unconditionally and adds its coverage; the additions commute (no counter can saturate:
//     n * ocv_w < 2^32, checked by the caller), so a tile's unconditional chains add with LDS / global atomics;
//   * a contained chain below the threshold is rescued when >= 70 % of its length lies in windows that are not full yet.  More coverage
//     can only turn windows full, so a chain that fails the test against the coverage at the START of its tile fails it against the exact
//     coverage too: all 64 candidates of a tile are tested in parallel against that lower bound, and only the (rare, once the windows
//     fill up) survivors are replayed one by one, in order, with the exact coverage.
template<bool CCLDS>
__device__ int64_t hao_select_prune_wave(const hao_sel_args &A, const hao_sel_ctx &S, int64_t n, int lch, uint64_t r, int *lch_out, uint64_t *l_cc)
{
	const uint64_t rl = A.len[A.rid_lo + r], max_n_chain = A.max_n_chain, ocv_w = A.ocv_w; const uint32_t chain_cutoff = A.chain_cutoff;
	const int lane = hao_lane(); const unsigned long long ltm = (1ULL << lane) - 1;
	*lch_out = lch;
	if ((uint64_t)n <= max_n_chain) return n;
	unsigned long long *cc = (unsigned long long*)(CCLDS ? l_cc : A.cc + A.cc_off[r]);
	// thresholds: score of the max_n_chain-th chain of each overlap type, in score order
	int32_t s[4] = {0, 0, 0, 0}; uint64_t nn[4] = {0, 0, 0, 0};
	for (int64_t b = 0; b < n; b += 64) {
		const int64_t i = b + lane; const bool act = i < n;
		const uint32_t me = act ? S.pm[i] : 0; const int w = act ? hao_ov_type(S.xs[me], (uint32_t)rl) : -1; const int32_t sc = act ? S.sc[me] : 0;
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const unsigned long long m = __ballot(w == t); const uint64_t c = __popcll(m);
			if (nn[t] < max_n_chain && nn[t] + c >= max_n_chain) {
				const uint64_t need = max_n_chain - nn[t];       // the need-th set bit of m
				const unsigned long long hit = __ballot(w == t && (uint64_t)__popcll(m & ltm) + 1 == need);
				s[t] = hao_bcast(sc, __ffsll((long long)hit) - 1);
			}
			nn[t] += c;
		}
	}
	if (!(s[0] > 0 || s[1] > 0 || s[2] > 0 || s[3] > 0)) return n;
	uint64_t cwn = 0;
	if (nn[3] >= max_n_chain && rl >= ocv_w) {
		cwn = rl / ocv_w + (rl % ocv_w ? 1 : 0);
		for (uint64_t mm = lane; mm < cwn; mm += 64) {
			const uint64_t cws = mm * ocv_w; uint64_t cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
			uint64_t v = (cwe - cws) * (max_n_chain >> 1); if (v > UINT32_MAX) v = UINT32_MAX;
			cc[mm] = v << 32;
		}
		HAO_WFENCE();
	}
	uint32_t *out = S.pm2; int64_t kk = 0; int lch2 = 0;
	for (int64_t b = 0; b < n; b += 64) {
		const int64_t i = b + lane; const bool act = i < n;
		const uint32_t me = act ? S.pm[i] : 0; const uint64_t xs = act ? S.xs[me] : 0;
		const uint64_t rs = xs >> 32, re = (uint64_t)(uint32_t)xs + 1;
		const int w = act ? hao_ov_type(xs, (uint32_t)rl) : 0;
		const bool unc = act && S.sc[me] >= s[w];
		bool maybe = false;
		auto rescue_test = [&]() -> bool {         // >= 70 % of the chain in windows that it does not fill (anchor.cpp:2016-2032)
			uint64_t cw0 = 0, cw1 = 0, mm = rs / ocv_w, cws = mm * ocv_w;
			for (; mm < cwn; ++mm, cws += ocv_w) {
				uint64_t cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
				const uint64_t os = rs >= cws ? rs : cws, oe = re <= cwe ? re : cwe;
				if (oe <= os) break;
				const unsigned long long v = cc[mm];
				if ((oe - os) + (uint64_t)(uint32_t)v >= (v >> 32)) cw1 += oe - os; else cw0 += oe - os;
			}
			return (double)cw0 >= (double)(cw0 + cw1) * 0.7;
		};
		auto cov_add = [&]() {                     // coverage of [rs, re) into its windows; counters cannot saturate here (caller's bound)
			uint64_t mm = rs / ocv_w, cws = mm * ocv_w;
			for (; mm < cwn; ++mm, cws += ocv_w) {
				uint64_t cwe = cws + ocv_w; if (cwe > rl) cwe = rl;
				const uint64_t os = rs >= cws ? rs : cws, oe = re <= cwe ? re : cwe;
				if (oe <= os) break;
				atomicAdd(cc + mm, (unsigned long long)(oe - os));
			}
		};
		if (act && !unc && w == 3 && cwn > 0) maybe = rescue_test();       // against the coverage before this tile: a lower bound
		const unsigned long long um = __ballot(unc); unsigned long long mm_ = __ballot(maybe), keepm = um;
		if (cwn) {
			int pos = 0;
			for (;;) {
				const unsigned long long rest = pos < 64 ? mm_ & ~((1ULL << pos) - 1) : 0ULL;
				const int c = rest ? __ffsll((long long)rest) - 1 : 64;
				if (unc && lane >= pos && lane < c) cov_add();
				HAO_WFENCE();
				if (c == 64) break;
				bool ok = false;
				if (lane == c) { ok = rescue_test(); if (ok) cov_add(); }      // exact coverage: everything kept before this chain has been added
				HAO_WFENCE();
				if (__ballot(ok)) keepm |= 1ULL << c;
				pos = c + 1;
			}
		}
		const bool keep = (keepm >> lane) & 1;
		if (keep) out[kk + __popcll(keepm & ltm)] = me;
		if (__ballot(keep && S.al[me] < chain_cutoff)) lch2 = 1;
		kk += __popcll(keepm);
	}
	HAO_WFENCE();
	for (int64_t i = lane; i < kk; i += 64) S.pm[i] = out[i];
	HAO_WFENCE();
	*lch_out = lch2;
	return kk;
}

// weak-chain filter (anchor.cpp:2061-2096), whole wave: control flow is uniform (keys in LDS), the count of a strong
// chain's hits inside the overlap interval is a strided wave reduction (the reference stops counting at ocn; only
// "count >= ocn" is observable), permutation swaps by lane 0.
__device__ int64_t hao_select_weak(const hao_sel_args &A, const hao_sel_ctx &S, int64_t n, const hao_ovlp_t *rec, const hao_cdesc *cd)
{
	const uint32_t chain_cutoff = A.chain_cutoff; const int lane = hao_lane();
#define XS(i) S.xs[S.pm[i]]
#define SC(i) S.sc[S.pm[i]]
#define AL(i) S.al[S.pm[i]]
	int64_t i, kk, ll;
	// Only chains with >= chain_cutoff hits can cover a weak one, and they are never dropped: list them once, in x_pos_s order (pm2 is free
	// after the sort).  The reference scans the whole (partly compacted) list up to the first entry with x_pos_s >= ze; every entry before
	// the weak chain has x_pos_s <= its own and the rest is still sorted, so that scan visits exactly the entries with x_pos_s < ze -
	// and only "some strong chain covers it" is observable.  On repeat-rich reads most of the several hundred chains are weak ones.
	uint32_t *sl = S.pm2; int64_t ns = 0;
	for (kk = 0; kk < n; kk += 64) {
		const int64_t kq = kk + lane; const bool st = kq < n && !(AL(kq) < chain_cutoff);
		const unsigned long long m = __ballot(st);
		if (st) sl[ns + __popcll(m & ((1ULL << lane) - 1))] = S.pm[kq];
		ns += __popcll(m);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	for (i = ll = 0; i < n; ++i) {
		bool drop = false;
		if (AL(i) < chain_cutoff) {
			uint64_t zs = XS(i) >> 32, ze = (uint64_t)(uint32_t)XS(i) + 1, ob = (uint64_t)((double)(ze - zs) * 0.95), ocn = (uint64_t)AL(i) << 4;
			int64_t osc = (int64_t)SC(i) * 16;
			if (ob < 16) ob = 16;
			// candidates are examined 64 at a time: the cheap tests (strong enough, overlaps >= ob of the weak chain) run one per lane,
			// only the survivors get the wave-wide hit count; the scan stops at the first chain that covers the weak one
			bool covered = false;
			for (kk = 0; kk < ns && !covered; kk += 64) {
				const int64_t kq = kk + lane; bool inr = false, cand = false; uint64_t os = 0, oe = 0; uint32_t ci = 0;
				if (kq < ns) {
					ci = sl[kq];
					const uint64_t xq = S.xs[ci]; inr = ze > (xq >> 32);
					if (inr && !(S.al[ci] < ocn || (int64_t)S.sc[ci] < osc)) {
						const uint64_t rs = xq >> 32, re = (uint64_t)(uint32_t)xq + 1; os = rs >= zs ? rs : zs; oe = re <= ze ? re : ze;
						cand = oe > os && oe - os >= ob;
					}
				}
				const unsigned long long inm = __ballot(inr);
				const int nin = __popcll(inm);                      // the strong list is sorted by x_pos_s: the in-range lanes are a prefix
				// Every candidate counts its own hits inside [os, oe], all candidates at once.  (One candidate after the other, the wave sweeping all of its hits,
				// cost two dependent memory round trips per candidate - and a single-hit weak chain is covered by nobody, so every overlapping strong chain
				// of the read was swept: 150 us per read, nearly all of chain_select_kernel.)  A chain's hits ascend in self_offset: a hit in front of the
				// window start counts only when me - span wraps (a k-mer at the read's first bases: me < span <= 255), the rest is a binary search for the
				// window's first hit and a walk to its last - the same count as the sweep's.
				// Few candidates, or a read with more than 128 chains (repeat-rich: most chains weak, the first strong one usually covers): the wave sweeps them one after
				// the other as before - two round trips each and an early exit beat the ~10 - 25 dependent loads of a lane's search and walk (repeat-rich 250 Mb twin:
				// 776 ms per pass with the sweep, 859 with the lane-wise form for every batch, 818 with it for batches of more than six candidates).
				const unsigned long long cm0 = __ballot(cand && lane < nin);
				if (n > HAO_WEAK_LANEWISE_N || __popcll(cm0) <= HAO_WEAK_SWEEP_MAX) {
					for (unsigned long long cm = cm0; cm; cm &= cm - 1) {
						const int l = __ffsll((long long)cm) - 1;
						const uint32_t cc_ = (uint32_t)__shfl((int)ci, l); const uint64_t cos = __shfl(os, l), coe = __shfl(oe, l);
						const hao_hit_t *ch = hao_cd_src(cd[cc_], A.hits, A.ohits); const uint64_t nh = S.al[cc_];
						uint64_t kn = 0;
						for (uint64_t b = 0; b < nh && kn < ocn; b += 64) {
							bool in = false;
							if (b + lane < nh) { uint64_t me = ch[b + lane].self_offset, ms = me - (ch[b + lane].cnt & 0xffu); in = ms >= cos && me <= coe; }
							kn += __popcll(__ballot(in));
						}
						if (kn >= ocn) { covered = true; break; }
					}
					if (covered || nin < 64) break;
					continue;
				}
				bool hit = false;
				if (cand && lane < nin) {
					const hao_hit_t *ch = hao_cd_src(cd[ci], A.hits, A.ohits); const uint64_t nh = S.al[ci];    // the chain's hits (the run of cl->list with its ordinal tag)
					uint64_t kn = 0, b = 0;
					for (; b < nh && kn < ocn; ++b) { const uint64_t me = ch[b].self_offset; if (me >= os || me >= 256) break; if (me - (ch[b].cnt & 0xffu) >= os) ++kn; }
					uint64_t lo = b, hi = nh;
					while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)ch[mid].self_offset < os) lo = mid + 1; else hi = mid; }
					for (b = lo; b < nh && kn < ocn; ++b) { const uint64_t me = ch[b].self_offset; if (me > oe) break; if (me - (ch[b].cnt & 0xffu) >= os) ++kn; }
					hit = kn >= ocn;
				}
				if (__any(hit)) { covered = true; break; }
				if (nin < 64) break;
			}
			drop = covered;
		}
		if (drop) continue;
		HAO_LOCKSTEP();      // every lane is done with chain i's record
		if (ll != i && lane == 0) hao_sw(S, ll, i);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		++ll;
	}
#undef XS
#undef SC
#undef AL
	return ll;
}

// INLDS as a template parameter: the sort / prune / filter code then addresses its keys with DS instructions (a run-time
// "LDS or global" pointer select would make every access a flat one).
template<int CAP, bool INLDS>
__device__ __forceinline__ void hao_select_body(const hao_sel_args &A, const uint64_t r, int64_t n, const uint64_t o0, const uint64_t cl0, const uint64_t cn,
		uint64_t *l_xs, int32_t *l_sc, uint32_t *l_al, uint32_t *l_pm, uint32_t *l_pm2, uint32_t *l_lp, uint32_t *l_rp, int32_t *l_stack, uint64_t *l_cc)
{
	const int lane = hao_lane();
	const hao_ovlp_t *rec = A.ol + o0;
	uint64_t *xs = INLDS ? l_xs : A.key_xs + o0; int32_t *sc = INLDS ? l_sc : A.key_sc + o0;
	uint32_t *al = INLDS ? l_al : A.key_al + o0, *pm = INLDS ? l_pm : A.perm + o0;
	int lch = 0;
	for (int64_t i = lane; i < n; i += 64) {
		const hao_ovlp_t q = rec[i];
		xs[i] = (uint64_t)q.x_pos_s << 32 | q.x_pos_e; sc[i] = q.shared_seed; al[i] = q.align_length; pm[i] = (uint32_t)i;
		if (q.align_length < A.chain_cutoff) lch = 1;
	}
	lch = __any(lch);
	__threadfence_block();
	hao_sel_ctx S; S.xs = xs; S.sc = sc; S.al = al; S.pm = pm; S.stack = l_stack;
	S.pm2 = INLDS ? l_pm2 : A.key_tmp + 5 * o0; S.lpos = INLDS ? l_lp : A.key_tmp + 5 * o0 + 2 * n; S.rasc = INLDS ? l_rp : A.key_tmp + 5 * o0 + 4 * n;
	int64_t nf = n; int lch2 = lch;
	unsigned long long tk0 = A.dbg ? wall_clock64() : 0, tk1 = tk0, tk2 = tk0, tk3, tk4;
	if ((uint64_t)n > A.max_n_chain) {
		hao_wave_intro_sort<0>(S, n);
		if (A.dbg) tk1 = wall_clock64();
		const bool cc_lds = (uint64_t)A.len[A.rid_lo + r] / A.ocv_w + 2 <= HAO_SEL_CCAP;
		if ((uint64_t)n * A.ocv_w < UINT32_MAX && !A.dbg_seq_prune) nf = cc_lds ? hao_select_prune_wave<true>(A, S, n, lch, r, &lch2, l_cc) : hao_select_prune_wave<false>(A, S, n, lch, r, &lch2, l_cc);
		else {      // a coverage counter could saturate: keep the order-dependent scan on one lane
			if (lane == 0) nf = cc_lds ? hao_select_prune<true>(A, S, n, lch, r, &lch2, l_cc) : hao_select_prune<false>(A, S, n, lch, r, &lch2, l_cc);
			nf = __shfl(nf, 0); lch2 = __shfl(lch2, 0);
		}
		HAO_WFENCE();
		if (A.dbg) tk2 = wall_clock64();
	}
	hao_wave_intro_sort<1>(S, nf);
	__threadfence_block();
	tk3 = A.dbg ? wall_clock64() : 0;
	if (lch2) nf = hao_select_weak(A, S, nf, rec, A.cd + o0);
	__threadfence_block();
	if (A.dbg && lane == 0) { tk4 = wall_clock64(); atomicAdd(A.dbg, tk1 - tk0); atomicAdd(A.dbg + 1, tk2 - tk1); atomicAdd(A.dbg + 2, tk3 - tk2); atomicAdd(A.dbg + 3, tk4 - tk3); atomicAdd(A.dbg + 4, 1ULL); }
	uint64_t fct = 0;
	for (int64_t i = lane; i < nf; i += 64) { uint32_t pi = pm[i]; if (INLDS) A.perm[o0 + i] = pi; fct += rec[pi].fc_len; }
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) fct += __shfl_xor(fct, d);
	if (lane == 0) { A.n_final[r] = (uint32_t)nf; A.fc_final[r] = fct; }
}

// WPB waves per workgroup, each wave one read; CAP = chains whose keys fit the wave's LDS slice (reads with more chains keep
// their keys in global scratch).
template<int WPB, int CAP>
__global__ __launch_bounds__(WPB * 64) void chain_select_kernel(hao_sel_args A, int64_t n_lo, int64_t n_hi)
{
	__shared__ uint64_t l_xs[WPB][CAP]; __shared__ int32_t l_sc[WPB][CAP]; __shared__ uint32_t l_al[WPB][CAP], l_pm[WPB][CAP], l_pm2[WPB][CAP], l_lp[WPB][CAP], l_rp[WPB][CAP]; __shared__ int32_t l_stack[WPB][3 * 72]; __shared__ uint64_t l_cc[WPB][HAO_SEL_CCAP];
	const int wv = threadIdx.x >> 6, lane = hao_lane();
	const uint64_t r = (uint64_t)blockIdx.x * WPB + wv;
	if (r > A.n_sel) return;
	if (r == A.n_sel) { if (lane == 0 && n_lo == 0) { A.n_final[r] = 0; A.fc_final[r] = 0; } return; }
	const uint64_t g0 = A.g_off[r], g1 = A.g_off[r + 1], o0 = A.ch_base[g0];
	const int64_t n = (int64_t)(A.ch_base[g1] - o0);
	if (n < n_lo || n >= n_hi) return;                          // this read belongs to another launch
	const uint64_t cl0 = A.cl_base[g0], cn = A.cl_base[g1] - cl0;
	if (n <= CAP) hao_select_body<CAP, true>(A, r, n, o0, cl0, cn, l_xs[wv], l_sc[wv], l_al[wv], l_pm[wv], l_pm2[wv], l_lp[wv], l_rp[wv], l_stack[wv], l_cc[wv]);
	else hao_select_body<CAP, false>(A, r, n, o0, cl0, cn, l_xs[wv], l_sc[wv], l_al[wv], l_pm[wv], l_pm2[wv], l_lp[wv], l_rp[wv], l_stack[wv], l_cc[wv]);
}

// Reads with hundreds of chains (repeat-rich): one workgroup of four waves per read.  The two sorts run on all four waves
// (hao_block_intro_sort); pruning and the weak-chain filter are single-wave code (wave 0), the other waves wait at the barriers.
template<int CAP>
__global__ __launch_bounds__(256) void chain_select4_kernel(hao_sel_args A, int64_t n_lo, int64_t n_hi)
{
	__shared__ uint64_t l_xs[CAP]; __shared__ int32_t l_sc[CAP]; __shared__ uint32_t l_al[CAP], l_pm[CAP], l_pm2[CAP], l_lp[CAP], l_rp[CAP];
	__shared__ int32_t l_stack[3 * 72]; __shared__ uint64_t l_cc[HAO_SEL_CCAP];
	__shared__ int32_t l_segs[2 * 3 * HAO_BSORT_MAXSEG]; __shared__ uint32_t l_segn[2]; __shared__ int l_flag; __shared__ int64_t l_nf; __shared__ int l_lch;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t r = blockIdx.x;
	if (r >= A.n_sel) return;
	const uint64_t g0 = A.g_off[r], g1 = A.g_off[r + 1], o0 = A.ch_base[g0];
	const int64_t n = (int64_t)(A.ch_base[g1] - o0);
	if (n < n_lo || n >= n_hi) return;                          // this read belongs to another launch (n <= CAP here)
	const hao_ovlp_t *rec = A.ol + o0;
	int lch = 0;
	for (int64_t i = tid; i < n; i += 256) {
		const hao_ovlp_t q = rec[i];
		l_xs[i] = (uint64_t)q.x_pos_s << 32 | q.x_pos_e; l_sc[i] = q.shared_seed; l_al[i] = q.align_length; l_pm[i] = (uint32_t)i;
		if (q.align_length < A.chain_cutoff) lch = 1;
	}
	if (tid == 0) l_lch = 0;
	__syncthreads();
	if (lch) l_lch = 1;
	__syncthreads();
	lch = l_lch;
	hao_sel_ctx S; S.xs = l_xs; S.sc = l_sc; S.al = l_al; S.pm = l_pm; S.stack = l_stack; S.pm2 = l_pm2; S.lpos = l_lp; S.rasc = l_rp;
	int64_t nf = n; int lch2 = lch;
	if ((uint64_t)n > A.max_n_chain) {
		hao_block_intro_sort<0, 4>(S, n, l_segs, l_segn, &l_flag);
		if (wv == 0) {
			const bool cc_lds = (uint64_t)A.len[A.rid_lo + r] / A.ocv_w + 2 <= HAO_SEL_CCAP;
			if ((uint64_t)n * A.ocv_w < UINT32_MAX) nf = cc_lds ? hao_select_prune_wave<true>(A, S, n, lch, r, &lch2, l_cc) : hao_select_prune_wave<false>(A, S, n, lch, r, &lch2, l_cc);
			else { if (lane == 0) nf = cc_lds ? hao_select_prune<true>(A, S, n, lch, r, &lch2, l_cc) : hao_select_prune<false>(A, S, n, lch, r, &lch2, l_cc); nf = __shfl(nf, 0); lch2 = __shfl(lch2, 0); }
			if (lane == 0) { l_nf = nf; l_lch = lch2; }
		}
		__syncthreads();
		nf = l_nf; lch2 = l_lch;
	}
	hao_block_intro_sort<1, 4>(S, nf, l_segs, l_segn, &l_flag);
	__syncthreads();
	if (wv != 0) return;
	if (lch2) nf = hao_select_weak(A, S, nf, rec, A.cd + o0);
	__threadfence_block();
	uint64_t fct = 0;
	for (int64_t i = lane; i < nf; i += 64) { const uint32_t pi = l_pm[i]; A.perm[o0 + i] = pi; fct += rec[pi].fc_len; }
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) fct += __shfl_xor(fct, d);
	if (lane == 0) { A.n_final[r] = (uint32_t)nf; A.fc_final[r] = fct; }
}

// final gather: records in final order (align_length zeroed, anchor.cpp:2098) + fake cigars in that order. One wave per read.
// With fcw != nullptr (delivery) the cigars are also written as they travel (hao_deliver.cuh: 4 bytes per entry after an overlap's first): overlap J of the batch,
// whose entries start at fc_out[fo], owns words [fo - J, fo - J + fc_len - 1) of the main region - every overlap before it saved exactly one entry - so no
// offsets have to be computed; an overlap with a step that does not fit the packed word (rare) appends its entries raw, two words each, behind the main region
// (n_main = all entries - all overlaps: both totals are on the device before this kernel runs) and says so in bit 63 of its offset.
__device__ __forceinline__ bool hao_fc_step(uint64_t prev, uint64_t cur, uint32_t *word)
{
	const uint32_t ps = (uint32_t)(prev >> 32), cs = (uint32_t)(cur >> 32), pl = (uint32_t)prev, cl = (uint32_t)cur;
	const int64_t psh = (pl & 1) ? -(int64_t)(pl >> 1) : (int64_t)(pl >> 1), csh = (cl & 1) ? -(int64_t)(cl >> 1) : (int64_t)(cl >> 1), dsh = csh - psh;
	if (cs < ps || cs - ps >= (1u << 20) || dsh < -2048 || dsh > 2047) return false;
	*word = (cs - ps) | (uint32_t)((dsh << 1) ^ (dsh >> 63)) << 20;
	return true;
}
__global__ __launch_bounds__(256) void chain_final_kernel(const hao_ovlp_t *ol, const uint64_t *ol_fc_off, const uint64_t *fc_raw, const uint32_t *perm,
		const uint64_t *g_off, const uint64_t *ch_base, const uint64_t *fin_off, const uint64_t *fcf_off, uint64_t n_sel,
		hao_ovlp_t *ol_out, uint64_t *fc_out, uint64_t *fc_out_off, uint32_t *fcw, uint64_t *fcw_off, unsigned long long *fcw_raw_words, uint32_t raw_every /* tests: every n-th overlap travels raw */)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_sel) return;
	const int lane = hao_lane();
	const uint64_t o0 = ch_base[g_off[r]], d0 = fin_off[r], n = fin_off[r + 1] - d0; uint64_t run = fcf_off[r];
	const uint64_t n_main = fcw ? fcf_off[n_sel] - fin_off[n_sel] : 0;
	for (uint64_t base = 0; base < n; base += 64) {      // one kept chain per lane; fake-cigar destinations from a wave scan of the lengths
		const uint64_t i = base + lane; const bool act = i < n;
		hao_ovlp_t o; uint64_t fs = 0; uint32_t fl = 0;
		if (act) { const uint64_t src = o0 + perm[o0 + i]; o = ol[src]; fs = ol_fc_off[src]; fl = o.fc_len; }
		const uint32_t inc = hao_wave_incl_scan_u32(fl);
		const uint64_t fo = run + inc - fl, wo = fo - (d0 + i);      // first entry in fc_out / first word on the wire
		bool ok = !(raw_every && (d0 + i) % raw_every == raw_every - 1);      // the cigar fits the packed words so far
		if (act) {
			o.align_length = 0; ol_out[d0 + i] = o; fc_out_off[d0 + i] = fo;
			// the packed layout (wo = fo - J, n_main = entries - overlaps) needs at least one entry per overlap; gen_fake_cigar runs with apend_be = 1, so there always
			// is one - should that ever change, the batch fails (bit 63 of the raw-word counter, read by the host) instead of shipping overlapping word ranges
			if (fcw && fl == 0) atomicOr(fcw_raw_words, 1ULL << 63);
			if (fl <= 16) {      // every entry requested before the first is stored (one memory round trip for the cigar instead of one per entry)
				uint64_t ev[16];
#pragma unroll
				for (uint32_t j = 0; j < 16; ++j) if (j < fl) ev[j] = fc_raw[fs + j];
#pragma unroll
				for (uint32_t j = 0; j < 16; ++j) if (j < fl) {
					const uint64_t e = ev[j]; fc_out[fo + j] = e;
					if (fcw) { uint32_t w = 0; if (j == 0) ok = ok && e == (uint64_t)o.x_pos_s << 32; else if (ok && hao_fc_step(ev[j ? j - 1 : 0], e, &w)) fcw[wo + j - 1] = w; else ok = false; }
				}
			}
		}
		// long cigars (noisy reads: hundreds of entries per chain): the wave copies them together, one chain after the other
		for (unsigned long long big = __ballot(act && fl > 16); big; big &= big - 1) {
			const int l = __ffsll((long long)big) - 1;
			const uint64_t cfs = hao_readlane_i64((int64_t)fs, l), cfo = hao_readlane_i64((int64_t)fo, l), cwo = hao_readlane_i64((int64_t)wo, l); const uint32_t cfl = hao_bcast(fl, l), cxs = hao_bcast(o.x_pos_s, l);
			bool okw = true;
			for (uint32_t j = lane; j < cfl; j += 64) {
				const uint64_t e = fc_raw[cfs + j]; fc_out[cfo + j] = e;
				if (fcw) { uint32_t w = 0; if (j == 0) okw = e == (uint64_t)cxs << 32; else if (hao_fc_step(fc_raw[cfs + j - 1], e, &w)) fcw[cwo + j - 1] = w; else okw = false; }
			}
			if (__ballot(!okw) && lane == l) ok = false;
		}
		if (fcw) {
			// overlaps that do not fit: raw, behind the main region (one atomic per wave reserves their words)
			const uint32_t need = (act && !ok) ? 2 * fl : 0, rinc = hao_wave_incl_scan_u32(need), rtot = hao_bcast(rinc, 63);
			unsigned long long rb = 0;
			if (rtot) { if (lane == 63) rb = atomicAdd(fcw_raw_words, (unsigned long long)rtot); rb = (unsigned long long)hao_readlane_i64((int64_t)rb, 63); }
			if (act) {
				if (ok) fcw_off[d0 + i] = wo;
				else {
					const uint64_t at = n_main + rb + rinc - need; fcw_off[d0 + i] = at | HAO_FC_RAW;
					for (uint32_t j = 0; j < fl; ++j) { const uint64_t e = fc_raw[fs + j]; fcw[at + 2 * j] = (uint32_t)e; fcw[at + 2 * j + 1] = (uint32_t)(e >> 32); }
				}
			}
		}
		run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
	}
}
