// Host-side scalar pieces of the product path (tiny, must be exact): histogram -> peaks,
// occurrence thresholds, seed weights, chain penalties.  libm calls (expf, pow) stay on the
// host exactly as in the reference (SURVEY.md 8c "libm note").
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include "hao.h"

// With a prior homozygous coverage (--hg-size): of the three candidate peaks pick the one nearest to the prior (ties go to the highest
// bin `top`); a nearest peak that lies below the prior by more than half its own position is taken as the heterozygous peak and the
// prior itself is returned; otherwise the next candidate to the left becomes the heterozygous peak (adj_m_peak_hom, hist.cpp:46-72).
static int hao_adjust_to_prior(int prior, int top, int left, int right, int *peak_het)
{
	const int64_t cand[3] = { left, top, right };
	int best = -1; int64_t best_d = -1;
	for (int i = 0; i < 3; ++i) {
		if (cand[i] <= 0) continue;
		const int64_t d = cand[i] >= prior ? cand[i] - prior : prior - cand[i];
		if (best_d == -1 || d < best_d || (d == best_d && i == 1)) { best_d = d; best = i; }
	}
	if (best < 0) return prior;
	if (cand[best] < prior && (double)(prior - cand[best]) >= cand[best] * 0.51) { *peak_het = (int)cand[best]; return prior; }
	for (int i = best - 1; i >= 0; --i) if (cand[i] > 0) { *peak_het = (int)cand[i]; break; }
	return (int)cand[best];
}

// Peak finder of the k-mer / minimizer count histogram: ha_analyze_count (hist.cpp:74-157).  cnt[i] = #distinct k-mers seen i times;
// prior_hom = total_bases / hg_size when --hg-size is given, else <= 0 (htab.cpp:1156,1254).
// Returns peak_hom (or -1 when the histogram never rises: coverage too low), *peak_het = -1 if none.
static int hao_find_peaks(const int64_t *cnt, int n_cnt, int start_cnt, int *peak_het, int prior_hom = -1)
{
	*peak_het = -1;
	const int first = cnt[1] > 0 ? 1 : 2;
	int valley = std::max(first, start_cnt) + 1;
	while (valley < n_cnt && cnt[valley] <= cnt[valley - 1]) ++valley;      // walk down the error tail
	--valley;
	if (valley == n_cnt - 1) return -1;
	int top = valley + 1;                                                     // highest bin right of the valley, leftmost on ties
	for (int i = valley + 1; i < n_cnt; ++i) if (cnt[i] > cnt[top]) top = i;
	const int64_t top_v = cnt[top];
	auto is_local_max = [&](int i) { return cnt[i] >= cnt[i - 1] && cnt[i] >= cnt[i + 1]; };
	auto min_between = [&](int a, int b) { int64_t m = top_v; for (int i = a; i < b; ++i) if (cnt[i] < m) m = cnt[i]; return m; };
	// strongest local maximum strictly between valley and top (scanning from top downwards, strict improvement)
	int left = -1; int64_t left_v = -1;
	for (int i = top - 1; i > valley; --i) if (is_local_max(i) && cnt[i] > left_v) { left_v = cnt[i]; left = i; }
	if (left > valley && left < top) {
		if (left_v < top_v * 0.05 || min_between(left + 1, top) > left_v * 0.95) left = -1;   // too small, or no real dip
	}
	// strongest local maximum right of top
	int right = -1; int64_t right_v = -1;
	for (int i = top + 1; i < n_cnt - 1; ++i) if (is_local_max(i) && cnt[i] > right_v) { right_v = cnt[i]; right = i; }
	if (right > top) {
		if (right_v < top_v * 0.05 || min_between(top + 1, right) > right_v * 0.95 || right > top * 2.5) right = -1;
	}
	if (prior_hom > 0) return hao_adjust_to_prior(prior_hom, top, left, right, peak_het);
	if (right > 0) { *peak_het = top; return right; }      // top was the heterozygous peak
	if (left > 0) *peak_het = left;
	return top;
}

// high_occ / low_occ of a pass: ecovlp.cpp:3237-3238 (HA_KMER_GOOD_RATIO = 0.333, anchor.cpp:11)
static inline void hao_occ_thresholds(int hom_cov, uint32_t *high_occ, uint32_t *low_occ)
{
	*high_occ = (uint32_t)(hom_cov * (2.0 - 0.333));
	*low_occ = (uint32_t)(hom_cov * 0.333);
}

// weight of a seed whose minimizer occurs n times in the index (anchor.cpp:1066-1075):
// 1 inside (low, high), 2 at or below low, else floor(pow(1 + ceil(n / 2 high), 1.1)); table over n = 0..4095
static inline void hao_seed_weight_table(uint32_t high_occ, uint32_t low_occ, std::vector<uint32_t> &tab)
{
	const uint64_t hi = high_occ < 2 ? 2 : high_occ, lo = low_occ < 2 ? 2 : low_occ;
	tab.resize(4096);
	for (uint64_t n = 0; n < 4096; ++n) {
		uint32_t wgt;
		if (n < hi && n > lo) wgt = 1;
		else if (n <= lo) wgt = 2;
		else { wgt = (uint32_t)(1 + ((n + (hi << 1) - 1) / (hi << 1))); wgt = (uint32_t)pow((double)wgt, 1.1); }
		if (wgt > 0xffffffu) wgt = 0xffffffu;
		tab[n] = wgt;
	}
}

struct hao_chain_par {          // set_lchain_dp_op(is_accurate = 1), anchor.cpp:2272-2285 + h_ec_lchain arguments (ecovlp.cpp:3274)
	double pen_gap, pen_skip, bw;
	int64_t max_skip, max_iter, max_dis;
	int64_t mcopy_num; double mcopy_rate; int64_t mcopy_khit_cut;
	uint32_t chain_cutoff; uint64_t ocv_w; uint64_t max_n_chain;
};

static inline hao_chain_par hao_chain_params(int k, const hao_pass_t &ps)
{
	hao_chain_par p;
	double tmp = expf(-0.01 * (double)k);       // float expf of a double argument, result widened: as the reference
	p.pen_gap = 0.5f * tmp; p.pen_skip = 0.0005f * tmp;
	p.max_skip = 25; p.max_iter = 5000; p.max_dis = 5000;
	p.bw = ps.bw_thres;
	p.mcopy_num = ps.mcopy_num; p.mcopy_rate = ps.mcopy_rate; p.mcopy_khit_cut = ps.mcopy_khit_cut; p.chain_cutoff = ps.chain_cutoff; p.ocv_w = ps.ocv_w;
	p.max_n_chain = (uint64_t)ps.max_n_chain;
	return p;
}
