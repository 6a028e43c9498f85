// f3 on the device end to end: the window / candidate pairs of a batch, generated ON THE DEVICE from the batch's final ol->list on the reference's fixed window grid
// (windows of WINDOW = 375 query bases starting at multiples of WINDOW: Hash_Table.h:9, Correct.cpp:5645, 5993; a pair per overlap and grid window it covers, the
// window clipped to the overlap at its two ends, the pattern = the target interval on the overlap's diagonal padded by thre on both sides and clipped at the read ends,
// abs_diag = the bases clipped at the start: Correct.cpp:3897's call of ed_band_cal_semi_64_w_absent_diag without the fake-cigar shift - the same pairs as
// tests/helpers.py ed_tasks_grid).  Tasks come out in TEXT order - (query read, grid window, position in ol->list) - which is the order the window-alignment kernels
// want (hao_align.cuh: a wave takes 64 neighbours, which share their text), so nothing is uploaded, sorted or downloaded: two counting kernels, two scans, one fill.
#pragma once
#include "hao_common.cuh"

__global__ void ed_grid_nwin_kernel(const uint32_t *len, uint64_t rid_lo, uint64_t n, uint32_t wl, uint64_t *nwin)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n) return;
	nwin[r] = r < n ? (len[rid_lo + r] + wl - 1) / wl : 0;
}

// the pair of overlap z and grid window w (helpers.ed_tasks_grid); false: the overlap does not cover the window, or the pair is empty / not expressible
__device__ __forceinline__ bool hao_grid_pair(const hao_ovlp_t &z, uint32_t w, uint32_t wl, uint32_t thre, uint32_t nword, const uint32_t *len, hao_ed_task_t *t)
{
	const int64_t xs = z.x_pos_s, xe = z.x_pos_e, g0 = (int64_t)w * wl;
	if (xs / wl > (int64_t)w || xe / wl < (int64_t)w) return false;
	const int64_t ws = g0 > xs ? g0 : xs, we = g0 + wl - 1 < xe ? g0 + wl - 1 : xe, tn = we + 1 - ws, tl = len[z.y_id];
	int64_t p0 = (int64_t)z.y_pos_s + (ws - xs) - (int64_t)thre, p1 = p0 + tn + 2 * (int64_t)thre, ad = 0;
	if (p0 < 0) { ad = -p0 < 2 * (int64_t)thre ? -p0 : 2 * (int64_t)thre; p0 = 0; }
	if (p1 > tl) p1 = tl;
	if (p1 <= p0 || tn <= 0) return false;
	// bands of more than one word: the final scan reads bit i of VP / VN for i < p_len - t_len + abs_diag, which must lie inside the band's words (hao_window_ed_batch refuses such a task)
	if (nword > 1 && (p1 - p0) - tn + ad > 64 * (int64_t)nword) return false;
	t->p_rid = z.y_id; t->p_pos = (uint32_t)p0; t->p_len = (uint32_t)(p1 - p0); t->p_rev = z.y_pos_strand;
	t->t_rid = z.x_id; t->t_pos = (uint32_t)ws; t->t_len = (uint32_t)tn; t->t_rev = 0; t->thre = thre; t->abs_diag = (uint32_t)ad;
	return true;
}

// one wave per read of the batch, a lane per grid window (64 at a time); FILL = false: pairs per window -> cnt[wbase[r] + w]; FILL = true: the pairs themselves at off[wbase[r] + w] ..
template<bool FILL>
__global__ __launch_bounds__(256) void ed_grid_kernel(const hao_ovlp_t *ol, const uint64_t *fin_off, const uint32_t *len, uint64_t rid_lo, uint64_t n, uint32_t wl, uint32_t thre, uint32_t nword,
		const uint64_t *wbase, uint64_t *cnt_or_off, hao_ed_task_t *tasks)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n) return;
	const int lane = hao_lane();
	const uint64_t o0 = fin_off[r], o1 = fin_off[r + 1], wb = wbase[r]; const uint32_t nw = (uint32_t)(wbase[r + 1] - wb);
	for (uint32_t w = lane; w < nw; w += 64) {
		uint64_t k = 0; const uint64_t at = FILL ? cnt_or_off[wb + w] : 0;
		for (uint64_t i = o0; i < o1; ++i) {
			hao_ed_task_t t;
			if (!hao_grid_pair(ol[i], w, wl, thre, nword, len, &t)) continue;
			if (FILL) tasks[at + k] = t;
			++k;
		}
		if (!FILL) cnt_or_off[wb + w] = k;
	}
}
