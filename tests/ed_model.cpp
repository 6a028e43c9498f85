// Executable model of the window-alignment kernels (hifiasm_amd/csrc/hao_align.cuh) for the CPU test suite: the lane-level functions of that file - text
// staging, pattern streaming, the per-column step, the final scans, the traceback - are compiled here by g++ exactly as the device compiles them, and
// hao_al_kernel's wave is replaced by loops over 64 lanes (the lanes of a wave share nothing but the staged text, which they only read, so a sequential
// walk over the lanes is exact).  The host flow of hao_window_ed_batch / hao_window_trace_batch is mirrored as well: tasks sorted by text window, tiles of 64,
// text segments inside a tile, and for the traced alignments the column-free first sweep, the selection and the second sweep in slices.
// TEST INFRASTRUCTURE: built and called by tests/test_ed_model_cpu.py only, compared there with the oracle (itself pinned to the reference's functions).
#define HAO_ALIGN_HOST_MODEL
#include "hao.h"
#include "hao_align.cuh"
#include <algorithm>
#include <vector>

template<typename WT, int MODE, bool TRACE>
static void model_launch(const hao_ed_reads &R, const hao_ed_task_t *task, const uint32_t *order, uint64_t n_order, uint64_t *path, uint64_t stride,
		hao_ed_result_t *out_ed, hao_trace_result_t *out_tr, uint8_t *want_trace, uint16_t *cig, uint32_t cap)
{
	uint8_t codes[HAO_AL_CH];
	for (uint64_t tile = 0; tile * 64 < n_order; ++tile) {
		hao_ed_task_t T[64]; bool have[64], mine[64], head[64]; uint32_t ti[64]; hao_al_state<WT> S[64];
		for (int lane = 0; lane < 64; ++lane) {
			const uint64_t slot = tile * 64 + lane;
			have[lane] = slot < n_order; ti[lane] = have[lane] ? order[slot] : 0;
			if (have[lane]) T[lane] = task[ti[lane]]; else memset(&T[lane], 0, sizeof(hao_ed_task_t));
			mine[lane] = have[lane] && hao_al_mine<WT>(T[lane].thre);
		}
		for (int lane = 0; lane < 64; ++lane) {
			hao_ed_task_t L = T[lane];
			if (lane == 0) { L.t_rid = 0xffffffffu; L.t_pos = 0; L.t_len = 0; L.t_rev = 0; L.thre = 0; }
			else { L.t_rid = T[lane - 1].t_rid; L.t_pos = T[lane - 1].t_pos; L.t_len = T[lane - 1].t_len; L.t_rev = T[lane - 1].t_rev; L.thre = T[lane - 1].thre; }
			head[lane] = mine[lane] && (lane == 0 || !hao_al_same_text(T[lane], L) || !hao_al_mine<WT>(L.thre));
			S[lane].alive = 0; S[lane].dead = 0;
			if (mine[lane]) hao_al_init<WT, MODE>(S[lane], R, T[lane]);
		}
		for (int h0 = 0; h0 < 64; ++h0) {
			if (!head[h0]) continue;
			int h1 = 64; for (int x = h0 + 1; x < 64; ++x) if (head[x]) { h1 = x; break; }
			const hao_al_walk tw = hao_al_walk_of(R, T[h0].t_rid, T[h0].t_pos, T[h0].t_len, T[h0].t_rev, MODE == HAO_AL_EXT_BWD);
			const int64_t t_len = T[h0].t_len;
			for (int64_t k0 = 0; k0 < t_len; k0 += HAO_AL_CH) {
				const int32_t n = (int32_t)(t_len - k0 < HAO_AL_CH ? t_len - k0 : HAO_AL_CH);
				for (int lane = 0; lane < 64; ++lane) hao_al_stage_bases(tw, k0, n, codes, lane, 64);
				for (int lane = 0; lane < 64; ++lane) hao_al_stage_nsites(tw, k0, n, codes, lane, 64);
				for (int lane = h0; lane < h1; ++lane) {
					if (!mine[lane] || !S[lane].alive || S[lane].dead) continue;
					uint64_t *col = path + (tile * 64 + lane);
					for (int32_t i = (int32_t)k0; i < (int32_t)k0 + n && i < S[lane].tn && !S[lane].dead; ++i) hao_al_column<WT, MODE, TRACE>(S[lane], codes[i - (int32_t)k0], i, col, stride);
				}
			}
		}
		for (int lane = 0; lane < 64; ++lane) {
			if (!mine[lane]) continue;
			hao_trace_result_t res; uint64_t *col = path + (tile * 64 + lane);
			const bool tr = hao_al_finish<WT, MODE, TRACE>(S[lane], T[lane], res, col, stride, TRACE ? cig + (uint64_t)ti[lane] * cap : nullptr, cap);
			if (MODE == HAO_AL_ED) { hao_ed_result_t r2; r2.err = res.err; r2.pe = res.pe; out_ed[ti[lane]] = r2; }
			else { out_tr[ti[lane]] = res; if (!TRACE) want_trace[ti[lane]] = tr ? 1 : 0; }
		}
	}
}

// the instantiations a batch needs: band class 0 = one- and two-word bands (what the device library launches), 1 = three- and four-word bands (thre 64 .. 127)
template<int MODE, bool TRACE>
static void model_launch_all(int band, bool two_types, const hao_ed_reads &R, const hao_ed_task_t *task, const uint32_t *order, uint64_t n_order, uint64_t *path, uint64_t stride,
		hao_ed_result_t *out_ed, hao_trace_result_t *out_tr, uint8_t *want_trace, uint16_t *cig, uint32_t cap)
{
	if (band == 0) {
		model_launch<uint64_t, MODE, TRACE>(R, task, order, n_order, path, stride, out_ed, out_tr, want_trace, cig, cap);
		if (two_types) model_launch<hao_u128, MODE, TRACE>(R, task, order, n_order, path, stride, out_ed, out_tr, want_trace, cig, cap);
	} else {
		model_launch<hao_wide<3>, MODE, TRACE>(R, task, order, n_order, path, stride, out_ed, out_tr, want_trace, cig, cap);
		model_launch<hao_wide<4>, MODE, TRACE>(R, task, order, n_order, path, stride, out_ed, out_tr, want_trace, cig, cap);
	}
}

template<int MODE>
static void model_trace(int band, const hao_ed_reads &R, const hao_ed_task_t *task, const std::vector<uint32_t> &order, bool wide, uint64_t tn_max, hao_trace_result_t *out, uint16_t *cig, uint32_t cap,
		uint64_t slice_bytes)
{
	const uint64_t n = order.size();
	std::vector<uint8_t> want(n, 0);
	model_launch_all<MODE, false>(band, wide, R, task, order.data(), n, nullptr, 0, nullptr, out, want.data(), nullptr, 0);
	std::vector<uint32_t> sel; for (uint32_t i : order) if (want[i]) sel.push_back(i);
	const uint64_t n_sel = sel.size();
	if (!n_sel) return;
	const uint64_t cw = band ? 20 : (wide ? 10 : 5);      // 64-bit words kept per text column: 5 vectors x the widest band word of the batch
	const uint64_t slice = std::max<uint64_t>(256, std::min<uint64_t>((n_sel + 255) & ~255ULL, (slice_bytes / (8 * cw * tn_max)) & ~255ULL));
	std::vector<uint64_t> path(cw * tn_max * slice + 1);
	for (uint64_t lo = 0; lo < n_sel; lo += slice) {
		const uint64_t m = std::min<uint64_t>(slice, n_sel - lo);
		model_launch_all<MODE, true>(band, wide, R, task, sel.data() + lo, m, path.data(), slice, nullptr, out, nullptr, cig, cap);
	}
}

// mode 0 .. 3: hao_window_trace_batch's modes (out_tr, cig); 4: hao_window_ed_batch (out_ed).  slice_bytes: column scratch of one second-sweep slice
// (the product uses 4 GB; tests pass something small to cross slice boundaries).  band: 0 = bands of one / two words (thre <= 63: what libhao.so launches),
// 1 = three / four words (thre 64 .. 127: the reference's *_infi_* functions; model only so far).
extern "C" int hao_model_window(int mode, const uint8_t *packed, const uint64_t *pk_off, const uint32_t *len, const uint64_t *nsite_off, const uint32_t *nsite,
		const hao_ed_task_t *tasks, uint64_t n, hao_ed_result_t *out_ed, hao_trace_result_t *out_tr, uint16_t *cig, uint32_t cap, uint64_t slice_bytes, int band)
{
	hao_ed_reads R; R.packed = packed; R.pk_off = pk_off; R.len = len; R.nsite_off = nsite_off; R.nsite = nsite;
	std::vector<uint32_t> order(n); for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hao_al_sort_key(tasks[a]) < hao_al_sort_key(tasks[b]); });
	bool wide = false; uint64_t tn_max = 1;
	for (uint64_t i = 0; i < n; ++i) {
		const uint32_t nw = hao_al_nword(tasks[i].thre);
		if (band == 0 ? nw > 2 : (nw < 3 || nw > 4)) return -2;      // a task of another band class
		if (nw == 2) wide = true;
		if (tasks[i].t_len > tn_max) tn_max = tasks[i].t_len;
	}
	if (mode == HAO_AL_ED) model_launch_all<HAO_AL_ED, false>(band, wide, R, tasks, order.data(), n, nullptr, 0, out_ed, nullptr, nullptr, nullptr, 0);
	else if (mode == HAO_AL_GLOBAL) model_trace<HAO_AL_GLOBAL>(band, R, tasks, order, wide, tn_max, out_tr, cig, cap, slice_bytes);
	else if (mode == HAO_AL_EXT_FWD) model_trace<HAO_AL_EXT_FWD>(band, R, tasks, order, wide, tn_max, out_tr, cig, cap, slice_bytes);
	else if (mode == HAO_AL_EXT_BWD) model_trace<HAO_AL_EXT_BWD>(band, R, tasks, order, wide, tn_max, out_tr, cig, cap, slice_bytes);
	else if (mode == HAO_AL_SEMI) model_trace<HAO_AL_SEMI>(band, R, tasks, order, wide, tn_max, out_tr, cig, cap, slice_bytes);
	else return -1;
	return 0;
}
