// libhao.so, second translation unit: f3, the window-alignment batches (hao_align.cuh) - 36 instantiations of hao_al_kernel (five modes, with and without
// traceback, bands of one to four words) that the rest of the library does not depend on; compiled beside hao_capi.hip (hifiasm_amd/build.py).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "hao_ctx.hpp"
#include "hao_comm.hpp"
#include "hao_align.cuh"

// ---- f3 (hao_align.cuh): host side of the window-alignment batches ----
// tasks -> device, and their order by text window (hao_align.cuh: a wave takes 64 neighbours of that order, which mostly share one text)
static int hao_al_upload_sorted(hao_ctx *c, const hao_ed_task_t *tasks, uint64_t n)
{
	HIP_TRY(c->al_task.reserve(n)); HIP_TRY(c->al_k1.reserve(n)); HIP_TRY(c->al_k2.reserve(n)); HIP_TRY(c->al_i1.reserve(n)); HIP_TRY(c->al_order.reserve(n));
	HIP_TRY(hipMemcpyAsync(c->al_task.p, tasks, n * sizeof(hao_ed_task_t), hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(hao_al_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->al_task.p, n, c->al_k1.p, c->al_i1.p); HAO_CHECK_LAUNCH();
	size_t tb = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, c->al_k1.p, c->al_k2.p, c->al_i1.p, c->al_order.p, n, 0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, c->al_k1.p, c->al_k2.p, c->al_i1.p, c->al_order.p, n, 0, 64, c->stream));
	return HAO_OK;
}
static hao_ed_reads hao_al_reads_of(hao_ctx *c)
{
	hao_ed_reads R; R.packed = c->d_packed.p; R.pk_off = c->d_pk_off.p; R.len = c->d_len.p; R.nsite_off = c->has_n ? c->d_nsite_off.p : nullptr; R.nsite = c->has_n ? c->d_nsite.p : nullptr;
	return R;
}

template<int MODE> static int hao_al_trace_run(hao_ctx *c, const hao_ed_reads &R, const hao_ed_task_t *dt, const uint32_t *order, uint64_t n, uint32_t words /* bit (nword - 1): some task's band has nword words */, uint64_t tn_max,
		hao_trace_result_t *dr, uint8_t *want, uint16_t *dc, uint32_t cap)
{
	// first sweep: no column storage, every task; decides which tasks end within their threshold (want[])
	const dim3 g_((unsigned)((n + 255) / 256)), b_(256);
	// (one launch per band word count that occurs: a launch skips the tasks of the other widths, hao_al_mine)
	if (words & 1u) { hipLaunchKernelGGL((hao_al_kernel<uint64_t, MODE, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, (hao_ed_result_t*)nullptr, dr, want, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 2u) { hipLaunchKernelGGL((hao_al_kernel<hao_u128, MODE, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, (hao_ed_result_t*)nullptr, dr, want, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 4u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<3>, MODE, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, (hao_ed_result_t*)nullptr, dr, want, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 8u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<4>, MODE, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, (hao_ed_result_t*)nullptr, dr, want, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	// the tasks of the second sweep, still in text order
	DevBuf<uint32_t> &sel = c->al_sel; DevBuf<uint64_t> &path = c->al_path; uint64_t n_sel = 0;
	HIP_TRY(sel.reserve(n + 1)); HIP_TRY(c->d_cursor.reserve(2));
	size_t tb = 0;
	HIP_TRY(rocprim::select(nullptr, tb, order, sel.p, (uint64_t*)c->d_cursor.p, n, hao_al_flagged{want}, c->stream)); HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::select(c->d_tmp.p, tb, order, sel.p, (uint64_t*)c->d_cursor.p, n, hao_al_flagged{want}, c->stream));
	HIP_TRY(hipMemcpyAsync(&n_sel, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (n_sel) {
		// second sweep: 40 bytes per band word, text column and selected pair, in slices whose columns fit ~4 GB; then the traceback
		const uint64_t cw = 5 * (uint64_t)((words & 8u) ? 4 : (words & 4u) ? 3 : (words & 2u) ? 2 : 1);
		const uint64_t slice = std::max<uint64_t>(256, std::min<uint64_t>((n_sel + 255) & ~255ULL, ((4ULL << 30) / (8 * cw * tn_max)) & ~255ULL));
		HIP_TRY(path.reserve(cw * tn_max * slice + 1));
		for (uint64_t lo = 0; lo < n_sel; lo += slice) {
			const uint64_t m = std::min<uint64_t>(slice, n_sel - lo);
			const dim3 g2((unsigned)((m + 255) / 256));
			if (words & 1u) { hipLaunchKernelGGL((hao_al_kernel<uint64_t, MODE, true>), g2, b_, 0, c->stream, R, dt, sel.p + lo, m, path.p, slice, (hao_ed_result_t*)nullptr, dr, (uint8_t*)nullptr, dc, cap); HAO_CHECK_LAUNCH(); }
			if (words & 2u) { hipLaunchKernelGGL((hao_al_kernel<hao_u128, MODE, true>), g2, b_, 0, c->stream, R, dt, sel.p + lo, m, path.p, slice, (hao_ed_result_t*)nullptr, dr, (uint8_t*)nullptr, dc, cap); HAO_CHECK_LAUNCH(); }
			if (words & 4u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<3>, MODE, true>), g2, b_, 0, c->stream, R, dt, sel.p + lo, m, path.p, slice, (hao_ed_result_t*)nullptr, dr, (uint8_t*)nullptr, dc, cap); HAO_CHECK_LAUNCH(); }
			if (words & 8u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<4>, MODE, true>), g2, b_, 0, c->stream, R, dt, sel.p + lo, m, path.p, slice, (hao_ed_result_t*)nullptr, dr, (uint8_t*)nullptr, dc, cap); HAO_CHECK_LAUNCH(); }
		}
	}
	return HAO_OK;
}

// the distance-only window alignment over n tasks that already lie in c->al_task in text order (hao_window_ed_grid, hao_batch.hpp): results in c->al_res
int hao_al_ed_resident(hao_ctx *c, uint64_t n, uint32_t nword)
{
	HIP_TRY(c->al_order.reserve(n + 1)); HIP_TRY(c->al_res.reserve(n + 1));
	hipLaunchKernelGGL(hao_al_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->al_order.p, n); HAO_CHECK_LAUNCH();
	const hao_ed_reads R = hao_al_reads_of(c);
	const dim3 g_((unsigned)((n + 255) / 256)), b_(256);
	hao_ed_task_t *dt = c->al_task.p; uint32_t *order = c->al_order.p; hao_ed_result_t *dr = c->al_res.p;
	if (nword == 1) hipLaunchKernelGGL((hao_al_kernel<uint64_t, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, dr, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u);
	else if (nword == 2) hipLaunchKernelGGL((hao_al_kernel<hao_u128, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, dr, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u);
	else if (nword == 3) hipLaunchKernelGGL((hao_al_kernel<hao_wide<3>, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, dr, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u);
	else hipLaunchKernelGGL((hao_al_kernel<hao_wide<4>, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt, order, n, (uint64_t*)nullptr, (uint64_t)0, dr, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u);
	HAO_CHECK_LAUNCH();
	return HAO_OK;
}

extern "C" {

int hao_window_ed_batch(hao_ctx *c, const hao_ed_task_t *tasks, uint64_t n_tasks, hao_ed_result_t *out)
{
	if (!c || (!tasks && n_tasks) || (!out && n_tasks)) return HAO_EINVAL;
	if (int rc = hao_view_refresh(c)) return rc;
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_window_ed_batch needs the bases of both reads: single-device mode only"); return HAO_EUNSUPP; }
	c->al_grid_n = 0;      // (the task / result scratch is shared with hao_window_ed_grid: what that call left is gone)
	if (n_tasks == 0) return HAO_OK;
	if (n_tasks >= (1ULL << 32)) { hao_set_err(c, "hao_window_ed_batch: more than 2^32 tasks in one call"); return HAO_EUNSUPP; }
	uint32_t words = 0;      // bit (nword - 1): some band needs nword 64-bit words (the reference's cal_exz_infi picks nword = ceil((2 thre + 1) / 64), Correct.cpp:14508-14565)
	for (uint64_t i = 0; i < n_tasks; ++i) {      // the reference indexes its strings unchecked; a device kernel must not
		const hao_ed_task_t &t = tasks[i];
		if (t.p_rid >= c->n_reads || t.t_rid >= c->n_reads || (uint64_t)t.p_pos + t.p_len > c->h_len[t.p_rid] || (uint64_t)t.t_pos + t.t_len > c->h_len[t.t_rid] ||
			t.thre > HAO_ED_MAX_THRE || t.abs_diag > 2 * t.thre) { hao_set_err(c, "hao_window_ed_batch: task " + std::to_string(i) + " out of range"); return HAO_EINVAL; }
		const uint32_t nw = hao_al_nword(t.thre);
		if (nw > 1 && (int64_t)t.p_len - (int64_t)t.t_len + (int64_t)t.abs_diag > 64 * (int64_t)nw) {      // the final scan would read VP / VN bits beyond the band's words (the reference then indexes the neighbouring vectors of its bit_extz_t)
			hao_set_err(c, "hao_window_ed_batch: task " + std::to_string(i) + ": p_len - t_len + abs_diag beyond the band's words"); return HAO_EINVAL; }
		words |= 1u << (nw - 1);
	}
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_al_upload_sorted(c, tasks, n_tasks)) return rc;
	HIP_TRY(c->al_res.reserve(n_tasks));
	DevBuf<hao_ed_task_t> &dt = c->al_task; DevBuf<uint32_t> &order = c->al_order; DevBuf<hao_ed_result_t> &dr = c->al_res;
	const hao_ed_reads R = hao_al_reads_of(c);
	const dim3 g_((unsigned)((n_tasks + 255) / 256)), b_(256);
	if (words & 1u) { hipLaunchKernelGGL((hao_al_kernel<uint64_t, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt.p, order.p, n_tasks, (uint64_t*)nullptr, (uint64_t)0, dr.p, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 2u) { hipLaunchKernelGGL((hao_al_kernel<hao_u128, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt.p, order.p, n_tasks, (uint64_t*)nullptr, (uint64_t)0, dr.p, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 4u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<3>, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt.p, order.p, n_tasks, (uint64_t*)nullptr, (uint64_t)0, dr.p, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	if (words & 8u) { hipLaunchKernelGGL((hao_al_kernel<hao_wide<4>, HAO_AL_ED, false>), g_, b_, 0, c->stream, R, dt.p, order.p, n_tasks, (uint64_t*)nullptr, (uint64_t)0, dr.p, (hao_trace_result_t*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr, 0u); HAO_CHECK_LAUNCH(); }
	HIP_TRY(hipMemcpyAsync(out, dr.p, n_tasks * sizeof(hao_ed_result_t), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HAO_OK;
}

int hao_window_trace_batch(hao_ctx *c, int mode, const hao_ed_task_t *tasks, uint64_t n_tasks, hao_trace_result_t *out, uint16_t *cigars, uint32_t cigar_cap)
{
	if (!c || (mode < HAO_ALIGN_GLOBAL || mode > HAO_ALIGN_SEMI) || (!tasks && n_tasks) || (!out && n_tasks) || (!cigars && n_tasks && cigar_cap)) return HAO_EINVAL;
	if (int rc = hao_view_refresh(c)) return rc;
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_window_trace_batch needs the bases of both reads: single-device mode only"); return HAO_EUNSUPP; }
	c->al_grid_n = 0;
	if (n_tasks == 0) return HAO_OK;
	if (n_tasks >= (1ULL << 32)) { hao_set_err(c, "hao_window_trace_batch: more than 2^32 tasks in one call"); return HAO_EUNSUPP; }
	uint64_t tn_max = 1; uint32_t words = 0;      // bit (nword - 1): some band needs nword 64-bit words
	for (uint64_t i = 0; i < n_tasks; ++i) {      // the reference indexes its strings unchecked; a device kernel must not
		const hao_ed_task_t &t = tasks[i];
		if (t.p_rid >= c->n_reads || t.t_rid >= c->n_reads || (uint64_t)t.p_pos + t.p_len > c->h_len[t.p_rid] || (uint64_t)t.t_pos + t.t_len > c->h_len[t.t_rid] ||
			t.thre > HAO_ED_MAX_THRE) { hao_set_err(c, "hao_window_trace_batch: task " + std::to_string(i) + " out of range"); return HAO_EINVAL; }
		words |= 1u << (hao_al_nword(t.thre) - 1);
		if (mode == HAO_ALIGN_SEMI) {
			const int64_t ai = (int64_t)t.p_len - (int64_t)t.t_len + (int64_t)t.abs_diag;
			if (ai < 0 || ai > 2 * (int64_t)t.thre || t.t_len <= t.abs_diag || t.abs_diag > 2 * t.thre) { hao_set_err(c, "hao_window_trace_batch: task " + std::to_string(i) + ": the band does not cover the pattern"); return HAO_EINVAL; }
		}
		if (t.t_len > tn_max) tn_max = t.t_len;
	}
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_al_upload_sorted(c, tasks, n_tasks)) return rc;
	DevBuf<hao_ed_task_t> &dt = c->al_task; DevBuf<uint32_t> &order = c->al_order; DevBuf<hao_trace_result_t> &dr = c->al_tres; DevBuf<uint16_t> &dc = c->al_cig; DevBuf<uint8_t> &want = c->al_want;
	HIP_TRY(dr.reserve(n_tasks)); HIP_TRY(want.reserve(n_tasks)); HIP_TRY(dc.reserve(n_tasks * (uint64_t)cigar_cap + 1));
	HIP_TRY(hipMemsetAsync(want.p, 0, n_tasks, c->stream));
	if (cigar_cap) HIP_TRY(hipMemsetAsync(dc.p, 0, n_tasks * (uint64_t)cigar_cap * 2, c->stream));      // tasks without an alignment get no cigar: their rows read as zeros, not as an earlier call's entries
	const hao_ed_reads R = hao_al_reads_of(c);
	int rc;
	if (mode == HAO_ALIGN_EXT_FWD) rc = hao_al_trace_run<HAO_AL_EXT_FWD>(c, R, dt.p, order.p, n_tasks, words, tn_max, dr.p, want.p, dc.p, cigar_cap);
	else if (mode == HAO_ALIGN_EXT_BWD) rc = hao_al_trace_run<HAO_AL_EXT_BWD>(c, R, dt.p, order.p, n_tasks, words, tn_max, dr.p, want.p, dc.p, cigar_cap);
	else if (mode == HAO_ALIGN_SEMI) rc = hao_al_trace_run<HAO_AL_SEMI>(c, R, dt.p, order.p, n_tasks, words, tn_max, dr.p, want.p, dc.p, cigar_cap);
	else rc = hao_al_trace_run<HAO_AL_GLOBAL>(c, R, dt.p, order.p, n_tasks, words, tn_max, dr.p, want.p, dc.p, cigar_cap);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(out, dr.p, n_tasks * sizeof(hao_trace_result_t), hipMemcpyDeviceToHost, c->stream));
	if (cigar_cap) HIP_TRY(hipMemcpyAsync(cigars, dc.p, n_tasks * (uint64_t)cigar_cap * 2, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->al_path.cap > (1ULL << 27)) c->al_path.release();      // (more than 1 GB of column scratch is not kept between calls)
	if (c->al_cig.cap > (1ULL << 29)) c->al_cig.release();        // (nor more than 1 GB of cigar rows)
	return HAO_OK;
}

}
