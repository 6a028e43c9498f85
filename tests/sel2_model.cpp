// Executable model of sketch_select2_kernel (hifiasm_amd/csrc/hao_select2.cuh) for the CPU test suite: the element functions of that file compiled by g++,
// the kernel's phases walked in the kernel's order with a loop over the 64 lanes where the device has a wave (lanes only meet through the arrays between
// phases).  TEST INFRASTRUCTURE: built and called by tests/test_select_model_cpu.py, compared there with the oracle.
#define HAO_SEL2_HOST_MODEL
#include "hao_select2.cuh"
#include <vector>
#include <string.h>

// in: the candidate list (x, info with the count in the rid field, ord), read length, tot_l; out: kept[] (indices, ascending).  Returns the number kept, or
// -2 if it has no high-count candidate; -1 / -3 / -4 if the read is outside the closed form's reach (more than HAO_S2_CAP candidates / ordinals that restart / a window of more
// than 1 << HAO_S2_LOG candidates): the kernel then runs the sequential routine.
extern "C" int hao_sel2_model_cap(uint64_t *x, uint64_t *info, uint32_t *ord, int n, int len, int tot_l, int sample_dist, int rewin, int k, int32_t *kept, int CAP)
{
	bool any = false; for (int i = 0; i < n; ++i) if ((info[i] & 0xfffffffu) > 0) any = true;
	if (!any) return -2;
	if (n > CAP) return -1;
	std::vector<uint16_t> idx(CAP), rank(CAP), start(CAP), wm(CAP), mn((HAO_S2_LOG + 1) * CAP), mx((HAO_S2_LOG + 1) * CAP);
	std::vector<uint8_t> flag(CAP);
	hao_s2_view V; V.x = x; V.info = info; V.ord = ord; V.idx = idx.data(); V.rank = rank.data(); V.start = start.data(); V.wm = wm.data(); V.mn = mn.data(); V.mx = mx.data(); V.flag = flag.data();
	V.n = n; V.cap = CAP; V.len = len; V.sample_dist = sample_dist; V.w = rewin; V.k = k; V.tot_l = tot_l;
	int P = 64; while (P < n) P <<= 1; V.P = P;
	int s_i0 = n, s_bad = 0, s_anyq = 0;
	for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) {
		if (i > 0 && ord[i] < ord[i - 1]) s_bad = 1;
		if (hao_s2_cnt(V, i) > 0 && (i == 0 || hao_s2_cnt(V, i - 1) == 0)) { int e, span; if (hao_s2_run(V, i, e, span) > 0) s_anyq = 1; }
		if (hao_s2_first_window(V, i) && i < s_i0) s_i0 = i;
	}
	if (s_bad) return -3;
	if (!s_anyq) { for (int i = 0; i < n; ++i) kept[i] = i; return n; }
	const int i0 = s_i0 < n ? s_i0 : -1;
	if (i0 >= 0) {
		for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < P; i += 64) idx[i] = i < n ? (uint16_t)i : (uint16_t)HAO_S2_PAD;
		for (int kk = 2; kk <= P; kk <<= 1)
			for (int j = kk >> 1; j > 0; j >>= 1) for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < P; i += 64) hao_s2_bitonic(V, i, j, kk);
		for (int lane = 0; lane < 64; ++lane) for (int p = lane; p < P; p += 64) hao_s2_rank(V, p);
		for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) { hao_s2_start(V, i); mn[i] = rank[i]; }
		for (int L = 1; L <= HAO_S2_LOG; ++L) for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) hao_s2_level(V.mn, CAP, n, L, i, false);
		int bad = 0;
		for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) { if (!hao_s2_window_min(V, i, i0)) bad = 1; mx[i] = wm[i]; }
		if (bad) return -4;
		for (int L = 1; L <= HAO_S2_LOG; ++L) for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) hao_s2_level(V.mx, CAP, n, L, i, true);
		const int s_last = n - 1 > i0 ? (int)start[n - 1] : 0, tail_hi = hao_s2_tail_hi(V, s_last);
		for (int lane = 0; lane < 64; ++lane) for (int j = lane; j < n; j += 64) hao_s2_mark(V, j, i0, s_last, tail_hi);
	} else for (int j = 0; j < n; ++j) flag[j] = 0;
	for (int lane = 0; lane < 64; ++lane) for (int i = lane; i < n; i += 64) {
		if (hao_s2_cnt(V, i) == 0) flag[i] |= 2;
		else if (i0 >= 0 && (i == 0 || hao_s2_cnt(V, i - 1) == 0)) { int e, span; const int q = hao_s2_run(V, i, e, span); if (q > 0) hao_s2_finish_run(V, i, e, span, q); }
	}
	int m = 0;
	for (int i = 0; i < n; ++i) if (flag[i] & 2) kept[m++] = i;
	return m;
}

// the kernel's two instantiations: a read goes to the smaller capacity that holds it
extern "C" int hao_sel2_model(uint64_t *x, uint64_t *info, uint32_t *ord, int n, int len, int tot_l, int sample_dist, int rewin, int k, int32_t *kept)
{ return hao_sel2_model_cap(x, info, ord, n, len, tot_l, sample_dist, rewin, k, kept, n <= HAO_S2_CAP_SMALL ? HAO_S2_CAP_SMALL : HAO_S2_CAP); }
