// High-count thinning of a read's minimizer candidates (mz1_select_mz_h, sketch.cpp:247-330; mz1_qfw :226-246; mz1_hf_select :194-216), data-parallel.
//
// The reference (and hao_select_high, hao_sketch.cuh: one LANE per read) replays a newest-wins state machine over the candidate list.  What that machine computes
// (tests/sel_model.py derives it, tests/test_select_model_cpu.py checks it against the oracle): a high-count candidate is MARKED iff its (count, hash) key
// equals the maximum, over the second-level windows that contain it, of the window's minimum key - windows W_i = { m <= i : ord[m] + w > ord[i] } for every
// entry i from the first full window on (that first window is [0, i0]), plus the tail windows [s, n - 1] while ord[s] + w <= tot_l + 1.  Marked candidates
// of a run (a stretch of high-count candidates) long enough to be sampled survive; a run without any takes its min(16, q) smallest through the reference's
// heap; candidates that are not high-count always survive.
// Here one WAVE owns a read (its <= 1024 candidates in LDS; reads of <= 512 candidates - a 15 kb read has ~430 - run in an instantiation with half the arrays: 5 waves per CU instead of 2): keys are replaced by their rank (bitonic sort of 16-bit indices), window starts are a binary
// search per entry, window minima a sparse-table range-min, the "maximum of the minima of my windows" a second sparse table, runs are walked by the lane
// that owns their first entry, survivors are compacted by ballots.  Reads the closed form does not cover - more than 1024 candidates, k-mer ordinals that
// restart (N bases), a window of more than 128 candidates - take the sequential routine.
// tests/sel2_model.cpp compiles the element functions below with g++ and walks the kernel's phases with loops, tests/test_select_model_cpu.py compares with the
// oracle on the CPU; tests/test_gpu_zz_new.py compares the kernel with the oracle and the reference's digests on the device.
#pragma once
#include <stdint.h>
#ifdef HAO_SEL2_HOST_MODEL
#define HAO_S2_FN static inline
#else
#include "hao_common.cuh"
#define HAO_S2_FN __host__ __device__ __forceinline__
#endif

#define HAO_S2_CAP 1024         // most candidates of a read the closed form takes (two launches: reads of up to HAO_S2_CAP_SMALL candidates in 30 KB of LDS, the longer ones in 61 KB)
#define HAO_S2_CAP_SMALL 512
#define HAO_S2_LOG 7            // windows of up to 1 << HAO_S2_LOG candidates
#define HAO_S2_ZERO 0xfffeu     // rank of every candidate that is not high-count: above every real rank
#define HAO_S2_PAD 0xffffu      // index of a padding slot of the sort

struct hao_s2_view {
	uint64_t *x, *info; uint32_t *ord;                 // the read's candidates, info.rid = filter-table count (0: not high-count); survivors are compacted in place
	uint16_t *idx, *rank, *start, *wm, *mn, *mx;       // [CAP] sort permutation, rank, window start, window minimum; [(LOG + 1) * CAP] sparse tables (min of rank, max of wm)
	uint8_t *flag;                                     // [CAP] bit 0: marked, bit 1: kept
	int n, P, cap;                                     // candidates; P = power of two >= n (sort width); cap = capacity of the arrays = stride of the sparse tables' levels
	int len, sample_dist, w, k, tot_l;
};
HAO_S2_FN uint32_t hao_s2_cnt(const hao_s2_view &V, int i) { return (uint32_t)(V.info[i] & 0xfffffffu); }
HAO_S2_FN int32_t hao_s2_pos(const hao_s2_view &V, int i) { return (int32_t)((V.info[i] >> 28) & 0x7ffffffu); }

// ---- phase 1: ranks ----
// sort key of slot content a (a candidate index, or HAO_S2_PAD): (class, count, hash); class 0 high-count, 1 the others (all equal), 2 padding
HAO_S2_FN void hao_s2_key(const hao_s2_view &V, uint32_t a, uint32_t &c, uint64_t &x)
{
	if (a == HAO_S2_PAD) { c = 0x20000000u; x = 0; return; }
	const uint32_t cn = hao_s2_cnt(V, (int)a);
	if (cn == 0) { c = 0x10000000u; x = 0; } else { c = cn; x = V.x[a]; }
}
HAO_S2_FN bool hao_s2_key_gt(const hao_s2_view &V, uint32_t a, uint32_t b)
{ uint32_t ca, cb; uint64_t xa, xb; hao_s2_key(V, a, ca, xa); hao_s2_key(V, b, cb, xb); return ca > cb || (ca == cb && xa > xb); }
HAO_S2_FN bool hao_s2_key_eq(const hao_s2_view &V, uint32_t a, uint32_t b)
{ uint32_t ca, cb; uint64_t xa, xb; hao_s2_key(V, a, ca, xa); hao_s2_key(V, b, cb, xb); return ca == cb && xa == xb; }
// one compare-exchange of the bitonic network: slot i with its partner i ^ j in the stage of width k (callers pass every i < P once per (k, j), then fence)
HAO_S2_FN void hao_s2_bitonic(const hao_s2_view &V, int i, int j, int k)
{
	const int p = i ^ j;
	if (p > i) { const uint16_t a = V.idx[i], b = V.idx[p]; if (hao_s2_key_gt(V, a, b) == ((i & k) == 0)) { V.idx[i] = b; V.idx[p] = a; } }
}
// rank of the candidate at sorted slot p: 1 + the first slot that holds an equal key (equal keys share a rank; ties are repeats of one k-mer: short)
HAO_S2_FN void hao_s2_rank(const hao_s2_view &V, int p)
{
	const uint32_t a = V.idx[p];
	if (a == HAO_S2_PAD) return;
	if (hao_s2_cnt(V, (int)a) == 0) { V.rank[a] = HAO_S2_ZERO; return; }
	int r = p; while (r > 0 && hao_s2_key_eq(V, V.idx[r - 1], a)) --r;
	V.rank[a] = (uint16_t)(r + 1);
}

// ---- phase 2: windows ----
// first m <= i with ord[m] + w > ord[i] (ordinals ascend along the list: checked by the caller)
HAO_S2_FN void hao_s2_start(const hao_s2_view &V, int i)
{
	int lo = 0, hi = i; const int64_t oi = (int64_t)V.ord[i];
	while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)V.ord[m] + V.w > oi) hi = m; else lo = m + 1; }
	V.start[i] = (uint16_t)lo;
}
// does entry i close the first full second-level window (mz1_qfw's stop condition, sketch.cpp:226-246)?
HAO_S2_FN bool hao_s2_first_window(const hao_s2_view &V, int i)
{
	const int64_t ws = (int64_t)V.w + V.k - 1, oi = (int64_t)V.ord[i];
	return oi >= ws || (i + 1 < V.n && oi < ws && (int64_t)V.ord[i + 1] > ws) || (i + 1 == V.n && (int64_t)V.tot_l >= ws && oi < ws);
}
// sparse tables: level L (>= 1) of table t from level L - 1; op = min (is_max false) / max
HAO_S2_FN void hao_s2_level(uint16_t *t, int cap, int n, int L, int i, bool is_max)
{
	const int h = 1 << (L - 1); const uint16_t a = t[(L - 1) * cap + i], b = i + h < n ? t[(L - 1) * cap + i + h] : a;
	t[L * cap + i] = is_max ? (a > b ? a : b) : (a < b ? a : b);
}
HAO_S2_FN uint16_t hao_s2_query(const uint16_t *t, int cap, int a, int b, bool is_max)      // over [a, b], b - a + 1 <= 1 << HAO_S2_LOG
{
	int j = 0; while ((2 << j) <= b - a + 1) ++j;
	const uint16_t u = t[j * cap + a], v = t[j * cap + b - (1 << j) + 1];
	return is_max ? (u > v ? u : v) : (u < v ? u : v);
}
HAO_S2_FN int hao_s2_wstart(const hao_s2_view &V, int i, int i0) { return i == i0 ? 0 : (int)V.start[i]; }
// minimum rank of the window that ends at entry i (0: i closes no window); returns false if the window is longer than the tables reach
HAO_S2_FN bool hao_s2_window_min(const hao_s2_view &V, int i, int i0)
{
	if (i0 < 0 || i < i0) { V.wm[i] = 0; return true; }
	const int a = hao_s2_wstart(V, i, i0);
	if (i - a + 1 > (1 << HAO_S2_LOG)) { V.wm[i] = 0; return false; }
	V.wm[i] = hao_s2_query(V.mn, V.cap, a, i, false);
	return true;
}
// last start s of a tail window [s, n - 1] (s_last - 1: none): tail windows exist while ord[s] + w <= tot_l + 1
HAO_S2_FN int hao_s2_tail_hi(const hao_s2_view &V, int s_last)
{
	int lo = s_last, hi = V.n;      // first t >= s_last with ord[t] + w > tot_l + 1
	while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)V.ord[m] + V.w > (int64_t)V.tot_l + 1) hi = m; else lo = m + 1; }
	return lo - 1;
}
// is high-count entry j the minimum (ties included) of one of the windows that contain it?
HAO_S2_FN void hao_s2_mark(const hao_s2_view &V, int j, int i0, int s_last, int tail_hi)
{
	V.flag[j] = 0;
	if (hao_s2_cnt(V, j) == 0 || i0 < 0) return;
	const int lo = j > i0 ? j : i0;
	int a = lo, b = V.n - 1, hi = lo - 1;      // last i >= lo whose window still starts at or before j
	while (a <= b) { const int m = (a + b) >> 1; if (hao_s2_wstart(V, m, i0) <= j) { hi = m; a = m + 1; } else b = m - 1; }
	uint16_t cand = hi >= lo ? hao_s2_query(V.mx, V.cap, lo, hi, true) : 0;
	if (tail_hi >= s_last && j >= s_last) { const uint16_t t = hao_s2_query(V.mn, V.cap, j < tail_hi ? j : tail_hi, V.n - 1, false); if (t > cand) cand = t; }
	if (cand == V.rank[j]) V.flag[j] = 1;
}

// ---- phase 3: runs ----
// [i, e) = the run of high-count entries that starts at i (cnt[i] > 0 and i == 0 or cnt[i - 1] == 0): its bounding positions and sampling quota
HAO_S2_FN int hao_s2_run(const hao_s2_view &V, int i, int &e, int &span)
{
	e = i; while (e < V.n && hao_s2_cnt(V, e) > 0) ++e;
	const int ps = i == 0 ? 0 : hao_s2_pos(V, i - 1), pe = e == V.n ? V.len : hao_s2_pos(V, e);
	span = pe - ps;
	return (int)((double)(pe - ps) / V.sample_dist + .499);
}
struct hao_s2_hent { uint64_t x; uint32_t c; int idx; };
HAO_S2_FN bool hao_s2_hent_lt(const hao_s2_hent &a, const hao_s2_hent &b) { return a.c < b.c || (a.c == b.c && a.x < b.x); }
HAO_S2_FN void hao_s2_heap_down(int i, int n, hao_s2_hent *l)      // ksort.h:43-52 (max-heap on (count, hash)): its sift order decides which of several equal keys stays
{
	int k = i; hao_s2_hent tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && hao_s2_hent_lt(l[k], l[k + 1])) ++k;
		if (hao_s2_hent_lt(l[k], tmp)) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}
// the run [i, e) of quota q > 0: its marked entries are kept; without any, the min(16, q) smallest (count, hash) of the run that are rarer than the run is long (mz1_hf_select)
HAO_S2_FN void hao_s2_finish_run(const hao_s2_view &V, int i, int e, int span, int q)
{
	int nm = 0;
	for (int m = i; m < e; ++m) if (V.flag[m] & 1) { V.flag[m] |= 2; ++nm; }
	if (nm || e - i < 1) return;
	if (q > 16) q = 16;
	hao_s2_hent b[16]; int j, kk;
	for (j = i, kk = 0; j < e && kk < q; ++j, ++kk) { b[kk].x = V.x[j]; b[kk].c = hao_s2_cnt(V, j); b[kk].idx = j; }
	for (int t = (kk >> 1) - 1; t >= 0; --t) hao_s2_heap_down(t, kk, b);
	for (; j < e; ++j) { hao_s2_hent h; h.x = V.x[j]; h.c = hao_s2_cnt(V, j); h.idx = j; if (hao_s2_hent_lt(h, b[0])) { b[0] = h; hao_s2_heap_down(0, kk, b); } }
	for (j = 0; j < kk; ++j) if ((int)b[j].c < span) V.flag[b[j].idx] |= 2;
}

#ifndef HAO_SEL2_HOST_MODEL
// ---------------------------------------------------------------------------------------
// One wave per read (blocks of 64 threads; ~61 KB of LDS).  Reads outside the closed form's reach run
// hao_select_high (hao_sketch.cuh) on lane 0, as that kernel does.
// ---------------------------------------------------------------------------------------
// the sequential routine for the reads the closed form does not cover: kept out of line so that its registers are not the kernel's
__device__ __attribute__((noinline)) int hao_s2_sequential(uint64_t *x, uint64_t *info, uint32_t *ord, int n, int len, int sample_dist, int rewin, int k, int tot_l)
{ hao_sel_view v; v.n = n; v.x = x; v.info = info; v.ord = ord; return hao_select_high(v, len, sample_dist, rewin, k, tot_l); }
// NT threads per read: 64 (one wave, fences only) for the small instantiation; the 61 KB instantiation runs 256 threads per read - at two workgroups per CU one
// wave per read left the GPU with 512 waves in flight (125 ms for the 250 Mb repeat-rich set, where half of the reads have more than 512 candidates)
template<int NT> __device__ __forceinline__ void hao_s2_sync() { if (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } else __syncthreads(); }
template<int NT> __device__ __forceinline__ bool hao_s2_any(bool v) { if (NT == 64) return __any(v) != 0; return __syncthreads_or(v ? 1 : 0) != 0; }
template<int CAP, int NT>      // candidates per read this instantiation stages: reads of (CAP / 2, CAP] (the smallest CAP: [0, CAP]; the largest CAP also takes the longer reads, sequentially)
__global__ __launch_bounds__(NT) void sketch_select2_kernel(uint64_t *x, uint64_t *info, uint32_t *ord, const uint64_t *mz_off, const uint32_t *len, const uint32_t *tot_l,
		uint64_t rid_lo, uint64_t n_sel, int sample_dist, int rewin, int k, uint32_t *new_n, const int *err)
{
	if (*err) return;
	__shared__ uint64_t l_x[CAP], l_info[CAP]; __shared__ uint32_t l_ord[CAP];
	__shared__ uint16_t l_idx[CAP], l_rank[CAP], l_start[CAP], l_wm[CAP], l_mn[(HAO_S2_LOG + 1) * CAP], l_mx[(HAO_S2_LOG + 1) * CAP];
	__shared__ uint8_t l_flag[CAP]; __shared__ int s_i0, s_bad, s_anyq, s_m;
	const int tid = threadIdx.x;
	const uint64_t r = blockIdx.x;
	if (r >= n_sel) return;
	const uint64_t o = mz_off[r]; const int n = (int)(mz_off[r + 1] - o);
	if ((CAP < HAO_S2_CAP && n > CAP) || (CAP > HAO_S2_CAP_SMALL && n <= CAP / 2)) return;      // the other instantiation's read
	bool any = false;
	for (int i = tid; i < n; i += NT) if ((info[o + i] & 0xfffffffu) > 0) any = true;
	if (!hao_s2_any<NT>(any)) { if (tid == 0) new_n[r] = (uint32_t)n; return; }
	const bool in_lds = n <= CAP;
	if (in_lds) for (int i = tid; i < n; i += NT) { l_x[i] = x[o + i]; l_info[i] = info[o + i]; l_ord[i] = ord[o + i]; }
	if (tid == 0) { s_i0 = n; s_bad = in_lds ? 0 : 1; s_anyq = 0; s_m = 0; }
	hao_s2_sync<NT>();
	hao_s2_view V; V.x = l_x; V.info = l_info; V.ord = l_ord; V.idx = l_idx; V.rank = l_rank; V.start = l_start; V.wm = l_wm; V.mn = l_mn; V.mx = l_mx; V.flag = l_flag;
	V.n = n; V.cap = CAP; V.len = (int)len[rid_lo + r]; V.sample_dist = sample_dist; V.w = rewin; V.k = k; V.tot_l = (int)tot_l[r];
	int P = 64; while (P < n) P <<= 1; V.P = P;
	if (in_lds) {
		// ordinals must ascend; the runs' quotas; the first full window
		for (int i = tid; i < n; i += NT) {
			if (i > 0 && l_ord[i] < l_ord[i - 1]) s_bad = 1;
			if (hao_s2_cnt(V, i) > 0 && (i == 0 || hao_s2_cnt(V, i - 1) == 0)) { int e, span; if (hao_s2_run(V, i, e, span) > 0) s_anyq = 1; }
			if (hao_s2_first_window(V, i)) atomicMin(&s_i0, i);
		}
		hao_s2_sync<NT>();
	}
	// the sequential routine (hao_select_high), by thread 0 (in place on the staged list, or on the global arrays of a very long read); survivors copied back by all
	auto sequential = [&](bool staged) {
		if (tid == 0) s_m = staged ? hao_s2_sequential(l_x, l_info, l_ord, n, V.len, sample_dist, rewin, k, V.tot_l) : hao_s2_sequential(x + o, info + o, ord + o, n, V.len, sample_dist, rewin, k, V.tot_l);
		hao_s2_sync<NT>();
		const int m = s_m;
		if (staged) for (int i = tid; i < m; i += NT) { x[o + i] = l_x[i]; info[o + i] = l_info[i]; }
		if (tid == 0) new_n[r] = (uint32_t)m;
	};
	if (s_bad) { sequential(in_lds); return; }
	if (!s_anyq) { if (tid == 0) new_n[r] = (uint32_t)n; return; }      // no run is long enough to be sampled: everything stays (sketch.cpp:266)
	const int i0 = s_i0 < n ? s_i0 : -1;
	if (i0 >= 0) {
		// ranks: bitonic sort of the candidate indices by key, then the first slot of every key
		for (int i = tid; i < P; i += NT) l_idx[i] = i < n ? (uint16_t)i : (uint16_t)HAO_S2_PAD;
		hao_s2_sync<NT>();
		for (int kk = 2; kk <= P; kk <<= 1)
			for (int j = kk >> 1; j > 0; j >>= 1) { for (int i = tid; i < P; i += NT) hao_s2_bitonic(V, i, j, kk); hao_s2_sync<NT>(); }
		for (int p = tid; p < P; p += NT) hao_s2_rank(V, p);
		hao_s2_sync<NT>();
		// window starts, the sparse table of ranks, window minima, their sparse table
		for (int i = tid; i < n; i += NT) { hao_s2_start(V, i); l_mn[i] = l_rank[i]; }
		hao_s2_sync<NT>();
		for (int L = 1; L <= HAO_S2_LOG; ++L) { for (int i = tid; i < n; i += NT) hao_s2_level(l_mn, CAP, n, L, i, false); hao_s2_sync<NT>(); }
		bool bad = false;
		for (int i = tid; i < n; i += NT) { if (!hao_s2_window_min(V, i, i0)) bad = true; l_mx[i] = l_wm[i]; }
		if (hao_s2_any<NT>(bad)) { hao_s2_sync<NT>(); sequential(true); return; }      // a window beyond the tables: sequential routine (the staged list is untouched so far)
		hao_s2_sync<NT>();
		for (int L = 1; L <= HAO_S2_LOG; ++L) { for (int i = tid; i < n; i += NT) hao_s2_level(l_mx, CAP, n, L, i, true); hao_s2_sync<NT>(); }
		const int s_last = n - 1 > i0 ? (int)l_start[n - 1] : 0, tail_hi = hao_s2_tail_hi(V, s_last);
		for (int j = tid; j < n; j += NT) hao_s2_mark(V, j, i0, s_last, tail_hi);
	} else for (int j = tid; j < n; j += NT) l_flag[j] = 0;
	hao_s2_sync<NT>();
	// runs: the thread that owns a run's first entry decides for the run; entries that are not high-count always stay
	for (int i = tid; i < n; i += NT) {
		if (hao_s2_cnt(V, i) == 0) l_flag[i] |= 2;
		else if (i0 >= 0 && (i == 0 || hao_s2_cnt(V, i - 1) == 0)) { int e, span; const int q = hao_s2_run(V, i, e, span); if (q > 0) hao_s2_finish_run(V, i, e, span, q); }
	}
	hao_s2_sync<NT>();
	// survivors, in order (the first wave)
	if (tid < 64) {
		int m = 0;
		for (int b = 0; b < n; b += 64) {
			const int i = b + tid; const bool kp = i < n && (l_flag[i] & 2);
			const unsigned long long bal = __ballot(kp);
			if (kp) { const int d = m + __popcll(bal & ((1ULL << tid) - 1)); x[o + d] = l_x[i]; info[o + d] = l_info[i]; }
			m += __popcll(bal);
		}
		if (tid == 0) new_n[r] = (uint32_t)m;
	}
}
#endif
