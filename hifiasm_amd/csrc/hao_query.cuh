// K5-K7: seed lookup, anchor expansion, per-read ordering and k_mer_hit construction
// (minimizers_qgen0, anchor.cpp:987-1081) for gfx950.
//
// The reference materialises 24-byte anchors and radix-sorts them by
// (target id, strand, query pos) then by target offset.  Here an anchor is an 8-byte
// key that is also its own payload:
//
//     key = tid:28 | rev:1 | qidx:16 | jj:12        (bit 0 = LSB of jj)
//
// qidx = index of the query minimizer inside its read (strictly increasing with query
// position), jj = index inside the minimizer's hit list, reversed for opposite-strand
// hits.  Because a key's hit list is ordered by (rid,pos) and two hits of one k-mer in
// one target have k-mer starts ordered like their ends, sorting the keys numerically
// yields exactly the reference order (tid, strand, self_offset, other_off); the hit is
// rebuilt from (qidx, jj) afterwards.  Sorting moves 8 B instead of 24 B per anchor and
// only the bits that vary.
#pragma once
#include "hao_common.cuh"
#include "hao_index.cuh"

#define HAO_KEY_JJ_BITS 12
#define HAO_KEY_QI_BITS 16
#define HAO_KEY_REV_BIT 28
#define HAO_KEY_TID_SHIFT 29

// Q1: one thread per query minimizer of the batch: index lookup (ha_pt_get, anchor.cpp:1013)
__global__ void seed_count_kernel(const uint64_t *mz_x, uint64_t mz0, uint64_t n_mz, hao_pt_dev pt, uint64_t *s_start, uint32_t *s_n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n_mz) return;
	if (i == n_mz) { s_n[i] = 0; return; }
	uint64_t st = 0; uint32_t n = hao_pt_lookup(pt, mz_x[mz0 + i], &st);
	s_start[i] = st; s_n[i] = n;
}

// per-read anchor segment bounds from the per-minimizer scan
__global__ void seed_segments_kernel(const uint64_t *mz_off, uint64_t rid_lo, uint64_t n_sel, uint64_t mz0, const uint64_t *a_off, uint64_t *seg)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	seg[r] = a_off[mz_off[rid_lo + r] - mz0];
}

// Q2: one workgroup per read, one wave per minimizer: write the anchor keys
__global__ __launch_bounds__(256) void seed_expand_kernel(const uint64_t *mz_off, const uint64_t *mz_info, uint64_t rid_lo, uint64_t mz0,
		const uint64_t *s_start, const uint32_t *s_n, const uint64_t *a_off, const uint64_t *sinfo, uint64_t *keys, int *err)
{
	const uint64_t r = blockIdx.x, rid = rid_lo + r;
	const uint64_t m0 = mz_off[rid], m1 = mz_off[rid + 1];
	if (m1 - m0 > (1u << HAO_KEY_QI_BITS)) { if (threadIdx.x == 0) *err = 2; return; }
	for (uint64_t m = m0 + (threadIdx.x >> 6); m < m1; m += 4) {
		const uint64_t li = m - mz0; const uint32_t n = s_n[li];
		if (n == 0) continue;
		const uint64_t st = s_start[li], ao = a_off[li], z = mz_info[m]; const uint32_t zrev = hao_info_rev(z), qidx = (uint32_t)(m - m0);
		for (uint32_t j = hao_lane(); j < n; j += 64) {
			uint64_t y = sinfo[st + j]; uint32_t rev = zrev != hao_info_rev(y);
			uint32_t jj = rev ? n - 1 - j : j;
			keys[ao + j] = (uint64_t)hao_info_rid(y) << HAO_KEY_TID_SHIFT | (uint64_t)rev << HAO_KEY_REV_BIT | (uint64_t)qidx << HAO_KEY_JJ_BITS | jj;
		}
	}
}

// Q3: per-read stable LSD radix pass on one 8-bit digit of the key.  The keys of a read are generated in
// (qidx, j) order, so only the (tid, rev) bits need sorting (stable); the rare runs of equal (tid, rev=1, qidx)
// come out with jj descending and are mirrored in hits_build_kernel.  One workgroup per read, each wave owns a
// contiguous quarter of the segment; digit ranks inside a 64-key tile come from an 8-ballot match, per-wave
// digit counters live in LDS (no atomics: one leader lane per distinct digit), the keys themselves stream
// through L2 (a read's ~100 KB segment stays cache-resident between the count and the scatter sweep).
__device__ __forceinline__ unsigned long long hao_match8(uint32_t d, bool act)
{
	unsigned long long m = __ballot(act);
#pragma unroll
	for (int b = 0; b < 8; ++b) { unsigned long long bal = __ballot((d >> b) & 1); m &= ((d >> b) & 1) ? bal : ~bal; }
	return m;
}

__global__ __launch_bounds__(256) void seg_radix_pass_kernel(const uint64_t *in, uint64_t *out, const uint64_t *seg, int shift)
{
	__shared__ uint32_t cnt[4][256]; __shared__ uint32_t tot[256];
	const uint64_t r = blockIdx.x, s = seg[r], e = seg[r + 1]; const uint32_t n = (uint32_t)(e - s);
	const int wv = threadIdx.x >> 6, lane = hao_lane();
	if (n == 0) return;
	const uint32_t chunk = ((n + 3) / 4 + 63) & ~63u, c0 = min(n, wv * chunk), c1 = min(n, c0 + chunk);
	for (int i = threadIdx.x; i < 1024; i += 256) cnt[i >> 8][i & 255] = 0;
	__syncthreads();
	for (uint32_t t0 = c0; t0 < c1; t0 += 64) {
		const uint32_t i = t0 + lane; const bool act = i < c1;
		const uint32_t d = act ? (uint32_t)(in[s + i] >> shift) & 255u : 0;
		const unsigned long long m = hao_match8(d, act);
		if (act && (m & ((1ULL << lane) - 1)) == 0) cnt[wv][d] += __popcll(m);      // leader of its digit in this tile
	}
	__syncthreads();
	{	// exclusive offsets in (digit major, wave minor) order
		const int d = threadIdx.x; uint32_t run = 0;
		for (int x = 0; x < 4; ++x) { uint32_t c = cnt[x][d]; cnt[x][d] = run; run += c; }
		tot[d] = run;
		__syncthreads();
		uint32_t v = tot[d], tsum; uint32_t ex = hao_wave_excl_scan(v, &tsum);
		__shared__ uint32_t wsum[4];
		if (lane == 63) wsum[wv] = ex + v;
		__syncthreads();
		uint32_t add = 0; for (int x = 0; x < wv; ++x) add += wsum[x];
		ex += add;
		for (int x = 0; x < 4; ++x) cnt[x][d] += ex;
	}
	__syncthreads();
	for (uint32_t t0 = c0; t0 < c1; t0 += 64) {
		const uint32_t i = t0 + lane; const bool act = i < c1;
		const uint64_t key = act ? in[s + i] : 0; const uint32_t d = act ? (uint32_t)(key >> shift) & 255u : 0;
		const unsigned long long m = hao_match8(d, act);
		uint32_t base = act ? cnt[wv][d] : 0;
		if (act) out[s + base + __popcll(m & ((1ULL << lane) - 1))] = key;
		if (act && (m & ((1ULL << lane) - 1)) == 0) cnt[wv][d] = base + __popcll(m);
	}
}

// Q4: sorted key -> k_mer_hit (anchor.cpp:1055-1076). One workgroup per read.
__global__ __launch_bounds__(256) void hits_build_kernel(const uint64_t *keys, const uint64_t *seg, const uint64_t *mz_off, const uint64_t *mz_info, uint64_t rid_lo, uint64_t mz0,
		const uint64_t *s_start, const uint32_t *s_n, const uint64_t *sinfo, const uint32_t *len, const uint32_t *wgt_tab, hao_hit_t *hits, int mirror)
{
	const uint64_t r = blockIdx.x, rid = rid_lo + r, m0 = mz_off[rid];
	const uint64_t run_mask = ~(uint64_t)((1u << HAO_KEY_JJ_BITS) - 1);        // (tid, rev, qidx)
	for (uint64_t i = seg[r] + threadIdx.x; i < seg[r + 1]; i += 256) {
		uint64_t key = keys[i];
		if (mirror && (key >> HAO_KEY_REV_BIT & 1)) {      // opposite-strand hits of one minimizer in one target arrive in reverse list order: mirror the run
			uint64_t a = i, b = i;
			while (a > seg[r] && (keys[a - 1] & run_mask) == (key & run_mask)) --a;
			while (b + 1 < seg[r + 1] && (keys[b + 1] & run_mask) == (key & run_mask)) ++b;
			if (a != b) key = keys[a + b - i];
		}
		const uint32_t jj = (uint32_t)(key & ((1u << HAO_KEY_JJ_BITS) - 1)), qidx = (uint32_t)(key >> HAO_KEY_JJ_BITS & ((1u << HAO_KEY_QI_BITS) - 1));
		const uint32_t rev = (uint32_t)(key >> HAO_KEY_REV_BIT & 1), tid = (uint32_t)(key >> HAO_KEY_TID_SHIFT);
		const uint64_t m = m0 + qidx, li = m - mz0; const uint32_t n = s_n[li];
		const uint64_t z = mz_info[m], y = sinfo[s_start[li] + (rev ? n - 1 - jj : jj)];
		hao_hit_t h;
		h.w0 = tid | rev << 31;
		h.offset = rev ? len[tid] - 1 - (hao_info_pos(y) + 1 - hao_info_span(y)) : hao_info_pos(y);
		h.self_offset = hao_info_pos(z);
		h.cnt = wgt_tab[n] << 8 | hao_info_span(z);
		hits[i] = h;
	}
}

// Q5: target groups of each read (hits are sorted by target id inside a read).
// pass 0: count groups per read; pass 1: write group starts at g_off[r] + rank.
__global__ __launch_bounds__(256) void groups_kernel(const hao_hit_t *hits, const uint64_t *seg, uint64_t n_sel, const uint64_t *g_off, uint64_t *g_cnt, uint64_t *g_start, uint32_t *g_read, int pass)
{
	const uint64_t r = blockIdx.x; const uint64_t s = seg[r], e = seg[r + 1];
	__shared__ uint32_t s_w[4]; __shared__ uint64_t s_run;
	if (threadIdx.x == 0) s_run = 0;
	__syncthreads();
	for (uint64_t b = s; b < e; b += 256) {
		uint64_t i = b + threadIdx.x; bool st = false;
		if (i < e) st = (i == s) || ((hits[i].w0 & 0x7fffffffu) != (hits[i - 1].w0 & 0x7fffffffu));
		unsigned long long bal = __ballot(st);
		if (hao_lane() == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
		__syncthreads();
		uint32_t before = 0; for (int x = 0; x < (int)(threadIdx.x >> 6); ++x) before += s_w[x];
		uint32_t rank = before + __popcll(bal & ((1ULL << hao_lane()) - 1));
		if (pass == 1 && st) { uint64_t g = g_off[r] + s_run + rank; g_start[g] = i; g_read[g] = (uint32_t)r; }
		__syncthreads();
		if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
		__syncthreads();
	}
	if (pass == 0 && threadIdx.x == 0) g_cnt[r] = s_run;
	if (pass == 0 && r == 0 && threadIdx.x == 1) g_cnt[n_sel] = 0;
}
