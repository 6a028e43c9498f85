// Seed stage by MERGE (K5-K7 in one pass; minimizers_qgen0, anchor.cpp:987-1081): one wave per read, no hash table, no counting pass, no staging.
//
// The reference orders a read's seed hits by (target id, strand, query position, target offset) - anchor.cpp:1011-1076 materialises them and radix-sorts.  Two facts
// of the position index make that order computable in ONE walk:
//   * every position list is sorted by (rid, pos) (ha_pt_gen, htab.cpp:380-460), so the lists of a read's minimizers are R sorted runs over the same key (rid);
//   * inside a (target, strand) bin the order is (query minimizer, list order) - hao_query.cuh's header - and minimizers are already in query order.
// So the hits of a read are the R-way MERGE of its lists by target, written target by target: a wave keeps one cursor per list (row), RPL rows per lane
// (row = block * 64 + lane, so that row order = (block, lane) order); a step takes the smallest target T under any cursor (one wave-min), every row whose head record
// belongs to T emits it - forward-strand hits first, then the opposite strand, each ranked by ballot + mbcnt in row order - and advances.  The output position is a
// running counter (bins come out in ascending (target, strand) order, which IS the layout of the read's segment of hits[]), the group list (one entry per target)
// is written as the targets go by, and every index record is read exactly once.  Stores of a step are consecutive 16-byte hits from consecutive active lanes.
//
// A row with SEVERAL records of one target (a k-mer twice in a target: rare) is found after the fact - the next minimum equals T again - and the step is redone by
// the general routine (per-row runs read from the lists, prefix sums over rows, opposite-strand records of a run in reverse list order: anchor.cpp:1023).
//
// Rows: the read's minimizers that have a list (compacted).  A read with more than 64 * RPL of them, or with more than 4096 minimizers, is left to the table
// kernels (hao_query.cuh, hao_query3.cuh) through the overflow list; so is a read with more than max_n seed hits, and the host gives whole batches whose reads
// average more than 14 000 hits to the table kernels (reads that cross repeat families meet hundreds of targets, a step each: the tables are faster there;
// hao_batch.hpp).  The id 2^28 - 1 is the merge's end mark: a read set that uses it takes the tables.
//
// HBM traffic per anchor: 8 bytes in (once), 16 bytes out; LDS: 12 bytes per row (the two query words of the row's hits, its list length, its minimizer); no barriers.
// The rows' registers are named variables (ROW(i) below), not arrays: the compiler turned arrays under this control flow into register tuples it copied whole.
#pragma once
#include "hao_query.cuh"

#define HAO_MRG_SENT (~0ULL)                 // an exhausted row: rid = 2^28 - 1 (no read has it: n_total < 2^28), so it never wins the minimum before the end
#define HAO_MRG_END 0xfffffffu
#define HAO_MRG_MAXRPL 8

template<int RPL> struct hao_seed4_lds {      // dynamic LDS per wave: q words uint2[64 * RPL], list length u16[64 * RPL], minimizer index u16[64 * RPL]
	static_assert(RPL >= 1 && RPL <= HAO_MRG_MAXRPL, "rows per lane");
	static constexpr uint32_t ROWS = 64u * RPL, PER_WAVE = ROWS * 12, TOTAL = 4 * PER_WAVE;
};

__device__ __forceinline__ uint32_t hao_mbcnt(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// one hit (anchor.cpp:1021-1023, 1059-1076): y = the index record with the HIT's strand in bit 55; offset = target coordinate in the strand of the hit
__device__ __forceinline__ hao_hit_t hao_mrg_hit(uint64_t y, uint32_t T, uint32_t rv, uint32_t tlen, uint2 qw)
{
	hao_hit_t h; h.w0 = T | rv << 31;
	h.offset = rv ? tlen - 1 - (hao_info_pos(y) + 1 - hao_info_span(y)) : hao_info_pos(y);
	h.self_offset = qw.x; h.cnt = qw.y;
	return h;
}
struct hao_rec4 { uint64_t a, b, c, d; };      // four index records: one 32-byte read of a row's list
struct hao_rec8 { uint64_t a, b, c, d, e, f, g, h; };      // eight: one aligned 64-byte block

// every row i < RPL: ROWS_DO(X) expands X(0) ... X(7) under `if constexpr`
#define HAO_MRG_ROWS_DO(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// a row's bookkeeping word: records of the list not loaded yet (bits 0-11), records in the buffer behind the head (bits 12-15), strand of the minimizer (bit 31)
#define HAO_MRG_REM(rz) ((rz) & 0xfffu)
#define HAO_MRG_BC(rz) ((rz) >> 12 & 0xfu)
#define HAO_MRG_Z(rz) ((rz) >> 31)

// Launch order.  The reads of a batch come from all over the genome, and the ~5 of them that contain a given k-mer (a batch is a sixth of the reads at 30x) walk
// the same position list - each from its own wave, at its own time, on any XCD: the list crosses the memory fabric once per read that uses it.  Reads of one locus
// share most of their lists AND most of their targets, so started together on one XCD they ask for the same lines at about the same time and the XCD's L2 answers.
// A read's locus is not known, but its smallest target is (what the merge's first step computes): reads with the same smallest target X overlap X, i.e. lie within
// a read length of each other, and X's position in the hit orders them along X.  seed_locus_kernel writes key = (smallest target << 27 | its position) per read;
// the host sorts the batch's reads by key and the merge kernel takes them in that order, an eighth of the sorted list per XCD (block b runs on XCD b % 8).
__global__ __launch_bounds__(256) void seed_locus_kernel(hao_seed_args S, const uint64_t *__restrict__ sinfo, uint32_t max_q, uint64_t *key, uint32_t *idx)
{
	const int wv = threadIdx.x >> 6, lane = hao_lane();
	const uint64_t r = (uint64_t)blockIdx.x * 4 + wv;
	if (r >= S.n_sel) return;
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	uint64_t best = ~0ULL;
	const uint32_t nqs = max_q && max_q < nq ? max_q : nq;      // (max_q: only the read's first minimizers - its first two kilobases - are asked: 64 of them cost one load per lane)
	for (uint32_t q = lane; q < nqs; q += 64)
		if (S.s_n[li0 + q]) { const uint64_t y = sinfo[S.s_start[li0 + q]]; best = min(best, (uint64_t)hao_info_rid(y) << 27 | hao_info_pos(y)); }
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) best = min(best, (uint64_t)__shfl_xor((unsigned long long)best, d));
	if (lane == 0) { key[r] = best; idx[r] = (uint32_t)r; }
}

// BUF = 1: a row reads its list one 8-byte record at a time; BUF = 4: 32 bytes at a time.  A lane's reads are its own (the lists of a read's minimizers lie anywhere
// in the index), so every read moves a whole cache line through the memory fabric however little of it is used - and a line comes around again only after the other
// ~500 rows of every wave of the XCD had their turn, by when the 4 MB L2 has lost it: with 8-byte reads a 128-byte line crosses the fabric up to 16 times
// (measured: 122 ms per configs[2] pass against 59 ms for the table kernels), with 32-byte reads 4 times (54.7 ms: the default).  AL: the 32-byte reads on 32-byte
// boundaries (measured slower: a row's first read then brings fewer records).  What the variants measured, side by side: profiles/r05/seed_ab.txt, DESIGN 5.
template<int RPL, int BUF, bool AL = false>
__global__ __launch_bounds__(256, BUF == 4 ? (RPL == 8 ? 3 : 5) : 4) void seed_merge_kernel(hao_seed_args S, const uint64_t *__restrict__ sinfo, const uint32_t *__restrict__ len, const uint32_t *__restrict__ order, uint32_t max_n, uint32_t *ovf_list, unsigned long long *ovf_cnt)
{
	static_assert(BUF == 1 || BUF == 4, "records per list read");
	constexpr uint32_t ROWS = hao_seed4_lds<RPL>::ROWS, mrg_row0 = 0;      // (mrg_row0: first row of the wave - the row macros are shared with the four-wave kernel below)
	extern __shared__ uint32_t mg_smem[];
	const int wv = threadIdx.x >> 6, lane = hao_lane();
	uint2 *l_q = (uint2*)((char*)mg_smem + wv * hao_seed4_lds<RPL>::PER_WAVE);      // [ROWS] self_offset, cnt of the row's hits (anchor.cpp:1065-1076)
	uint16_t *l_cnt = (uint16_t*)(l_q + ROWS);                                        // [ROWS] length of the row's list (< 4096: the index caps a list at 4095 records)
	uint16_t *l_qi = l_cnt + ROWS;                                                     // [ROWS] the row's minimizer: index in the read's full minimizer list
	// the wave's read: in launch order, or - with a sorted order list (the grid is then a multiple of 8 blocks) - the next of its XCD's eighth of the list
	const uint64_t slot = order ? ((uint64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + wv : (uint64_t)blockIdx.x * 4 + wv;
	if (slot == 0 && lane == 0) S.g_cnt[S.n_sel] = 0;
	if (slot >= S.n_sel) return;                                                        // (no workgroup barriers anywhere: waves are independent)
	const uint64_t r = order ? order[slot] : slot;
	const uint64_t s = S.seg[r]; const uint32_t n = (uint32_t)(S.seg[r + 1] - s);
	if (n == 0) { if (lane == 0) S.g_cnt[r] = 0; return; }
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	// rows = the minimizers with a list, in order (stable compaction by ballot)
	uint32_t nk = 0;
	if (nq <= HAO_QTAB_CAP)
		for (uint32_t b = 0; b < nq && nk <= ROWS; b += 64) {
			const uint32_t q = b + lane; const bool ne = q < nq && S.s_n[li0 + q] != 0;
			const unsigned long long bal = __ballot(ne); const uint32_t k = nk + hao_mbcnt(bal);
			if (ne && k < ROWS) l_qi[k] = (uint16_t)q;
			nk += (uint32_t)__popcll(bal);
		}
	if (nq > HAO_QTAB_CAP || nk > ROWS || n > max_n) { if (lane == 0) ovf_list[atomicAdd(ovf_cnt, 1ULL)] = (uint32_t)r; return; }      // left to the table kernels (max_n: a read with that many
	                                                                                                                                       // seed hits crosses repeat families - hundreds of targets, a step each: the tables are faster there)
	HAO_LOCKSTEP();      // the rows' minimizers are in LDS: every lane reads the ones of its rows
	// per row: head record e0 (with the HIT's strand in bit 55; rid = HAO_MRG_END beyond the list), the BUF records behind it as the index holds them (b0 first; the
	// strand is folded in when a record becomes the head - a row's read is then waited for a step after it was issued, not at once), index of the next record to
	// load, the bookkeeping word
#define HAO_MRG_DECL(i) uint64_t e0_##i = HAO_MRG_SENT, b0_##i = HAO_MRG_SENT, b1_##i = HAO_MRG_SENT, b2_##i = HAO_MRG_SENT, b3_##i = HAO_MRG_SENT, b4_##i = HAO_MRG_SENT, b5_##i = HAO_MRG_SENT, b6_##i = HAO_MRG_SENT, b7_##i = HAO_MRG_SENT, nx_##i = 0; uint32_t rz_##i = 0;
	HAO_MRG_ROWS_DO(HAO_MRG_DECL)
	// (re)fill the buffer of row i from record index nx: up to BUF of the `left` records not loaded yet.  A 32-byte read may run up to three records past the list
	// (into the next list or the slack behind the array: every allocation of the index keeps at least eight records of it); the count says which are real
#define HAO_MRG_FILL(i, left, z) { uint32_t take_ = min((uint32_t)BUF, (left)); \
			if constexpr (BUF == 8) {      /* the ALIGNED 64-byte block that holds record nx: every sector of a list crosses the fabric once.  Only a row's first read (and the one \
			                                  after a redo) starts inside a block: its records move down to b0 (a three-stage shift), later reads start on a block boundary */ \
				const uint32_t off_ = (uint32_t)nx_##i & 7u; take_ = min(8u - off_, (left)); \
				const hao_rec8 v_ = *(const hao_rec8*)(sinfo + (nx_##i - off_)); \
				b0_##i = v_.a; b1_##i = v_.b; b2_##i = v_.c; b3_##i = v_.d; b4_##i = v_.e; b5_##i = v_.f; b6_##i = v_.g; b7_##i = v_.h; \
				if (off_) { \
					if (off_ & 4) { b0_##i = b4_##i; b1_##i = b5_##i; b2_##i = b6_##i; b3_##i = b7_##i; } \
					if (off_ & 2) { b0_##i = b2_##i; b1_##i = b3_##i; b2_##i = b4_##i; b3_##i = b5_##i; b4_##i = b6_##i; b5_##i = b7_##i; } \
					if (off_ & 1) { b0_##i = b1_##i; b1_##i = b2_##i; b2_##i = b3_##i; b3_##i = b4_##i; b4_##i = b5_##i; b5_##i = b6_##i; b6_##i = b7_##i; } \
				} \
			} \
			else if constexpr (BUF == 4 && AL) {      /* the ALIGNED 32-byte block that holds record nx (a read never straddles a sector); as with BUF = 8 only a row's first read starts inside a block */ \
				const uint32_t off_ = (uint32_t)nx_##i & 3u; take_ = min(4u - off_, (left)); \
				const hao_rec4 v_ = *(const hao_rec4*)(sinfo + (nx_##i - off_)); b0_##i = v_.a; b1_##i = v_.b; b2_##i = v_.c; b3_##i = v_.d; \
				if (off_) { \
					if (off_ & 2) { b0_##i = b2_##i; b1_##i = b3_##i; } \
					if (off_ & 1) { b0_##i = b1_##i; b1_##i = b2_##i; b2_##i = b3_##i; } \
				} \
			} \
			else if constexpr (BUF == 4) { const hao_rec4 v_ = *(const hao_rec4*)(sinfo + nx_##i); b0_##i = v_.a; b1_##i = v_.b; b2_##i = v_.c; b3_##i = v_.d; } \
			else b0_##i = sinfo[nx_##i]; \
			nx_##i += take_; rz_##i = ((left) - take_) | take_ << 12 | (z) << 31; }
#define HAO_MRG_INIT(i) if constexpr (i < RPL) { \
		const uint32_t row = i * 64 + lane; \
		if (row < nk) { \
			const uint32_t q = l_qi[row], c = S.s_n[li0 + q], z = hao_info_rev(S.mz_info[m0 + q]); \
			const uint64_t st = S.s_start[li0 + q]; \
			l_q[row] = make_uint2(S.q_pos[li0 + q], S.q_cnt[li0 + q]); l_cnt[row] = (uint16_t)c; \
			e0_##i = sinfo[st] ^ (uint64_t)z << 55; nx_##i = st + 1; rz_##i = z << 31; \
			if (c > 1) HAO_MRG_FILL(i, c - 1, z) \
		} }
	HAO_MRG_ROWS_DO(HAO_MRG_INIT)
	hao_hit_t *hits = S.hits + s; uint64_t *g_tmp = S.g_tmp + s; uint16_t *hq = S.hq ? S.hq + s : nullptr;
	uint32_t run = 0, ngr = 0;
#define HAO_MRG_MIN(i) if constexpr (i < RPL) mn = min(mn, (uint32_t)e0_##i & 0xfffffffu);
#define HAO_MRG_NEXT(out) { uint32_t mn = HAO_MRG_END; HAO_MRG_ROWS_DO(HAO_MRG_MIN) out = hao_wave_min_u32(mn); }
	uint32_t T; HAO_MRG_NEXT(T)
	while (T != HAO_MRG_END) {
		const uint32_t tlen = len[T];
		uint32_t c0 = 0;
#define HAO_MRG_MASK(i) unsigned long long h_##i = 0, v_##i = 0; if constexpr (i < RPL) { \
			h_##i = __ballot(((uint32_t)e0_##i & 0xfffffffu) == T); v_##i = __ballot(((uint32_t)(e0_##i >> 32) & 0x800000u) != 0) & h_##i; \
			c0 += (uint32_t)__popcll(h_##i & ~v_##i); }
		HAO_MRG_ROWS_DO(HAO_MRG_MASK)      // h: rows whose head is a record of T; v: those of them on the opposite strand
		const uint32_t base0 = run;
		if (lane == 0) g_tmp[ngr] = (uint64_t)T << 32 | base0;
		++ngr;
		uint32_t p0 = base0, p1 = base0 + c0;
#define HAO_MRG_EMIT(i) if constexpr (i < RPL) { if (h_##i) {      /* (wave-uniform) */ \
			const uint64_t y = e0_##i; const unsigned long long f_ = h_##i & ~v_##i; \
			const uint32_t rv = (uint32_t)(y >> 55) & 1, a_ = hao_mbcnt(f_), t_ = hao_mbcnt(h_##i); \
			const uint32_t at = rv ? p1 + (t_ - a_) : p0 + a_; \
			p0 += (uint32_t)__popcll(f_); p1 += (uint32_t)__popcll(v_##i); \
			if (((uint32_t)y & 0xfffffffu) == T) { \
				const uint32_t row = i * 64 + lane; \
				hits[at] = hao_mrg_hit(y, T, rv, tlen, l_q[row]); \
				if (hq) hq[at] = l_qi[row]; \
				/* advance the row: the first buffered record becomes the head, the buffer moves up, and an emptied buffer is read again */ \
				const uint32_t bc_ = HAO_MRG_BC(rz_##i), z_ = HAO_MRG_Z(rz_##i); \
				e0_##i = bc_ ? b0_##i ^ (uint64_t)z_ << 55 : HAO_MRG_SENT; \
				if constexpr (BUF >= 4) { b0_##i = b1_##i; b1_##i = b2_##i; b2_##i = b3_##i; } \
				if constexpr (BUF == 8) { b3_##i = b4_##i; b4_##i = b5_##i; b5_##i = b6_##i; b6_##i = b7_##i; } \
				if (bc_ > 1) rz_##i -= 1u << 12; \
				else if (HAO_MRG_REM(rz_##i)) HAO_MRG_FILL(i, HAO_MRG_REM(rz_##i), z_) \
				else rz_##i = z_ << 31; \
			} } }
		HAO_MRG_ROWS_DO(HAO_MRG_EMIT)
		run = p1;
		uint32_t Tn; HAO_MRG_NEXT(Tn)
		if (Tn == T) {
			// ---- some row holds several records of T: redo the target in full (the rows that took part above stand one record behind their head) ----
			// pass 1: forward-strand records of T over all rows (where the opposite strand starts)
			uint32_t f_mine = 0;
#define HAO_MRG_RUN(i, ...) { const uint32_t row = mrg_row0 + i * 64 + lane, c = l_cnt[row], left = HAO_MRG_REM(rz_##i), z = HAO_MRG_Z(rz_##i); \
				const uint64_t *lst = sinfo + (nx_##i - (c - left));      /* the row's list */ \
				const uint32_t held = (((uint32_t)e0_##i & 0xfffffffu) != HAO_MRG_END) + HAO_MRG_BC(rz_##i), j0 = c - left - held - 1; uint32_t j1, nf = 0, nr = 0; \
				for (j1 = j0; j1 < c && hao_info_rid(lst[j1]) == T; ++j1) { if (z ^ hao_info_rev(lst[j1])) ++nr; else ++nf; } \
				__VA_ARGS__ }
#define HAO_MRG_CNT(i) if constexpr (i < RPL) { if (h_##i >> lane & 1) HAO_MRG_RUN(i, f_mine += nf; (void)nr; (void)lst;) }
			HAO_MRG_ROWS_DO(HAO_MRG_CNT)
			uint32_t f_tot = hao_wave_incl_scan_u32(f_mine); f_tot = (uint32_t)__builtin_amdgcn_readlane((int)f_tot, 63);
			p0 = base0; p1 = base0 + f_tot;
#define HAO_MRG_REDO(i) if constexpr (i < RPL) { if (h_##i) { \
				const bool mine = h_##i >> lane & 1; uint32_t nf_ = 0, nr_ = 0; \
				if (mine) HAO_MRG_RUN(i, nf_ = nf; nr_ = nr; (void)lst;) \
				const uint32_t inf = hao_wave_incl_scan_u32(nf_), inr = hao_wave_incl_scan_u32(nr_); \
				const uint32_t tf = (uint32_t)__builtin_amdgcn_readlane((int)inf, 63), tr = (uint32_t)__builtin_amdgcn_readlane((int)inr, 63); \
				if (mine) HAO_MRG_RUN(i, \
					const uint2 qw = l_q[row]; const uint16_t qi = l_qi[row]; \
					uint32_t af = p0 + inf - nf, ar = p1 + inr;      /* forward records in list order; opposite-strand records of the run in REVERSE list order (anchor.cpp:1023) */ \
					for (uint32_t j = j0; j < j1; ++j) { \
						const uint64_t y = lst[j] ^ (uint64_t)z << 55; const uint32_t rv = (uint32_t)(y >> 55) & 1, at = rv ? --ar : af++; \
						hits[at] = hao_mrg_hit(y, T, rv, tlen, qw); \
						if (hq) hq[at] = qi; \
					} \
					/* the row continues behind the run */ \
					e0_##i = j1 < c ? lst[j1] ^ (uint64_t)z << 55 : HAO_MRG_SENT; \
					nx_##i = (uint64_t)(lst - sinfo) + min(c, j1 + 1); rz_##i = z << 31; \
					if (j1 + 1 < c) HAO_MRG_FILL(i, c - (j1 + 1), z)) \
				p0 += tf; p1 += tr; } }
			HAO_MRG_ROWS_DO(HAO_MRG_REDO)
			run = p1;
			HAO_MRG_NEXT(Tn)
		}
		T = Tn;
	}
	if (lane == 0) S.g_cnt[r] = ngr;

}

// ---- the same merge with FOUR waves per read (one workgroup = one read) ----
// One wave per read needs eight rows per lane: 168 registers, three waves per SIMD, and every step is one wave's ~500 dependent instructions - the kernel sits at the
// table kernels' speed (9.7 ms per 993 M-anchor launch) with the SIMDs a third busy.  Here wave w owns rows [w * 64 RPL, (w + 1) * 64 RPL) of the read (RPL = 2: 60-odd
// registers, eight waves per SIMD, a step is a quarter of the work per wave) and the four waves agree on the step through ONE exchange: every wave posts its own
// smallest head target m_w with the forward / opposite-strand counts of its rows that stand on m_w, a barrier, everybody reads the four posts: T = min m_w, the waves
// with m_w = T emit (wave order = row order: their output starts are prefix sums of the posted counts) and advance, the others sit the step out.  The posts alternate
// between two sets of LDS words, so one barrier per step is enough (a wave can be at most one step ahead of the slowest).  A target with several records in a row is
// redone as in the one-wave kernel, with one more exchange for the per-wave run totals.
template<int RPL> struct hao_seed4w_lds { static constexpr uint32_t ROWS_W = 64u * RPL, ROWS = 4 * ROWS_W, TOTAL = ROWS * 12; };

template<int RPL, int BUF>
__global__ __launch_bounds__(256, RPL <= 2 ? (BUF == 8 ? 5 : 6) : 4) void seed_mergew_kernel(hao_seed_args S, const uint64_t *__restrict__ sinfo, const uint32_t *__restrict__ len, const uint32_t *__restrict__ order, uint32_t max_n, uint32_t *ovf_list, unsigned long long *ovf_cnt)
{
	static_assert(BUF == 1 || BUF == 4 || BUF == 8, "records per list read");
	constexpr bool AL = false;      // (aligned 32-byte reads exist in the one-wave kernel only; BUF = 8 reads are aligned by construction)
	constexpr uint32_t ROWS_W = hao_seed4w_lds<RPL>::ROWS_W, ROWS = hao_seed4w_lds<RPL>::ROWS;
	extern __shared__ uint32_t mg_smem[];
	__shared__ uint4 s_x[2][4];      // [set][wave] the wave's post: smallest head target, forward hits, opposite-strand hits of its rows on it
	__shared__ uint2 s_y[4];         // [wave] redo: forward / opposite-strand records of the target in the wave's rows
	const int wv = threadIdx.x >> 6, lane = hao_lane();
	uint2 *l_q = (uint2*)mg_smem;                        // [ROWS] self_offset, cnt of the row's hits
	uint16_t *l_cnt = (uint16_t*)(l_q + ROWS);           // [ROWS] length of the row's list
	uint16_t *l_qi = l_cnt + ROWS;                       // [ROWS] the row's minimizer
	const uint64_t slot = order ? (uint64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (uint64_t)blockIdx.x;
	if (slot == 0 && threadIdx.x == 0) S.g_cnt[S.n_sel] = 0;
	if (slot >= S.n_sel) return;
	const uint64_t r = order ? order[slot] : slot;
	const uint64_t s = S.seg[r]; const uint32_t n = (uint32_t)(S.seg[r + 1] - s);
	if (n == 0) { if (threadIdx.x == 0) S.g_cnt[r] = 0; return; }
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	// rows = the minimizers with a list, in order; every wave runs the whole (short) compaction and keeps the rows of its own range
	const uint32_t row0 = wv * ROWS_W, mrg_row0 = row0;
	uint32_t nk = 0;
	if (nq <= HAO_QTAB_CAP)
		for (uint32_t b = 0; b < nq && nk <= ROWS; b += 64) {
			const uint32_t q = b + lane; const bool ne = q < nq && S.s_n[li0 + q] != 0;
			const unsigned long long bal = __ballot(ne); const uint32_t k = nk + hao_mbcnt(bal);
			if (ne && k >= row0 && k < row0 + ROWS_W) l_qi[k] = (uint16_t)q;
			nk += (uint32_t)__popcll(bal);
		}
	if (nq > HAO_QTAB_CAP || nk > ROWS || n > max_n) { if (threadIdx.x == 0) ovf_list[atomicAdd(ovf_cnt, 1ULL)] = (uint32_t)r; return; }      // (the same decision in every wave) left to the table kernels
	HAO_LOCKSTEP();
	HAO_MRG_ROWS_DO(HAO_MRG_DECL)
#define HAO_MRGW_INIT(i) if constexpr (i < RPL) { \
		const uint32_t row = row0 + i * 64 + lane; \
		if (row < nk) { \
			const uint32_t q = l_qi[row], c = S.s_n[li0 + q], z = hao_info_rev(S.mz_info[m0 + q]); \
			const uint64_t st = S.s_start[li0 + q]; \
			l_q[row] = make_uint2(S.q_pos[li0 + q], S.q_cnt[li0 + q]); l_cnt[row] = (uint16_t)c; \
			e0_##i = sinfo[st] ^ (uint64_t)z << 55; nx_##i = st + 1; rz_##i = z << 31; \
			if (c > 1) HAO_MRG_FILL(i, c - 1, z) \
		} }
	HAO_MRG_ROWS_DO(HAO_MRGW_INIT)
	hao_hit_t *hits = S.hits + s; uint64_t *g_tmp = S.g_tmp + s; uint16_t *hq = S.hq ? S.hq + s : nullptr;
	uint32_t run = 0, ngr = 0, par = 0, T_prev = HAO_MRG_END, run_prev = 0;
#define HAO_MRGW_HP(i) unsigned long long hp_##i = 0;      /* rows of this wave that took part in the last step */
	HAO_MRG_ROWS_DO(HAO_MRGW_HP)
	for (;;) {
		uint32_t mw; HAO_MRG_NEXT(mw)
		uint32_t c0 = 0, c1 = 0;
#define HAO_MRGW_MASK(i) unsigned long long h_##i = 0, v_##i = 0; if constexpr (i < RPL) { \
			h_##i = __ballot(((uint32_t)e0_##i & 0xfffffffu) == mw); v_##i = __ballot(((uint32_t)(e0_##i >> 32) & 0x800000u) != 0) & h_##i; \
			c0 += (uint32_t)__popcll(h_##i & ~v_##i); c1 += (uint32_t)__popcll(v_##i); }
		HAO_MRG_ROWS_DO(HAO_MRGW_MASK)
		if (mw == HAO_MRG_END) c0 = c1 = 0;      // (exhausted rows all "stand on" the end mark)
		if (lane == 0) s_x[par][wv] = make_uint4(mw, c0, c1, 0);
		__syncthreads();
		const uint4 x0 = s_x[par][0], x1 = s_x[par][1], x2 = s_x[par][2], x3 = s_x[par][3];
		par ^= 1;
		const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(min(x0.x, x1.x), min(x2.x, x3.x)));
		if (T == HAO_MRG_END) break;
		if (T == T_prev) {
			// ---- some row holds several records of T_prev: redo that target in full (the rows that took part stand one record behind their head) ----
			const uint32_t tlen = len[T];
			uint32_t f_mine = 0, r_mine = 0;
#define HAO_MRGW_CNT(i) if constexpr (i < RPL) { if (hp_##i >> lane & 1) HAO_MRG_RUN(i, f_mine += nf; r_mine += nr; (void)lst;) }
			HAO_MRG_ROWS_DO(HAO_MRGW_CNT)
			uint32_t f_w = hao_wave_incl_scan_u32(f_mine), r_w = hao_wave_incl_scan_u32(r_mine);
			f_w = (uint32_t)__builtin_amdgcn_readlane((int)f_w, 63); r_w = (uint32_t)__builtin_amdgcn_readlane((int)r_w, 63);
			if (lane == 0) s_y[wv] = make_uint2(f_w, r_w);
			__syncthreads();
			uint32_t p0 = run_prev, p1 = run_prev, f_all = 0, r_all = 0;
#pragma unroll
			for (int w = 0; w < 4; ++w) { const uint2 y = s_y[w]; if (w < wv) { p0 += y.x; p1 += y.y; } f_all += y.x; r_all += y.y; }
			p1 += f_all;
#define HAO_MRGW_REDO(i) if constexpr (i < RPL) { if (hp_##i) { \
				const bool mine = hp_##i >> lane & 1; uint32_t nf_ = 0, nr_ = 0; \
				if (mine) HAO_MRG_RUN(i, nf_ = nf; nr_ = nr; (void)lst;) \
				const uint32_t inf = hao_wave_incl_scan_u32(nf_), inr = hao_wave_incl_scan_u32(nr_); \
				const uint32_t tf = (uint32_t)__builtin_amdgcn_readlane((int)inf, 63), tr = (uint32_t)__builtin_amdgcn_readlane((int)inr, 63); \
				if (mine) HAO_MRG_RUN(i, \
					const uint2 qw = l_q[row]; const uint16_t qi = l_qi[row]; \
					uint32_t af = p0 + inf - nf, ar = p1 + inr; \
					for (uint32_t j = j0; j < j1; ++j) { \
						const uint64_t y = lst[j] ^ (uint64_t)z << 55; const uint32_t rv = (uint32_t)(y >> 55) & 1, at = rv ? --ar : af++; \
						hits[at] = hao_mrg_hit(y, T, rv, tlen, qw); \
						if (hq) hq[at] = qi; \
					} \
					e0_##i = j1 < c ? lst[j1] ^ (uint64_t)z << 55 : HAO_MRG_SENT; \
					nx_##i = (uint64_t)(lst - sinfo) + min(c, j1 + 1); rz_##i = z << 31; \
					if (j1 + 1 < c) HAO_MRG_FILL(i, c - (j1 + 1), z)) \
				p0 += tf; p1 += tr; } }
			HAO_MRG_ROWS_DO(HAO_MRGW_REDO)
			run = run_prev + f_all + r_all;
			T_prev = HAO_MRG_END;
#define HAO_MRGW_CLR(i) hp_##i = 0;
			HAO_MRG_ROWS_DO(HAO_MRGW_CLR)
			continue;      // the heads have moved: the waves post again
		}
		// this step's output starts: the waves that stand on T, in wave order
		const bool my = mw == T;
		uint32_t p0 = run, p1 = run, c0_all = 0, c1_all = 0;
		{ const uint4 xs[4] = {x0, x1, x2, x3};
#pragma unroll
		  for (int w = 0; w < 4; ++w) if (xs[w].x == T) { if (w < wv) { p0 += xs[w].y; p1 += xs[w].z; } c0_all += xs[w].y; c1_all += xs[w].z; } }
		p1 += c0_all;
		if (threadIdx.x == 0) g_tmp[ngr] = (uint64_t)T << 32 | run;
		++ngr; run_prev = run; run += c0_all + c1_all; T_prev = T;
		if (my) {
			const uint32_t tlen = len[T];
#define HAO_MRGW_EMIT(i) if constexpr (i < RPL) { hp_##i = h_##i; if (h_##i) { \
				const uint64_t y = e0_##i; const unsigned long long f_ = h_##i & ~v_##i; \
				const uint32_t rv = (uint32_t)(y >> 55) & 1, a_ = hao_mbcnt(f_), t_ = hao_mbcnt(h_##i); \
				const uint32_t at = rv ? p1 + (t_ - a_) : p0 + a_; \
				p0 += (uint32_t)__popcll(f_); p1 += (uint32_t)__popcll(v_##i); \
				if (((uint32_t)y & 0xfffffffu) == T) { \
					const uint32_t row = row0 + i * 64 + lane; \
					hits[at] = hao_mrg_hit(y, T, rv, tlen, l_q[row]); \
					if (hq) hq[at] = l_qi[row]; \
					const uint32_t bc_ = HAO_MRG_BC(rz_##i), z_ = HAO_MRG_Z(rz_##i); \
					e0_##i = bc_ ? b0_##i ^ (uint64_t)z_ << 55 : HAO_MRG_SENT; \
					if constexpr (BUF >= 4) { b0_##i = b1_##i; b1_##i = b2_##i; b2_##i = b3_##i; } \
				if constexpr (BUF == 8) { b3_##i = b4_##i; b4_##i = b5_##i; b5_##i = b6_##i; b6_##i = b7_##i; } \
					if (bc_ > 1) rz_##i -= 1u << 12; \
					else if (HAO_MRG_REM(rz_##i)) HAO_MRG_FILL(i, HAO_MRG_REM(rz_##i), z_) \
					else rz_##i = z_ << 31; \
				} } }
			HAO_MRG_ROWS_DO(HAO_MRGW_EMIT)
		} else {
			HAO_MRG_ROWS_DO(HAO_MRGW_CLR)
		}
	}
	if (threadIdx.x == 0) S.g_cnt[r] = ngr;
#undef HAO_MRGW_INIT
#undef HAO_MRGW_HP
#undef HAO_MRGW_MASK
#undef HAO_MRGW_CNT
#undef HAO_MRGW_REDO
#undef HAO_MRGW_CLR
#undef HAO_MRGW_EMIT
#undef HAO_MRG_DECL
#undef HAO_MRG_FILL
#undef HAO_MRG_INIT
#undef HAO_MRG_MIN
#undef HAO_MRG_NEXT
#undef HAO_MRG_MASK
#undef HAO_MRG_EMIT
#undef HAO_MRG_RUN
#undef HAO_MRG_CNT
#undef HAO_MRG_REDO
}
