// Seed stage, LIST-MAJOR through LDS (K5-K7 in one pass; minimizers_qgen0, anchor.cpp:987-1081): one persistent workgroup per CU, one read at a time,
// every position list of the read read ONCE with adjacent lanes on adjacent records, the R-way merge by target run out of LDS.
//
// Why: round 5's one-wave merge kernel (one wave per read, the same merge with lane-private list reads; profiles/r05) read its lists lane-privately (32 bytes at a time), which moves whole 128-byte lines through the fabric several times - it
// fetched 2.5 - 5 x its records and sat at 0.33 of the HBM peak, where a coalesced walk of the same lists with the same stores runs at 0.53
// (tools/ubench_gather.hip, profiles/r05).  The transposition between "list-major in" and "target-major out" needs the read's records on chip at once:
// ~12 000 records of a 15 kb read at 30 x.  Here they are, 6 bytes each (key word: target << 1 | strand of the HIT; offset word: 16 bits while every read is
// shorter than 64 kb, else 32), in up to 124 KB of the CU's 160 KB (20 592 record slots for reads of up to 1024 rows).
//
// One workgroup of 8 waves owns the CU's LDS, so nothing else hides its memory latency: the kernel is a software pipeline over the workgroup's reads
// (the first is read blockIdx.x, the next ones come from a cursor shared by the workgroups, asked one step ahead).  While read e is merged,
//   the records of read e + 1 are in flight into REGISTERS (two adjacent 8-byte records = one 16-byte load per lane and load slot; a slot = up to 16 consecutive
//   records of one list, read by 8 adjacent lanes), issued from a slot table that the preparation of read e + 1 left in LDS;
//   the minimizer words of read e + 2 (list start | length | strand, query position, weight) are in flight into registers;
//   the four offsets of read e + 3 are in flight.
// A step: write read e's records from the registers into LDS (strand folded, offset precomputed), barrier, prepare read e + 1 from its minimizer words
// (stable compaction of the minimizers that have a list, three block scans in one, slot table), barrier, issue the three sets of loads, merge read e.
//
// The merge.  The hits of a read in the reference's order (target, strand, query minimizer, list order) are the merge of its lists by target (every list is
// sorted by (rid, pos), htab.cpp:380-460).  The eight waves split the TARGET range: while the records are staged, a 256-bin histogram of their targets (bins =
// equal slices of the read-id range) is counted in LDS; its prefix sums give seven bin boundaries with about an eighth of the read's hits between them, wave w takes
// the targets in [s_w, s_w+1): a 4-ary search per row in LDS finds where its range starts in every list, the sum of those positions is where its output starts.
// (First version: splitters from 64 sampled records - eight samples per range: the slowest wave took twice the mean, profiles/r06/seed_ab.txt.)
// Inside its range a wave steps bin by bin - the smallest (target, strand) key under any cursor by one wave-min, one ballot per 64 rows, the rows that stand on it
// ranked by mbcnt, one 16-byte store per hit at a running position - with heads and cursors in registers (2 per row, up to 24 rows per lane: reads with up to 1536 minimizers
// that have a list) and everything else read from LDS: no buffers to refill, an advance is a cursor increment and a 4-byte LDS read, the end of a list is a
// sentinel record.  A row with several records of one target (a k-mer twice in a target) shows as "the next minimum equals T again" and the target is redone
// by the general routine (runs per row, forward records in list order, opposite-strand records in reverse list order: anchor.cpp:1023).  Group entries
// (target, first hit) go to a per-wave LDS list and are written out behind each other after the merge.
//
// Reads this kernel leaves to the table kernels (overflow list): more than 1536 minimizers, more records than the LDS holds, more load
// slots than the slot table, more than max_n hits (reads across repeat families: hundreds of targets, a step each), more than 64 groups in one wave's range.
// HBM traffic per seed hit: 8 bytes in (once, coalesced), 16 bytes out.
#pragma once
#include <type_traits>
#include "hao_query3.cuh"

#define HAO_L5_THREADS 512
#define HAO_L5_W 8                    // waves per workgroup = target ranges per read
// template parameter QPT: minimizers per thread (2 or 3): a read has at most 1024 / 1536; the LDS tables are laid out for that many rows (hao_l5_lds)
#define HAO_L5_CH 16                  // records per load slot
#define HAO_L5_LPS (HAO_L5_CH / 2)     // lanes per slot: a lane reads two consecutive records with one 16-byte load (half the load instructions of 8-byte loads: the
                                      // vector memory pipeline's address processing, not the bytes, was what the record loads of a read cost - 4 us per read)
#define HAO_L5_SUB (HAO_L5_THREADS / HAO_L5_LPS)      // slot groups per workgroup (64)
struct hao_rec2 { uint64_t a, b; };
// template parameter NPF: slots per group held in registers while the previous read is merged (16: 1024 slots, four registers each; the rest of a read's slots are loaded when it is staged).
// Registers are what bounds this kernel (one workgroup of 512 = two waves per SIMD = 256 VGPRs): the record registers (2 NPF), the merge's rows (about 5 per row:
// 48 / 80 / 121 for 8 / 16 / 24 rows per lane) and the minimizer words (8 QPT) are all alive while a read is merged, and a single spilled register is fatal here -
// its reload is a scratch LOAD, and the s_waitcnt vmcnt(0) in front of its use waits for every record load in flight (the first device run spent the merge waiting
// for the prefetch it was meant to hide).  So: <QPT 2, NPF 16> for batches whose reads have at most 1024 minimizers (HiFi reads up to ~35 kb), <QPT 3, NPF 8> beyond.
#define HAO_L5_GW 64                  // group entries per wave
#define HAO_L5_HB 256                 // bins of the target histogram that splits a read's target range between the waves
// the unrolled load / staging loops: stop the scheduler from hoisting all 32 address computations (64 more live registers) in front of the first access
#ifndef HAO_L5_SCHED_FENCE
#define HAO_L5_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define HAO_L5_MASK 0xfffffffu      // the read id 2^28 - 1 is the merge's end mark: a read set that uses it (exactly 2^28 reads) takes the table kernels
#define HAO_L5_SENT 0xffffffffu

template<bool B16, int QPT> struct hao_l5_lds {
	typedef typename std::conditional<B16, uint16_t, uint32_t>::type off_t;
	static constexpr uint32_t R = QPT * HAO_L5_THREADS;      // rows (minimizers with a list)
	static constexpr uint32_t SL = 2048;                     // slot table (16 records per slot; a list of c records takes ceil(c / 16))
	static constexpr uint32_t FIXED = SL * 8 + HAO_L5_W * HAO_L5_GW * 8 + R * 8 /* qw */ + (QPT * HAO_L5_W + 4) * 8 /* scan */ + (R + 4) * 4 /* ao */ + 64 /* small words */
		+ 5 * HAO_L5_HB * 4 /* hist x 2, bin_tid, bin_len (one word per THREAD: see the preparation) */ + R * 2 /* qi */ + 64;
	static constexpr uint32_t TOTAL = 160 * 1024;
	static constexpr uint32_t CAP = ((TOTAL - FIXED) / (B16 ? 6 : 8) - 8) & ~7u;      // record slots (records + one sentinel per row + the guard slot 0): 20 592 at 6 bytes and 1024 rows, 19 568 at 1536
};

__device__ __forceinline__ uint64_t hao_wave_incl_scan_u64(uint64_t v)
{
#define HAO_RED_STEP(CTRL, RM) { const uint32_t lo2 = (uint32_t)hao_dpp<CTRL, RM>(0, (int)(uint32_t)v), hi2 = (uint32_t)hao_dpp<CTRL, RM>(0, (int)(uint32_t)(v >> 32)); v += (uint64_t)hi2 << 32 | lo2; }
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return v;
}
__device__ __forceinline__ uint64_t hao_readlane_u64(uint64_t v, int l) { return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l); }

// a read's place in the pipeline: what the offsets say (alpha) and what the preparation found (gamma)
struct hao_l5_read { uint64_t m0, s; uint32_t nq, n; uint32_t nk, nslots, nlds; bool valid, skip; };

// the pointers of the workgroup's LDS block
template<bool B16, int QPT> struct hao_l5_ptr {
	typedef hao_l5_lds<B16, QPT> LD;
	uint64_t *slots, *grp, *scan; uint2 *qw; uint32_t *recA, *ao, *sm, *hist, *bin_tid, *bin_len; uint16_t *qi; typename LD::off_t *recB;
	__device__ __forceinline__ hao_l5_ptr(void *base) {
		slots = (uint64_t*)base; grp = slots + LD::SL; qw = (uint2*)(grp + HAO_L5_W * HAO_L5_GW); scan = (uint64_t*)(qw + LD::R);
		ao = (uint32_t*)(scan + QPT * HAO_L5_W + 4); sm = ao + LD::R + 4; hist = sm + 16;      /* (hist is read 16 bytes at a time: every array before it is a multiple of 16 bytes) */
		bin_tid = hist + 2 * HAO_L5_HB; bin_len = bin_tid + HAO_L5_HB; recA = bin_len + 2 * HAO_L5_HB;
		recB = (typename LD::off_t*)(recA + LD::CAP); qi = (uint16_t*)(recB + LD::CAP);
	}
};

// slot word: list position of the slot's first record (40 bits) | records - 1 (4 bits) << 40 | strand of the minimizer << 45 | LDS position of the first record << 46
#define HAO_L5_SLOT_G(e) ((e) & ((1ULL << 40) - 1))
#define HAO_L5_SLOT_C(e) (((uint32_t)((e) >> 40) & 15u) + 1u)
#define HAO_L5_SLOT_Z(e) ((uint32_t)((e) >> 45) & 1u)
#define HAO_L5_SLOT_O(e) ((uint32_t)((e) >> 46))

// an index record as the merge wants it: key = target << 1 | strand of the HIT (29 bits; the order of the keys is the order of the output's bins), off = the hit's offset on the forward
// strand, or - opposite strand - the k-mer's start in the target, from which the emission takes tlen - 1 - start (anchor.cpp:1021-1023, 1059-1064).  The
// target's length: every record also leaves its target id in bin_tid[its histogram bin] (some target of the bin survives), the preparation phase reads len[] for the
// non-empty bins (one gather of at most 256 lanes per read) and the merge takes a step's tlen from LDS when the bin's survivor is the step's target (9 in 10 steps:
// ~60 targets in 256 bins), else by a scalar load.  (A scalar load per step cost its cache miss in front of the step's first LDS wait: 1.8 us per wave and read;
// a gather per opposite-strand record while staging - 64 different lines per instruction - cost 4 us per read.)
__device__ __forceinline__ void hao_l5_fold(uint64_t y, uint32_t z, uint32_t &a, uint32_t &b)
{
	const uint32_t rv = z ^ hao_info_rev(y);
	a = hao_info_rid(y) << 1 | rv;
	b = rv ? hao_info_pos(y) + 1 - hao_info_span(y) : hao_info_pos(y);
}

// ---- the merge of one read out of LDS, one wave = one target range [t_lo, t_hi) ----
template<int RPL, bool B16, int QPT>
__device__ __forceinline__ uint32_t hao_l5_merge(const hao_l5_ptr<B16, QPT> &L, const uint32_t nk, const uint32_t t_lo, const uint32_t t_hi, const int wv, const int lane,
		hao_hit_t *__restrict__ hits, uint16_t *__restrict__ hq, const uint32_t *__restrict__ len, const uint32_t hshift, uint32_t &ngr_out)
{
	uint32_t hd[RPL], cur[RPL];
	// where the wave's range starts in every row: lower bound of t_lo in the row's records (wave 0: the row's first record).  During the search cur[] is the
	// lower end and hd[] the upper end of a row's interval (no arrays of their own: the registers are what this kernel is short of)
	uint32_t before = 0;
#pragma unroll
	for (int i = 0; i < RPL; ++i) {
		const uint32_t row = i * 64 + lane;
		if (row < nk) { cur[i] = L.ao[row]; hd[i] = t_lo ? L.ao[row + 1] - 1 : cur[i]; }      // (the slot before the next row's first record is this row's sentinel)
		else cur[i] = hd[i] = 0;                                                               // (rows beyond nk stand on the guard slot 0: a sentinel)
		before -= cur[i];
	}
	if (t_lo)      // three pivots per step (the interval shrinks to a quarter: half the dependent LDS round trips of a binary search - the search was 2.5 us of a wave's 10.5 per read)
		for (;;) {
			bool more = false;
#pragma unroll
			for (int i = 0; i < RPL; ++i)
				if (cur[i] < hd[i]) {
					const uint32_t n_ = hd[i] - cur[i], p2 = cur[i] + (n_ >> 1), p1 = cur[i] + (n_ >> 2), p3 = p2 + ((hd[i] - p2) >> 1);
					const bool l1 = (L.recA[p1] >> 1) < t_lo, l2 = (L.recA[p2] >> 1) < t_lo, l3 = (L.recA[p3] >> 1) < t_lo;
					hd[i] = l3 ? hd[i] : l2 ? p3 : l1 ? p2 : p1;
					cur[i] = l3 ? p3 + 1 : l2 ? p2 + 1 : l1 ? p1 + 1 : cur[i];
					more = true;
				}
			if (!__any(more)) break;
		}
	// a row's head in registers: key word hd, offset word hb (read when the row advances: an emission then needs no LDS round trip - the first version read the
	// offset and the row's query words inside every block of every step and waited for them there, ~130 cycles per block); QREG (up to 8 rows per lane, i.e. every
	// read with at most 512 rows): the row's two query words in registers too
	constexpr bool QREG = RPL <= 8, HBREG = RPL <= 16;      // (24 rows per lane: the offset word is read at the emission - no registers left for it)
	uint32_t hb[HBREG ? RPL : 1], qx[QREG ? RPL : 1], qy[QREG ? RPL : 1];
#pragma unroll
	for (int i = 0; i < RPL; ++i) {
		before += cur[i]; hd[i] = L.recA[cur[i]]; if constexpr (HBREG) hb[i] = (uint32_t)L.recB[cur[i]];
		if constexpr (QREG) { const uint2 q_ = L.qw[min((uint32_t)(i * 64 + lane), (uint32_t)(hao_l5_lds<B16, QPT>::R - 1))]; qx[i] = q_.x; qy[i] = q_.y; }
	}
	uint32_t run = hao_wave_incl_scan_u32(before); run = (uint32_t)__builtin_amdgcn_readlane((int)run, 63);
	uint32_t ngr = 0;
	uint64_t *grp = L.grp + wv * HAO_L5_GW;
#define HAO_L5_NEXT(out) { uint32_t mn = HAO_L5_SENT; _Pragma("unroll") for (int i = 0; i < RPL; ++i) mn = min(mn, hd[i]); out = hao_wave_min_u32(mn); }
#define HAO_L5_PUT(at_, w0_, rv_, b_, qx_, qy_, row_) { \
		hao_hit_t h_; h_.w0 = (w0_); h_.offset = (rv_) ? tlen - 1 - (b_) : (b_); h_.self_offset = (qx_); h_.cnt = (qy_); \
		hits[at_] = h_; if (hq) hq[at_] = L.qi[row_]; }
	// A step = one BIN: the smallest key K = (target, strand) under any cursor; the rows whose head is K emit it in row order and advance.  (The first version stepped
	// by target and ranked forward and opposite-strand hits in one pass: a count pass over all blocks in front of every step and two ballots, two mbcnt pairs and four
	// selects per block - 45 instructions per emitting block where this takes 16, and at two waves per SIMD the step loop runs at the latency of its dependent
	// instructions, ~7 cycles each: profiles/r06/seed_ab.txt.)  A row's records of one target are in POSITION order, so behind a head (T, opposite) there can be a
	// (T, forward) record that the forward bin's step did not see, and a row can hold a key twice: both show after the step as "the next key is not above this one,
	// same target", and the target is redone in full by the general routine (per-row runs, forward records in list order, opposite-strand records in reverse list
	// order: anchor.cpp:1023) from where its first bin started.
	uint32_t K; HAO_L5_NEXT(K)
	uint32_t T_cur = HAO_L5_SENT, base_tid = run, tlen = 0;
	while ((K >> 1) < t_hi) {
		const uint32_t T = K >> 1, rv = K & 1;
		if (T != T_cur) {      // (wave-uniform) a new target: its group entry and its length - from the bin table when the bin kept this target (9 in 10), else a scalar load
			// (HAO_SLOAD_U32: inline assembly - written as two C++ loads the compiler selected between the two ADDRESSES and issued one flat load behind an
			// s_waitcnt vmcnt(0), i.e. every step waited for every record load and hit store in flight)
			T_cur = T; base_tid = run;
			tlen = L.bin_len[T >> hshift];
			if (L.bin_tid[T >> hshift] != T) HAO_SLOAD_U32(tlen, len + T);
			if (lane == 0 && ngr < HAO_L5_GW) grp[ngr] = (uint64_t)T << 32 | run;
			++ngr;
		}
		const uint32_t w0 = T | rv << 31;
#pragma unroll
		for (int i = 0; i < RPL; ++i) {
			const bool act = hd[i] == K;
			const unsigned long long h = __ballot(act);
			if (h) {      // (wave-uniform)
				const uint32_t at = run + hao_mbcnt(h);
				run += (uint32_t)__popcll(h);
				if (act) {
					const uint32_t row = i * 64 + lane;
					if constexpr (QREG) HAO_L5_PUT(at, w0, rv, hb[i], qx[i], qy[i], row)
					else if constexpr (HBREG) { const uint2 q_ = L.qw[row]; HAO_L5_PUT(at, w0, rv, hb[i], q_.x, q_.y, row) }
					else { const uint2 q_ = L.qw[row]; const uint32_t b_ = (uint32_t)L.recB[cur[i]]; HAO_L5_PUT(at, w0, rv, b_, q_.x, q_.y, row) }
					++cur[i]; hd[i] = L.recA[cur[i]]; if constexpr (HBREG) hb[i] = (uint32_t)L.recB[cur[i]];      // (first needed by the next step's minimum / emission)
				}
			}
		}
		uint32_t Kn; HAO_L5_NEXT(Kn)
		if ((Kn >> 1) == T && Kn <= K) {
			// ---- a row holds a key of T twice, or a forward record of T behind an opposite-strand one: the target again, in full ----
			// every row's run of T: from the first record of T at or before its cursor (some rows have emitted records of T, some stand on one) to the first record behind T
			uint32_t f_mine = 0;
#pragma unroll
			for (int i = 0; i < RPL; ++i) {
				uint32_t j0 = cur[i];
				if ((uint32_t)(i * 64 + lane) < nk) while ((L.recA[j0 - 1] >> 1) == T) --j0;      // (the slot before a row's first record is a sentinel: the walk stops there; rows beyond nk stand on slot 0)
				for (uint32_t j = j0; (L.recA[j] >> 1) == T; ++j) f_mine += !(L.recA[j] & 1);
				cur[i] = j0;
			}
			uint32_t f_tot = hao_wave_incl_scan_u32(f_mine); f_tot = (uint32_t)__builtin_amdgcn_readlane((int)f_tot, 63);
			uint32_t p0 = base_tid, p1 = base_tid + f_tot;
#pragma unroll
			for (int i = 0; i < RPL; ++i) {
				const bool mine = (L.recA[cur[i]] >> 1) == T;
				if (__ballot(mine)) {
					uint32_t nf = 0, nr = 0, j1 = cur[i];
					if (mine) for (; (L.recA[j1] >> 1) == T; ++j1) { if (L.recA[j1] & 1) ++nr; else ++nf; }
					const uint32_t inf = hao_wave_incl_scan_u32(nf), inr = hao_wave_incl_scan_u32(nr);
					const uint32_t tf = (uint32_t)__builtin_amdgcn_readlane((int)inf, 63), tr = (uint32_t)__builtin_amdgcn_readlane((int)inr, 63);
					if (mine) {
						const uint32_t row = i * 64 + lane; const uint2 q_ = L.qw[row];
						uint32_t af = p0 + inf - nf, ar = p1 + inr;      // forward records in list order; opposite-strand records of the run in REVERSE list order (anchor.cpp:1023)
						for (uint32_t j = cur[i]; j < j1; ++j) { const uint32_t r_ = L.recA[j] & 1, at = r_ ? --ar : af++, b_ = (uint32_t)L.recB[j]; HAO_L5_PUT(at, T | r_ << 31, r_, b_, q_.x, q_.y, row) }
						cur[i] = j1; hd[i] = L.recA[j1]; if constexpr (HBREG) hb[i] = (uint32_t)L.recB[j1];
					}
					p0 += tf; p1 += tr;
				}
			}
			run = p1;
			HAO_L5_NEXT(Kn)
		}
		K = Kn;
	}
#undef HAO_L5_NEXT
#undef HAO_L5_PUT
	ngr_out = ngr;
	return run;
}

template<bool B16, int QPT, int NPF, bool DBG>
__global__ __launch_bounds__(HAO_L5_THREADS, 2) void seed_lds_kernel(hao_seed_args S, const uint64_t *__restrict__ sinfo, const uint32_t *__restrict__ len, const uint64_t *__restrict__ s_pk,
		uint32_t max_n, uint32_t w0_share, uint32_t *ovf_list, unsigned long long *ovf_cnt, unsigned long long *next_read)
{
	constexpr uint32_t CAP = hao_l5_lds<B16, QPT>::CAP;
	extern __shared__ uint64_t l5_smem[];
	const hao_l5_ptr<B16, QPT> L((void*)l5_smem);
	const uint32_t tid = threadIdx.x; const int wv = tid >> 6, lane = hao_lane();
	const uint32_t sg = tid / HAO_L5_LPS, sj = 2 * (tid % HAO_L5_LPS);      // this thread's slot group and its first record inside a slot (it takes records sj and sj + 1)
	const uint32_t hshift = S.tb > 8 ? (uint32_t)S.tb - 8u : 0u;      // read ids have tb bits: 256 histogram bins over the id range
	const uint64_t G = gridDim.x, b0 = blockIdx.x;
	if (b0 == 0 && tid == 0) S.g_cnt[S.n_sel] = 0;
	if (b0 >= S.n_sel) return;
	if (tid < 2 * HAO_L5_HB) L.hist[tid] = 0;
	// this workgroup's reads: b0, then whatever the batch's cursor hands out (*next_read, zero at launch: read G + its value).  A fixed share - b0, b0 + G, ... - cost 1 - 2 % of the
	// stage (same box, three runs each: 43.3 against 44.0 ms per pass; 325 reads per workgroup whose hit counts vary by a quarter).  The cursor is asked one step ahead by thread 0 and its
	// answer travels through LDS with the step's last barrier.
	const uint32_t NONE = 0xffffffffu, n_sel32 = (uint32_t)S.n_sel;
	uint32_t r_a = NONE, r_b = NONE, r_d = NONE, r_e = NONE; uint64_t nrd = 0;
	// pipeline registers
	uint64_t av = 0;                                                        // alpha: lanes 0 - 3 of every wave hold mz_off[r], mz_off[r + 1], seg[r], seg[r + 1]
	uint64_t raw_s[QPT]; uint32_t raw_p[QPT], raw_c[QPT];      // beta: start | n << 48 | strand << 63, query position, cnt word of the thread's minimizers
	uint32_t t_kc[QPT], t_ao[QPT], t_p[QPT], t_c[QPT];      // gamma: row | list length << 16 (or ~0), first LDS slot, the two query words
	hao_rec2 rec[NPF];                                               // delta: the records of this thread's slots
	hao_l5_read rb, rd, re;
	rb.valid = rd.valid = re.valid = false; rb.skip = rd.skip = re.skip = true; rb.m0 = rb.s = rd.m0 = rd.s = re.m0 = re.s = 0; rb.nq = rb.n = rd.nq = rd.n = re.nq = re.n = 0;
	rb.nk = rb.nslots = rb.nlds = rd.nk = rd.nslots = rd.nlds = re.nk = re.nslots = re.nlds = 0;
#pragma unroll
	for (int m = 0; m < QPT; ++m) { raw_s[m] = 0; raw_p[m] = raw_c[m] = 0; t_kc[m] = HAO_L5_SENT; t_ao[m] = t_p[m] = t_c[m] = 0; }
#pragma unroll
	for (int i = 0; i < NPF; ++i) rec[i].a = rec[i].b = 0;

	// DBG instances: wall-clock ticks (100 MHz) of wave 0 per phase, summed over the workgroups into S.dbg[0 .. 9] (HAO_DBG_PRINT=seed)
	unsigned long long tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_last = DBG ? wall_clock64() : 0;
#define HAO_L5_TICK(k) if constexpr (DBG) { const unsigned long long now_ = wall_clock64(); tk_acc[k] += now_ - tk_last; tk_last = now_; }
	for (uint64_t step = 0; ; ++step) {
		// reads of this step: e (taken three steps ago) is staged and merged, d prepared and its records requested, b: its minimizer words requested, a (new): its offsets
		r_e = r_d; r_d = r_b; r_b = r_a;
		r_a = step == 0 ? (uint32_t)b0 : (r_b < n_sel32 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.sm[12]) : NONE);      // (once the cursor has run past the batch it is not asked again)
		if (r_a >= n_sel32) r_a = NONE;
		if ((r_a & r_b & r_d & r_e) == NONE) break;
		unsigned long long nx_ = 0;
		if (tid == 0 && r_a != NONE) nx_ = atomicAdd(next_read, 1ULL);      // the read after r_a; lands in LDS at the end of the step
		if (r_a != NONE) ++nrd;
		re = rd; rd = rb;
		rb.valid = r_b != NONE; rb.skip = true;
		if (rb.valid) {
			const uint64_t m0 = hao_readlane_u64(av, 0), m1 = hao_readlane_u64(av, 1), s0 = hao_readlane_u64(av, 2), s1 = hao_readlane_u64(av, 3);
			rb.m0 = m0; rb.nq = (uint32_t)(m1 - m0); rb.s = s0; rb.n = (uint32_t)(s1 - s0);
			rb.skip = rb.n == 0 || rb.nq > (uint32_t)(QPT * HAO_L5_THREADS) || rb.n > max_n;      // (skip: nothing to load; a read that is left to the table kernels is listed when it reaches the merge step)
		}
		// ---- stage read e: the row table from the registers the preparation left, the records from the registers the loads filled ----
		uint32_t *hist_e = L.hist + ((step + 1) & 1) * HAO_L5_HB;      // (this step's histogram; the other one is cleared below for the next step)
		if (re.valid && !re.skip) {
#pragma unroll
			for (int m = 0; m < QPT; ++m)
				if (t_kc[m] != HAO_L5_SENT) {
					const uint32_t k = t_kc[m] & 0xffffu, c = t_kc[m] >> 16;
					L.ao[k] = t_ao[m]; L.qw[k] = make_uint2(t_p[m], t_c[m]); L.qi[k] = (uint16_t)(m * HAO_L5_THREADS + tid);
					L.recA[t_ao[m] + c] = HAO_L5_SENT;
				}
			if (tid == 0) { L.recA[0] = HAO_L5_SENT; L.ao[re.nk] = re.nlds + 1; }
#pragma unroll
			for (int i = 0; i < NPF; ++i) {
				const uint32_t sl = i * HAO_L5_SUB + sg;
				if (sl < re.nslots) {
					const uint64_t e = L.slots[sl]; const uint32_t c = HAO_L5_SLOT_C(e), z = HAO_L5_SLOT_Z(e), o = HAO_L5_SLOT_O(e) + sj;
					if (sj < c) { uint32_t a, b; hao_l5_fold(rec[i].a, z, a, b); const uint32_t t = a >> 1; L.recA[o] = a; L.recB[o] = (typename hao_l5_lds<B16, QPT>::off_t)b; atomicAdd(&hist_e[t >> hshift], 1u); L.bin_tid[t >> hshift] = t; }
					if (sj + 1 < c) { uint32_t a, b; hao_l5_fold(rec[i].b, z, a, b); L.recA[o + 1] = a; L.recB[o + 1] = (typename hao_l5_lds<B16, QPT>::off_t)b; }      // (the histogram is a sample: every other record)
				}
				if ((i & 1) == 1) HAO_L5_SCHED_FENCE();
			}
			for (uint32_t sl = NPF * HAO_L5_SUB + sg; sl < re.nslots; sl += HAO_L5_SUB) {      // the slots beyond the registers: loaded now
				const uint64_t e = L.slots[sl];
				const uint32_t c = HAO_L5_SLOT_C(e), z = HAO_L5_SLOT_Z(e), o = HAO_L5_SLOT_O(e) + sj;
#pragma unroll
				for (uint32_t x = 0; x < 2; ++x)
					if (sj + x < c) { uint32_t a, b; hao_l5_fold(sinfo[HAO_L5_SLOT_G(e) + sj + x], z, a, b); const uint32_t t = a >> 1; L.recA[o + x] = a; L.recB[o + x] = (typename hao_l5_lds<B16, QPT>::off_t)b; atomicAdd(&hist_e[t >> hshift], 1u); L.bin_tid[t >> hshift] = t; }
			}
		}
		HAO_L5_TICK(0)
		__syncthreads();
		HAO_L5_TICK(1)
		// ---- prepare read d from its minimizer words: rows = the minimizers that have a list, in order; one scan for (rows, load slots, LDS slots) ----
		{
			uint64_t v[QPT], inc[QPT];
			const bool go = rd.valid && !rd.skip;
			if (tid < HAO_L5_HB) L.hist[(step & 1) * HAO_L5_HB + tid] = 0;      // next step's histogram (last read by the step before this one)
			// the length of the target this step's records left in bin tid, written to LDS at the end of the preparation (the load has the scans to land).  EVERY thread
			// loads and stores (threads 256 .. 511 a word nobody reads): under a condition, the compiler's s_waitcnt pass keeps the load's register "pending" on the path
			// around the store and puts an s_waitcnt vmcnt(0) in front of the next write to that register - which was in the merge, with every record load in flight
			const bool blw_ = tid < HAO_L5_HB && re.valid && !re.skip && hist_e[tid] != 0;
			const uint32_t bl_ = len[blw_ ? L.bin_tid[tid] : 0u];
			if (tid < HAO_L5_HB && !blw_) L.bin_tid[tid] = HAO_L5_SENT;      // (no sampled record in the bin: whatever an earlier read left there must not pair up with the word stored below)
#pragma unroll
			for (int m = 0; m < QPT; ++m) {
				const uint32_t q = m * HAO_L5_THREADS + tid, c = go && q < rd.nq ? (uint32_t)(raw_s[m] >> 48) & 0xfffu : 0u;
				v[m] = c ? 1ULL | (uint64_t)((c + HAO_L5_CH - 1) / HAO_L5_CH) << 16 | (uint64_t)(c + 1) << 40 : 0ULL;
				inc[m] = hao_wave_incl_scan_u64(v[m]);
				if (lane == 63) L.scan[m * HAO_L5_W + wv] = inc[m];
			}
			__syncthreads();
			uint64_t run = 0, basev[QPT];
#pragma unroll
			for (int m = 0; m < QPT; ++m)
#pragma unroll
				for (int w = 0; w < HAO_L5_W; ++w) { if (w == wv) basev[m] = run; run += L.scan[m * HAO_L5_W + w]; }
			rd.nk = (uint32_t)run & 0xffffu; rd.nslots = (uint32_t)(run >> 16) & 0xffffffu; rd.nlds = (uint32_t)(run >> 40);
			if (go && (rd.nslots > hao_l5_lds<B16, QPT>::SL || rd.nlds + 2 > CAP)) rd.skip = true;
			const bool go2 = rd.valid && !rd.skip;
#pragma unroll
			for (int m = 0; m < QPT; ++m) {
				t_kc[m] = HAO_L5_SENT; t_ao[m] = 0; t_p[m] = raw_p[m]; t_c[m] = raw_c[m];
				if (go2 && v[m]) {
					const uint64_t ex = basev[m] + inc[m] - v[m];
					const uint32_t k = (uint32_t)ex & 0xffffu, sl0 = (uint32_t)(ex >> 16) & 0xffffffu, o0 = (uint32_t)(ex >> 40) + 1, c = (uint32_t)(raw_s[m] >> 48) & 0xfffu, z = (uint32_t)(raw_s[m] >> 63);
					const uint64_t st = raw_s[m] & ((1ULL << 48) - 1);
					t_kc[m] = k | c << 16; t_ao[m] = o0;
					for (uint32_t x = 0, sl = sl0; x < c; x += HAO_L5_CH, ++sl)
						L.slots[sl] = (st + x) | (uint64_t)(min(c - x, (uint32_t)HAO_L5_CH) - 1) << 40 | (uint64_t)z << 45 | (uint64_t)(o0 + x) << 46;
				}
			}
			L.bin_len[tid] = bl_;
		}
		HAO_L5_TICK(2)
		__syncthreads();
		HAO_L5_TICK(3)
		// ---- loads: the records of read d (first NPF slots per group), the minimizer words of read b, the offsets of read a ----
		{
			const uint32_t nsl = rd.valid && !rd.skip ? rd.nslots : 0u;
#pragma unroll
			for (int i = 0; i < NPF; ++i) {
				const uint32_t sl = i * HAO_L5_SUB + sg;
				rec[i].a = rec[i].b = 0;      // (every register is written in every step: a conditional assignment alone would keep last step's value alive across the whole loop body)
				if (sl < nsl) {
					const uint64_t e = L.slots[sl]; const uint32_t c = HAO_L5_SLOT_C(e);
					if (sj + 1 < c) rec[i] = *(const hao_rec2*)(sinfo + HAO_L5_SLOT_G(e) + sj);      // (16 bytes at an 8-byte boundary)
					else if (sj < c) rec[i].a = sinfo[HAO_L5_SLOT_G(e) + sj];                        // the odd record at the end of a list: nothing is read past it
				}
				if ((i & 1) == 1) HAO_L5_SCHED_FENCE();
			}
		}
		{
			const uint64_t li0 = rb.m0 - S.mz0; const uint32_t nqb = rb.valid && !rb.skip ? rb.nq : 0u;
#pragma unroll
			for (int m = 0; m < QPT; ++m) {
				const uint32_t q = m * HAO_L5_THREADS + tid;
				raw_s[m] = 0; raw_p[m] = 0; raw_c[m] = 0;
				if (q < nqb) { raw_s[m] = s_pk[li0 + q]; raw_p[m] = S.q_pos[li0 + q]; raw_c[m] = S.q_cnt[li0 + q]; }
			}
		}
		if (r_a != NONE) {
			const uint64_t r = r_a;
			if (lane < 4) av = lane < 2 ? S.mz_off[S.rid_lo + r + lane] : S.seg[r + lane - 2];
		}
		HAO_L5_TICK(4)
		// ---- merge read e ----
		if (re.valid) {
			const uint64_t r = r_e;
			uint32_t ngr = 0; bool gover = false;
			if (!re.skip) {
				// seven splitters from the target histogram: wave w starts at the first bin whose exclusive prefix sum reaches w / 8 of the read's hits
				uint32_t t_lo = 0, t_hi = HAO_L5_MASK;
				{
					const uint4 hv = *(const uint4*)(hist_e + 4 * lane);
					const uint32_t s4 = hv.x + hv.y + hv.z + hv.w, ex0 = hao_wave_incl_scan_u32(s4) - s4, ex1 = ex0 + hv.x, ex2 = ex1 + hv.y, ex3 = ex2 + hv.z;
					// shares in 1/1024 of the read's hits: wave 0 needs no search for its start (about 4 % of a read's time), and the second wave of every SIMD (waves 4 - 7)
					// loses the issue arbitration to the first (measured: 11.4 against 10.5 us per read for equal shares, profiles/r06/seed_ab.txt)
					// shares in 1/1024 of the (sampled) hits: wave 0 needs no search for its start and gets w0_share (default 176), the others share the rest - the second
					// wave of every SIMD (waves 4 - 7) 4 % less than the first: it loses the issue arbitration (profiles/r06/seed_ab.txt)
					const uint32_t tot_ = (uint32_t)__builtin_amdgcn_readlane((int)(ex0 + s4), 63), rest_ = 1024u - w0_share;
					auto cumw = [&](uint32_t w) -> uint32_t { return w == 0 ? 0u : w >= HAO_L5_W ? 1024u : w0_share + (w <= 4 ? (w - 1) * (rest_ * 37u >> 8) : 3u * (rest_ * 37u >> 8) + (w - 4) * (rest_ * 35u >> 8)); };
					const uint32_t th_lo = (uint32_t)(((uint64_t)tot_ * cumw((uint32_t)wv)) >> 10), th_hi = (uint32_t)(((uint64_t)tot_ * cumw((uint32_t)wv + 1)) >> 10);
					const uint32_t b_lo = (uint32_t)(__popcll(__ballot(ex0 < th_lo)) + __popcll(__ballot(ex1 < th_lo)) + __popcll(__ballot(ex2 < th_lo)) + __popcll(__ballot(ex3 < th_lo)));
					const uint32_t b_hi = (uint32_t)(__popcll(__ballot(ex0 < th_hi)) + __popcll(__ballot(ex1 < th_hi)) + __popcll(__ballot(ex2 < th_hi)) + __popcll(__ballot(ex3 < th_hi)));
					t_lo = b_lo << hshift;      // (wave 0: th_lo = 0, no bin below it: t_lo = 0)
					if (wv < HAO_L5_W - 1 && b_hi < HAO_L5_HB) t_hi = b_hi << hshift;
				}
				hao_hit_t *hits = S.hits + re.s; uint16_t *hq = S.hq ? S.hq + re.s : nullptr;
				HAO_L5_TICK(5)
				if (t_lo < t_hi) {
					if (re.nk <= 8 * 64) (void)hao_l5_merge<8, B16, QPT>(L, re.nk, t_lo, t_hi, wv, lane, hits, hq, len, hshift, ngr);
					else if (QPT == 2 || re.nk <= 16 * 64) (void)hao_l5_merge<16, B16, QPT>(L, re.nk, t_lo, t_hi, wv, lane, hits, hq, len, hshift, ngr);
					else if constexpr (QPT > 2) (void)hao_l5_merge<24, B16, QPT>(L, re.nk, t_lo, t_hi, wv, lane, hits, hq, len, hshift, ngr);
				}
				if (lane == 0) L.sm[wv] = ngr;
				HAO_L5_TICK(6)
			}
			__syncthreads();
			if (!re.skip) {
				uint32_t gb = 0, tot = 0;
#pragma unroll
				for (int w = 0; w < HAO_L5_W; ++w) { const uint32_t g = L.sm[w]; if (w < wv) gb += g; tot += g; gover |= g > HAO_L5_GW; }
				if (!gover) {
					uint64_t *g_tmp = S.g_tmp + re.s; const uint64_t *grp = L.grp + wv * HAO_L5_GW;
					for (uint32_t k = lane; k < ngr; k += 64) g_tmp[gb + k] = grp[k];
					if (tid == 0) S.g_cnt[r] = tot;
				}
			}
			if (tid == 0) {
				if (re.n == 0) S.g_cnt[r] = 0;
				else if (re.skip || gover) ovf_list[atomicAdd(ovf_cnt, 1ULL)] = (uint32_t)r;      // left to the table kernels
			}
		}
		HAO_L5_TICK(7)
		if (tid == 0 && r_a != NONE) L.sm[12] = G + nx_ < (unsigned long long)NONE ? (uint32_t)(G + nx_) : NONE;
		__syncthreads();
		HAO_L5_TICK(8)
	}
	if constexpr (DBG) if (S.dbg) {
		if (tid == 0) { for (int k = 0; k < 9; ++k) atomicAdd(S.dbg + k, tk_acc[k]); atomicAdd(S.dbg + 9, (unsigned long long)nrd); }
		if (lane == 0) atomicAdd(S.dbg + 16 + wv, tk_acc[6]);      // every wave's own time in the merge (search + steps)
	}
#undef HAO_L5_TICK
}
