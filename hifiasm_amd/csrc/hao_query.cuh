// K5-K7: seed lookup, per-read ordering of the seed hits and k_mer_hit construction
// (minimizers_qgen0, anchor.cpp:987-1081) for gfx950.
//
// The reference materialises 24-byte anchors per read and radix-sorts them by (target id, strand, query pos), then by target
// offset.  Here nothing is materialised between the position index and the sorted k_mer_hits:
//   * an index list is ordered by (rid, pos), and two hits of one k-mer in one target have k-mer starts ordered like their
//     ends, so the reference order (tid, strand, self_offset, other_off) is: (tid, rev), then the query minimizer's rank q,
//     then list order for same-strand hits / reverse list order for opposite-strand hits;
//   * walking a read's anchors in generation order (q, list order) therefore only needs a STABLE partition by (tid, rev);
//   * a read meets few distinct (tid, rev) values (the reads that overlap it: ~2 x coverage) although read ids need 15-29 bits,
//     so the partition is not digit-wise: one workgroup per read builds the set of distinct (tid, rev) bins in an LDS hash table
//     (counting per wave while inserting), sorts the distinct bins, and places every hit with one ranked stable scatter.  The
//     sorted bin table also IS the list of target groups of the read (consecutive bins with the same tid).
#pragma once
#include "hao_common.cuh"
#include "hao_index.cuh"

// Q1 (ha_pt_get, anchor.cpp:1013, and the two words every hit of a minimizer shares: self_offset and cnt = weight(n) << 8 | span, anchor.cpp:1065-1076):
// the lookup results were computed when the index was built (hao_index_finish_kernel): unpack them for the batch's minimizers
__global__ void seed_unpack_kernel(const uint64_t *lk, const uint64_t *mz_info, uint64_t mz0, uint64_t n_mz, const uint32_t *wgt_tab,
		uint64_t *s_start, uint32_t *s_n, uint32_t *q_pos, uint32_t *q_cnt, uint64_t *s_pk = nullptr)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n_mz) return;
	if (i == n_mz) { s_n[i] = 0; if (s_pk) s_pk[i] = 0; return; }
	const uint64_t v = lk[mz0 + i], z = mz_info[mz0 + i]; const uint32_t n = (uint32_t)(v >> 48);
	s_start[i] = v & ((1ULL << 48) - 1); s_n[i] = n; q_pos[i] = hao_info_pos(z); q_cnt[i] = wgt_tab[n] << 8 | hao_info_span(z);
	if (s_pk) s_pk[i] = (v & ((1ULL << 60) - 1)) | (uint64_t)hao_info_rev(z) << 63;      // seed_lds_kernel's minimizer word: list start | list length << 48 | strand of the minimizer << 63
}

// per-read anchor segment bounds from the per-minimizer scan
__global__ void seed_segments_kernel(const uint64_t *mz_off, uint64_t rid_lo, uint64_t n_sel, uint64_t mz0, const uint64_t *a_off, uint64_t *seg)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	seg[r] = a_off[mz_off[rid_lo + r] - mz0];
}

// rank support: lanes of a wave whose d agree on the low nbits bits (nbits ballots)
__device__ __forceinline__ unsigned long long hao_match_bits(uint32_t d, bool act, int nbits)
{
	unsigned long long m = __ballot(act);
	for (int b = 0; b < nbits; ++b) { unsigned long long bal = __ballot((d >> b) & 1); m &= ((d >> b) & 1) ? bal : ~bal; }
	return m;
}

// the same for a key of a fixed number of bits, fully unrolled: per bit one sign-extended bit field (0 / -1), one compare (the ballot) and one three-input bit
// operation per half of the mask, m & ~(ballot ^ my bit) = v_bitop3 0x90 - 5 vector instructions per bit where the loop above costs 13.  The scatter passes
// match on the SLOT of a hit's bin (slots and bins correspond one to one), so they need neither the bins' ranks nor a run-time bit count.
template<int NB> __device__ __forceinline__ unsigned long long hao_match_key(uint32_t key, bool act)
{
	const unsigned long long m0 = __ballot(act);
	uint32_t m_lo = (uint32_t)m0, m_hi = (uint32_t)(m0 >> 32);
#pragma unroll
	for (int b = 0; b < NB; ++b) {
		const int32_t t = __builtin_amdgcn_sbfe((int)key, b, 1);
		const unsigned long long bal = __ballot(t != 0);
		m_lo = __builtin_amdgcn_bitop3_b32(m_lo, (uint32_t)bal, (uint32_t)t, 0x90);
		m_hi = __builtin_amdgcn_bitop3_b32(m_hi, (uint32_t)(bal >> 32), (uint32_t)t, 0x90);
	}
	return (unsigned long long)m_hi << 32 | m_lo;
}

// Which minimizer holds each anchor of a 64-anchor window?  QL kernels stage only the read's NON-EMPTY minimizers, so their first-anchor offsets
// ao[] increase strictly and a window of 64 anchors sees at most 64 list boundaries: lane i looks at boundary kc + 1 + i and pushes a flag to the lane of
// its window position (ds_permute: lanes nobody writes to read 0; boundaries beyond the window are parked on lane 0, whose own anchor - the window's
// first - is never a boundary because kc holds it), one ballot turns the flags into the window's boundary mask and a lane's minimizer is kc + the
// boundaries at or before it.  Replaces a per-lane `while (ao[q + 1] <= x) ++q` (a divergent loop with an LDS round trip per step).
// kc: compact index of the minimizer that holds anchor x0 (wave-uniform); on return, of the one that holds x0 + 64.  ao[nk] = the read's anchor count.
__device__ __forceinline__ uint32_t hao_seed_locate(const uint32_t *ao, uint32_t nk, uint32_t &kc, uint32_t x0, int lane)
{
	const uint32_t cand = kc + 1 + lane, p = ao[cand < nk ? cand : nk] - x0;      // >= 1
	const int got = __builtin_amdgcn_ds_permute((int)((p < 64 ? p : 0u) << 2), 1);
	const unsigned long long M = __ballot(got != 0) & ~1ULL;
	const uint32_t k = kc + __popcll(M & ((2ULL << lane) - 1));
	kc += __popcll(M) + (__ballot(p == 64) != 0 ? 1u : 0u);
	return k;
}

#define HAO_QTAB_CAP 4096     // most query minimizers whose offsets / list starts are staged in LDS per read; longer reads read them from global memory
#define HAO_BIN_EMPTY 0xffffffffu
// the workgroup's "table full" flag: relaxed LDS accesses (a volatile pointer to it compiled to flat loads, each followed by a wait for every outstanding index read)
#define HAO_OVF() __hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define HAO_OVF_SET() __hip_atomic_store(&s_ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// ---- pieces the seed kernels share (seed_bin_kernel here, seed_bin3_kernel in hao_query3.cuh); all 256 threads call them ----
// QL staging: the read's minimizers that have anchors, in order (stable compaction): l_ao[k] = first anchor of the k-th of them (relative to the read), l_ss[k] = its list
// start | its index in the read's full minimizer list << 48 | its strand << 63; l_ao[nk] = n.  Returns nk.  (The caller synchronises before it reads the tables.)
__device__ __forceinline__ uint32_t hao_seed_stage_nonempty(uint32_t *l_ao, uint64_t *l_ss, uint32_t *s_wt /* [4] */, const uint64_t *g_ao, const uint64_t *g_ss, const uint64_t *g_info,
		uint64_t s, uint32_t nq, uint32_t n)
{
	const uint32_t tid = threadIdx.x; const int wv = tid >> 6, lane = tid & 63;
	uint32_t nk = 0;
	for (uint32_t b = 0; b < nq; b += 256) {
		const uint32_t q = b + tid; uint32_t a0 = 0, a1 = 0;
		if (q < nq) { a0 = (uint32_t)(g_ao[q] - s); a1 = (uint32_t)(g_ao[q + 1] - s); }      // (a_off has an entry past the batch's last minimizer)
		const bool ne = a1 > a0; const unsigned long long bal = __ballot(ne);
		if (lane == 0) s_wt[wv] = (uint32_t)__popcll(bal);
		__syncthreads();
		uint32_t k = nk + (uint32_t)__popcll(bal & ((1ULL << lane) - 1)); for (int w = 0; w < wv; ++w) k += s_wt[w];
		if (ne) { l_ao[k] = a0; l_ss[k] = g_ss[q] | (uint64_t)q << 48 | (uint64_t)hao_info_rev(g_info[q]) << 63; }
		nk += s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
		__syncthreads();
	}
	if (tid == 0) l_ao[nk] = n;
	return nk;
}
// the D distinct bins of the table hk[CAP] as (bin key << 32 | slot), ascending, in sk[0 .. D) (bitonic sort over the next power of two P, padded with ~0); *s_c = 0 on entry.
// Returns P; ends with a barrier.
template<uint32_t CAP> __device__ __forceinline__ uint32_t hao_seed_sort_bins(const uint32_t *hk, uint64_t *sk, uint32_t *s_c, uint32_t D)
{
	const uint32_t tid = threadIdx.x;
	uint32_t P = 2; while (P < D) P <<= 1;
	for (uint32_t i = tid; i < CAP; i += 256) if (hk[i] != 0xffffffffu) sk[atomicAdd(s_c, 1u)] = (uint64_t)hk[i] << 32 | i;
	for (uint32_t i = D + tid; i < P; i += 256) sk[i] = ~0ULL;
	__syncthreads();
	for (uint32_t k = 2; k <= P; k <<= 1)
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t i = tid; i < P; i += 256) {
				const uint32_t x = i ^ j;
				if (x > i) { const uint64_t a = sk[i], b = sk[x]; if ((a > b) == ((i & k) == 0)) { sk[i] = b; sk[x] = a; } }
			}
			__syncthreads();
		}
	return P;
}

// Q2-Q5 fused, one workgroup per read: pass A walks the read's anchors (minimizer q, list entry j) and counts bins, pass B walks them again
// (the read's slices of the position lists are L2-resident by then) and writes each k_mer_hit to its final place.
// HBM traffic per anchor: one 8-byte index record in, one 16-byte hit out.
// Anchor x of the read (generation order = (q, list order)) is located through the read's exclusive anchor offsets, staged in LDS
// with the list starts; a wave walks its chunk in order, so a lane only steps forward from the tile's first minimizer.
struct hao_seed_args {
	const uint64_t *mz_off, *mz_info; uint64_t rid_lo, mz0;
	const uint64_t *s_start; const uint32_t *s_n; const uint64_t *a_off, *seg, *sinfo; const uint32_t *len, *q_pos, *q_cnt;
	hao_hit_t *hits; uint64_t *g_tmp, *g_cnt; uint64_t n_sel; uint32_t qcap; int tb;
	unsigned long long *dbg;      // optional: per-phase wall-clock ticks summed over workgroups (HAO_DBG_PRINT=seed)
	uint16_t *hq;      // optional (delivery path): index of the query minimizer of every hit (saturating at 65535), next to hits[]: the wire packer's codes need it (hao_deliver.cuh)
};

// Three launches cover a batch: <512 slots, tier 0> takes every read and gives up (appends the read to ovf_list) when its bins do not fit in
// one round - the small table keeps many workgroups per CU for the common reads; <1024 slots, tier 1> takes the listed reads and gives up the same
// way (a read that crosses repeat families meets hundreds of targets: on the 250 Mb repeat-rich set nearly every read overflowed the small table, and the
// 2048-slot kernel - 60 KB of LDS, 121 VGPRs: two workgroups per CU - took 23 ms per batch against 1.6 ms for the first launch); <2048 slots, tier 2>
// takes what is left, in as many (tid, rev) range rounds as it needs.
//
// Pass B writes through LDS.  A hit is 16 bytes and a read feeds ~100 bins, so storing hits where they are produced costs one 16-byte write
// request per hit (the request slots of the L2, not its bytes, are what ran out: the same stores issued in generation order made the whole
// kernel 1/3 faster).  The workgroup therefore walks the read in tiles of TILE (512) anchors: every wave ranks the hits of its quarter
// tile (same ranked scatter as before, per-wave counts per bin), one scan over the bins turns the counts into tile offsets, the hits are
// parked in LDS grouped by bin (in generation order inside a bin), and the tile is written out with consecutive lanes on consecutive
// addresses: ~5 hits of a bin at a time (TILE = 1024 doubles that but its LDS and registers leave 4 instead of 6 workgroups per CU: slower).
struct hao_stage_t { uint32_t offset, self_offset, cnt; };
// TIER 0: every read of the batch, gives up on overflow (-> ovf_list); TIER 1: the reads of in_list, gives up on overflow (-> ovf_list); TIER 2: the reads of in_list, in
// as many (tid, rev) range rounds as their bins need
// QL: every read of the batch has at most qcap minimizers (the host knows), so the per-minimizer tables always live in LDS: no uniform LDS / global
// branch at every use, only the non-empty minimizers are staged (list start | query minimizer index << 48 | strand << 63), located by hao_seed_locate
template<int CAPLOG, int TIER, uint32_t HAO_SEED_TILE, bool QL>
__global__ __launch_bounds__(256, CAPLOG == 9 ? 6 : CAPLOG == 10 ? 3 : 1) void seed_bin_kernel(hao_seed_args S, const uint32_t *in_list, const unsigned long long *in_cnt, uint32_t *ovf_list, unsigned long long *ovf_cnt)
{
	constexpr bool FIRST = TIER == 0, GIVEUP = TIER < 2;
	constexpr uint32_t CAP = 1u << CAPLOG, MAXD = CAP - 288;      // at most MAXD + 256 bins are ever inserted (one per thread after the table fills), so probing terminates; CAP >= 512
	constexpr int UA = 4;                        // tiles per lane in flight in the counting pass (chunks are multiples of 64 * UA anchors)
	constexpr uint32_t STAGE_BYTES = HAO_SEED_TILE * (sizeof(hao_stage_t) + 4), SORT_BYTES = CAP * 12, UNION_BYTES = STAGE_BYTES > SORT_BYTES ? STAGE_BYTES : SORT_BYTES;
	extern __shared__ uint32_t bs_smem[];
	uint32_t *hk = bs_smem;                      // [CAP]    bin key (tid << 1 | rev) per slot
	uint32_t *cwd = hk + CAP;                    // [CAP]    hits of the bin (pass A), its first output position (between the passes); pass B, per tile:
	                                             //          output position of the bin minus its offset in the staged tile
	uint32_t *bl = cwd + CAP;                    // [CAP]    read length of the slot's target (opposite-strand offsets)
	uint16_t *wc = (uint16_t*)(bl + CAP);        // [4][CAP] pass B, per tile: per-wave counts, then each wave's first staged entry of the bin
	uint16_t *rk = wc + 4 * CAP;                 // [CAP]    (unused since the scatter pass matches on slots; kept so that the host's LDS size formula stands)
	uint64_t *sk = (uint64_t*)(rk + CAP);        // [CAP]    (bin key << 32 | slot), sorted          } between the passes
	uint32_t *tot = (uint32_t*)(sk + CAP);       // [CAP]    per-rank totals                          }
	hao_stage_t *stage = (hao_stage_t*)sk;       // [TILE]   pass B: the tile's hits grouped by bin   } same memory
	uint16_t *sslot = (uint16_t*)(stage + HAO_SEED_TILE);   // [TILE] slot of the staged hit          }
	uint16_t *sq = sslot + HAO_SEED_TILE;                   // [TILE] query minimizer of the staged hit }
	uint64_t *l_ss = (uint64_t*)((char*)sk + UNION_BYTES);  // [qcap]   list start of minimizer q in the position index | strand of the minimizer << 63
	uint32_t *l_ao = (uint32_t*)(l_ss + S.qcap); // [qcap+1] first anchor of minimizer q, relative to the read
	__shared__ uint32_t s_nd, s_ovf, s_c, s_wt[4]; __shared__ uint64_t s_ws[4], s_all;
	uint64_t *g_tmp = S.g_tmp;
	if (!FIRST && blockIdx.x >= *in_cnt) return;
	const uint64_t r = FIRST ? blockIdx.x : in_list[blockIdx.x], s = S.seg[r], e = S.seg[r + 1]; const uint32_t n = (uint32_t)(e - s);
	const int wv = threadIdx.x >> 6, lane = hao_lane(); const uint32_t tid = threadIdx.x;
	if (FIRST && r == 0 && tid == 0) S.g_cnt[S.n_sel] = 0;
	if (n == 0) { if (tid == 0) S.g_cnt[r] = 0; return; }
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	const bool qlds = QL || nq <= S.qcap;          // very long reads keep the per-minimizer table in global memory (uniform branches, no flat accesses)
	const uint64_t *g_ao = S.a_off + li0, *g_ss = S.s_start + li0, *g_info = S.mz_info + m0;
	uint32_t nk = nq;                              // staged minimizers (QL: the non-empty ones)
	if (QL) {
		nk = hao_seed_stage_nonempty(l_ao, l_ss, s_wt, g_ao, g_ss, g_info, s, nq, n);      // only the minimizers that have anchors are staged
	} else if (qlds) {
		for (uint32_t q = tid; q < nq; q += 256) { l_ss[q] = g_ss[q] | (uint64_t)hao_info_rev(g_info[q]) << 63; l_ao[q] = (uint32_t)(g_ao[q] - s); }
		if (tid == 0) l_ao[nq] = n;
	}
	__syncthreads();
	unsigned long long tk0 = S.dbg ? wall_clock64() : 0, tk1 = 0, tk2 = 0;
#define HAO_AO(q) (QL ? l_ao[q] : (qlds ? l_ao[q] : ((q) >= nq ? n : (uint32_t)(g_ao[q] - s))))
#define HAO_SS(q) (QL ? l_ss[q] : (qlds ? l_ss[q] : (g_ss[q] | (uint64_t)hao_info_rev(g_info[q]) << 63)))
#define HAO_START(sv) ((sv) & (QL ? (1ULL << 48) - 1 : ~(1ULL << 63)))      /* list start of a staged minimizer word */
#define HAO_QIDX(q, sv) (QL ? (uint32_t)((sv) >> 48) & 0xfffu : (q))      /* index of the minimizer in the read's full list */
	const uint32_t chunk = ((n + 3) / 4 + 64 * UA - 1) / (64 * UA) * (64 * UA), c0 = min(n, wv * chunk), c1 = min(n, c0 + chunk);
	uint32_t q_c0 = 0;       // minimizer holding anchor c0: last q with AO(q) <= c0 (binary search, uniform in the wave)
	if (c0 < c1) { uint32_t lo_ = 0, hi_ = nk; while (hi_ - lo_ > 1) { const uint32_t md = (lo_ + hi_) >> 1; if (HAO_AO(md) <= c0) lo_ = md; else hi_ = md; } q_c0 = lo_; }
	const uint32_t k_end = 2u << S.tb;
	uint32_t lo = 0, placed = 0, ngr = 0, last_tid = 0xffffffffu;
	while (lo < k_end) {
		uint32_t hi = k_end;
		for (;;) {      // count the bins of [lo, hi); shrink the range until they fit the table
			for (uint32_t i = tid; i < CAP; i += 256) { hk[i] = HAO_BIN_EMPTY; cwd[i] = 0; ((uint32_t*)wc)[i] = 0; ((uint32_t*)wc)[CAP + i] = 0; }
			if (tid == 0) { s_nd = 0; s_ovf = 0; s_c = 0; }
			__syncthreads();
			uint32_t qc = q_c0;
			for (uint32_t t0 = c0; t0 < c1; t0 += 64 * UA) {      // UA independent index reads in flight per lane
				uint64_t yv[UA]; uint32_t zr[UA];
#pragma unroll
				for (int u = 0; u < UA; ++u) {
					const uint32_t x = t0 + u * 64 + lane; const bool act = x < c1; uint32_t q = qc;
					if (QL) q = hao_seed_locate(l_ao, nk, qc, t0 + u * 64, lane);
					else { if (act) { while (HAO_AO(q + 1) <= x) ++q; } qc = (uint32_t)__builtin_amdgcn_readlane((int)q, 63); }
					const uint64_t sv = HAO_SS(q);
					yv[u] = act ? S.sinfo[HAO_START(sv) + (x - HAO_AO(q))] : 0; zr[u] = (uint32_t)(sv >> 63);
				}
				if (HAO_OVF()) break;
				HAO_LOCKSTEP();      // the wave has read the flag as one: its lanes may set it from here on
#pragma unroll
				for (int u = 0; u < UA; ++u) {
					const uint32_t x = t0 + u * 64 + lane, kk = hao_info_rid(yv[u]) << 1 | (zr[u] ^ hao_info_rev(yv[u]));
					if (x < c1 && kk >= lo && kk < hi && !HAO_OVF()) {        // a thread starts at most one insertion after the table was declared full
						uint32_t slot = (kk * 2654435761u) >> (32 - CAPLOG);
						for (uint32_t pr = 0; ; ++pr) {
							if (pr == CAP) { HAO_OVF_SET(); break; }
							const uint32_t old = atomicCAS(&hk[slot], HAO_BIN_EMPTY, kk);
							if (old == HAO_BIN_EMPTY) { if (atomicAdd(&s_nd, 1u) >= MAXD) HAO_OVF_SET(); break; }
							if (old == kk) break;
							slot = (slot + 1) & (CAP - 1);
						}
						atomicAdd(&cwd[slot], 1u);
					}
				}
			}
			__syncthreads();
			const bool ovf = HAO_OVF() != 0;
			__syncthreads();
			if (!ovf) break;
			if (GIVEUP) { if (tid == 0) ovf_list[atomicAdd(ovf_cnt, 1ULL)] = (uint32_t)r; return; }      // left to the launch with the bigger table
			hi = lo + (hi - lo) / 2;      // hi - lo >= 2 here: one bin always fits
		}
		if (S.dbg) tk1 = wall_clock64();
		const uint32_t D = s_nd;
		if (D) {
			const uint32_t P = hao_seed_sort_bins<CAP>(hk, sk, &s_c, D);
			for (uint32_t d = tid; d < D; d += 256) { const uint32_t slot = (uint32_t)sk[d]; bl[slot] = S.len[(uint32_t)(sk[d] >> 33)]; tot[d] = cwd[slot]; }
			__syncthreads();
			// exclusive scan over the sorted bins of (hits, group starts), packed as starts << 32 | hits; thread t owns bins [t*per, (t+1)*per)
			const uint32_t per = P >= 256 ? P / 256 : 1, d0 = tid * per; uint64_t mine = 0;
			for (uint32_t d = d0; d < d0 + per && d < D; ++d) {
				const uint32_t t_k = (uint32_t)(sk[d] >> 33), t_p = d ? (uint32_t)(sk[d - 1] >> 33) : last_tid;
				mine += (uint64_t)(t_k != t_p) << 32 | tot[d];
			}
			uint64_t inc = mine;
#pragma unroll
			for (int dl = 1; dl < 64; dl <<= 1) { const uint64_t y = __shfl_up(inc, dl); if (lane >= dl) inc += y; }
			if (lane == 63) s_ws[wv] = inc;
			__syncthreads();
			uint64_t ex = inc - mine; for (int x = 0; x < wv; ++x) ex += s_ws[x];
			if (tid == 255) s_all = ex + mine;
			for (uint32_t d = d0; d < d0 + per && d < D; ++d) {
				const uint32_t slot = (uint32_t)sk[d], t_k = (uint32_t)(sk[d] >> 33), t_p = d ? (uint32_t)(sk[d - 1] >> 33) : last_tid;
				if (t_k != t_p) { g_tmp[s + ngr + (uint32_t)(ex >> 32)] = (uint64_t)t_k << 32 | (placed + (uint32_t)ex); ex += 1ULL << 32; }
				cwd[slot] = placed + (uint32_t)ex;
				ex += tot[d];
			}
			const uint32_t last_tid_next = (uint32_t)(sk[D - 1] >> 33);      // (the staged tiles reuse sk / tot)
			__syncthreads();
			const uint64_t all = s_all;
			if (S.dbg) tk2 = wall_clock64();
			constexpr int NU = HAO_SEED_TILE / 256;      // 64-anchor sub-tiles per wave and tile
			uint16_t *wcw = wc + wv * CAP;
			constexpr uint32_t SPT = CAP / 256;      // thread t owns slots [t * SPT, (t + 1) * SPT) in the per-tile scan and keeps their next output positions
			uint32_t ob[SPT];
#pragma unroll
			for (uint32_t k = 0; k < SPT; ++k) ob[k] = cwd[tid * SPT + k];
			uint32_t qw = 0;      // minimizer holding the first anchor of this wave's quarter of the tile
			uint64_t yv[NU], ype[NU], yne[NU]; uint32_t qv[NU], qz[NU];      // qz: the minimizer's index in the read's full list | its strand << 31 (kept in a register:
			                                                                // every LDS store in between would make the compiler read the staged word again)
			// (requesting a tile's records one tile ahead costs 12 VGPRs = one wave per SIMD less, and loses: 70.7 against 65.8 ms per step)
			auto request = [&](const uint32_t T0) {
				const uint32_t x0 = T0 + wv * (HAO_SEED_TILE / 4);
				if (x0 < n)      // gallop: 64 candidates per step (the offsets do not decrease)
					for (;;) {
						const uint32_t t = qw + 1 + lane;
						const unsigned long long le = __ballot(t <= nk && HAO_AO(t) <= x0);
						const int adv = le == ~0ULL ? 64 : __ffsll((long long)~le) - 1;
						qw += adv; if (adv < 64) break;
					}
				uint32_t qc = qw;
#pragma unroll
				for (int u = 0; u < NU; ++u) {
					const uint32_t x = x0 + u * 64 + lane; const bool act = x < n; uint32_t q = qc;
					if (QL) q = hao_seed_locate(l_ao, nk, qc, x0 + u * 64, lane);
					else { if (act) { while (HAO_AO(q + 1) <= x) ++q; } qc = (uint32_t)__builtin_amdgcn_readlane((int)q, 63); }
					qv[u] = q;
					const uint64_t sv = HAO_SS(q); qz[u] = HAO_QIDX(q, sv) | (uint32_t)(sv >> 63) << 31;
					const uint32_t a0 = HAO_AO(q), j = x - a0; const uint64_t ad = HAO_START(sv) + j;
					yv[u] = act ? S.sinfo[ad] : 0;
					// list neighbours normally sit in the adjacent lanes; only the lanes at a tile edge fetch theirs
					ype[u] = (act && lane == 0 && j > 0) ? S.sinfo[ad - 1] : ~0ULL;
					yne[u] = (act && (lane == 63 || x + 1 == n) && j + 1 < HAO_AO(q + 1) - a0) ? S.sinfo[ad + 1] : ~0ULL;
				}
			};
			for (uint32_t T0 = 0; T0 < n; T0 += HAO_SEED_TILE) {
				const uint32_t x0 = T0 + wv * (HAO_SEED_TILE / 4);
				request(T0);
				uint32_t qp[NU], qn[NU];      // the two words of the query minimizer: first needed when the hit is staged
#pragma unroll
				for (int u = 0; u < NU; ++u) { const bool act = x0 + u * 64 + lane < n; const uint32_t qi = qz[u] & 0x7fffffffu; qp[u] = act ? S.q_pos[li0 + qi] : 0; qn[u] = act ? S.q_cnt[li0 + qi] : 0; }
				uint32_t ps[NU], po[NU];      // slot | rank inside the wave's quarter tile << 16 (or ~0: no hit); k_mer_hit::offset
#pragma unroll
				for (int u = 0; u < NU; ++u) {
					const uint32_t x = x0 + u * 64 + lane, q = qv[u]; uint64_t y = yv[u];
					const uint32_t zrev = qz[u] >> 31;
					const uint32_t tidk = hao_info_rid(y), rev = zrev ^ hao_info_rev(y), kk = tidk << 1 | rev;
					const bool inr = x < n && kk >= lo && kk < hi;
					// target of the previous / next entry of my list (0xffffffff: none): lane - 1 / lane + 1 hold them unless they belong to another
					// minimizer (then I am the first / last entry of my list) or I sit at a tile edge (fetched above)
					const uint32_t yt = hao_info_rid(y);
					uint32_t t_up = hao_wave_shr1(yt, 0u), t_dn = hao_wave_shl1(yt, 0u);             // cross-lane moves (DPP): all lanes, before any branch
					const uint32_t q_up = hao_wave_shr1(q, 0xffffffffu), q_dn = hao_wave_shl1(q, 0xffffffffu);
					if (lane == 0) t_up = ype[u] == ~0ULL ? 0xffffffffu : hao_info_rid(ype[u]);
					else if (q_up != q) t_up = 0xffffffffu;
					if (lane == 63 || x + 1 >= n) t_dn = yne[u] == ~0ULL ? 0xffffffffu : hao_info_rid(yne[u]);
					else if (q_dn != q) t_dn = 0xffffffffu;
					if (inr && rev) {
						// opposite-strand hits of one k-mer in one target must come out by DEscending target position (ascending other_off,
						// anchor.cpp:1023): inside the (rare) run of list entries with the same target, the anchor at rev position k takes the
						// record of rev entry R-1-k
						const uint32_t a0 = HAO_AO(q), nl = HAO_AO(q + 1) - a0, j = x - a0;
						const bool pv = t_up == tidk, nx = t_dn == tidk;
						if (pv || nx) {
							const uint64_t st = HAO_START(HAO_SS(q));
							uint32_t ja = j, jb = j;
							while (ja > 0 && hao_info_rid(S.sinfo[st + ja - 1]) == tidk) --ja;
							while (jb + 1 < nl && hao_info_rid(S.sinfo[st + jb + 1]) == tidk) ++jb;
							uint32_t k = 0, R = 0, z;
							for (z = ja; z <= jb; ++z) if (zrev != hao_info_rev(S.sinfo[st + z])) { if (z < j) ++k; ++R; }
							const uint32_t want = R - 1 - k; uint32_t seen = 0;
							for (z = ja; z <= jb; ++z) if (zrev != hao_info_rev(S.sinfo[st + z])) { if (seen == want) { y = S.sinfo[st + z]; break; } ++seen; }
						}
					}
					uint32_t slot = (kk * 2654435761u) >> (32 - CAPLOG);
					if (inr) while (hk[slot] != kk) slot = (slot + 1) & (CAP - 1);
					// (slot < CAP on every lane: the LDS words of a lane without a hit are read unconditionally and never used - no divergent branch around a load)
					const unsigned long long m = hao_match_key<CAPLOG>(slot, inr);
					const uint32_t before = __popcll(m & ((1ULL << lane) - 1)), base = wcw[slot], tlen = bl[slot];
					ps[u] = inr ? (slot | (base + before) << 16) : 0xffffffffu;
					// k_mer_hit::offset (anchor.cpp:1021-1023,1059-1064): target coordinate in the strand of the hit
					po[u] = rev ? tlen - 1 - (hao_info_pos(y) + 1 - hao_info_span(y)) : hao_info_pos(y);
					HAO_LOCKSTEP();      // every lane of the match group has read wcw[slot]
					if (inr && before == 0) wcw[slot] = (uint16_t)(base + __popcll(m));
				}
				__syncthreads();
				// tile offsets of the bins (slot order: any order keeps a bin's hits together)
				uint32_t c_[SPT][4], mine_t = 0;
#pragma unroll
				for (uint32_t k = 0; k < SPT; ++k) {
#pragma unroll
					for (int w = 0; w < 4; ++w) { c_[k][w] = wc[w * CAP + tid * SPT + k]; mine_t += c_[k][w]; }
				}
				const uint32_t inc_t = hao_wave_incl_scan_u32(mine_t);
				if (lane == 63) s_wt[wv] = inc_t;
				__syncthreads();
				uint32_t ex_t = inc_t - mine_t; for (int x = 0; x < wv; ++x) ex_t += s_wt[x];
				const uint32_t tile_n = s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
#pragma unroll
				for (uint32_t k = 0; k < SPT; ++k) {
					const uint32_t sl = tid * SPT + k, tt = c_[k][0] + c_[k][1] + c_[k][2] + c_[k][3];
					if (tt) {
						cwd[sl] = ob[k] - ex_t; ob[k] += tt;
						wc[sl] = (uint16_t)ex_t; wc[CAP + sl] = (uint16_t)(ex_t + c_[k][0]); wc[2 * CAP + sl] = (uint16_t)(ex_t + c_[k][0] + c_[k][1]); wc[3 * CAP + sl] = (uint16_t)(ex_t + c_[k][0] + c_[k][1] + c_[k][2]);
						ex_t += tt;
					}
				}
				__syncthreads();
#pragma unroll
				for (int u = 0; u < NU; ++u)
					if (ps[u] != 0xffffffffu) {
						const uint32_t slot = ps[u] & 0xffffu, at = wcw[slot] + (ps[u] >> 16);
						hao_stage_t z; z.offset = po[u]; z.self_offset = qp[u]; z.cnt = qn[u];
						const uint32_t qi = qz[u] & 0x7fffffffu;
						stage[at] = z; sslot[at] = (uint16_t)slot; sq[at] = (uint16_t)(qi < 65535u ? qi : 65535u);
					}
				__syncthreads();
				for (uint32_t at = tid; at < tile_n; at += 256) {
					const uint32_t slot = sslot[at], kk = hk[slot]; const hao_stage_t z = stage[at];
					hao_hit_t h; h.w0 = kk >> 1 | kk << 31; h.offset = z.offset; h.self_offset = z.self_offset; h.cnt = z.cnt;
					S.hits[s + (uint32_t)(cwd[slot] + at)] = h;      // (32-bit sum: cwd may have wrapped below zero)
					if (S.hq) S.hq[s + (uint32_t)(cwd[slot] + at)] = sq[at];
				}
				for (uint32_t i = lane; i < CAP / 2; i += 64) ((uint32_t*)wcw)[i] = 0;      // own row only: nobody else reads it before the next tile's barrier
			}
			last_tid = last_tid_next;
			placed += (uint32_t)all; ngr += (uint32_t)(all >> 32);
			__syncthreads();
		}
		lo = hi;
	}
	if (tid == 0) S.g_cnt[r] = ngr;
	if (S.dbg && tid == 0) { const unsigned long long tk3 = wall_clock64(); atomicAdd(S.dbg, tk1 - tk0); atomicAdd(S.dbg + 1, tk2 - tk1); atomicAdd(S.dbg + 2, tk3 - tk2); atomicAdd(S.dbg + 3, 1ULL); }
#undef HAO_AO
#undef HAO_SS
#undef HAO_START
#undef HAO_QIDX
}

// ---- group table ----
// seg_bin_sort_kernel leaves the groups of read r at g_tmp[seg[r] ..) as (tid << 32 | first hit, relative to the read).
// Groups are then handed to the chain kernels through per-size-class work lists of self-contained 32-byte entries
// (no dependent index loads in the consumers), the biggest classes first: a slow group of a big class is found early
// and its sequential DP overlaps the quick checks of the smaller classes.
#define HAO_NCLS 7
#define HAO_TINY_MAX 8          // groups up to this many hits: class 0, one LANE per group (chain_tiny_kernel)
struct hao_gent { uint32_t g, r; uint64_t start; uint32_t n, yid, xl, yl; };      // 32 bytes
__host__ __device__ __forceinline__ int hao_size_class(uint32_t n) { return n <= HAO_TINY_MAX ? 0 : n <= 64 ? 1 : n <= 128 ? 2 : n <= 256 ? 3 : n <= 512 ? 4 : n <= 2048 ? 5 : 6; }

// per-read group counts by class, laid out class-major: cc[x * (n_sel + 1) + r] (entry n_sel of every class = 0).  ONE exclusive scan of this
// array gives every (class, read) its slot range in the concatenated class lists (deterministic, no atomics).  One wave per read.
__global__ __launch_bounds__(256) void groups_classify_kernel(const uint64_t *g_tmp, const uint64_t *seg, const uint64_t *g_cnt, uint64_t n_sel, uint64_t *cc)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r > n_sel) return;
	if (r == n_sel) { if (hao_lane() < HAO_NCLS) cc[hao_lane() * (n_sel + 1) + n_sel] = 0; return; }
	const uint64_t s = seg[r], ng = g_cnt[r]; const uint32_t n = (uint32_t)(seg[r + 1] - s);
	uint32_t c[HAO_NCLS] = {0, 0, 0, 0, 0, 0, 0};
	for (uint64_t k = hao_lane(); k < ng; k += 64) {
		const uint32_t st = (uint32_t)g_tmp[s + k], en = k + 1 < ng ? (uint32_t)g_tmp[s + k + 1] : n;
		const int cl = hao_size_class(en - st);
#pragma unroll
		for (int x = 0; x < HAO_NCLS; ++x) c[x] += cl == x;
	}
#pragma unroll
	for (int x = 0; x < HAO_NCLS; ++x) {
		uint32_t v = c[x];
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
		if (hao_lane() == 0) cc[x * (n_sel + 1) + r] = v;
	}
}

struct hao_cls_layout { uint64_t base[HAO_NCLS + 1]; };
// class list bases (and the group total) from the scanned table: out[x] = co[x * (n_sel + 1)], out[HAO_NCLS] = total
__global__ void groups_layout_kernel(const uint64_t *co, uint64_t n_sel, unsigned long long *out)
{
	const uint32_t x = threadIdx.x;
	if (x < HAO_NCLS) out[x] = co[x * (n_sel + 1)];
	else if (x == HAO_NCLS) out[x] = co[HAO_NCLS * (n_sel + 1) - 1];      // last entry is a zero slot: exclusive sum there = all groups
}

// group arrays (g_off, g_start, g_read) + the class work lists; one wave per read.  co = exclusive scan of the class-major count table.
__global__ __launch_bounds__(256) void groups_compact_kernel(const uint64_t *g_tmp, const uint64_t *seg, const uint64_t *co, uint64_t n_sel, uint64_t rid_lo, const uint32_t *len,
		uint64_t *g_off, uint64_t *g_start, uint32_t *g_read, uint8_t *g_cls, hao_gent *glist)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r > n_sel) return;
	const int lane = hao_lane();
	uint64_t pos[HAO_NCLS], g0 = 0, g1 = 0;           // next list slot of each class for this read; groups before / through this read
#pragma unroll
	for (int x = 0; x < HAO_NCLS; ++x) { const uint64_t b = co[x * (n_sel + 1)]; pos[x] = co[x * (n_sel + 1) + r]; g0 += pos[x] - b; if (r < n_sel) g1 += co[x * (n_sel + 1) + r + 1] - b; }
	if (lane == 0) g_off[r] = g0;
	if (r == n_sel) return;
	const uint64_t s = seg[r], ng = g1 - g0; const uint32_t n = (uint32_t)(seg[r + 1] - s), xl = len[rid_lo + r];
	for (uint64_t kb = 0; kb < ng; kb += 64) {
		const uint64_t k = kb + lane; const bool act = k < ng;
		hao_gent e; int cl = -1;
		if (act) {
			const uint64_t w = g_tmp[s + k]; const uint32_t st = (uint32_t)w, en = k + 1 < ng ? (uint32_t)g_tmp[s + k + 1] : n;
			e.g = (uint32_t)(g0 + k); e.r = (uint32_t)r; e.start = s + st; e.n = en - st; e.yid = (uint32_t)(w >> 32); e.xl = xl; e.yl = len[e.yid];
			g_start[g0 + k] = e.start; g_read[g0 + k] = (uint32_t)r;
			cl = hao_size_class(e.n); g_cls[g0 + k] = (uint8_t)cl;
		}
#pragma unroll
		for (int x = 0; x < HAO_NCLS; ++x) {
			const unsigned long long m = __ballot(cl == x);
			if (cl == x) glist[pos[x] + __popcll(m & ((1ULL << lane) - 1))] = e;
			pos[x] += __popcll(m);
		}
	}
}

