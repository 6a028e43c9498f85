// Seed kernel for reads that meet MANY targets (the 1024- and 2048-slot launches: reads that cross repeat families - on the repeat-rich 250 Mb set nearly
// every read, 300 - 1700 (target, strand) bins each).  Same algorithm as seed_bin_kernel (hao_query.cuh: bins in an LDS hash table, sorted bins = the group list,
// one ranked stable scatter), but the scatter pass is cut for that regime:
//   * with ~1000 bins a 512-anchor tile holds less than one hit per bin, so parking the tile in LDS to write a bin's hits together buys nothing - while the
//     per-tile scan over all 1024 / 2048 slots (8 - 32 LDS reads per thread and tile), its four barriers and the staging traffic are what the kernel spends its
//     time on (profiles/r04: 7.6 + 13.4 ms per batch for the two launches against 1.5 ms for the 512-slot launch that precedes them);
//   * so: pass A counts the hits of every bin PER WAVE (a wave owns a contiguous quarter of the read's anchors), the sort + scan give the bin starts, wave w
//     starts a bin at the bin's start + the counts of the waves before it, and pass B is wave-private and barrier-free: NU x 64 index records in flight per wave,
//     hits ranked inside their 64-anchor window by ballot match groups, the bin's running position in an LDS word per (wave, slot), every hit stored where it
//     belongs.  (The same re-cut WITH per-wave staging lost on the 512-slot launch - round 4's hao_query2.cuh, removed in round 6 - where tiles do hold several hits per bin.)
// QL only: every read's minimizer table fits the LDS (the host launches seed_bin_kernel<.., false> otherwise).
#pragma once
#include "hao_query.cuh"

template<int CAPLOG> struct hao_seed3_lds {      // byte layout of the dynamic LDS (host and device agree through this struct)
	static constexpr uint32_t CAP = 1u << CAPLOG;
	static constexpr uint32_t FIXED = CAP * 34;      // hk u32[CAP], cur u32[4][CAP], tb u32[CAP] (totals by rank, then target lengths by slot), sk u64[CAP], rk u16[CAP]; + 12 B per staged minimizer + 16
};

template<int CAPLOG, int TIER, int NU>
__global__ __launch_bounds__(256) void seed_bin3_kernel(hao_seed_args S, const uint32_t *in_list, const unsigned long long *in_cnt, uint32_t *ovf_list, unsigned long long *ovf_cnt)
{
	constexpr bool FIRST = TIER == 0, GIVEUP = TIER < 2;
	constexpr uint32_t CAP = 1u << CAPLOG, MAXD = CAP - 288;      // at most MAXD + 256 bins are ever inserted (one per thread after the table fills), so probing terminates
	constexpr int UA = 4;
	extern __shared__ uint32_t bs3_smem[];
	const int wv = threadIdx.x >> 6, lane = hao_lane(); const uint32_t tid = threadIdx.x;
	uint32_t *hk = bs3_smem;                     // [CAP]    bin key (tid << 1 | rev) per slot
	uint32_t *cur = hk + CAP;                    // [4][CAP] pass A: hits of the bin in each wave's quarter of the read; pass B: the wave's next output position in the bin
	uint32_t *tb = cur + 4 * CAP;                // [CAP]    hits per RANK (sort .. scan), then read length of the SLOT's target (opposite-strand offsets)
	uint64_t *sk = (uint64_t*)(tb + CAP);        // [CAP]    (bin key << 32 | slot), sorted
	uint16_t *rk = (uint16_t*)(sk + CAP);        // [CAP]    (unused since pass B matches on slots; the layout - hao_seed3_lds - is kept)
	uint64_t *l_ss = (uint64_t*)(rk + CAP);      // [qcap]   non-empty minimizers: list start | index in the read's minimizer list << 48 | strand << 63
	uint32_t *l_ao = (uint32_t*)(l_ss + S.qcap); // [qcap+1] their first anchor, relative to the read
	__shared__ uint32_t s_nd, s_ovf, s_c, s_wt[4]; __shared__ uint64_t s_ws[4], s_all;
	uint64_t *g_tmp = S.g_tmp;
	if (!FIRST && blockIdx.x >= *in_cnt) return;
	const uint64_t r = FIRST ? blockIdx.x : in_list[blockIdx.x], s = S.seg[r], e = S.seg[r + 1]; const uint32_t n = (uint32_t)(e - s);
	if (FIRST && r == 0 && tid == 0) S.g_cnt[S.n_sel] = 0;
	if (n == 0) { if (tid == 0) S.g_cnt[r] = 0; return; }
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	const uint64_t *g_ao = S.a_off + li0, *g_ss = S.s_start + li0, *g_info = S.mz_info + m0;
	const uint32_t nk = hao_seed_stage_nonempty(l_ao, l_ss, s_wt, g_ao, g_ss, g_info, s, nq, n);      // only the minimizers that have anchors are staged (hao_query.cuh)
	__syncthreads();
	unsigned long long tk0 = S.dbg ? wall_clock64() : 0, tk1 = 0, tk2 = 0;
	// wave wv owns anchors [c0, c1) of the read in BOTH passes (a multiple of 64 * max(UA, NU) anchors)
	constexpr uint32_t CH = 64 * (UA > NU ? UA : NU);
	const uint32_t chunk = ((n + 3) / 4 + CH - 1) / CH * CH, c0 = min(n, wv * chunk), c1 = min(n, c0 + chunk);
	uint32_t q_c0 = 0;       // minimizer holding anchor c0: last k with ao[k] <= c0 (binary search, uniform in the wave)
	if (c0 < c1) { uint32_t lo_ = 0, hi_ = nk; while (hi_ - lo_ > 1) { const uint32_t md = (lo_ + hi_) >> 1; if (l_ao[md] <= c0) lo_ = md; else hi_ = md; } q_c0 = lo_; }
	const uint32_t k_end = 2u << S.tb;
	uint32_t lo = 0, placed = 0, ngr = 0, last_tid = 0xffffffffu;
	uint32_t *cw = cur + wv * CAP;
	while (lo < k_end) {
		uint32_t hi = k_end;
		for (;;) {      // count the bins of [lo, hi), per wave; shrink the range until they fit the table
			for (uint32_t i = tid; i < CAP; i += 256) { hk[i] = HAO_BIN_EMPTY; cur[i] = 0; cur[CAP + i] = 0; cur[2 * CAP + i] = 0; cur[3 * CAP + i] = 0; }
			if (tid == 0) { s_nd = 0; s_ovf = 0; s_c = 0; }
			__syncthreads();
			uint32_t qc = q_c0;
			for (uint32_t t0 = c0; t0 < c1; t0 += 64 * UA) {      // UA independent index reads in flight per lane
				uint64_t yv[UA]; uint32_t zr[UA];
#pragma unroll
				for (int u = 0; u < UA; ++u) {
					const uint32_t x = t0 + u * 64 + lane; const bool act = x < c1;
					const uint32_t q = hao_seed_locate(l_ao, nk, qc, t0 + u * 64, lane);
					const uint64_t sv = l_ss[q];
					yv[u] = act ? S.sinfo[(sv & ((1ULL << 48) - 1)) + (x - l_ao[q])] : 0; zr[u] = (uint32_t)(sv >> 63);
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
						atomicAdd(&cw[slot], 1u);
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
			for (uint32_t d = tid; d < D; d += 256) {
				const uint32_t slot = (uint32_t)sk[d];
				tb[d] = cur[slot] + cur[CAP + slot] + cur[2 * CAP + slot] + cur[3 * CAP + slot];
			}
			__syncthreads();
			// exclusive scan over the sorted bins of (hits, group starts), packed as starts << 32 | hits; thread t owns bins [t*per, (t+1)*per)
			const uint32_t per = P >= 256 ? P / 256 : 1, d0 = tid * per; uint64_t mine = 0;
			for (uint32_t d = d0; d < d0 + per && d < D; ++d) {
				const uint32_t t_k = (uint32_t)(sk[d] >> 33), t_p = d ? (uint32_t)(sk[d - 1] >> 33) : last_tid;
				mine += (uint64_t)(t_k != t_p) << 32 | tb[d];
			}
			uint64_t inc = mine;
#pragma unroll
			for (int dl_ = 1; dl_ < 64; dl_ <<= 1) { const uint64_t y = __shfl_up(inc, dl_); if (lane >= dl_) inc += y; }
			if (lane == 63) s_ws[wv] = inc;
			__syncthreads();
			uint64_t ex = inc - mine; for (int x = 0; x < wv; ++x) ex += s_ws[x];
			if (tid == 255) s_all = ex + mine;
			for (uint32_t d = d0; d < d0 + per && d < D; ++d) {
				const uint32_t slot = (uint32_t)sk[d], t_k = (uint32_t)(sk[d] >> 33), t_p = d ? (uint32_t)(sk[d - 1] >> 33) : last_tid;
				if (t_k != t_p) { g_tmp[s + ngr + (uint32_t)(ex >> 32)] = (uint64_t)t_k << 32 | (placed + (uint32_t)ex); ex += 1ULL << 32; }
				// the four waves fill consecutive parts of the bin, in wave order = generation order
				const uint32_t b0 = placed + (uint32_t)ex, n0 = cur[slot], n1 = cur[CAP + slot], n2 = cur[2 * CAP + slot];
				cur[slot] = b0; cur[CAP + slot] = b0 + n0; cur[2 * CAP + slot] = b0 + n0 + n1; cur[3 * CAP + slot] = b0 + n0 + n1 + n2;
				ex += tb[d];
			}
			const uint32_t last_tid_next = (uint32_t)(sk[D - 1] >> 33);
			__syncthreads();      // the totals by rank are dead: the same words take the target lengths by slot
			for (uint32_t d = tid; d < D; d += 256) tb[(uint32_t)sk[d]] = S.len[(uint32_t)(sk[d] >> 33)];
			const uint64_t all = s_all;
			__syncthreads();
			if (S.dbg) tk2 = wall_clock64();
			// ---- pass B: wave-private, no barrier until the end of the round ----
			uint32_t qc = q_c0;
			for (uint32_t t0 = c0; t0 < c1; t0 += 64 * NU) {
				uint64_t yv[NU]; uint32_t tpe[NU], tne[NU], qv[NU], qz[NU], qp[NU], qn[NU];      // qz: index of the minimizer in the read's full list | its strand << 31
#pragma unroll
				for (int u = 0; u < NU; ++u) {      // the tile's index records (+ the list neighbours a 64-anchor window's edges need, + the two words of the query minimizer)
					const uint32_t x = t0 + u * 64 + lane; const bool act = x < c1;
					const uint32_t q = hao_seed_locate(l_ao, nk, qc, t0 + u * 64, lane);
					qv[u] = q;
					const uint64_t sv = l_ss[q]; const uint32_t a0 = l_ao[q], j = x - a0, qi = (uint32_t)(sv >> 48) & 0xfffu; const uint64_t ad = (sv & ((1ULL << 48) - 1)) + j;
					qz[u] = qi | (uint32_t)(sv >> 63) << 31;
					yv[u] = act ? S.sinfo[ad] : 0;
					tpe[u] = (act && lane == 0 && j > 0) ? hao_info_rid(S.sinfo[ad - 1]) : 0xffffffffu;
					tne[u] = (act && (lane == 63 || x + 1 == n) && j + 1 < l_ao[q + 1] - a0) ? hao_info_rid(S.sinfo[ad + 1]) : 0xffffffffu;
					qp[u] = act ? S.q_pos[li0 + qi] : 0; qn[u] = act ? S.q_cnt[li0 + qi] : 0;
				}
#pragma unroll
				for (int u = 0; u < NU; ++u) {
					const uint32_t x = t0 + u * 64 + lane, q = qv[u]; uint64_t y = yv[u];
					const uint32_t zrev = qz[u] >> 31;
					const uint32_t tidk = hao_info_rid(y), rev = zrev ^ hao_info_rev(y), kk = tidk << 1 | rev;
					const bool inr = x < c1 && kk >= lo && kk < hi;
					// target of the previous / next entry of my list (0xffffffff: none): lane - 1 / lane + 1 hold them unless they belong to another
					// minimizer (then I am the first / last entry of my list) or I sit at a window edge (fetched with the records)
					uint32_t t_up = hao_wave_shr1(tidk, 0u), t_dn = hao_wave_shl1(tidk, 0u);             // cross-lane moves (DPP): all lanes, before any branch
					const uint32_t q_up = hao_wave_shr1(q, 0xffffffffu), q_dn = hao_wave_shl1(q, 0xffffffffu);
					if (lane == 0) t_up = tpe[u];
					else if (q_up != q) t_up = 0xffffffffu;
					if (lane == 63 || x + 1 >= n) t_dn = tne[u];
					else if (q_dn != q) t_dn = 0xffffffffu;
					if (inr && rev) {
						// opposite-strand hits of one k-mer in one target must come out by DEscending target position (ascending other_off,
						// anchor.cpp:1023): inside the (rare) run of list entries with the same target, the anchor at rev position k takes the
						// record of rev entry R-1-k
						const uint32_t a0 = l_ao[q], nl = l_ao[q + 1] - a0, j = x - a0;
						const bool pv = t_up == tidk, nx = t_dn == tidk;
						if (pv || nx) {
							const uint64_t st = l_ss[q] & ((1ULL << 48) - 1);
							uint32_t ja = j, jb = j;
							while (ja > 0 && hao_info_rid(S.sinfo[st + ja - 1]) == tidk) --ja;
							while (jb + 1 < nl && hao_info_rid(S.sinfo[st + jb + 1]) == tidk) ++jb;
							uint32_t k = 0, R_ = 0, z;
							for (z = ja; z <= jb; ++z) if (zrev != hao_info_rev(S.sinfo[st + z])) { if (z < j) ++k; ++R_; }
							const uint32_t want = R_ - 1 - k; uint32_t seen = 0;
							for (z = ja; z <= jb; ++z) if (zrev != hao_info_rev(S.sinfo[st + z])) { if (seen == want) { y = S.sinfo[st + z]; break; } ++seen; }
						}
					}
					uint32_t slot = (kk * 2654435761u) >> (32 - CAPLOG);
					if (inr) while (hk[slot] != kk) slot = (slot + 1) & (CAP - 1);
					const unsigned long long m = hao_match_key<CAPLOG>(slot, inr);      // (slots and bins correspond one to one: hao_query.cuh)
					const uint32_t before = __popcll(m & ((1ULL << lane) - 1)), base = cw[slot];
					HAO_LOCKSTEP();      // every lane of the match group has read cw[slot]
					if (inr && before == 0) cw[slot] = base + (uint32_t)__popcll(m);
					if (inr) {
						// k_mer_hit::offset (anchor.cpp:1021-1023,1059-1064): target coordinate in the strand of the hit
						hao_hit_t h; h.w0 = kk >> 1 | kk << 31; h.offset = rev ? tb[slot] - 1 - (hao_info_pos(y) + 1 - hao_info_span(y)) : hao_info_pos(y); h.self_offset = qp[u]; h.cnt = qn[u];
						const uint64_t at = s + (uint32_t)(base + before);
						S.hits[at] = h;
						if (S.hq) S.hq[at] = (uint16_t)(qz[u] & 0xfffu);
					}
				}
			}
			last_tid = last_tid_next;
			placed += (uint32_t)all; ngr += (uint32_t)(all >> 32);
			__syncthreads();
		}
		lo = hi;
	}
	if (tid == 0) S.g_cnt[r] = ngr;
	if (S.dbg && tid == 0) { const unsigned long long tk3 = wall_clock64(); atomicAdd(S.dbg, tk1 - tk0); atomicAdd(S.dbg + 1, tk2 - tk1); atomicAdd(S.dbg + 2, tk3 - tk2); atomicAdd(S.dbg + 3, 1ULL); }
}
