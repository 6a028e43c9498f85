// Seed kernel, round-4 form (minimizers_qgen0, anchor.cpp:987-1081): same bins-instead-of-digits algorithm as seed_bin_kernel (hao_query.cuh), with a
// scatter pass that has NO workgroup barrier.
//
// What the counters said about seed_bin_kernel (profiles/r04/seed_counters.txt): waves wait on something 56 % of their cycles but on an instruction
// issue only 14 %, on LDS 2.5 % (bank conflicts 2.7 %, 3 LDS atomics per 64 anchors): the kernel is neither issue- nor LDS-bound, it is bound by what a
// wave waits for between its four barriers per 512-anchor tile - the 8-byte gathers of the index records (L2 hit rate 35 %) with two of them in flight per
// lane, and the other three waves of its workgroup.  So the scatter pass is re-cut along the wave instead of along the tile:
//   * pass A counts the hits of a bin PER WAVE (each wave owns a contiguous quarter of the read's anchors, as before);
//   * the bin starts come from the same sort + scan, and wave w starts a bin at the bin's start plus the counts of waves < w: the four waves fill
//     disjoint, consecutive parts of every bin, in generation order;
//   * pass B is then wave-private: a wave walks its quarter in tiles of 256 anchors (4 index reads per lane in flight, the next tile's reads requested
//     before the current tile is ranked), ranks the hits per bin with the same ballot match groups, keeps the bins' running output positions in the
//     registers of the lane that owns the slot, parks the tile's hits in its own LDS stage grouped by bin and writes them out with consecutive lanes on
//     consecutive addresses.  No __syncthreads between the scan and the end of the round; LDS ordering inside a wave is program order.
// LDS: 10 B per slot shared + 7 KB per wave (512 slots) = 33.5 KB + the read's minimizer table -> 3-4 workgroups per CU, each wave with up to 8 gathers
// in flight (the old kernel: 6 workgroups with 2).
#pragma once
#include "hao_query.cuh"

template<int CAPLOG> struct hao_seed2_lds {      // byte layout of the dynamic LDS (host and device agree through this struct)
	static constexpr uint32_t CAP = 1u << CAPLOG, TILE = 256;
	static constexpr uint32_t SHARED = CAP * 10;                       // hk u32[CAP], bl u32[CAP], rk u16[CAP]
	static constexpr uint32_t PHASE_A = CAP * 32;                      // cnt u32[4][CAP], sk u64[CAP], tot u32[CAP], cwd u32[CAP]
	static constexpr uint32_t PER_WAVE = CAP * 6 + TILE * 16;          // tc u16[CAP], dl u32[CAP], stage 12 B x TILE, sslot u16[TILE], sq u16[TILE]
	static constexpr uint32_t UNION = PHASE_A > 4 * PER_WAVE ? PHASE_A : 4 * PER_WAVE;
	static constexpr uint32_t FIXED = SHARED + UNION;                  // + 12 B per staged query minimizer + 16
};

template<int CAPLOG, int TIER, bool PF>
__global__ __launch_bounds__(256) void seed_bin2_kernel(hao_seed_args S, const uint32_t *in_list, const unsigned long long *in_cnt, uint32_t *ovf_list, unsigned long long *ovf_cnt)
{
	constexpr bool FIRST = TIER == 0, GIVEUP = TIER < 2;
	using L = hao_seed2_lds<CAPLOG>;
	constexpr uint32_t CAP = L::CAP, MAXD = CAP - 288, TILE = L::TILE;      // at most MAXD + 256 bins are ever inserted (one per thread after the table fills), so probing terminates
	constexpr int UA = 4, NU = TILE / 64;
	constexpr uint32_t SPW = CAP / 64;           // slots per lane in the per-tile scan; the lane keeps their next output positions
	extern __shared__ uint32_t bs2_smem[];
	const int wv = threadIdx.x >> 6, lane = hao_lane(); const uint32_t tid = threadIdx.x;
	uint32_t *hk = bs2_smem;                     // [CAP]    bin key (tid << 1 | rev) per slot
	uint32_t *bl = hk + CAP;                     // [CAP]    read length of the slot's target (opposite-strand offsets)
	uint16_t *rk = (uint16_t*)(bl + CAP);        // [CAP]    rank of the slot's bin among the bins of the round
	char *uni = (char*)(rk + CAP);
	uint32_t *cnt = (uint32_t*)uni;              // [4][CAP] pass A: hits of the bin in each wave's quarter of the read   } until the cursors are in registers
	uint64_t *sk = (uint64_t*)(uni + 16 * CAP);  // [CAP]    (bin key << 32 | slot), sorted                               }
	uint32_t *tot = (uint32_t*)(uni + 24 * CAP); // [CAP]    per-rank totals                                              }
	uint32_t *cwd = (uint32_t*)(uni + 28 * CAP); // [CAP]    first output position of the bin                             }
	char *pw = uni + wv * L::PER_WAVE;           // pass B, this wave's own:
	uint16_t *tc = (uint16_t*)pw;                // [CAP]    hits of the bin in the tile so far, then the bin's offset in the staged tile; zero between tiles
	uint32_t *dl = (uint32_t*)(pw + 2 * CAP);    // [CAP]    output position of the bin's hits of this tile minus their offset in the staged tile
	hao_stage_t *stage = (hao_stage_t*)(pw + 6 * CAP);                  // [TILE] the tile's hits grouped by bin
	uint16_t *sslot = (uint16_t*)(pw + 6 * CAP + sizeof(hao_stage_t) * TILE);      // [TILE] slot of the staged hit
	uint16_t *sq = sslot + TILE;                 // [TILE]   query minimizer of the staged hit
	uint64_t *l_ss = (uint64_t*)(uni + L::UNION);   // [qcap]   list start of minimizer q in the position index | strand of the minimizer << 63
	uint32_t *l_ao = (uint32_t*)(l_ss + S.qcap); // [qcap+1] first anchor of minimizer q, relative to the read
	__shared__ uint32_t s_nd, s_ovf, s_c; __shared__ uint64_t s_ws[4], s_all;
	uint64_t *g_tmp = S.g_tmp;
	if (!FIRST && blockIdx.x >= *in_cnt) return;
	const uint64_t r = FIRST ? blockIdx.x : in_list[blockIdx.x], s = S.seg[r], e = S.seg[r + 1]; const uint32_t n = (uint32_t)(e - s);
	if (FIRST && r == 0 && tid == 0) S.g_cnt[S.n_sel] = 0;
	if (n == 0) { if (tid == 0) S.g_cnt[r] = 0; return; }
	const uint64_t m0 = S.mz_off[S.rid_lo + r], li0 = m0 - S.mz0; const uint32_t nq = (uint32_t)(S.mz_off[S.rid_lo + r + 1] - m0);
	const bool qlds = nq <= S.qcap;                // very long reads keep the per-minimizer table in global memory (uniform branches, no flat accesses)
	const uint64_t *g_ao = S.a_off + li0, *g_ss = S.s_start + li0, *g_info = S.mz_info + m0;
	if (qlds) {
		for (uint32_t q = tid; q < nq; q += 256) { l_ss[q] = g_ss[q] | (uint64_t)hao_info_rev(g_info[q]) << 63; l_ao[q] = (uint32_t)(g_ao[q] - s); }
		if (tid == 0) l_ao[nq] = n;
	}
	__syncthreads();
	unsigned long long tk0 = S.dbg ? wall_clock64() : 0, tk1 = 0, tk2 = 0;
#define HAO_AO(q) (qlds ? l_ao[q] : ((q) >= nq ? n : (uint32_t)(g_ao[q] - s)))
#define HAO_SS(q) (qlds ? l_ss[q] : (g_ss[q] | (uint64_t)hao_info_rev(g_info[q]) << 63))
	// wave wv owns anchors [c0, c1) of the read in BOTH passes (a multiple of TILE = 64 * UA anchors, so a tile never straddles two waves)
	const uint32_t chunk = ((n + 3) / 4 + 64 * UA - 1) / (64 * UA) * (64 * UA), c0 = min(n, wv * chunk), c1 = min(n, c0 + chunk);
	uint32_t q_c0 = 0;       // minimizer holding anchor c0: last q with AO(q) <= c0 (binary search, uniform in the wave)
	if (c0 < c1) { uint32_t lo_ = 0, hi_ = nq; while (hi_ - lo_ > 1) { const uint32_t md = (lo_ + hi_) >> 1; if (HAO_AO(md) <= c0) lo_ = md; else hi_ = md; } q_c0 = lo_; }
	const uint32_t k_end = 2u << S.tb;
	uint32_t lo = 0, placed = 0, ngr = 0, last_tid = 0xffffffffu;
	while (lo < k_end) {
		uint32_t hi = k_end;
		for (;;) {      // count the bins of [lo, hi), per wave; shrink the range until they fit the table
			for (uint32_t i = tid; i < CAP; i += 256) { hk[i] = HAO_BIN_EMPTY; cnt[i] = 0; cnt[CAP + i] = 0; cnt[2 * CAP + i] = 0; cnt[3 * CAP + i] = 0; }
			if (tid == 0) { s_nd = 0; s_ovf = 0; s_c = 0; }
			__syncthreads();
			uint32_t qc = q_c0; uint32_t *cw = cnt + wv * CAP;
			for (uint32_t t0 = c0; t0 < c1; t0 += 64 * UA) {      // UA independent index reads in flight per lane
				uint64_t yv[UA]; uint32_t zr[UA];
#pragma unroll
				for (int u = 0; u < UA; ++u) {
					const uint32_t x = t0 + u * 64 + lane; const bool act = x < c1; uint32_t q = qc;
					if (act) { while (HAO_AO(q + 1) <= x) ++q; }
					qc = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
					const uint64_t sv = HAO_SS(q);
					yv[u] = act ? S.sinfo[(sv & ~(1ULL << 63)) + (x - HAO_AO(q))] : 0; zr[u] = (uint32_t)(sv >> 63);
				}
				if (HAO_OVF()) break;
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
				const uint32_t slot = (uint32_t)sk[d]; rk[slot] = (uint16_t)d; bl[slot] = S.len[(uint32_t)(sk[d] >> 33)];
				tot[d] = cnt[slot] + cnt[CAP + slot] + cnt[2 * CAP + slot] + cnt[3 * CAP + slot];
			}
			__syncthreads();
			// exclusive scan over the sorted bins of (hits, group starts), packed as starts << 32 | hits; thread t owns bins [t*per, (t+1)*per)
			const uint32_t per = P >= 256 ? P / 256 : 1, d0 = tid * per; uint64_t mine = 0;
			for (uint32_t d = d0; d < d0 + per && d < D; ++d) {
				const uint32_t t_k = (uint32_t)(sk[d] >> 33), t_p = d ? (uint32_t)(sk[d - 1] >> 33) : last_tid;
				mine += (uint64_t)(t_k != t_p) << 32 | tot[d];
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
				cwd[slot] = placed + (uint32_t)ex;
				ex += tot[d];
			}
			const uint32_t last_tid_next = (uint32_t)(sk[D - 1] >> 33);      // (the per-wave stages reuse sk / tot)
			__syncthreads();
			const uint64_t all = s_all;
			int nbits = 0; while ((1u << nbits) < D) ++nbits;
			// this wave's first output position of every bin: the bin's start + what the waves before it put there.  Lane l keeps slots [l * SPW, (l + 1) * SPW)
			uint32_t ob[SPW];
#pragma unroll
			for (uint32_t k = 0; k < SPW; ++k) {
				const uint32_t sl = lane * SPW + k; uint32_t v = cwd[sl];
				for (int w = 0; w < wv; ++w) v += cnt[w * CAP + sl];
				ob[k] = v;      // (an empty slot holds whatever the LDS held: never used, its tile counts stay zero)
			}
			__syncthreads();      // cnt / sk / tot / cwd are dead from here: the same memory holds the four waves' private stages
			for (uint32_t i = lane; i < CAP / 2; i += 64) ((uint32_t*)tc)[i] = 0;
			if (S.dbg) tk2 = wall_clock64();
			// ---- pass B: wave-private from here to the end of the round ----
			struct req_t { uint64_t yv[NU]; uint32_t tpe[NU], tne[NU], qv[NU], qp[NU], qn[NU]; };
			uint32_t qc = q_c0;
			auto request = [&](const uint32_t t0, req_t &R) {      // the tile's index records (+ the list neighbours the tile edges need, + the two words of the query minimizer)
#pragma unroll
				for (int u = 0; u < NU; ++u) {
					const uint32_t x = t0 + u * 64 + lane; const bool act = x < c1; uint32_t q = qc;
					if (act) { while (HAO_AO(q + 1) <= x) ++q; }
					qc = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
					R.qv[u] = q;
					const uint32_t a0 = HAO_AO(q), j = x - a0; const uint64_t ad = (HAO_SS(q) & ~(1ULL << 63)) + j;
					R.yv[u] = act ? S.sinfo[ad] : 0;
					// list neighbours normally sit in the adjacent lanes; only the lanes at a sub-tile edge fetch theirs (0xffffffff: no such entry)
					R.tpe[u] = (act && lane == 0 && j > 0) ? hao_info_rid(S.sinfo[ad - 1]) : 0xffffffffu;
					R.tne[u] = (act && (lane == 63 || x + 1 == n) && j + 1 < HAO_AO(q + 1) - a0) ? hao_info_rid(S.sinfo[ad + 1]) : 0xffffffffu;
					R.qp[u] = act ? S.q_pos[li0 + q] : 0; R.qn[u] = act ? S.q_cnt[li0 + q] : 0;
				}
			};
			req_t A;
			request(c0, A);
			for (uint32_t t0 = c0; t0 < c1; t0 += TILE) {
				req_t Bn;
				if (PF) request(t0 + TILE, Bn);      // (past the end of the quarter every lane is inactive: nothing is read)
				uint32_t ps[NU], po[NU];      // slot | rank inside the tile's bin << 16 (or ~0: no hit); k_mer_hit::offset
#pragma unroll
				for (int u = 0; u < NU; ++u) {
					const uint32_t x = t0 + u * 64 + lane, q = A.qv[u]; uint64_t y = A.yv[u];
					const uint64_t sv = HAO_SS(q), st = sv & ~(1ULL << 63); const uint32_t zrev = (uint32_t)(sv >> 63);
					const uint32_t tidk = hao_info_rid(y), rev = zrev ^ hao_info_rev(y), kk = tidk << 1 | rev;
					const bool inr = x < c1 && kk >= lo && kk < hi;
					// target of the previous / next entry of my list (0xffffffff: none): lane - 1 / lane + 1 hold them unless they belong to another
					// minimizer (then I am the first / last entry of my list) or I sit at a sub-tile edge (fetched with the records)
					uint32_t t_up = hao_wave_shr1(tidk, 0u), t_dn = hao_wave_shl1(tidk, 0u);             // cross-lane moves (DPP): all lanes, before any branch
					const uint32_t q_up = hao_wave_shr1(q, 0xffffffffu), q_dn = hao_wave_shl1(q, 0xffffffffu);
					if (lane == 0) t_up = A.tpe[u];
					else if (q_up != q) t_up = 0xffffffffu;
					if (lane == 63 || x + 1 >= n) t_dn = A.tne[u];
					else if (q_dn != q) t_dn = 0xffffffffu;
					if (inr && rev) {
						// opposite-strand hits of one k-mer in one target must come out by DEscending target position (ascending other_off,
						// anchor.cpp:1023): inside the (rare) run of list entries with the same target, the anchor at rev position k takes the
						// record of rev entry R-1-k
						const uint32_t a0 = HAO_AO(q), nl = HAO_AO(q + 1) - a0, j = x - a0;
						const bool pv = t_up == tidk, nx = t_dn == tidk;
						if (pv || nx) {
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
					const uint32_t d = inr ? rk[slot] : 0;
					const unsigned long long m = hao_match_bits(d, inr, nbits);
					const uint32_t before = __popcll(m & ((1ULL << lane) - 1)), base = inr ? tc[slot] : 0;
					ps[u] = inr ? (slot | (base + before) << 16) : 0xffffffffu;
					// k_mer_hit::offset (anchor.cpp:1021-1023,1059-1064): target coordinate in the strand of the hit
					po[u] = inr ? (rev ? bl[slot] - 1 - (hao_info_pos(y) + 1 - hao_info_span(y)) : hao_info_pos(y)) : 0;
					if (inr && before == 0) tc[slot] = (uint16_t)(base + __popcll(m));
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				// the bins' offsets in the staged tile (slot order: any order keeps a bin's hits together) and where the tile's part of each bin goes
				uint32_t c_[SPW], mine_t = 0;
#pragma unroll
				for (uint32_t k = 0; k < SPW; ++k) { c_[k] = tc[lane * SPW + k]; mine_t += c_[k]; }
				const uint32_t inc_t = hao_wave_incl_scan_u32(mine_t), tile_n = (uint32_t)__builtin_amdgcn_readlane((int)inc_t, 63);
				uint32_t ex_t = inc_t - mine_t;
#pragma unroll
				for (uint32_t k = 0; k < SPW; ++k)
					if (c_[k]) { const uint32_t sl = lane * SPW + k; dl[sl] = ob[k] - ex_t; tc[sl] = (uint16_t)ex_t; ob[k] += c_[k]; ex_t += c_[k]; }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
				for (int u = 0; u < NU; ++u)
					if (ps[u] != 0xffffffffu) {
						const uint32_t slot = ps[u] & 0xffffu, at = tc[slot] + (ps[u] >> 16);
						hao_stage_t z; z.offset = po[u]; z.self_offset = A.qp[u]; z.cnt = A.qn[u];
						stage[at] = z; sslot[at] = (uint16_t)slot; sq[at] = (uint16_t)(A.qv[u] < 65535u ? A.qv[u] : 65535u);
					}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				for (uint32_t at = lane; at < tile_n; at += 64) {
					const uint32_t slot = sslot[at], kk = hk[slot]; const hao_stage_t z = stage[at];
					hao_hit_t h; h.w0 = kk >> 1 | kk << 31; h.offset = z.offset; h.self_offset = z.self_offset; h.cnt = z.cnt;
					S.hits[s + (uint32_t)(dl[slot] + at)] = h;      // (32-bit sum: dl may have wrapped below zero)
					if (S.hq) S.hq[s + (uint32_t)(dl[slot] + at)] = sq[at];
				}
#pragma unroll
				for (uint32_t k = 0; k < SPW; ++k) tc[lane * SPW + k] = 0;
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				if (PF) A = Bn; else request(t0 + TILE, A);
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
}
