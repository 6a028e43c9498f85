// K3, unit kernel: (count,hash) window minimizers, one WAVE per unit of 1024 window ordinals, no workgroup barrier.
//
// Same closed form as hao_sketch.cuh (ordinal j is emitted iff its key is the minimum of at least one valid window of W consecutive
// ordinals that contains it - all ties -, plus the first-window quirk and the short-read flush of sketch.cpp:523-573), evaluated in two steps
// so that the expensive part runs on 32-bit values (tests/sk3_model.py restates every formula below lane by lane and is checked against the
// oracle on the CPU):
//
//   1. candidates.  Every window entry gets a 32-bit PROXY p of its key, monotone (non-strict) in the key order.  Sliding minimum of p over the
//      W entries ending at each entry, sliding maximum of that over the W windows starting at each entry: p == max-of-min marks a SUPERSET of
//      the answer (a true window minimum always qualifies; a false candidate needs two different keys with one proxy).  Blocked layout, 16
//      consecutive entries per lane: prefix / suffix minima in registers, whole predecessor lanes by three wave_shr:1 DPP moves of the lane
//      total, the partial far lane by ONE ds_bpermute per entry - 32-bit v_min_u32 / v_max_u32 where the previous kernel spent a 64-bit
//      compare and two selects (22 issue cycles instead of 4, tools/ubench_valu.hip) and two 32-bit shuffles per 64-bit key.
//   2. verification on the sparse candidate list (about 40 of 1024 entries) with the full keys: a window's minimum over all keys equals its
//      minimum over the candidates in it, so a candidate is emitted iff some valid window containing it holds no candidate with a strictly
//      smaller key - two short neighbour scans in an LDS list.  Exact for any proxy, so the rare case is not a separate code path.
//
// Only the marked entries ever need a position: pos / span come from a select (ordinal -> base) in the per-word run counts of the unit's
// decode, not from a per-run table; keys are hashed once for the proxy (16 per lane, straight-line) and once more for the ~40 candidates.
// The wave decodes its own runs (16 bases per lane and step; dense 2-bit run codes by a funnel shift per base; OR-ed into two bit planes in
// LDS), so the four waves of a workgroup share nothing and the grid is one wave per 924 decided ordinals.
//
// Reads whose runs are unusually long (a 1024-base tile with < 544 runs: the unit would need more than 3 decode steps; or 17 consecutive
// words with <= 67 run ends: a k-mer span could reach 256, sketch.cpp:505) are flagged by hpc_index_kernel and take the exact scalar kernel.
#pragma once
#include "hao_sketch.cuh"

#define SK3_E 16
#define SK3_NENT 1024
#define SK3_SMAX 3
#define SK3_RING 192
#define SK3_SLOT 64          // pool entries reserved per unit (one round of candidates): no allocation, no atomic; bigger candidate lists go behind the slots
template<int K, int W> struct hao_sk3 {
	static constexpr int MW = SK3_NENT - 2 * (W - 1);                    // ordinals decided per unit
	static constexpr int NB = SK3_NENT + K - 1;                           // runs (plane bits) a unit needs
	static constexpr int NCELL = ((NB + 16 + 31) >> 5) + 2;
};

struct hao_sk3_lds {                                                     // per wave
	unsigned long long cell[48];                                         // bit planes: {plane 0 word, plane 1 word} per 32 runs
	uint32_t tab_o[SK3_SMAX * 64], tab_eb[SK3_SMAX * 64], tab_g0[SK3_SMAX];   // per decoded word: ordinal of the first run ending in it, its run-end bits; first base of the step
	unsigned long long kx[SK3_RING]; uint32_t kc[SK3_RING]; uint16_t kq[SK3_RING];   // candidate ring: key, count | rev << 31, entry
};

// (sk3_run_ends, sk3_load16: hao_sketch.cuh, next to hpc_index_kernel)
__device__ __forceinline__ uint32_t sk3_even_bits(uint32_t x)            // bits 0,2,4,..,30 -> bits 0..15
{
	x = (x | x >> 1) & 0x33333333u; x = (x | x >> 2) & 0x0f0f0f0fu; x = (x | x >> 4) & 0x00ff00ffu; x = (x | x >> 8) & 0xffffu;
	return x;
}
// index (0..15, in base order) of the r-th (0-based) run end among the run-end bits eb
__device__ __forceinline__ uint32_t sk3_select_end(uint32_t eb, uint32_t r)
{
	uint32_t j = 0, c;
	c = __popc(eb >> 16); if (r >= c) { r -= c; j += 8; eb <<= 16; }
	c = __popc(eb >> 24); if (r >= c) { r -= c; j += 4; eb <<= 8; }
	c = __popc(eb >> 28); if (r >= c) { r -= c; j += 2; eb <<= 4; }
	c = __popc(eb >> 30); if (r >= c) { j += 1; }
	return j;
}

template<int K> __device__ __forceinline__ uint64_t sk3_mask() { return K >= 64 ? ~0ULL : (1ULL << K) - 1; }

// key of the k-mer whose runs are bits [q, q+K) of the planes: hash of the canonical strand (yak_hash_long, htab.h:161-166); rev as in sketch.cpp:503
template<int K> __device__ __forceinline__ uint64_t sk3_key_dyn(const unsigned long long *cell, int q, uint32_t &rev)
{
	const int wi = q >> 5; const uint32_t sh = (uint32_t)q & 31u;
	const unsigned long long c0 = cell[wi], c1 = cell[wi + 1], c2 = cell[wi + 2];
	const uint64_t mask = sk3_mask<K>();
	const uint64_t W0 = ((uint64_t)__builtin_amdgcn_alignbit((uint32_t)c2, (uint32_t)c1, sh) << 32 | __builtin_amdgcn_alignbit((uint32_t)c1, (uint32_t)c0, sh)) & mask;
	const uint64_t W1 = ((uint64_t)__builtin_amdgcn_alignbit((uint32_t)(c2 >> 32), (uint32_t)(c1 >> 32), sh) << 32 | __builtin_amdgcn_alignbit((uint32_t)(c1 >> 32), (uint32_t)(c0 >> 32), sh)) & mask;
	const uint64_t f1 = __brevll(W1) >> (64 - K), r1 = ~W1 & mask;
	rev = f1 < r1 ? 0u : 1u;
	return f1 < r1 ? hao_hash_planes<K>(__brevll(W0) >> (64 - K), f1) : hao_hash_planes<K>(~W0 & mask, r1);
}

template<bool HAS_FT> __device__ __forceinline__ uint32_t sk3_proxy(uint64_t x, uint32_t c)
{
	if (HAS_FT) return c > 0 ? (0x80000000u | (c < 0x7fffu ? c : 0x7fffu) << 16 | (uint32_t)(x >> 48)) : (uint32_t)(x >> 33);
	return (uint32_t)(x >> 32);
}
__device__ __forceinline__ bool sk3_lt(uint64_t xa, uint32_t ca, uint64_t xb, uint32_t cb) { return ca < cb || (ca == cb && xa < xb); }

template<bool HAS_FT, int K, int W>
__global__ __launch_bounds__(256) void sketch_unit_kernel(hao_sk_args a)
{
	static_assert(SK3_E == 16 && (W - 1) / 16 == 3 && (W - 16) / 16 == 2 && K + 15 <= 96 && K > 32 && K < 64, "lane-window rules below are written for 48 < W <= 64, 32 < K < 64");
	constexpr int MW = hao_sk3<K, W>::MW, NB = hao_sk3<K, W>::NB, NCELL = hao_sk3<K, W>::NCELL;
	static_assert(NCELL <= 48, "plane cells");
	__shared__ hao_sk3_lds lds_all[4];
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	hao_sk3_lds &S = lds_all[wv];
	const uint64_t un = (uint64_t)blockIdx.x * 4 + (uint64_t)wv;          // this wave's unit
	if (un >= a.n_units) return;
	const uint64_t r = a.unit_rid[un];                                   // (hao_unit_rid_kernel: no search in chunk_off)
	if (a.scalar_flag[r]) return;                                        // record written by the scalar kernel
	const int ui = (int)(un - a.chunk_off[r]);
	const uint64_t rid = a.rid_lo + r; const uint8_t *rd = a.packed + a.pk_off[rid]; const uint32_t L = a.len[rid];
	const int T = (int)a.n_runs[r];
	const int jw0 = K + ui * MW, jw1 = min(jw0 + MW, T + 1);            // marks are decided for ordinals [jw0, jw1)
	if (jw0 > T) { if (lane == 0) { a.chunk_base[un] = 0; a.chunk_cnt[un] = 0; } return; }     // read with < K runs (its single unit)
	const int kw0 = max(K, jw0 - (W - 1));                               // entry q <-> ordinal kw0 + q
	const int kk1 = min(T, jw1 - 1 + W - 1);                             // last ordinal whose key matters
	const int tm0 = max(jw0, W + K - 1);                                 // first valid window
	const int rbase = kw0 - K;                                           // plane bit b <-> run rbase + 1 + b
	const uint64_t mask = sk3_mask<K>();
	// ---- decode: run ends of up to SK3_SMAX steps of 1024 bases -> select table + dense bit planes ----
	if (lane < NCELL) S.cell[lane] = 0;
	{
		const uint32_t *tord = a.tile_ord + a.tile_off[r]; const uint32_t ntile = (L + HAO_SK_TILE - 1) / HAO_SK_TILE;
		const uint32_t first = (uint32_t)rbase + 1u;
		// tile that holds run `first` = the last tile with tord[ti] < first (tord[ti] = runs that end before tile ti).  Runs are spread evenly enough
		// that a proportional guess is at most one tile off: six neighbouring entries are requested at once and the tile is picked among them
		// (one round of scalar-load latency instead of a binary search); anything else falls back to the search.
		uint32_t tlo, tbv[SK3_SMAX];
		{
			uint32_t gt = (uint32_t)(((uint64_t)first * ntile) / (uint32_t)(T + 1)); if (gt >= ntile) gt = ntile - 1;
			const uint32_t b0 = gt > 0 ? gt - 1 : 0;                        // entries b0 .. b0+5 (tord has ntile + 1 entries; clamp to ntile)
			uint32_t v[6];
#pragma unroll
			for (int j = 0; j < 6; ++j) v[j] = tord[min(b0 + (uint32_t)j, ntile)];
			int pick = -1;
#pragma unroll
			for (int j = 0; j < 3; ++j) if (pick < 0 && b0 + (uint32_t)j < ntile && v[j] < first && (b0 + (uint32_t)j + 1 >= ntile || v[j + 1] >= first)) pick = j;
			if (pick >= 0) {
				tlo = b0 + (uint32_t)pick;
#pragma unroll
				for (int s2 = 0; s2 < SK3_SMAX; ++s2) tbv[s2] = pick == 0 ? v[s2] : (pick == 1 ? v[s2 + 1] : v[s2 + 2]);
			} else {
				uint32_t lo2 = 0, hi2 = ntile;
				while (hi2 - lo2 > 1) { const uint32_t m = (lo2 + hi2) >> 1; if (tord[m] < first) lo2 = m; else hi2 = m; }
				tlo = lo2;
#pragma unroll
				for (int s2 = 0; s2 < SK3_SMAX; ++s2) tbv[s2] = tord[min(tlo + (uint32_t)s2, ntile)];
			}
		}
		bool live[SK3_SMAX]; uint32_t Wd[SK3_SMAX], nxt[SK3_SMAX];
#pragma unroll
		for (int s = 0; s < SK3_SMAX; ++s) {                               // the (up to three) 16-base words of this lane: all loads in flight before the first is used
			const uint32_t ti = tlo + (uint32_t)s, g0 = ti * HAO_SK_TILE + (uint32_t)lane * 16u;
			live[s] = ti < ntile && tbv[s] < (uint32_t)kk1; Wd[s] = 0; nxt[s] = 0;
			if (live[s] && g0 < L) sk3_load16(rd, g0, Wd[s], nxt[s]);
		}
#pragma unroll
		for (int s = 0; s < SK3_SMAX; ++s) {
			const uint32_t ti = tlo + (uint32_t)s, tb = tbv[s];
			const uint32_t g0 = ti * HAO_SK_TILE + (uint32_t)lane * 16u;
			const uint32_t eb = (live[s] && g0 < L) ? sk3_run_ends(Wd[s], nxt[s], L - g0, a.hpc) : 0u;
			const uint32_t n = __popc(eb), incl = hao_wave_incl_scan_u32(n);
			const uint32_t o = live[s] ? tb + incl - n + 1u : 0xffffffffu;   // ordinal of the first run that ends in this word (dead steps sort last)
			S.tab_o[s * 64 + lane] = o; S.tab_eb[s * 64 + lane] = eb;
			if (lane == 0) S.tab_g0[s] = ti * HAO_SK_TILE;
			if (!live[s]) continue;
			// dense 2-bit codes of the word's runs, first run lowest: one funnel shift per base pulls the code in from the top iff the base ends a run
			uint32_t acc = 0; const uint32_t E2 = eb << 1;
#pragma unroll
			for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_alignbit(Wd[s] >> (30 - 2 * j), acc, (E2 >> (30 - 2 * j)) & 3u);
			const uint32_t dense = n ? acc >> (32 - 2 * n) : 0u;
			uint32_t c0 = sk3_even_bits(dense & 0x55555555u), c1 = sk3_even_bits((dense >> 1) & 0x55555555u);
			int p = (int)o - (int)first;                                  // plane bit of the word's first run
			if (p < 0) { const int d = -p; c0 = d < 16 ? c0 >> d : 0u; c1 = d < 16 ? c1 >> d : 0u; p = 0; }
			if ((c0 | c1) && p < NB) {
				const int wi = p >> 5; const uint32_t sh = (uint32_t)p & 31u;
				const unsigned long long lo64 = (unsigned long long)(c1 << sh) << 32 | (unsigned long long)(c0 << sh);
				atomicOr(&S.cell[wi], lo64);
				if (sh > 16) { const unsigned long long hi64 = (unsigned long long)(c1 >> (32 - sh)) << 32 | (unsigned long long)(c0 >> (32 - sh)); if (hi64) atomicOr(&S.cell[wi + 1], hi64); }
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	// ---- proxies of my 16 entries: q = 16 lane + i, k-mer = plane bits [q, q + K) ----
	const int q0 = lane * SK3_E, t0 = kw0 + q0;
	const bool full = kw0 + SK3_NENT - 1 <= kk1;                          // every entry of the unit is a real ordinal
	const int nvalid = max(0, min(SK3_E, kk1 - t0 + 1));
	uint32_t p[SK3_E];
	{
		const unsigned long long c0 = S.cell[lane >> 1], c1 = S.cell[(lane >> 1) + 1], c2 = S.cell[(lane >> 1) + 2];
		const uint32_t sh = (uint32_t)(lane & 1) * 16u;
		// B = bits [q0, q0 + 96) of a plane; RB = B reversed (the forward k-mer reads from its first run at the top); NB_ = ~B (the reverse complement)
		const uint32_t B0l = __builtin_amdgcn_alignbit((uint32_t)c1, (uint32_t)c0, sh), B0m = __builtin_amdgcn_alignbit((uint32_t)c2, (uint32_t)c1, sh), B0h = (uint32_t)c2 >> sh;
		const uint32_t B1l = __builtin_amdgcn_alignbit((uint32_t)(c1 >> 32), (uint32_t)(c0 >> 32), sh), B1m = __builtin_amdgcn_alignbit((uint32_t)(c2 >> 32), (uint32_t)(c1 >> 32), sh), B1h = (uint32_t)(c2 >> 32) >> sh;
		const uint32_t R0l = __brev(B0h), R0m = __brev(B0m), R0h = __brev(B0l), R1l = __brev(B1h), R1m = __brev(B1m), R1h = __brev(B1l);
		const uint32_t N0l = ~B0l, N0m = ~B0m, N0h = ~B0h, N1l = ~B1l, N1m = ~B1m, N1h = ~B1h;
		constexpr uint32_t MH = (uint32_t)(((1ULL << K) - 1) >> 32);
#pragma unroll
		for (int i = 0; i < SK3_E; ++i) {
			// forward strand: RB >> (96 - K - i); reverse complement: NB_ >> i  (both masked to K bits)
			constexpr int dummy = 0; (void)dummy;
			const int sf = 96 - K - i;
			uint32_t f0l, f0h, f1l, f1h;
			if (sf >= 32) { f0l = __builtin_amdgcn_alignbit(R0h, R0m, sf - 32); f0h = (R0h >> (sf - 32)) & MH; f1l = __builtin_amdgcn_alignbit(R1h, R1m, sf - 32); f1h = (R1h >> (sf - 32)) & MH; }
			else { f0l = __builtin_amdgcn_alignbit(R0m, R0l, sf); f0h = __builtin_amdgcn_alignbit(R0h, R0m, sf) & MH; f1l = __builtin_amdgcn_alignbit(R1m, R1l, sf); f1h = __builtin_amdgcn_alignbit(R1h, R1m, sf) & MH; }
			const uint32_t r0l = __builtin_amdgcn_alignbit(N0m, N0l, i), r0h = __builtin_amdgcn_alignbit(N0h, N0m, i) & MH;
			const uint32_t r1l = __builtin_amdgcn_alignbit(N1m, N1l, i), r1h = __builtin_amdgcn_alignbit(N1h, N1m, i) & MH;
			// canonical strand: forward iff f1 < r1 (sketch.cpp:503) = the borrow of f1 - r1 (two 32-bit subtracts; a 64-bit compare costs 14 issue cycles)
			uint32_t x0l, x0h, x1l, x1h, tmp;
			asm("v_sub_co_u32 %4, vcc, %5, %9\n\tv_subb_co_u32 %4, vcc, %6, %10, vcc\n\tv_cndmask_b32 %0, %11, %7, vcc\n\tv_cndmask_b32 %1, %12, %8, vcc\n\tv_cndmask_b32 %2, %9, %5, vcc\n\tv_cndmask_b32 %3, %10, %6, vcc"
				: "=&v"(x0l), "=&v"(x0h), "=&v"(x1l), "=&v"(x1h), "=&v"(tmp)
				: "v"(f1l), "v"(f1h), "v"(f0l), "v"(f0h), "v"(r1l), "v"(r1h), "v"(r0l), "v"(r0h) : "vcc");
			const uint64_t x0 = (uint64_t)x0h << 32 | x0l, x1 = (uint64_t)x1h << 32 | x1l;
			const uint64_t y = hao_hash_planes<K>(x0, x1);
			if (HAS_FT) { const int32_t cnt = hao_ft_lookup(a.ft, y); p[i] = cnt < (1 << 28) ? sk3_proxy<true>(y, (uint32_t)cnt) : 0xffffffffu; }
			else p[i] = sk3_proxy<false>(y, 0);
		}
		if (!full) {
#pragma unroll
			for (int i = 0; i < SK3_E; ++i) if (i >= nvalid) p[i] = 0xffffffffu;
		}
	}
	const uint32_t up3 = (uint32_t)(lane - 3) << 2, up4 = (uint32_t)(lane - 4) << 2, dn3 = (uint32_t)(lane + 3) << 2, dn4 = (uint32_t)(lane + 4) << 2;
	const uint32_t or3 = lane >= 3 ? 0u : 0xffffffffu, or4 = lane >= 4 ? 0u : 0xffffffffu, and3 = lane + 3 <= 63 ? 0xffffffffu : 0u, and4 = lane + 4 <= 63 ? 0xffffffffu : 0u;
	// ---- sliding minimum over the W entries ending at each entry ----
	uint32_t m[SK3_E];
	{
		uint32_t pre[SK3_E], suf[SK3_E];
		pre[0] = p[0];
#pragma unroll
		for (int i = 1; i < SK3_E; ++i) pre[i] = min(pre[i - 1], p[i]);
		suf[SK3_E - 1] = p[SK3_E - 1];
#pragma unroll
		for (int i = SK3_E - 2; i >= 0; --i) suf[i] = min(p[i], suf[i + 1]);
		const uint32_t t1 = hao_wave_shr1(pre[SK3_E - 1], 0xffffffffu), t2 = hao_wave_shr1(t1, 0xffffffffu), t3 = hao_wave_shr1(t2, 0xffffffffu);
		const uint32_t A2 = min(t1, t2), A3 = min(A2, t3);
#pragma unroll
		for (int i = 0; i < SK3_E; ++i) {
			const int rem = W - 1 - i, nfull = rem / SK3_E, part = rem % SK3_E;   // entries before my lane: nfull whole lanes + the last `part` of a farther one
			uint32_t v = min(pre[i], nfull == 3 ? A3 : A2);
			if (part > 0) {
				const uint32_t far = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nfull == 3 ? up4 : up3), (int)suf[SK3_E - part]) | (nfull == 3 ? or4 : or3);
				v = min(v, far);
			}
			m[i] = v;
		}
	}
	{	// windows exist for entries q >= W-1 that are real ordinals (t <= kk1); the rest counts as -inf in the maximum
		const int lo_i = W - 1 - q0;                                     // entries i < lo_i have q < W-1
		const uint32_t vm = (nvalid >= 16 ? 0xffffu : (1u << nvalid) - 1u) & (lo_i <= 0 ? 0xffffu : (lo_i >= 16 ? 0u : (0xffffu << lo_i) & 0xffffu));
		if (vm != 0xffffu) {
#pragma unroll
			for (int i = 0; i < SK3_E; ++i) if (!((vm >> i) & 1u)) m[i] = 0;
		}
	}
	// ---- sliding maximum over the W windows starting at each entry; candidates p == max ----
	uint32_t cm = 0;
	{
		uint32_t pre[SK3_E], suf[SK3_E];
		pre[0] = m[0];
#pragma unroll
		for (int i = 1; i < SK3_E; ++i) pre[i] = max(pre[i - 1], m[i]);
		suf[SK3_E - 1] = m[SK3_E - 1];
#pragma unroll
		for (int i = SK3_E - 2; i >= 0; --i) suf[i] = max(m[i], suf[i + 1]);
		const uint32_t n1 = hao_wave_shl1(pre[SK3_E - 1], 0u), n2 = hao_wave_shl1(n1, 0u), n3 = hao_wave_shl1(n2, 0u);
		const uint32_t B2 = max(n1, n2), B3 = max(B2, n3);
#pragma unroll
		for (int i = 0; i < SK3_E; ++i) {
			const int rem = W - (SK3_E - i), nfull = rem / SK3_E, part = rem % SK3_E;   // windows after my lane's suffix
			uint32_t v = max(suf[i], nfull == 3 ? B3 : B2);
			if (part > 0) {
				const uint32_t far = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nfull == 3 ? dn4 : dn3), (int)pre[part - 1]) & (nfull == 3 ? and4 : and3);
				v = max(v, far);
			}
			cm |= (p[i] == v ? 1u : 0u) << i;
		}
	}
	// ---- first-window quirk (sketch.cpp:523-534,543-547) / reads with fewer than W+K-1 runs (:571-573): unit 0 only ----
	int patch_on = 0, patch_prev = -1; uint64_t pkx = 0; uint32_t pkc = 0;
	if (ui == 0) {
		{	// keys of ordinals K .. K+63 (entries 0..63), one per lane, into the ring
			const int t = K + lane; uint32_t rev; uint64_t x = UINT64_MAX; uint32_t c = HAO_CNT_DUMMY;
			if (t <= kk1) {
				x = sk3_key_dyn<K>(S.cell, lane, rev); c = 0;
				if (HAS_FT) { const int32_t cnt = hao_ft_lookup(a.ft, x); if (cnt < (1 << 28)) c = (uint32_t)cnt; else { x = UINT64_MAX; c = HAO_CNT_DUMMY; } }
			}
			S.kx[lane] = x; S.kc[lane] = c;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		{	// newest minimum of the keys of ordinals [ta, tb] (= lanes ta-K .. tb-K): the lane whose proxy is the range's minimum - one DPP reduction - when
			// exactly one lane has it (then its key is the unique minimum); proxy ties go through the sequential scan
			const bool early = T >= W + K - 1;
			const int ta = early ? K : max(K, T - W + 1), tb = early ? W + K - 2 : T;
			const uint64_t mx = S.kx[lane]; const uint32_t mc = S.kc[lane];
			const bool inr = K + lane >= ta && K + lane <= tb;
			const uint32_t mp = inr && mx != UINT64_MAX ? sk3_proxy<HAS_FT>(mx, mc) : 0xffffffffu;
			const uint32_t pmin = hao_wave_min_u32(mp);
			const unsigned long long at = __ballot(mp == pmin && inr);
			int prev = -1; uint64_t px = UINT64_MAX; uint32_t pc = HAO_CNT_DUMMY;
			if (pmin != 0xffffffffu && __popcll(at) == 1) {
				const int src = __ffsll((long long)at) - 1; prev = K + src; px = S.kx[src]; pc = S.kc[src];
			} else if (pmin != 0xffffffffu || __popcll(at) > 0) {             // proxy tie (or a real key with the proxy of a dummy): exact scan, every lane the same
				for (int t = ta; t <= tb; ++t) { const uint64_t ox = S.kx[t - K]; const uint32_t oc = S.kc[t - K]; if (!sk3_lt(px, pc, ox, oc)) { px = ox; pc = oc; prev = t; } }
				if (px == UINT64_MAX) prev = -1;
			}
			if (early) { if (prev >= 0 && !sk3_lt(px, pc, S.kx[W - 1], S.kc[W - 1])) patch_on = 1; }      // key(t0) <= key(prev), t0 = W+K-1 = entry W-1
			else patch_on = 2;
			patch_prev = prev; pkx = px; pkc = pc;
		}
		patch_on = __builtin_amdgcn_readfirstlane(patch_on); patch_prev = __builtin_amdgcn_readfirstlane(patch_prev);
		pkx = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pkx) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pkx >> 32)) << 32;
		pkc = (uint32_t)__builtin_amdgcn_readfirstlane((int)pkc);
		if (patch_on == 1) {       // key(t0) <= key(prev): prev is never emitted, its older ties in [K, W+K-2] are: both must be on the candidate list
			if (q0 < W - 1) {
#pragma unroll 1
				for (int i = 0; i < SK3_E; ++i) { const int q = q0 + i; if (q < W - 1 && (K + q == patch_prev || (S.kx[q] == pkx && S.kc[q] == pkc))) cm |= 1u << i; }
			}
		} else if (patch_on == 2) {
			cm = (patch_prev >= 0 && (patch_prev - K) >> 4 == lane) ? 1u << ((patch_prev - K) & 15) : 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	// ---- candidate list, in entry order ----
	uint32_t ncand; const uint32_t my_n = __popc(cm), my_off = hao_wave_excl_scan(my_n, &ncand);
	unsigned long long base = 0;
	if (lane == 0) {
		// room for every candidate (the marks are written densely from the start of it): the unit's own slot, or - a degenerate read with more than
		// SK3_SLOT candidates in 1024 ordinals - an allocation behind the slots.  One atomic per unit on ONE address was what bounded the round-2
		// kernel (12 ns each: 3.4 M workgroups = 42 ms).
		base = ncand <= SK3_SLOT ? un * SK3_SLOT : a.pool_static + atomicAdd(a.pool_cursor, (unsigned long long)ncand);
		if (base + ncand > a.pool_cap) { *a.err = 1; base = ~0ULL; }
		a.chunk_base[un] = base == ~0ULL ? 0 : base;
	}
	base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32;
	if (base == ~0ULL || ncand == 0) { if (lane == 0) a.chunk_cnt[un] = 0; return; }
	const int nround = (int)((ncand + 63u) >> 6);
	uint32_t nout = 0;
	// fill ring slot (rd % 3) with the keys of candidates [64 rd, 64 rd + 64)
	auto fill = [&](int rdn) {
		const uint32_t lo_c = (uint32_t)rdn * 64u; const int slot = (rdn % 3) * 64;
		uint32_t rk = my_off, mm = cm;
		while (mm) { const int i = __builtin_ctz(mm); mm &= mm - 1; const uint32_t at = rk - lo_c; if (at < 64u) S.kq[slot + at] = (uint16_t)(q0 + i); ++rk; }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		if (lo_c + (uint32_t)lane < ncand) {
			const int q = S.kq[slot + lane]; uint32_t rev = 0; uint64_t x = UINT64_MAX; uint32_t c = HAO_CNT_DUMMY;
			if (kw0 + q <= kk1) {
				x = sk3_key_dyn<K>(S.cell, q, rev); c = 0;
				if (HAS_FT) { const int32_t cnt = hao_ft_lookup(a.ft, x); if (cnt < (1 << 28)) c = (uint32_t)cnt; else { x = UINT64_MAX; c = HAO_CNT_DUMMY; } }
			}
			S.kx[slot + lane] = x; S.kc[slot + lane] = c | rev << 31;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	};
	fill(0);
#pragma unroll 1
	for (int rdn = 0; rdn < nround; ++rdn) {
		if (rdn + 1 < nround) fill(rdn + 1);
		const int ci = rdn * 64 + lane; bool ok = false; int q = 0; uint64_t xj = 0; uint32_t cj = 0;
		if (ci < (int)ncand) {
			const int sl = (rdn % 3) * 64 + lane;
			q = S.kq[sl]; xj = S.kx[sl]; cj = S.kc[sl];
			const int tj = kw0 + q; const uint32_t cc = cj & 0x7fffffffu;
			if (xj != UINT64_MAX && tj >= jw0 && tj < jw1) {
				if (patch_on == 2) ok = tj == patch_prev;
				else {
					int lmax = -(1 << 30), rmin = 1 << 30;
					for (int b = ci - 1; b >= 0; --b) {
						const int sb = ((b >> 6) % 3) * 64 + (b & 63); const int qb = S.kq[sb];
						if (qb < q - (W - 1)) break;
						if (sk3_lt(S.kx[sb], S.kc[sb] & 0x7fffffffu, xj, cc)) { lmax = kw0 + qb; break; }
					}
					for (int b = ci + 1; b < (int)ncand; ++b) {
						const int sb = ((b >> 6) % 3) * 64 + (b & 63); const int qb = S.kq[sb];
						if (qb > q + (W - 1)) break;
						if (sk3_lt(S.kx[sb], S.kc[sb] & 0x7fffffffu, xj, cc)) { rmin = kw0 + qb; break; }
					}
					ok = max(max(tj, tm0), lmax + W) <= min(min(tj + W - 1, kk1), rmin - 1);
					if (patch_on == 1 && tj < W + K - 1) { if (tj == patch_prev) ok = false; else if (xj == pkx && cc == pkc) ok = true; }
				}
			}
		}
		const unsigned long long bal = __ballot(ok);
		if (ok) {
			// position of the k-mer's last base and its span: select run (rbase + K + q) and run (rbase + q) in the decode table
			const uint32_t ge = (uint32_t)(kw0 + q), gs = ge - (uint32_t)K;
			auto pos_of = [&](uint32_t g) -> uint32_t {                     // 0-based index of the last base of run g (g >= 1, decoded)
				int flo = 0, fhi = SK3_SMAX * 64;                            // last word with tab_o <= g
				while (fhi - flo > 1) { const int fm = (flo + fhi) >> 1; if (S.tab_o[fm] <= g) flo = fm; else fhi = fm; }
				return S.tab_g0[flo >> 6] + (uint32_t)(flo & 63) * 16u + sk3_select_end(S.tab_eb[flo], g - S.tab_o[flo]);
			};
			const uint32_t pe = pos_of(ge), ps1 = gs == 0 ? 0u : pos_of(gs) + 1u;      // end1[e] - 1, end1[e - K]
			const uint64_t o = base + nout + (uint64_t)__popcll(bal & ((1ULL << lane) - 1));
			a.pool_x[o] = xj;
			a.pool_info[o] = hao_info_pack(HAS_FT ? (cj & 0x7fffffffu) : 0u, pe, cj >> 31, pe + 1u - ps1);      // rid field carries the count until the end (sketch.cpp:515)
			a.pool_ord[o] = ge;
		}
		nout += (uint32_t)__popcll(bal);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	if (lane == 0) a.chunk_cnt[un] = nout;
}
