// K1+K3: HPC k-mer hashing and (count,hash) window minimizers for gfx950.
//
// Replaces the per-base state machine of mz1_ha_sketch (sketch.cpp:454-579) by a
// data-parallel closed form (validated there against the
// reference):
//
//   * one "iteration" of the reference = one homopolymer run (its last base); runs
//     get ordinals t = 1..T.  Ordinal t >= k carries a k-mer made of runs t-k+1..t.
//   * position j is emitted iff its key (count,hash) equals the minimum key of at
//     least one window of w consecutive ordinals [t-w+1, t], w+k-1 <= t <= T (all ties
//     emitted), plus the first-window quirk (sketch.cpp:523-534,543-547) and the
//     end-of-read flush for reads with fewer than w+k-1 ordinals (sketch.cpp:571-573).
//     Equivalently  key(j) == max_{t in [j, j+w-1] valid} min_{i in [t-w+1, t]} key(i):
//     a sliding minimum followed by a sliding maximum - both done by log-step doubling
//     over LDS, no per-lane divergence.
//   * non-candidate slots (span >= 256, sketch.cpp:505; filtered k-mers, :510) still
//     occupy a window position with the maximal key.
//
// Work decomposition: (read, chunk of HAO_SK_CHUNK window ordinals).  Every chunk is
// self-contained: it re-derives its runs from the packed bases using a per-read
// run-count index at 1024-base granularity (hpc_index_kernel), so long reads split
// over many workgroups and the grid is >> 256 workgroups for any realistic batch.
//
// Reads the fast path does not cover (N bases, even k: strand-symmetric k-mers,
// sketch.cpp:502) go through sketch_scalar_kernel: the exact state machine, one lane
// per read.  There is no host fallback.
#pragma once
#include "hao_common.cuh"

#define HAO_SK_CHUNK 1024      // window ordinals marked per workgroup
#define HAO_SK_TILE 1024       // bases per run-count index tile (= 64 lanes x 16 bases)
#define HAO_SK_THREADS 256

// High-count filter table, device view.  Every (HPC) k-mer of every read asks it for its count, and almost every answer is "not in the table", so the
// lookup is laid out for that: (1) a bitmap over the top hbits bits of the hash (a few MB: it lives in the L2s / the Infinity Cache), a clear bit = count 0;
// (2) a bucketed hash table addressed by the top hbits - 2 bits of the hash, ONE 16-byte load: two slots per bucket, a slot = the hash's low 51 bits << 13 | code
// (code = count 1 .. 4095, 8191 = "above max_kmer_cnt": ha_ft_cnt's INT32_MAX), 0 = empty; (3) a bucket that would need a third slot holds the mark ~0 in
// its first slot and sends its keys to the sorted array (binary search below a 64K-bucket index: the host view's layout, also what hao_index_save writes).
// (Round 3 did (3) for every k-mer: ~9 dependent global loads per lookup; on a repeat-rich 250 Mb genome - 5 M table entries - the sketch took 288 ms
// instead of the 20 ms of an empty table.)
struct hao_ft_dev {
	const uint64_t *keys; const int32_t *vals; const uint32_t *bucket; uint64_t n;      // sorted keys + values + 64K-bucket index on the top 16 hash bits
	const uint32_t *hbit; const ulonglong2 *hslot; int hbits;                             // bitmap (2^hbits bits) and buckets (2^(hbits - 2)) of the hash view; hbits >= 15
};
#define HAO_FT_CODE_BITS 13
#define HAO_FT_CODE_BIG 8191u
#define HAO_FT_MARK (~0ULL)

__device__ __forceinline__ int32_t hao_ft_lookup_sorted(const hao_ft_dev &ft, uint64_t y)
{
	uint32_t b = (uint32_t)(y >> 48), lo = ft.bucket[b], hi = ft.bucket[b + 1];
	while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (ft.keys[m] < y) lo = m + 1; else hi = m; }
	return (lo < ft.bucket[b + 1] && ft.keys[lo] == y) ? ft.vals[lo] : 0;
}
// ha_ft_cnt (htab.cpp:1064-1070)
__device__ __forceinline__ int32_t hao_ft_lookup(const hao_ft_dev &ft, uint64_t y)
{
	if (ft.n == 0) return 0;
	const uint32_t fb = (uint32_t)(y >> (64 - ft.hbits));
	if (!((ft.hbit[fb >> 5] >> (fb & 31)) & 1u)) return 0;
	const ulonglong2 e = ft.hslot[fb >> 2];
	if (e.x == HAO_FT_MARK) return hao_ft_lookup_sorted(ft, y);
	const uint64_t want = y << HAO_FT_CODE_BITS, msk = ~(uint64_t)HAO_FT_CODE_BIG;
	uint32_t code = 0;
	if (e.x && (e.x & msk) == want) code = (uint32_t)e.x & HAO_FT_CODE_BIG;
	else if (e.y && (e.y & msk) == want) code = (uint32_t)e.y & HAO_FT_CODE_BIG;
	return code == HAO_FT_CODE_BIG ? INT32_MAX : (int32_t)code;
}
// build of the hash view from the sorted table (one thread per key; slots and bitmap zeroed beforehand): keys of one bucket are neighbours in the sorted
// array, so a key sees its rank in its bucket and whether the bucket holds more than two keys by looking at most two keys back and ahead
__global__ void hao_ft_hash_kernel(const uint64_t *keys, const int32_t *vals, uint64_t n, int hbits, uint32_t *hbit, unsigned long long *hslot)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t y = keys[i]; const int sh = 64 - (hbits - 2); const uint64_t b = y >> sh;
	const uint32_t fb = (uint32_t)(y >> (64 - hbits));
	atomicOr(&hbit[fb >> 5], 1u << (fb & 31));
	int back = 0, fwd = 0;
	while (back < 2 && i >= (uint64_t)back + 1 && (keys[i - 1 - back] >> sh) == b) ++back;
	while (fwd < 2 && i + 1 + fwd < n && (keys[i + 1 + fwd] >> sh) == b) ++fwd;
	if (back + fwd + 1 > 2) { if (back == 0) { hslot[2 * b] = HAO_FT_MARK; hslot[2 * b + 1] = HAO_FT_MARK; } return; }
	const int32_t v = vals[i]; const uint32_t code = v == INT32_MAX ? HAO_FT_CODE_BIG : (uint32_t)v;      // (a kept k-mer has a count of at least 1, at most 4095)
	hslot[2 * b + back] = y << HAO_FT_CODE_BITS | code;
}

// 16 bases (one big-endian 32-bit word of the packed read) -> bit (30-2j) set iff base j is
// the last base of a homopolymer run.  g0 = index of base 0 of the word.
__device__ __forceinline__ uint32_t hao_run_ends16(const uint8_t *rd, uint32_t len, uint32_t g0, uint32_t *word_out)
{
	if (g0 >= len) { *word_out = 0; return 0; }
	uint32_t nb = (len >> 2) + 1, b = g0 >> 2;             // bytes available: len/4+1 (Process_Read.cpp:443)
	uint32_t W = (uint32_t)rd[b] << 24;
	if (b + 1 < nb) W |= (uint32_t)rd[b + 1] << 16;
	if (b + 2 < nb) W |= (uint32_t)rd[b + 2] << 8;
	if (b + 3 < nb) W |= (uint32_t)rd[b + 3];
	uint32_t nxt = (b + 4 < nb) ? (uint32_t)(rd[b + 4] >> 6) : 0;      // base g0+16
	uint32_t Y = W ^ ((W << 2) | nxt);
	uint32_t ne = (Y | (Y >> 1)) & 0x55555555u;            // base j != base j+1
	uint32_t rem = len - g0;                                // >= 1 bases of this word are inside the read
	uint32_t q = rem > 16 ? 16 : rem - 1;                   // comparisons j vs j+1 valid for j < q
	uint32_t m = q == 0 ? 0 : (q >= 16 ? 0x55555555u : ((0xFFFFFFFFu << (32 - 2 * q)) & 0x55555555u));
	ne &= m;
	if (rem <= 16) ne |= 1u << (30 - 2 * (rem - 1));       // the last base of the read always ends a run
	*word_out = W;
	return ne;
}

__device__ __forceinline__ uint32_t hao_wave_excl_scan(uint32_t v, uint32_t *total)
{	// (DPP row / bank moves: no LDS crossbar trips)
	const uint32_t x = hao_wave_incl_scan_u32(v);
	*total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
	return x - v;
}

// ---------------------------------------------------------------------------------------
// K_A: per-read run-count index. One wave per read; tile_ord[tile_off[r] + i] = number of
// runs that end before base 1024*i; last entry = T (total runs). flags[r] bit0 = needs the
// scalar path.
// ---------------------------------------------------------------------------------------
#define SK3_MIN_TILE_RUNS 544      // hao_sketch3.cuh: a unit of 1024 window ordinals must fit 3 decode steps
#define SK3_LONGSPAN_ENDS 67       // 17 consecutive 16-base words with <= 67 run ends: some 256 bases hold <= 51 runs, a k-mer span can reach 256
// run-end bits of 16 bases (bit 30-2j <-> base j) from the big-endian word W, the base after it (nxt) and the bases left in the read from base 0 of the word
__device__ __forceinline__ uint32_t sk3_run_ends(uint32_t Wd, uint32_t nxt, uint32_t rem, int hpc)
{
	if (rem == 0) return 0;
	const uint32_t full = 0x55555555u;
	if (!hpc) return rem >= 16 ? full : ((0xFFFFFFFFu << (32 - 2 * rem)) & full);
	const uint32_t Y = Wd ^ ((Wd << 2) | nxt);
	uint32_t ne = (Y | (Y >> 1)) & full;
	const uint32_t q = rem > 16 ? 16 : rem - 1;                          // comparisons j vs j+1 are valid for j < q
	const uint32_t m = q == 0 ? 0 : (q >= 16 ? full : ((0xFFFFFFFFu << (32 - 2 * q)) & full));
	ne &= m;
	if (rem <= 16) ne |= 1u << (30 - 2 * (rem - 1));                      // the last base of the read always ends a run
	return ne;
}
// the 16 bases starting at base g0 (a multiple of 16) of a read: one unaligned 8-byte load (the read store has 16 bytes of slack at its end)
__device__ __forceinline__ void sk3_load16(const uint8_t *rd, uint32_t g0, uint32_t &Wd, uint32_t &nxt)
{
	uint2 v; __builtin_memcpy(&v, rd + (g0 >> 2), 8);
	Wd = __builtin_bswap32(v.x); nxt = (v.y & 0xffu) >> 6;
}
__global__ __launch_bounds__(256) void hpc_index_kernel(const uint8_t *packed, const uint64_t *pk_off, const uint32_t *len,
		const uint64_t *tile_off, uint32_t *tile_ord, uint32_t *n_runs, uint64_t rid_lo, uint64_t n_sel, int hpc, uint8_t *slow_flag)
{
	uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_sel) return;
	uint64_t rid = rid_lo + r; const uint8_t *rd = packed + pk_off[rid]; uint32_t L = len[rid];
	uint32_t *to = tile_ord + tile_off[r]; uint32_t run = 0, nt = (L + HAO_SK_TILE - 1) / HAO_SK_TILE;
	uint32_t prev_cum = 0; bool slow = false;
	// four tiles per round: their words are requested before the first is counted (tile by tile the loop ran at one memory round trip per tile, 15 per read)
	for (uint32_t t4 = 0; t4 < nt; t4 += 4) {
		uint32_t cc[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t g0 = (t4 + u) * HAO_SK_TILE + hao_lane() * 16; uint32_t c = 0;
			if (g0 < L) {
				if (hpc) { uint32_t Wd, nxt; sk3_load16(rd, g0, Wd, nxt); c = __popc(sk3_run_ends(Wd, nxt, L - g0, 1)); }
				else c = L - g0 > 16 ? 16 : L - g0;
			}
			cc[u] = c;
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t ti = t4 + u;
			if (ti >= nt) break;
			const uint32_t g0 = ti * HAO_SK_TILE + hao_lane() * 16, c = cc[u];
			uint32_t tot; const uint32_t excl = hao_wave_excl_scan(c, &tot);
			if (hao_lane() == 0) to[ti] = run;
			if (slow_flag && hpc) {       // reads the unit kernel leaves to the scalar one: long runs (see hao_sketch3.cuh)
				const uint32_t cum = run + excl + c;                               // run ends up to and including this word
				if (ti * HAO_SK_TILE + HAO_SK_TILE <= L && tot < SK3_MIN_TILE_RUNS) slow = true;
				const int ln = hao_lane();
				const uint32_t a = (uint32_t)__shfl((int)cum, (ln + 47) & 63), b = (uint32_t)__shfl((int)prev_cum, (ln + 47) & 63);   // lane - 17 (mod 64)
				const uint32_t before = ln >= 17 ? a : (ti > 0 ? b : 0u);
				if (ti * 64 + (uint32_t)ln >= 16 && g0 + 16 <= L && cum - before <= SK3_LONGSPAN_ENDS) slow = true;
				prev_cum = cum;
			}
			run += tot;
		}
	}
	if (hao_lane() == 0) { to[nt] = run; n_runs[r] = run; }
	if (slow_flag && __any(slow) && hao_lane() == 0) slow_flag[r] = 1;
}

struct hao_sk_args {
	const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len;
	const uint64_t *tile_off; const uint32_t *tile_ord; const uint32_t *n_runs;
	const uint64_t *chunk_off;     // [n_sel+1] exclusive scan of chunks per read
	const uint8_t *scalar_flag;    // [n_sel] 1 = handled by sketch_scalar_kernel
	uint64_t rid_lo, n_sel; int k, w, hpc;
	hao_ft_dev ft;
	// output pool (append order arbitrary) + per-chunk record
	uint64_t *pool_x, *pool_info; uint32_t *pool_ord; unsigned long long *pool_cursor; uint64_t pool_cap;
	uint64_t *chunk_base; uint32_t *chunk_cnt; int *err;
	const uint32_t *unit_rid; uint64_t n_units;      // unit kernel: read of every unit
	uint64_t pool_static;          // unit kernel: the first pool_static pool entries are fixed slots of SK3_SLOT entries per unit; the cursor allocates behind them
};

struct hao_key { uint64_t x; uint32_t c; };
template<bool HAS_FT> __device__ __forceinline__ bool hao_key_lt(const hao_key &a, const hao_key &b)
{ if (HAS_FT) return a.c < b.c || (a.c == b.c && a.x < b.x); return a.x < b.x; }
template<bool HAS_FT> __device__ __forceinline__ bool hao_key_eq(const hao_key &a, const hao_key &b)
{ if (HAS_FT) return a.c == b.c && a.x == b.x; return a.x == b.x; }

#define HAO_SK_GMAX 6   // entries per lane: (CHUNK + 2*(w-1)) / 256 rounded up, w <= 255

// ---------------------------------------------------------------------------------------
// K_B: one workgroup per (read, chunk).
// LDS layout (dynamic): end1[NE] u32 | rcode[NE] u8 | planes[2][NW] u64 | kx[NKP] u64 | kc[NKP] u32 |
//                       bx[2][NKP] u64 | bc[2][NKP] u32 | misc
// ---------------------------------------------------------------------------------------
template<bool HAS_FT>
__global__ __launch_bounds__(HAO_SK_THREADS) void sketch_chunk_kernel(hao_sk_args a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	const int k = a.k, w = a.w, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int NE = HAO_SK_CHUNK + 2 * (w - 1) + k + 1;
	const int NW = (NE + 63) / 64 + 1;
	const int NKP = ((HAO_SK_CHUNK + 2 * (w - 1) + HAO_SK_THREADS - 1) / HAO_SK_THREADS) * HAO_SK_THREADS;
	uint64_t *pl0 = (uint64_t*)smem, *pl1 = pl0 + NW, *kx = pl1 + NW, *bx0 = kx + NKP, *bx1 = bx0 + NKP;
	uint32_t *end1 = (uint32_t*)(bx1 + NKP), *kc = end1 + NE, *bc0 = kc + NKP, *bc1 = bc0 + NKP;
	uint8_t *rcode = (uint8_t*)(bc1 + NKP);
	__shared__ uint32_t s_cnt[4]; __shared__ unsigned long long s_base; __shared__ int s_patch_prev; __shared__ int s_patch_on;

	// which (read, chunk)?
	uint64_t ch = blockIdx.x, lo = 0, hi = a.n_sel;
	if (ch >= a.chunk_off[a.n_sel]) return;          // the grid is sized from read lengths (an upper bound of the run counts): no host round trip for the exact count
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a.chunk_off[m + 1] <= ch) lo = m + 1; else hi = m; }
	const uint64_t r = lo;
	if (a.scalar_flag[r]) return;                                   // record is written by the scalar kernel
	const uint32_t ci = (uint32_t)(ch - a.chunk_off[r]);
	const uint64_t rid = a.rid_lo + r; const uint8_t *rd = a.packed + a.pk_off[rid]; const uint32_t L = a.len[rid];
	const int T = (int)a.n_runs[r];
	const uint32_t *tord = a.tile_ord + a.tile_off[r]; const uint32_t ntile = (L + HAO_SK_TILE - 1) / HAO_SK_TILE;
	const int j0 = k + (int)ci * HAO_SK_CHUNK, j1 = min(j0 + HAO_SK_CHUNK, T + 1);      // marks decided for ordinals [j0, j1)
	if (j0 > T) { if (tid == 0) { a.chunk_base[ch] = 0; a.chunk_cnt[ch] = 0; } return; } // read with < k runs (its single chunk)
	const int kk0 = max(k, j0 - w + 1), kk1 = min(T, j1 - 1 + w - 1);                  // keys needed for ordinals [kk0, kk1]
	const int rbase = kk0 - k;                                                         // end1[e] <-> ordinal rbase + e
	const int nE = kk1 - rbase + 1, nK = kk1 - kk0 + 1;
	const uint64_t mask = (1ULL << k) - 1;

	// ---- S2: run ends of the needed ordinal range -> rcode/end1 ----
	if (tid == 0 && rbase == 0) end1[0] = 0;                        // ordinal 0 "ends" before base 0
	{
		const uint32_t first = rbase > 0 ? (uint32_t)rbase : 1u;   // first ordinal to materialise
		uint32_t tlo = 0, thi = ntile;                              // last tile with tord[ti] < first
		while (thi - tlo > 1) { uint32_t m = (tlo + thi) >> 1; if (tord[m] < first) tlo = m; else thi = m; }
		for (uint32_t ti = tlo + wv; ti < ntile && tord[ti] < (uint32_t)kk1; ti += 4) {
			uint32_t W, g0 = ti * HAO_SK_TILE + lane * 16, tot;
			uint32_t eb = a.hpc ? hao_run_ends16(rd, L, g0, &W) : 0;
			if (!a.hpc) { hao_run_ends16(rd, L, g0, &W); uint32_t rem = g0 >= L ? 0 : (L - g0 > 16 ? 16 : L - g0); eb = rem == 0 ? 0 : (rem >= 16 ? 0x55555555u : ((0xFFFFFFFFu << (32 - 2 * rem)) & 0x55555555u)); }
			uint32_t o = tord[ti] + hao_wave_excl_scan(__popc(eb), &tot) + 1;
			while (eb) {
				int hb = 31 - __clz(eb); int j = (30 - hb) >> 1; eb &= ~(1u << hb);
				if (o >= first && o <= (uint32_t)kk1) { int e = (int)o - rbase; rcode[e] = (W >> (30 - 2 * j)) & 3; end1[e] = g0 + j + 1; }
				++o;
			}
		}
	}
	__syncthreads();
	// ---- S2b: bit planes of the run codes (bit e of plane <-> run rbase+e) ----
	for (int e0 = wv * 64; e0 < nE + 64; e0 += 256) {
		int e = e0 + lane; uint32_t c = (e >= 1 && e < nE) ? rcode[e] : 0;
		unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c >> 1);
		if (lane == 0 && (e0 >> 6) < NW) { pl0[e0 >> 6] = b0; pl1[e0 >> 6] = b1; }
	}
	__syncthreads();
	// ---- S3: keys of ordinals kk0..kk1 ----
	hao_key key[HAO_SK_GMAX];
	const int G = NKP / HAO_SK_THREADS;
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) {
		key[g].x = UINT64_MAX; key[g].c = HAO_CNT_DUMMY;
		int q = tid + g * HAO_SK_THREADS;                           // q <-> ordinal kk0 + q
		if (g < G && q < nK) {
			int e = q + k, s = e - k + 1;                            // runs s..e of the local arrays
			int wi = s >> 6, sh = s & 63;
			uint64_t W0 = pl0[wi] >> sh, W1 = pl1[wi] >> sh;
			if (sh) { W0 |= pl0[wi + 1] << (64 - sh); W1 |= pl1[wi + 1] << (64 - sh); }
			W0 &= mask; W1 &= mask;
			uint64_t f1 = __brevll(W1) >> (64 - k), r1 = ~W1 & mask;  // high bit-planes of forward / reverse-complement k-mer
			uint32_t span = end1[e] - end1[e - k];
			if (span < 256) {
				uint64_t y = f1 < r1 ? hao_hash64(__brevll(W0) >> (64 - k)) + hao_hash64(f1)
									 : hao_hash64(~W0 & mask) + hao_hash64(r1);
				if (HAS_FT) { int32_t cnt = hao_ft_lookup(a.ft, y); if (cnt < (1 << 28)) { key[g].x = y; key[g].c = (uint32_t)cnt; } }
				else { key[g].x = y; key[g].c = 0; }
			}
		}
		if (g < G) { kx[q] = key[g].x; if (HAS_FT) kc[q] = key[g].c; }
	}
	// ---- S4: sliding minimum over [t-w+1, t] by doubling; ping-pong bx0/bx1 ----
	hao_key v[HAO_SK_GMAX];
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) v[g] = key[g];
	int cur = 0, cover = 1;
	auto publish = [&](int which) {
		uint64_t *bx = which ? bx1 : bx0; uint32_t *bc = which ? bc1 : bc0;
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { int q = tid + g * HAO_SK_THREADS; bx[q] = v[g].x; if (HAS_FT) bc[q] = v[g].c; }
		__syncthreads();
	};
	auto fetch = [&](int which, int q, hao_key dflt) -> hao_key {
		const uint64_t *bx = which ? bx1 : bx0; const uint32_t *bc = which ? bc1 : bc0;
		if (q < 0 || q >= NKP) return dflt;
		hao_key o; o.x = bx[q]; o.c = HAS_FT ? bc[q] : 0; return o;
	};
	const hao_key kmax = { UINT64_MAX, HAO_CNT_DUMMY }, kmin = { 0, 0 };
	while (cover * 2 <= w) {
		publish(cur);
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { hao_key o = fetch(cur, tid + g * HAO_SK_THREADS - cover, kmax); if (hao_key_lt<HAS_FT>(o, v[g])) v[g] = o; }
		cover *= 2; cur ^= 1;
	}
	if (cover < w) {
		publish(cur);
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { hao_key o = fetch(cur, tid + g * HAO_SK_THREADS - (w - cover), kmax); if (hao_key_lt<HAS_FT>(o, v[g])) v[g] = o; }
		cur ^= 1;
	}
	// v = m(t) for t = kk0 + q.  Windows are valid only for w+k-1 <= t <= T: others count as -inf in the max pass.
	const int tm0 = max(j0, w + k - 1);
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { int t = kk0 + tid + g * HAO_SK_THREADS; if (t < tm0 || t > kk1) v[g] = kmin; }
	// ---- sliding maximum over [j, j+w-1] ----
	cover = 1;
	while (cover * 2 <= w) {
		publish(cur);
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { hao_key o = fetch(cur, tid + g * HAO_SK_THREADS + cover, kmin); if (hao_key_lt<HAS_FT>(v[g], o)) v[g] = o; }
		cover *= 2; cur ^= 1;
	}
	if (cover < w) {
		publish(cur);
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) if (g < G) { hao_key o = fetch(cur, tid + g * HAO_SK_THREADS + (w - cover), kmin); if (hao_key_lt<HAS_FT>(v[g], o)) v[g] = o; }
		cur ^= 1;
	}
	// ---- marks ----
	bool mk[HAO_SK_GMAX];
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) {
		int t = kk0 + tid + g * HAO_SK_THREADS;
		mk[g] = g < G && t >= j0 && t < j1 && key[g].x != UINT64_MAX && hao_key_eq<HAS_FT>(key[g], v[g]);
	}
	// first-window quirk (chunk 0 only) and short reads: decided by one lane over <= w keys in LDS (kx/kc are intact)
	if (tid == 0) { s_patch_on = 0; s_patch_prev = -1; }
	__syncthreads();
	if (ci == 0 && tid == 0) {
		auto kq = [&](int t) -> hao_key { hao_key o; o.x = kx[t - kk0]; o.c = HAS_FT ? kc[t - kk0] : 0; return o; };
		if (T >= w + k - 1) {
			const int t0 = w + k - 1; int prev = -1; hao_key pk = kmax;
			for (int t = k; t < t0; ++t) { hao_key o = kq(t); if (!hao_key_lt<HAS_FT>(pk, o)) { pk = o; prev = t; } }   // newest minimum of [k, t0-1]
			if (prev >= 0 && pk.x != UINT64_MAX) { hao_key o = kq(t0); if (!hao_key_lt<HAS_FT>(pk, o)) { s_patch_on = 1; s_patch_prev = prev; } }
		} else {                                                    // fewer than w+k-1 runs: only the end-of-read flush fires
			int prev = -1; hao_key pk = kmax;
			for (int t = max(k, T - w + 1); t <= T; ++t) { hao_key o = kq(t); if (!hao_key_lt<HAS_FT>(pk, o)) { pk = o; prev = t; } }
			s_patch_on = 2; s_patch_prev = (prev >= 0 && pk.x != UINT64_MAX) ? prev : -1;
		}
	}
	__syncthreads();
	if (s_patch_on == 1) {          // key(t0) <= key(prev): prev itself is never emitted, its older ties in [k, t0-1] are
		const int prev = s_patch_prev; hao_key pk; pk.x = kx[prev - kk0]; pk.c = HAS_FT ? kc[prev - kk0] : 0;
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) {
			int t = kk0 + tid + g * HAO_SK_THREADS;
			if (g < G && t >= k && t < w + k - 1) { if (t == prev) mk[g] = false; else if (hao_key_eq<HAS_FT>(key[g], pk)) mk[g] = true; }
		}
	} else if (s_patch_on == 2) {
#pragma unroll
		for (int g = 0; g < HAO_SK_GMAX; ++g) { int t = kk0 + tid + g * HAO_SK_THREADS; mk[g] = g < G && t == s_patch_prev; }
	}
	// ---- S5: ordered compaction + append ----
	// entry q = tid + 256 g: order by q. rank = (#marked with smaller q).  Count per (g, wave) via ballots.
	uint32_t myrank[HAO_SK_GMAX]; __shared__ uint32_t s_gw[HAO_SK_GMAX][4];
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) {
		unsigned long long bal = __ballot(mk[g]);
		myrank[g] = __popcll(bal & ((1ULL << lane) - 1));
		if (lane == 0) s_gw[g][wv] = __popcll(bal);
	}
	__syncthreads();
	if (tid == 0) {
		uint32_t run = 0;
		for (int g = 0; g < HAO_SK_GMAX; ++g) for (int x = 0; x < 4; ++x) { uint32_t c = s_gw[g][x]; s_gw[g][x] = run; run += c; }
		s_cnt[0] = run;
		unsigned long long base = run ? atomicAdd(a.pool_cursor, (unsigned long long)run) : 0ULL;
		if (base + run > a.pool_cap) { *a.err = 1; s_cnt[0] = 0; run = 0; }
		s_base = base; a.chunk_base[ch] = base; a.chunk_cnt[ch] = run;
	}
	__syncthreads();
	if (s_cnt[0] == 0) return;
#pragma unroll
	for (int g = 0; g < HAO_SK_GMAX; ++g) {
		if (!mk[g]) continue;
		int q = tid + g * HAO_SK_THREADS, e = q + k, s = e - k + 1, wi = s >> 6, sh = s & 63;
		uint64_t W1 = pl1[wi] >> sh; if (sh) W1 |= pl1[wi + 1] << (64 - sh); W1 &= mask;
		uint32_t rev = (__brevll(W1) >> (64 - k)) < (~W1 & mask) ? 0 : 1;
		uint32_t span = end1[e] - end1[e - k];
		uint64_t o = s_base + s_gw[g][wv] + myrank[g];
		a.pool_x[o] = key[g].x;
		a.pool_info[o] = hao_info_pack(HAS_FT ? key[g].c : 0, end1[e] - 1, rev, span);     // rid field carries the count until the end (sketch.cpp:515)
		a.pool_ord[o] = (uint32_t)(kk0 + q);
	}
}

// ---------------------------------------------------------------------------------------
// K_B, wave-local variant (used when w is the compile-time W and k + 7 <= 64): each wave owns 512 consecutive
// window ordinals in a BLOCKED layout (lane L holds entries 8L..8L+7) and decides the marks of the middle
// 512 - 2(W-1); the four waves of a workgroup share one decode of the runs (LDS) and nothing else.
//   * hashes: the lane extracts ONE (k+7)-bit window of each bit plane and shifts it for its 8 k-mers;
//   * sliding minimum over W ordinals = min(prefix of my 8, full predecessors lanes, suffix of one farther lane):
//     per-lane prefix/suffix minima in registers, lane-axis doubling with ds_bpermute shuffles, no LDS traffic,
//     no barrier; the sliding maximum is the mirror image;
//   * marks compact through two wave scans and one atomic per workgroup.
// ---------------------------------------------------------------------------------------
#define HAO_SK2_WENT 512
template<int W> struct hao_sk2 { static constexpr int MW = HAO_SK2_WENT - 2 * (W - 1); static constexpr int CHUNK = 4 * MW; };

template<bool HAS_FT> __device__ __forceinline__ hao_key hao_kmin(const hao_key &a, const hao_key &b) { return hao_key_lt<HAS_FT>(b, a) ? b : a; }
template<bool HAS_FT> __device__ __forceinline__ hao_key hao_kmax(const hao_key &a, const hao_key &b) { return hao_key_lt<HAS_FT>(a, b) ? b : a; }
template<bool HAS_FT> __device__ __forceinline__ hao_key hao_kshfl_up(const hao_key &a, int d)
{ hao_key o; o.x = (uint64_t)__shfl_up((unsigned long long)a.x, d); o.c = HAS_FT ? __shfl_up(a.c, d) : 0; return o; }
template<bool HAS_FT> __device__ __forceinline__ hao_key hao_kshfl_down(const hao_key &a, int d)
{ hao_key o; o.x = (uint64_t)__shfl_down((unsigned long long)a.x, d); o.c = HAS_FT ? __shfl_down(a.c, d) : 0; return o; }

// combine over the N lanes before (DIR = 0) / after (DIR = 1) me, of the per-lane value `all`; MIN selects min or max
template<bool HAS_FT, bool MIN, int DIR, int N> __device__ __forceinline__ hao_key hao_lane_window(const hao_key &all)
{
	auto comb = [](const hao_key &a, const hao_key &b) { return MIN ? hao_kmin<HAS_FT>(a, b) : hao_kmax<HAS_FT>(a, b); };
	auto sh = [](const hao_key &a, int d) { return DIR == 0 ? hao_kshfl_up<HAS_FT>(a, d) : hao_kshfl_down<HAS_FT>(a, d); };
	// p[j] covers the 2^j lanes next to me (excluding me)
	hao_key p[6]; p[0] = sh(all, 1);
#pragma unroll
	for (int j = 1; j < 6; ++j) if ((1 << j) <= N) p[j] = comb(p[j - 1], sh(p[j - 1], 1 << (j - 1)));
	hao_key res = p[0]; int covered = 0; bool first = true;
#pragma unroll
	for (int j = 5; j >= 0; --j) if ((N >> j) & 1) { hao_key piece = covered ? sh(p[j], covered) : p[j]; res = first ? piece : comb(res, piece); first = false; covered += 1 << j; }
	return res;
}

template<bool HAS_FT, int W>
__global__ __launch_bounds__(256) void sketch_chunk_wave_kernel(hao_sk_args a)
{
	constexpr int MW = hao_sk2<W>::MW, CHUNK = hao_sk2<W>::CHUNK, NE_MAX = CHUNK + 2 * (W - 1) + 64 + 2, NW_MAX = (NE_MAX + 63) / 64 + 2;
	__shared__ uint32_t end1[NE_MAX]; __shared__ uint8_t rcode[NE_MAX]; __shared__ uint64_t pl0[NW_MAX], pl1[NW_MAX];
	__shared__ uint64_t fx[64]; __shared__ uint32_t fc[64]; __shared__ uint32_t s_wcnt[4]; __shared__ unsigned long long s_base; __shared__ int s_patch_prev, s_patch_on;
	const int k = a.k, w = W, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	uint64_t ch = blockIdx.x, lo = 0, hi = a.n_sel;
	if (ch >= a.chunk_off[a.n_sel]) return;          // the grid is sized from read lengths (an upper bound of the run counts): no host round trip for the exact count
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a.chunk_off[m + 1] <= ch) lo = m + 1; else hi = m; }
	const uint64_t r = lo;
	if (a.scalar_flag[r]) return;
	const uint32_t ci = (uint32_t)(ch - a.chunk_off[r]);
	const uint64_t rid = a.rid_lo + r; const uint8_t *rd = a.packed + a.pk_off[rid]; const uint32_t L = a.len[rid];
	const int T = (int)a.n_runs[r];
	const uint32_t *tord = a.tile_ord + a.tile_off[r]; const uint32_t ntile = (L + HAO_SK_TILE - 1) / HAO_SK_TILE;
	const int j0 = k + (int)ci * CHUNK, j1 = min(j0 + CHUNK, T + 1);
	if (j0 > T) { if (tid == 0) { a.chunk_base[ch] = 0; a.chunk_cnt[ch] = 0; } return; }
	const int kk0 = max(k, j0 - w + 1), kk1 = min(T, j1 - 1 + w - 1);
	const int rbase = kk0 - k, nE = kk1 - rbase + 1;
	const uint64_t mask = (1ULL << k) - 1;
	// ---- run ends of the workgroup's ordinal range -> rcode / end1 (shared by the four waves) ----
	if (tid == 0 && rbase == 0) end1[0] = 0;
	{
		const uint32_t first = rbase > 0 ? (uint32_t)rbase : 1u;
		uint32_t tlo = 0, thi = ntile;
		while (thi - tlo > 1) { uint32_t m = (tlo + thi) >> 1; if (tord[m] < first) tlo = m; else thi = m; }
		for (uint32_t ti = tlo + wv; ti < ntile && tord[ti] < (uint32_t)kk1; ti += 4) {
			uint32_t Wd, g0 = ti * HAO_SK_TILE + lane * 16, tot;
			uint32_t eb = hao_run_ends16(rd, L, g0, &Wd);
			if (!a.hpc) { uint32_t rem = g0 >= L ? 0 : (L - g0 > 16 ? 16 : L - g0); eb = rem == 0 ? 0 : (rem >= 16 ? 0x55555555u : ((0xFFFFFFFFu << (32 - 2 * rem)) & 0x55555555u)); }
			uint32_t o = tord[ti] + hao_wave_excl_scan(__popc(eb), &tot) + 1;
			while (eb) {
				int hb = 31 - __clz(eb); int j = (30 - hb) >> 1; eb &= ~(1u << hb);
				if (o >= first && o <= (uint32_t)kk1) { int e = (int)o - rbase; rcode[e] = (Wd >> (30 - 2 * j)) & 3; end1[e] = g0 + j + 1; }
				++o;
			}
		}
	}
	__syncthreads();
	for (int e0 = wv * 64; e0 < nE + 128; e0 += 256) {
		int e = e0 + lane; uint32_t c = (e >= 1 && e < nE) ? rcode[e] : 0;
		unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c >> 1);
		if (lane == 0 && (e0 >> 6) < NW_MAX) { pl0[e0 >> 6] = b0; pl1[e0 >> 6] = b1; }
	}
	__syncthreads();
	// ---- this wave's 512 ordinals: marks for [jw0, jw1), keys for [kw0, kw0 + 512) ----
	const int jw0 = j0 + wv * MW, jw1 = min(jw0 + MW, j1);
	const int kw0 = max(k, jw0 - (w - 1));                       // entry q <-> ordinal kw0 + q
	const hao_key kmax = { UINT64_MAX, HAO_CNT_DUMMY }, kmin = { 0, 0 };
	hao_key key[8];
	{
		const int q0 = lane * 8, t0 = kw0 + q0;                    // my first ordinal
		const int e0 = t0 - rbase, s0 = e0 - k + 1;                // local run index of its last / first run
		uint64_t B0 = 0, B1 = 0;
		if (jw0 < j1 && t0 <= kk1) {
			const int wi = s0 >> 6, sh = s0 & 63;
			B0 = pl0[wi] >> sh; B1 = pl1[wi] >> sh;
			if (sh) { B0 |= pl0[wi + 1] << (64 - sh); B1 |= pl1[wi + 1] << (64 - sh); }
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			key[i] = kmax;
			const int t = t0 + i;
			if (jw0 < j1 && t <= kk1) {
				const uint64_t W0 = (B0 >> i) & mask, W1 = (B1 >> i) & mask;
				const uint64_t f1 = __brevll(W1) >> (64 - k), r1 = ~W1 & mask;
				const uint32_t span = end1[e0 + i] - end1[e0 + i - k];
				if (span < 256) {
					const uint64_t y = f1 < r1 ? hao_hash64(__brevll(W0) >> (64 - k)) + hao_hash64(f1) : hao_hash64(~W0 & mask) + hao_hash64(r1);
					if (HAS_FT) { int32_t cnt = hao_ft_lookup(a.ft, y); if (cnt < (1 << 28)) { key[i].x = y; key[i].c = (uint32_t)cnt; } }
					else { key[i].x = y; key[i].c = 0; }
				}
			}
		}
	}
	// ---- sliding minimum over the W ordinals ending at each entry ----
	hao_key m[8];
	{
		hao_key pre[8], suf[8];
		pre[0] = key[0];
#pragma unroll
		for (int i = 1; i < 8; ++i) pre[i] = hao_kmin<HAS_FT>(pre[i - 1], key[i]);
		suf[7] = key[7];
#pragma unroll
		for (int i = 6; i >= 0; --i) suf[i] = hao_kmin<HAS_FT>(key[i], suf[i + 1]);
		constexpr int QA = (W - 1) / 8, QB = (W - 8) / 8;            // number of full predecessor lanes for i = 0 and i = 7
		const hao_key AA = hao_lane_window<HAS_FT, true, 0, QA>(pre[7]);
		const hao_key AB = QB == QA ? AA : hao_lane_window<HAS_FT, true, 0, (QB > 0 ? QB : 1)>(pre[7]);
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			constexpr int dummy = 0; (void)dummy;
			const int rem = W - 1 - i, nfull = rem / 8, part = rem % 8;       // compile-time after unrolling
			hao_key v = pre[i];
			if (nfull > 0) v = hao_kmin<HAS_FT>(v, nfull == QA ? AA : AB);
			if (part > 0) v = hao_kmin<HAS_FT>(v, hao_kshfl_up<HAS_FT>(suf[8 - part], nfull + 1));
			m[i] = v;
		}
	}
	// windows are valid for ordinals t with max(jw0, w+k-1) <= t <= T (and inside the wave's 512): the rest is -inf for the maximum
	{
		const int tm0 = max(jw0, w + k - 1), q0 = lane * 8;
#pragma unroll
		for (int i = 0; i < 8; ++i) { const int t = kw0 + q0 + i; if (t < tm0 || t > kk1 || t - kw0 < W - 1) m[i] = kmin; }
	}
	// ---- sliding maximum over the W ordinals starting at each entry ----
	bool mk[8];
	{
		hao_key pre[8], suf[8];
		pre[0] = m[0];
#pragma unroll
		for (int i = 1; i < 8; ++i) pre[i] = hao_kmax<HAS_FT>(pre[i - 1], m[i]);
		suf[7] = m[7];
#pragma unroll
		for (int i = 6; i >= 0; --i) suf[i] = hao_kmax<HAS_FT>(m[i], suf[i + 1]);
		constexpr int QA = (W - 8) / 8, QB = (W - 1) / 8;            // full successor lanes for i = 0 and i = 7
		const hao_key AA = hao_lane_window<HAS_FT, false, 1, (QA > 0 ? QA : 1)>(pre[7]);
		const hao_key AB = QB == QA ? AA : hao_lane_window<HAS_FT, false, 1, QB>(pre[7]);
		const int q0 = lane * 8;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int rem = W - (8 - i), nfull = rem / 8, part = rem % 8;      // entries needed after my own suffix
			hao_key v = suf[i];
			if (nfull > 0) v = hao_kmax<HAS_FT>(v, nfull == QA ? AA : AB);
			if (part > 0) v = hao_kmax<HAS_FT>(v, hao_kshfl_down<HAS_FT>(pre[part - 1], nfull + 1));
			const int t = kw0 + q0 + i;
			mk[i] = t >= jw0 && t < jw1 && key[i].x != UINT64_MAX && hao_key_eq<HAS_FT>(key[i], v);
		}
	}
	// ---- first-window quirk / short reads (wave 0 of chunk 0): one lane over <= W keys staged in LDS ----
	if (tid == 0) { s_patch_on = 0; s_patch_prev = -1; }
	if (ci == 0 && wv == 0 && lane < 8) {
#pragma unroll
		for (int i = 0; i < 8; ++i) { fx[lane * 8 + i] = key[i].x; fc[lane * 8 + i] = key[i].c; }   // ordinals k .. k+63
	}
	__syncthreads();
	if (ci == 0 && tid == 0) {
		auto kq = [&](int t) -> hao_key { hao_key o; o.x = fx[t - k]; o.c = HAS_FT ? fc[t - k] : 0; return o; };
		if (T >= w + k - 1) {
			const int t0 = w + k - 1; int prev = -1; hao_key pk = kmax;
			for (int t = k; t < t0; ++t) { hao_key o = kq(t); if (!hao_key_lt<HAS_FT>(pk, o)) { pk = o; prev = t; } }
			if (prev >= 0 && pk.x != UINT64_MAX) { hao_key o = kq(t0); if (!hao_key_lt<HAS_FT>(pk, o)) { s_patch_on = 1; s_patch_prev = prev; } }
		} else {
			int prev = -1; hao_key pk = kmax;
			for (int t = max(k, T - w + 1); t <= T; ++t) { hao_key o = kq(t); if (!hao_key_lt<HAS_FT>(pk, o)) { pk = o; prev = t; } }
			s_patch_on = 2; s_patch_prev = (prev >= 0 && pk.x != UINT64_MAX) ? prev : -1;
		}
	}
	__syncthreads();
	if (s_patch_on && wv == 0) {
		const int prev = s_patch_prev, q0 = lane * 8;
		if (s_patch_on == 1) {
			hao_key pk; pk.x = fx[prev - k]; pk.c = HAS_FT ? fc[prev - k] : 0;
#pragma unroll
			for (int i = 0; i < 8; ++i) { const int t = kw0 + q0 + i; if (t >= k && t < w + k - 1) { if (t == prev) mk[i] = false; else if (hao_key_eq<HAS_FT>(key[i], pk)) mk[i] = true; } }
		} else {
#pragma unroll
			for (int i = 0; i < 8; ++i) { const int t = kw0 + q0 + i; mk[i] = t == prev; }
		}
	} else if (s_patch_on == 2) {
#pragma unroll
		for (int i = 0; i < 8; ++i) mk[i] = false;
	}
	// ---- ordered compaction: entries are in ordinal order inside a lane, lanes inside a wave, waves inside the workgroup ----
	uint32_t mine = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) mine += mk[i] ? 1u : 0u;
	uint32_t wtot; const uint32_t lane_off = hao_wave_excl_scan(mine, &wtot);
	if (lane == 0) s_wcnt[wv] = wtot;
	__syncthreads();
	if (tid == 0) {
		uint32_t run = 0;
		for (int x = 0; x < 4; ++x) { uint32_t c = s_wcnt[x]; s_wcnt[x] = run; run += c; }
		unsigned long long base = run ? atomicAdd(a.pool_cursor, (unsigned long long)run) : 0ULL;
		if (base + run > a.pool_cap) { *a.err = 1; run = 0; for (int x = 0; x < 4; ++x) s_wcnt[x] = 0xffffffffu; }
		s_base = base; a.chunk_base[ch] = base; a.chunk_cnt[ch] = run;
	}
	__syncthreads();
	if (s_wcnt[wv] == 0xffffffffu || wtot == 0) return;
	// A wave marks ~16 of its 512 entries.  Writing them from where they sit would run the output code eight times (once per register slot) with two or
	// three lanes active each time; instead the marked entries are listed in LDS (entry number, key) in output order and the first lanes of the wave write
	// one minimizer each: one pass, consecutive lanes on consecutive addresses.  (64 per round; more than 64 marks per wave means a degenerate read.)
	__shared__ uint16_t l_q[4][64]; __shared__ uint64_t l_x[4][64]; __shared__ uint32_t l_c[4][64];
	const uint64_t obase = s_base + s_wcnt[wv]; const int q0 = lane * 8;
	for (uint32_t done = 0; done < wtot; done += 64) {
		uint32_t rnk = lane_off;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (mk[i]) { const uint32_t at = rnk - done; if (at < 64u) { l_q[wv][at] = (uint16_t)(q0 + i); l_x[wv][at] = key[i].x; if (HAS_FT) l_c[wv][at] = key[i].c; } ++rnk; }
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const uint32_t b = done + (uint32_t)lane;
		if (b < wtot) {
			const int q = l_q[wv][lane], e = kw0 + q - rbase, s1 = e - k + 1, wi = s1 >> 6, sh = s1 & 63;
			uint64_t W1 = pl1[wi] >> sh; if (sh) W1 |= pl1[wi + 1] << (64 - sh); W1 &= mask;
			const uint32_t rev = (__brevll(W1) >> (64 - k)) < (~W1 & mask) ? 0 : 1;
			const uint64_t o = obase + b;
			a.pool_x[o] = l_x[wv][lane];
			a.pool_info[o] = hao_info_pack(HAS_FT ? l_c[wv][lane] : 0, end1[e] - 1, rev, end1[e] - end1[e - k]);
			a.pool_ord[o] = (uint32_t)(kw0 + q);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
}

static inline size_t hao_sk_smem_bytes(int w, int k)
{
	size_t NE = HAO_SK_CHUNK + 2 * (w - 1) + k + 1, NW = (NE + 63) / 64 + 1;
	size_t NKP = ((HAO_SK_CHUNK + 2 * (w - 1) + HAO_SK_THREADS - 1) / HAO_SK_THREADS) * HAO_SK_THREADS;
	return 2 * NW * 8 + 3 * NKP * 8 + NE * 4 + 3 * NKP * 4 + NE + 64;
}

// ---------------------------------------------------------------------------------------
// K_C: gather chunk outputs (pool, arbitrary order) into per-read lists in position order.
// One wave per chunk. dst offset = chunk_dst[ch] (exclusive scan of chunk_cnt).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sketch_gather_kernel(const uint64_t *pool_x, const uint64_t *pool_info, const uint32_t *pool_ord,
		const uint64_t *chunk_base, const uint32_t *chunk_cnt, const uint64_t *chunk_dst, uint64_t n_chunks,
		uint64_t *out_x, uint64_t *out_info, uint32_t *out_ord, uint64_t out_cap, int *err)
{
	uint64_t ch = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (ch >= n_chunks) return;
	uint64_t b = chunk_base[ch], d = chunk_dst[ch]; uint32_t n = chunk_cnt[ch];
	if (d + n > out_cap) { if (hao_lane() == 0 && n) *err = 1; return; }      // the gathered list is sized by an estimate: the host retries with the pool capacity
	for (uint32_t i = hao_lane(); i < n; i += 64) { out_x[d + i] = pool_x[b + i]; out_info[d + i] = pool_info[b + i]; out_ord[d + i] = pool_ord[b + i]; }
}

// K_C + final in one pass (no thinning to run in between): one wave per read walks its chunks' pool segments and writes the final list, the read id
// stamped into info.rid (sketch.cpp:577-578).  dst offsets = chunk_dst (exclusive scan of chunk_cnt).
__global__ __launch_bounds__(256) void sketch_gather_finish_kernel(const uint64_t *pool_x, const uint64_t *pool_info, const uint64_t *chunk_base, const uint32_t *chunk_cnt,
		const uint64_t *chunk_dst, const uint64_t *chunk_off, uint64_t rid_lo, uint64_t n_sel, int stamp_rid, uint64_t *ox, uint64_t *oinfo, uint64_t out_cap, int *err)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_sel) return;
	const uint64_t rid = stamp_rid ? ((rid_lo + r) & 0xfffffffULL) : 0ULL;
	// The read's units (~14 of ~40 minimizers each).  Unit by unit - three scalars, then the unit's entries, then the stores - the loop ran at two memory round trips per
	// unit; here the units' scalars are fetched by one lane each, and four units' entries are requested before the first is stored.
	const uint64_t c0 = chunk_off[r], c1 = chunk_off[r + 1]; const uint32_t lane = (uint32_t)hao_lane();
	for (uint64_t cb = c0; cb < c1; cb += 64) {
		const uint64_t ch = cb + lane; uint64_t b = 0, d = 0; uint32_t n = 0;
		if (ch < c1) { b = chunk_base[ch]; d = chunk_dst[ch]; n = chunk_cnt[ch]; }
		if (__any(n && d + n > out_cap)) { if (lane == 0) *err = 1; return; }
		const int m = (int)(c1 - cb < 64 ? c1 - cb : 64);
		for (int j0 = 0; j0 < m; j0 += 4) {
			uint64_t x[4], inf[4], bb[4], dd[4]; uint32_t nn[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int j = j0 + u < m ? j0 + u : m - 1;      // (a round's spare slots repeat the last unit with no entries)
				bb[u] = (uint64_t)hao_readlane_i64((int64_t)b, j); dd[u] = (uint64_t)hao_readlane_i64((int64_t)d, j); nn[u] = j0 + u < m ? hao_bcast(n, j) : 0u;
				if (lane < nn[u]) { x[u] = pool_x[bb[u] + lane]; inf[u] = pool_info[bb[u] + lane]; }
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				if (lane < nn[u]) { ox[dd[u] + lane] = x[u]; oinfo[dd[u] + lane] = (inf[u] & ~0xfffffffULL) | rid; }
				for (uint32_t i = 64 + lane; i < nn[u]; i += 64) { ox[dd[u] + i] = pool_x[bb[u] + i]; oinfo[dd[u] + i] = (pool_info[bb[u] + i] & ~0xfffffffULL) | rid; }      // (a unit with more than 64 minimizers: degenerate reads)
			}
		}
	}
}

// per-read list bounds from the chunk scan: mz_off[r] = chunk_dst[chunk_off[r]]
__global__ void sketch_read_off_kernel(const uint64_t *chunk_off, const uint64_t *chunk_dst, uint64_t n_sel, uint64_t *mz_off)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	mz_off[r] = chunk_dst[chunk_off[r]];          // chunk_dst has one entry past the last chunk (= total)
}

// ---------------------------------------------------------------------------------------
// High-count thinning (mz1_select_mz_h, sketch.cpp:247-330; mz1_hf_select :194-216).
// Sequential per read on a few hundred entries; one lane per read, in place on the gathered
// list.  info.rid carries the filter-table count (0 = not a high-count k-mer).  new_n[r] <- kept.
// ---------------------------------------------------------------------------------------
struct hao_sel_view { uint64_t *x, *info; uint32_t *ord; int n; };
#define SV_CNT(v, i) hao_info_rid((v).info[i])
#define SV_HIGH(v, i) ((i) >= 0 && SV_CNT(v, i) > 0)
#define SV_MARK 0x80000000u

__device__ __forceinline__ int hao_sel_cmp(const hao_sel_view &v, int ai, int bi)     // mz1_mzcmp_l (sketch.cpp:217-225)
{
	if (ai >= 0 && bi >= 0) {
		uint32_t ca = SV_CNT(v, ai), cb = SV_CNT(v, bi);
		if (ca > 0 && cb > 0) { if (ca != cb) return ca < cb ? -1 : 1; return (v.x[ai] > v.x[bi]) - (v.x[ai] < v.x[bi]); }
		return (int)(ca == 0) - (int)(cb == 0);
	}
	return (int)(ai < 0) - (int)(bi < 0);
}
__device__ __forceinline__ uint32_t hao_sel_ord(const hao_sel_view &v, int i) { return v.ord[i] & ~SV_MARK; }
__device__ __forceinline__ void hao_sel_zero(hao_sel_view &v, int i) { v.info[i] &= ~0xfffffffULL; }

__device__ void hao_sel_rescan(hao_sel_view &v, int si, int i, int *mi)
{
	int m; *mi = -1;
	for (m = si; m <= i; ++m) if (hao_sel_cmp(v, *mi, m) >= 0) *mi = m;
	if (SV_HIGH(v, *mi)) for (m = si; m <= i; ++m) if (SV_HIGH(v, m) && hao_sel_cmp(v, *mi, m) == 0) v.ord[m] |= SV_MARK;
}

struct hao_hent { uint64_t x; uint32_t c; int idx; };
__device__ __forceinline__ bool hao_hent_lt(const hao_hent &a, const hao_hent &b) { return a.c < b.c || (a.c == b.c && a.x < b.x); }
__device__ void hao_heap_down(int i, int n, hao_hent *l)                              // ksort.h:43-52 semantics (max-heap on (count,hash))
{
	int k = i; hao_hent tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && hao_hent_lt(l[k], l[k + 1])) ++k;
		if (hao_hent_lt(l[k], tmp)) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

__device__ void hao_sel_heap(hao_sel_view &v, int si, int ei, int n, int len, int sample_dist)   // mz1_hf_select
{
	if (ei - si <= 1) return;
	int ps = si < 0 ? 0 : (int)hao_info_pos(v.info[si]), pe = ei == n ? len : (int)hao_info_pos(v.info[ei]);
	int q = (int)((double)(pe - ps) / sample_dist + .499), j, kk;
	if (q > 16) q = 16;
	hao_hent b[16];
	for (j = si + 1, kk = 0; j < ei && kk < q; ++j, ++kk) { b[kk].x = v.x[j]; b[kk].c = SV_CNT(v, j); b[kk].idx = j; }
	for (int i = (kk >> 1) - 1; i >= 0; --i) hao_heap_down(i, kk, b);
	for (; j < ei; ++j) { hao_hent e; e.x = v.x[j]; e.c = SV_CNT(v, j); e.idx = j; if (hao_hent_lt(e, b[0])) { b[0] = e; hao_heap_down(0, kk, b); } }
	for (j = 0; j < kk; ++j) if ((int)b[j].c < pe - ps) hao_sel_zero(v, b[j].idx);
}

__device__ int hao_select_high(hao_sel_view v, int len, int sample_dist, int w /*rewin*/, int k, int tot_l)
{
	int n = v.n, i, m, mi = -1, si, last0, any = 0, ws = w + k - 1;
	if (n == 0) return 0;
	for (i = 0, last0 = -1; i <= n; ++i) {
		if (i == n || SV_CNT(v, i) == 0) {
			if (i - last0 > 1) {
				int ps = last0 < 0 ? 0 : (int)hao_info_pos(v.info[last0]), pe = i == n ? len : (int)hao_info_pos(v.info[i]);
				if ((int)((double)(pe - ps) / sample_dist + .499) > 0) { any = 1; break; }
			}
			last0 = i;
		}
	}
	if (!any) return n;
	for (i = 0; i < n; ++i) {
		int oi = (int)hao_sel_ord(v, i);
		if (oi >= ws || (i + 1 < n && oi < ws && (int)hao_sel_ord(v, i + 1) > ws) || (i + 1 == n && tot_l >= ws && oi < ws)) {
			for (m = 0; m <= i; ++m) if (SV_HIGH(v, m) && hao_sel_cmp(v, mi, m) >= 0) mi = m;
			if (mi >= 0 && SV_HIGH(v, mi)) for (m = 0; m <= i; ++m) if (SV_HIGH(v, m) && hao_sel_cmp(v, mi, m) == 0) v.ord[m] |= SV_MARK;
			break;
		}
	}
	if (i < n) {
		for (si = 0, ++i; i < n; ++i) {
			for (; si < i; ++si) if ((int)hao_sel_ord(v, si) + w > (int)hao_sel_ord(v, i)) break;
			if (hao_sel_cmp(v, i, mi) <= 0) { if (SV_HIGH(v, mi)) v.ord[mi] |= SV_MARK; mi = i; }
			else if (si > mi) { if (SV_HIGH(v, mi)) v.ord[mi] |= SV_MARK; hao_sel_rescan(v, si, i, &mi); }
		}
		if (SV_HIGH(v, mi)) v.ord[mi] |= SV_MARK;
		for (i = n - 1; si < n && (int)hao_sel_ord(v, si) + w <= tot_l + 1; ++si)
			if (si > mi) { if (SV_HIGH(v, mi)) v.ord[mi] |= SV_MARK; hao_sel_rescan(v, si, i, &mi); }
		for (i = 0, last0 = -1; i <= n; ++i) {
			if (i == n || SV_CNT(v, i) == 0) {
				if (i - last0 > 1) {
					int ps = last0 < 0 ? 0 : (int)hao_info_pos(v.info[last0]), pe = i == n ? len : (int)hao_info_pos(v.info[i]);
					if ((int)((double)(pe - ps) / sample_dist + .499) > 0) {
						int nm = 0;
						for (m = last0 + 1; m < i; ++m) if (v.ord[m] & SV_MARK) { hao_sel_zero(v, m); ++nm; }
						if (nm == 0) hao_sel_heap(v, last0, i, n, len, sample_dist);
					}
				}
				last0 = i;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i) if (SV_CNT(v, i) == 0) { v.x[m] = v.x[i]; v.info[m] = v.info[i]; v.ord[m] = v.ord[i]; ++m; }
	return m;
}

// final: compact the per-read lists (after thinning) and stamp the read id into info.rid (sketch.cpp:577-578)
__global__ __launch_bounds__(256) void sketch_finish_kernel(const uint64_t *x, const uint64_t *info, const uint64_t *src_off, const uint64_t *dst_off,
		uint64_t rid_lo, uint64_t n_sel, int stamp_rid, uint64_t *ox, uint64_t *oinfo, const int *err)
{
	if (*err) return;
	uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_sel) return;
	uint64_t s = src_off[r], d = dst_off[r], n = dst_off[r + 1] - d;
	for (uint64_t i = hao_lane(); i < n; i += 64) {
		ox[d + i] = x[s + i];
		oinfo[d + i] = (info[s + i] & ~0xfffffffULL) | (stamp_rid ? ((rid_lo + r) & 0xfffffffULL) : 0ULL);
	}
}

// ---------------------------------------------------------------------------------------
// Exact scalar state machine (sketch.cpp:454-573 restated), one lane per flagged read:
// N bases, even k (strand-symmetric k-mer skip), anything the chunk kernel does not cover.
// Emits into the same pool / chunk record as the chunk kernel (one record per read).
// ---------------------------------------------------------------------------------------
struct hao_cand { uint64_t x; uint32_t c, pos; uint8_t rev, span; };
__device__ __forceinline__ int hao_cand_cmp(const hao_cand &a, const hao_cand &b)
{ if (a.c != b.c) return a.c < b.c ? -1 : 1; return (a.x > b.x) - (a.x < b.x); }

struct hao_scalar_args {
	const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len; const uint64_t *nsite_off; const uint32_t *nsite;
	const uint64_t *chunk_off; const uint8_t *scalar_flag; const uint32_t *scalar_list; uint32_t n_scalar;
	uint64_t rid_lo; int k, w, hpc, use_ft; hao_ft_dev ft;
	hao_cand *ring_ws; uint32_t *ringord_ws;      // n_scalar * 256 entries of workspace
	uint64_t *pool_x, *pool_info; uint32_t *pool_ord; unsigned long long *pool_cursor; uint64_t pool_cap;
	uint64_t *chunk_base; uint32_t *chunk_cnt; uint32_t *tot_l; int *err;
	int pass;   // 0 = count only, 1 = emit
	uint32_t *cnt_ws; uint64_t pool_static;
};

__global__ void sketch_scalar_kernel(hao_scalar_args a)
{
	uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
	if (si >= a.n_scalar) return;
	const uint32_t r = a.scalar_list[si]; const uint64_t rid = a.rid_lo + r;
	const uint8_t *rd = a.packed + a.pk_off[rid]; const int len = (int)a.len[rid], w = a.w, k = a.k;
	const uint32_t *ns = a.nsite ? a.nsite + a.nsite_off[rid] : nullptr; const uint32_t nn = a.nsite ? (uint32_t)(a.nsite_off[rid + 1] - a.nsite_off[rid]) : 0; uint32_t np = 0;
	hao_cand *ring = a.ring_ws + (size_t)si * 256; uint32_t *rord = a.ringord_ws + (size_t)si * 256;
	const hao_cand dummy = { UINT64_MAX, HAO_CNT_DUMMY, 0, 0, 0 };
	uint64_t mask = (1ULL << k) - 1, pl[4] = {0, 0, 0, 0}; const int sh = k - 1;
	hao_cand mn = dummy; uint32_t mn_ord = (uint32_t)-1;
	int qrun[64], qf = 0, qc = 0, span = 0, l = 0, tl = 0, bp = 0, mbp = 0, j;
	uint64_t out = 0, base = 0; bool emit = a.pass == 1;
	if (emit) { base = a.chunk_base[a.chunk_off[r]]; if (a.chunk_cnt[a.chunk_off[r]] == 0) return; }
	for (j = 0; j < w; ++j) { ring[j].x = UINT64_MAX; ring[j].c = HAO_CNT_DUMMY; ring[j].pos = (1u << 27) - 1; ring[j].rev = 1; ring[j].span = 255; }
#define SC_PUSH(e, o) do { if (emit) { a.pool_x[base + out] = (e).x; a.pool_info[base + out] = hao_info_pack((e).c, (e).pos, (e).rev, (e).span); a.pool_ord[base + out] = (o); } ++out; } while (0)
	for (int i = 0; i < len; ++i) {
		while (np < nn && ns[np] < (uint32_t)i) ++np;
		int b = (np < nn && ns[np] == (uint32_t)i) ? 4 : (int)hao_base_at(rd, i);
		hao_cand info = dummy; bool skip = false;
		if (b < 4) {
			if (a.hpc) {
				int run = 1; uint32_t npp = np;
				while (i + run < len) {
					while (npp < nn && ns[npp] < (uint32_t)(i + run)) ++npp;
					if (npp < nn && ns[npp] == (uint32_t)(i + run)) break;
					if ((int)hao_base_at(rd, i + run) != b) break;
					++run;
				}
				i += run - 1;
				qrun[(qc++ + qf) & 63] = run; span += run;
				if (qc > k) { span -= qrun[qf]; qf = (qf + 1) & 63; --qc; }
			} else span = l + 1 < k ? l + 1 : k;
			pl[0] = (pl[0] << 1 | (uint64_t)(b & 1)) & mask; pl[1] = (pl[1] << 1 | (uint64_t)(b >> 1)) & mask;
			pl[2] = pl[2] >> 1 | (uint64_t)(1 - (b & 1)) << sh; pl[3] = pl[3] >> 1 | (uint64_t)(1 - (b >> 1)) << sh;
			if (pl[1] == pl[3]) skip = true;
			else {
				int z = pl[1] < pl[3] ? 0 : 1;
				++l; ++tl;
				if (l >= k && span < 256) {
					uint64_t y = hao_hash64(pl[z << 1]) + hao_hash64(pl[z << 1 | 1]);
					int32_t cnt = a.use_ft ? hao_ft_lookup(a.ft, y) : 0;
					if (cnt < (1 << 28)) { info.x = y; info.c = (uint32_t)cnt; info.pos = (uint32_t)i; info.rev = (uint8_t)z; info.span = (uint8_t)span; }
				}
			}
		} else { l = 0; qc = qf = 0; span = 0; }
		if (skip) continue;
		ring[bp] = info; rord[bp] = (uint32_t)l;
		if (l == w + k - 1 && mn.x != UINT64_MAX) {
			for (j = bp + 1; j < w; ++j) if (hao_cand_cmp(mn, ring[j]) == 0 && ring[j].pos != mn.pos) SC_PUSH(ring[j], rord[j]);
			for (j = 0; j < bp; ++j) if (hao_cand_cmp(mn, ring[j]) == 0 && ring[j].pos != mn.pos) SC_PUSH(ring[j], rord[j]);
		}
		if (hao_cand_cmp(mn, info) >= 0) {
			if (l >= w + k && mn.x != UINT64_MAX) SC_PUSH(mn, mn_ord);
			mn = info; mbp = bp; mn_ord = rord[bp];
		} else if (bp == mbp) {
			if (l >= w + k - 1 && mn.x != UINT64_MAX) SC_PUSH(mn, mn_ord);
			mn = dummy;
			for (j = bp + 1; j < w; ++j) if (hao_cand_cmp(mn, ring[j]) >= 0) { mn = ring[j]; mbp = j; mn_ord = rord[j]; }
			for (j = 0; j <= bp; ++j) if (hao_cand_cmp(mn, ring[j]) >= 0) { mn = ring[j]; mbp = j; mn_ord = rord[j]; }
			if (l >= w + k - 1 && mn.x != UINT64_MAX) {
				for (j = bp + 1; j < w; ++j) if (hao_cand_cmp(mn, ring[j]) == 0 && mn.pos != ring[j].pos) SC_PUSH(ring[j], rord[j]);
				for (j = 0; j <= bp; ++j) if (hao_cand_cmp(mn, ring[j]) == 0 && mn.pos != ring[j].pos) SC_PUSH(ring[j], rord[j]);
			}
		}
		if (++bp == w) bp = 0;
	}
	if (mn.x != UINT64_MAX) SC_PUSH(mn, mn_ord);
#undef SC_PUSH
	if (!emit) {
		a.cnt_ws[si] = (uint32_t)out; a.tot_l[r] = (uint32_t)tl;
		unsigned long long bs = out ? a.pool_static + atomicAdd(a.pool_cursor, (unsigned long long)out) : 0ULL;
		if (bs + out > a.pool_cap) { *a.err = 1; out = 0; }
		a.chunk_base[a.chunk_off[r]] = bs; a.chunk_cnt[a.chunk_off[r]] = (uint32_t)out;
	}
}
