// K1/K2 (all-k-mer counting -> histogram -> high-count filter table) and K4 (minimizer
// count + position index) for gfx950.
//
// Both tables are built by sort + run-length instead of the reference's 4096 sharded
// khashl tables (htab.cpp:122-214, 299-460): at -f0 the result is order-independent
// (exact counts saturating at 4095), and a stable sort of minimizers that were
// produced in read order reproduces the reference's per-key (rid,pos) insertion order
// (kt_pipeline ordering).
#pragma once
#include "hao_common.cuh"
#include "hao_sketch.cuh"

#define HAO_KH_CHUNK 2048

// One workgroup per (read, chunk of 2048 k-mer ordinals): every HPC k-mer hash of an N-free read
// (mz1_count_seq_buf_HPC / _count_seq_buf, htab.cpp:608-645; hash = yak_hash_long htab.h:161-166).
// out[kmer_off[r] + (t - k)] for ordinal t in [k, T].
struct hao_kh_args {
	const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len;
	const uint64_t *tile_off; const uint32_t *tile_ord; const uint32_t *n_runs;
	const uint64_t *chunk_off; const uint8_t *scalar_flag; const uint64_t *kmer_off;
	uint64_t rid_lo, n_sel; int k, hpc; uint64_t *out;
	uint64_t ch0;      // first workgroup of this launch in chunk_off's numbering (ha_ft_gen in passes hashes a range of reads per launch)
};

__global__ __launch_bounds__(256) void kmer_hash_chunk_kernel(hao_kh_args a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	const int k = a.k, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int NE = HAO_KH_CHUNK + k + 1, NW = (NE + 63) / 64 + 1;
	uint64_t *pl0 = (uint64_t*)smem, *pl1 = pl0 + NW; uint8_t *rcode = (uint8_t*)(pl1 + NW);
	uint64_t ch = blockIdx.x + a.ch0, lo = 0, hi = a.n_sel;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a.chunk_off[m + 1] <= ch) lo = m + 1; else hi = m; }
	const uint64_t r = lo;
	if (a.scalar_flag[r]) return;
	const uint32_t ci = (uint32_t)(ch - a.chunk_off[r]);
	const uint64_t rid = a.rid_lo + r; const uint8_t *rd = a.packed + a.pk_off[rid]; const uint32_t L = a.len[rid];
	const int T = (int)a.n_runs[r];
	const uint32_t *tord = a.tile_ord + a.tile_off[r]; const uint32_t ntile = (L + HAO_SK_TILE - 1) / HAO_SK_TILE;
	const int kk0 = k + (int)ci * HAO_KH_CHUNK, kk1 = min(T, kk0 + HAO_KH_CHUNK - 1);
	if (kk0 > T) return;
	const int rbase = kk0 - k, nE = kk1 - rbase + 1, nK = kk1 - kk0 + 1;
	const uint64_t mask = (1ULL << k) - 1;
	{
		const uint32_t first = rbase > 0 ? (uint32_t)rbase : 1u;
		uint32_t tlo = 0, thi = ntile;
		while (thi - tlo > 1) { uint32_t m = (tlo + thi) >> 1; if (tord[m] < first) tlo = m; else thi = m; }
		for (uint32_t ti = tlo + wv; ti < ntile && tord[ti] < (uint32_t)kk1; ti += 4) {
			uint32_t W, g0 = ti * HAO_SK_TILE + lane * 16, tot;
			uint32_t eb = hao_run_ends16(rd, L, g0, &W);
			if (!a.hpc) { uint32_t rem = g0 >= L ? 0 : (L - g0 > 16 ? 16 : L - g0); eb = rem == 0 ? 0 : (rem >= 16 ? 0x55555555u : ((0xFFFFFFFFu << (32 - 2 * rem)) & 0x55555555u)); }
			uint32_t o = tord[ti] + hao_wave_excl_scan(__popc(eb), &tot) + 1;
			while (eb) {
				int hb = 31 - __clz(eb); int j = (30 - hb) >> 1; eb &= ~(1u << hb);
				if (o >= first && o <= (uint32_t)kk1) rcode[(int)o - rbase] = (W >> (30 - 2 * j)) & 3;
				++o;
			}
		}
	}
	__syncthreads();
	for (int e0 = wv * 64; e0 < nE + 64; e0 += 256) {
		int e = e0 + lane; uint32_t c = (e >= 1 && e < nE) ? rcode[e] : 0;
		unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c >> 1);
		if (lane == 0 && (e0 >> 6) < NW) { pl0[e0 >> 6] = b0; pl1[e0 >> 6] = b1; }
	}
	__syncthreads();
	uint64_t *dst = a.out + a.kmer_off[r] + (uint64_t)(kk0 - k);
	for (int q = tid; q < nK; q += 256) {
		int s = q + 1, wi = s >> 6, sh = s & 63;
		uint64_t W0 = pl0[wi] >> sh, W1 = pl1[wi] >> sh;
		if (sh) { W0 |= pl0[wi + 1] << (64 - sh); W1 |= pl1[wi + 1] << (64 - sh); }
		W0 &= mask; W1 &= mask;
		uint64_t f1 = __brevll(W1) >> (64 - k), r1 = ~W1 & mask;
		dst[q] = f1 < r1 ? hao_hash64(__brevll(W0) >> (64 - k)) + hao_hash64(f1) : hao_hash64(~W0 & mask) + hao_hash64(r1);
	}
}

static inline size_t hao_kh_smem_bytes(int k) { size_t NE = HAO_KH_CHUNK + k + 1, NW = (NE + 63) / 64 + 1; return 2 * NW * 8 + NE + 64; }

// reads with N: exact scalar walk, one lane per read; slot = len entries pre-filled with the sentinel
__global__ void kmer_hash_scalar_kernel(const uint8_t *packed, const uint64_t *pk_off, const uint32_t *len, const uint64_t *nsite_off, const uint32_t *nsite,
		const uint32_t *scalar_list, uint32_t n_scalar, const uint64_t *kmer_off, uint64_t rid_lo, int k, int hpc, uint64_t *out, unsigned long long *n_real)
{
	uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
	if (si >= n_scalar) return;
	const uint32_t r = scalar_list[si]; const uint64_t rid = rid_lo + r; const uint8_t *rd = packed + pk_off[rid]; const int L = (int)len[rid];
	const uint32_t *ns = nsite + nsite_off[rid]; const uint32_t nn = (uint32_t)(nsite_off[rid + 1] - nsite_off[rid]); uint32_t np = 0;
	uint64_t pl[4] = {0, 0, 0, 0}, mask = (1ULL << k) - 1, *dst = out + kmer_off[r], n = 0; int sh = k - 1, last = -1, l = 0;
	for (int i = 0; i < L; ++i) {
		while (np < nn && ns[np] < (uint32_t)i) ++np;
		int b = (np < nn && ns[np] == (uint32_t)i) ? 4 : (int)hao_base_at(rd, i);
		if (b >= 4) { l = 0; last = -1; pl[0] = pl[1] = pl[2] = pl[3] = 0; continue; }
		if (hpc && b == last) continue;
		pl[0] = (pl[0] << 1 | (uint64_t)(b & 1)) & mask; pl[1] = (pl[1] << 1 | (uint64_t)(b >> 1)) & mask;
		pl[2] = pl[2] >> 1 | (uint64_t)(1 - (b & 1)) << sh; pl[3] = pl[3] >> 1 | (uint64_t)(1 - (b >> 1)) << sh;
		last = b;
		if (++l >= k) { int j = pl[1] < pl[3] ? 0 : 1; dst[n++] = hao_hash64(pl[j << 1]) + hao_hash64(pl[j << 1 | 1]); }
	}
	atomicAdd(n_real, (unsigned long long)n);
}

__global__ void hao_kmer_slots_kernel(const uint32_t *n_runs, const uint8_t *scalar_flag, const uint32_t *len, uint64_t rid_lo, uint64_t n_sel, int k,
		uint64_t *slots, uint64_t *chunks)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	if (r == n_sel) { slots[r] = 0; chunks[r] = 0; return; }
	uint64_t nk = n_runs[r] >= (uint32_t)k ? n_runs[r] - k + 1 : 0;
	slots[r] = scalar_flag[r] ? len[rid_lo + r] : nk;
	chunks[r] = scalar_flag[r] ? 0 : (nk + HAO_KH_CHUNK - 1) / HAO_KH_CHUNK;
}

// histogram of min(count, 4095) over distinct keys (ha_ct_hist, htab.cpp:240-254)
__global__ __launch_bounds__(256) void hao_count_hist_kernel(const uint32_t *cnt, uint64_t n, unsigned long long *hist)
{
	__shared__ unsigned int h[HAO_N_COUNTS];
	for (int i = threadIdx.x; i < HAO_N_COUNTS; i += 256) h[i] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { uint32_t c = cnt[i]; atomicAdd(&h[c > HAO_MAX_COUNT ? HAO_MAX_COUNT : c], 1u); }
	__syncthreads();
	for (int i = threadIdx.x; i < HAO_N_COUNTS; i += 256) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// flag[i] = 1 iff lo <= min(cnt,4095) <= hi
__global__ void hao_range_flag_kernel(const uint32_t *cnt, uint64_t n, int lo, int hi, uint64_t *flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i == n) { flag[i] = 0; return; }
	int c = cnt[i] > HAO_MAX_COUNT ? HAO_MAX_COUNT : (int)cnt[i];
	flag[i] = (c >= lo && c <= hi) ? 1 : 0;
}

// scatter the kept runs: keys, start offset in the sorted array, count
__global__ void hao_keep_scatter_kernel(const uint64_t *ukeys, const uint32_t *ucnt, const uint64_t *ustart, const uint64_t *flag, const uint64_t *kpos, uint64_t n,
		uint64_t *keys, uint64_t *start, uint32_t *cnt)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || !flag[i]) return;
	uint64_t d = kpos[i];
	keys[d] = ukeys[i]; if (start) start[d] = ustart[i]; cnt[d] = ucnt[i] > HAO_MAX_COUNT ? HAO_MAX_COUNT : ucnt[i];
}

// bucket[b] = first index with (key >> shift) >= b, b = 0 .. nb (nb = 1 << bits)
__global__ void hao_bucket_kernel(const uint64_t *keys, uint64_t n, int shift, uint32_t nb, uint32_t *bucket)
{
	uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > nb) return;
	if (b == nb) { bucket[b] = (uint32_t)n; return; }
	uint64_t lo = 0, hi = n;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if ((keys[m] >> shift) < b) lo = m + 1; else hi = m; }
	bucket[b] = (uint32_t)lo;
}

// ---- index build, single device: the lookup of every minimizer is a by-product of the sort ----
// The query pass looks every minimizer of every read up in the index (ha_pt_get, anchor.cpp:1013) - but the index was just built from those same
// minimizers: after the stable sort by hash, position j of run u knows its key's list (start of the run, run length).  Carrying the original
// (read-order) index through the sort as the 4-byte value turns 215 M dependent bucket + binary searches per pass (11.5 GB of scattered line
// fetches per batch of configs[2]) into one scatter at build time: lk[orig] = list start | count << 48 (count 0 = the key is not indexed).
__global__ void hao_iota_kernel(uint32_t *v, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = (uint32_t)i;
}
// ---- the index sort on 40 of the 64 hash bits ----
// The (hash, read-order index) sort is LSD radix, 8 bits per pass: 8 passes of 24 bytes per element.  Two minimizers whose hashes agree in the top 40
// bits but not below are rare (n^2 / 2^41 pairs of distinct keys: a few hundred on a 250 Mb genome, ~10^5 at human size), so the sort covers bits 24 .. 63
// in 5 passes and the few 40-bit runs that hold more than one key are put right afterwards: hao_sort40_mark_kernel lists the runs that have a descent
// (a run without one is already in full-key order, and stable), hao_sort40_fix_kernel - one wave per listed run - rewrites the run key by key in ascending
// order, each key's elements in their current (= read) order, through a scratch piece.  Same result as the stable 64-bit sort, bit for bit.
#define HAO_SORT40_LOWBITS 24
__global__ void hao_sort40_mark_kernel(const uint64_t *sx, uint64_t m, uint32_t *list, unsigned long long *cnt, uint64_t cap)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	if (j >= m) return;
	const uint64_t b = sx[j], a = sx[j - 1], top = b >> HAO_SORT40_LOWBITS;
	if ((a >> HAO_SORT40_LOWBITS) != top || a <= b) return;
	uint64_t i = j - 1;      // walk back to the head of the run; an earlier descent reports the run instead of this one
	while (i > 0 && (sx[i - 1] >> HAO_SORT40_LOWBITS) == top) { if (sx[i - 1] > sx[i]) return; --i; }
	const unsigned long long k = atomicAdd(cnt, 1ULL);
	if (k < cap) list[k] = (uint32_t)i;
}
// cnt[0] = listed runs, cnt[1] = scratch cursor, cnt[2] = 1 if the scratch ran out (the host then sorts all 64 bits)
__global__ __launch_bounds__(64) void hao_sort40_fix_kernel(uint64_t *sx, uint32_t *oi, uint64_t m, const uint32_t *list, unsigned long long *cnt, uint64_t cap, uint64_t *tx, uint32_t *to, uint64_t tcap)
{
	const uint64_t n_runs = cnt[0] < cap ? cnt[0] : cap;
	if (blockIdx.x >= n_runs) return;
	const int lane = hao_lane();
	const uint64_t s = list[blockIdx.x], top = sx[s] >> HAO_SORT40_LOWBITS;
	uint64_t e = s;
	for (;;) {      // end of the run, 64 positions per step
		const uint64_t i = e + lane; const unsigned long long bal = __ballot(i < m && (sx[i] >> HAO_SORT40_LOWBITS) == top);
		if (bal == ~0ULL) { e += 64; continue; }
		e += (uint64_t)(__ffsll((long long)~bal) - 1); break;
	}
	const uint64_t L = e - s;
	unsigned long long base = 0;
	if (lane == 0) base = atomicAdd(cnt + 1, (unsigned long long)L);
	base = (unsigned long long)__shfl((long long)base, 0);
	if (base + L > tcap) { if (lane == 0) cnt[2] = 1; return; }
	uint64_t last = 0, out = 0; bool first = true;
	while (out < L) {
		uint64_t mn = ~0ULL;      // the smallest key not yet written
		for (uint64_t i = s + lane; i < e; i += 64) { const uint64_t k = sx[i]; if ((first || k > last) && k < mn) mn = k; }
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = (uint64_t)__shfl_xor((long long)mn, d); if (o < mn) mn = o; }
		for (uint64_t i0 = s; i0 < e; i0 += 64) {      // its elements, in their current order
			const uint64_t i = i0 + lane; const bool hit = i < e && sx[i] == mn; const unsigned long long bal = __ballot(hit);
			if (hit) { const uint64_t pos = base + out + __popcll(bal & ((1ULL << lane) - 1)); tx[pos] = mn; to[pos] = oi[i]; }
			out += __popcll(bal);
		}
		last = mn; first = false;
	}
	__threadfence();
	for (uint64_t i = lane; i < L; i += 64) { sx[s + i] = tx[base + i]; oi[s + i] = to[base + i]; }
}

// j = position in hash order: gather its 8-byte record (the index lists) and scatter its lookup result to read order
// (sharded build: the owner of a hash range runs this over what it RECEIVED - sidx = arrival index, base = first position of its partition in the
// global index - and the lookup results travel back to the ranks the minimizers came from)
__global__ void hao_index_finish_kernel(uint64_t m, const uint32_t *sidx, const uint32_t *runid, const uint32_t *ucnt, const uint64_t *ustart, int lo, int hi,
		const uint64_t *info, uint64_t *sinfo, uint64_t *lk, uint64_t base)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint32_t o = sidx[j], u = runid[j] - 1;
	sinfo[j] = info[o];
	int c = ucnt[u] > HAO_MAX_COUNT ? HAO_MAX_COUNT : (int)ucnt[u];
	lk[o] = (c >= lo && c <= hi) ? ((ustart[u] + base) | (uint64_t)c << 48) : 0;
}
// The two halves of hao_index_finish_kernel for big indexes: the gather with the lookup results left in hash order (val[j]), then - after ONE radix pass has
// grouped the (read-order index, result) pairs by the top 8 bits of the index - a scatter whose writes of any moment fall into a window of 2^(bits - 8) results
// (8 MB at 215 M minimizers) that the memory-side cache merges into full lines, instead of 8-byte writes all over a 1.7 GB array.
__global__ void hao_index_gather_kernel(uint64_t m, const uint32_t *sidx, const uint32_t *runid, const uint32_t *ucnt, const uint64_t *ustart, int lo, int hi,
		const uint64_t *info, uint64_t *sinfo, uint64_t *val, uint64_t base)
{
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint32_t o = sidx[j], u = runid[j] - 1;
	sinfo[j] = info[o];
	int c = ucnt[u] > HAO_MAX_COUNT ? HAO_MAX_COUNT : (int)ucnt[u];
	val[j] = (c >= lo && c <= hi) ? ((ustart[u] + base) | (uint64_t)c << 48) : 0;
}
// out[idx[i]] = in[i]
__global__ void hao_scatter_u64_kernel(const uint64_t *in, const uint32_t *idx, uint64_t n, uint64_t *out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[idx[i]] = in[i];
}

// position index, device view: kept keys (sorted) -> (start,cnt) into the hash-sorted minimizer array
struct hao_pt_dev {
	const uint64_t *keys, *start; const uint32_t *cnt, *bucket; const uint64_t *sinfo; uint64_t n_keys; int bshift;
};

// ha_pt_get (htab.cpp:518-527)
__device__ __forceinline__ uint32_t hao_pt_lookup(const hao_pt_dev &pt, uint64_t x, uint64_t *start)
{
	uint32_t b = (uint32_t)(x >> pt.bshift), lo = pt.bucket[b], hi = pt.bucket[b + 1], e = hi;
	while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (pt.keys[m] < x) lo = m + 1; else hi = m; }
	if (lo < e && pt.keys[lo] == x) { *start = pt.start[lo]; return pt.cnt[lo]; }
	return 0;
}


// ---------------------------------------------------------------------------------------
// Bloom filter in front of the k-mer count table (ha_ft_gen at -f > 0): yak_bf_insert (htab.cpp:99-116) behind
// ha_ct_insert_list (htab.cpp:181-214).  One filter of 2^(bf_shift-12) bits per sub-table (sub-table = low 12 hash bits), 512-bit
// blocks, 4 probes inside ONE block; an occurrence reaches the count table only if all its probe bits were already set.
// The outcome depends on the order of insertion, which the reference fixes: per sub-table, global (read, position) order.  A block
// is touched only by the k-mers that map to it, so the exact result is a per-block sequential replay: sort the occurrences by
// block id (stable: keeps the (read, position) order inside a block), then one lane replays each block with its 512 bits in LDS.
// The 2^(bf_shift-3)-byte filter itself (16 GB at the default -f37) is never materialised.
// ---------------------------------------------------------------------------------------
// block id of a k-mer hash: sub-table << xb | (hash >> 12) & (2^xb - 1), xb = bf_shift - 21; sentinels (slots of N reads) sort last
// (grid-stride: there are more occurrences than the 2^32 work-items one launch can address - 5.6 G on BASELINE configs[2]; a launch of grid x block
// beyond that silently runs only the remainder modulo 2^32)
__global__ void hao_bf_block_kernel(const uint64_t *kh, uint64_t n, int xb, uint32_t *blk)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t h = kh[i];
		blk[i] = h == UINT64_MAX ? 1u << (12 + xb) : (uint32_t)((h & 4095) << xb | ((h >> 12) & ((1ULL << xb) - 1)));
	}
}

// One lane per block RUN of the block-sorted list (runs from a run-length pass: consecutive lanes take consecutive runs, every lane replays its ~n / 2^(bf_shift-9)
// occurrences - about 20 at the reference's default on 30x human-size input): flag[j] = 1 iff occurrence j found all four probe bits set.  (One lane per
// OCCURRENCE with only the run heads working left three lanes of a wave busy: minutes for the 5.6 G occurrences of BASELINE configs[2].)  flag[] is
// zeroed beforehand: the sentinel run is skipped.
__global__ __launch_bounds__(256) void hao_bf_replay_kernel(const uint32_t *run_blk, const uint32_t *run_len, const uint64_t *run_start, uint64_t n_runs, const uint64_t *kh, int xb, uint8_t *flag)
{
	__shared__ uint32_t st[256][17];
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_runs) return;
	if (run_blk[r] >> (12 + xb)) return;                       // sentinels
	uint32_t *w = st[threadIdx.x];
#pragma unroll
	for (int q = 0; q < 16; ++q) w[q] = 0;
	const int nsh = xb + 9;
	const uint64_t j0 = run_start[r], j1 = j0 + run_len[r];
	for (uint64_t j = j0; j < j1; ++j) {
		const uint64_t x = kh[j] >> 12;
		int h2 = (int)(x >> nsh & 511), z = (int)(x >> xb & 511), cnt = 0;
		if ((h2 & 31) == 0) h2 = (h2 + 1) & 511;
#pragma unroll
		for (int q = 0; q < 4; ++q, z = (z + h2) & 511) { const uint32_t u = 1u << (z & 31); cnt += (w[z >> 5] & u) != 0; w[z >> 5] |= u; }
		flag[j] = cnt == 4;
	}
}
