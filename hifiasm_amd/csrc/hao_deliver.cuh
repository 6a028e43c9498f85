// Result delivery kernels: per-read digests of a batch's results (integrity check across the boundary / full-size parity tests).
#pragma once
#include "hao_common.cuh"
#include "hao_chain.cuh"

// ---------------------------------------------------------------------------------------
// hao_batch_digest (include/hao.h).  Per read r:
//   digest[r]    = sum over the 64-bit words of  ol->list (stream 1, 6 words per overlap_region = the 12 u32 fields of hao_ovlp_t),
//                  the fake cigars in ol order (stream 2) and cl->list (stream 3, 2 words per k_mer_hit)
//   digest_kh[r] = the same over the seed hits before chaining (stream 4)
//   term(stream, i, w) = mix64(w + 0x9E3779B97F4A7C15 * (i + 1) + stream * 0xD6E8FEB86659FD93),   i = index of the word in its stream
// (mod 2^64; mix64 = the splitmix64 finaliser).  The word index is mixed in, so the sum is order-sensitive although it commutes -
// which is what lets a workgroup reduce it in parallel.  oracle/ref_harness.cpp computes the same value from the reference's own
// overlap_region / k_mer_hit structs.
// ---------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t hao_dg_mix(uint64_t z)
{ z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL; z ^= z >> 27; z *= 0x94d049bb133111ebULL; z ^= z >> 31; return z; }
__host__ __device__ __forceinline__ uint64_t hao_dg_term(uint64_t stream, uint64_t i, uint64_t w)
{ return hao_dg_mix(w + 0x9E3779B97F4A7C15ULL * (i + 1) + stream * 0xD6E8FEB86659FD93ULL); }

struct hao_digest_args {
	const uint64_t *fin_off, *fcf_off, *g_off, *cl_base, *seg;      // per read: ol range, fake-cigar range, group range (-> cl range), seed-hit range
	const uint64_t *ol, *fc, *cl, *hits;                            // viewed as 64-bit words
	uint64_t n_sel; uint64_t *dig, *dig_kh;
};

__device__ __forceinline__ uint64_t hao_dg_range(uint64_t stream, const uint64_t *w, uint64_t n_words, uint32_t tid)
{
	uint64_t s = 0;
	for (uint64_t i = tid; i < n_words; i += 256) s += hao_dg_term(stream, i, w[i]);
	return s;
}

__global__ __launch_bounds__(256) void hao_digest_kernel(hao_digest_args A)
{
	const uint64_t r = blockIdx.x; const uint32_t tid = threadIdx.x;
	__shared__ uint64_t part[2][4];
	const uint64_t o0 = A.fin_off[r], o1 = A.fin_off[r + 1], f0 = A.fcf_off[r], f1 = A.fcf_off[r + 1];
	const uint64_t c0 = A.cl_base[A.g_off[r]], c1 = A.cl_base[A.g_off[r + 1]], h0 = A.seg[r], h1 = A.seg[r + 1];
	uint64_t d = hao_dg_range(1, A.ol + o0 * 6, (o1 - o0) * 6, tid) + hao_dg_range(2, A.fc + f0, f1 - f0, tid) + hao_dg_range(3, A.cl + c0 * 2, (c1 - c0) * 2, tid);
	uint64_t k = A.dig_kh ? hao_dg_range(4, A.hits + h0 * 2, (h1 - h0) * 2, tid) : 0;
#pragma unroll
	for (int dl = 32; dl >= 1; dl >>= 1) { d += __shfl_xor(d, dl); k += __shfl_xor(k, dl); }
	if ((tid & 63) == 0) { part[0][tid >> 6] = d; part[1][tid >> 6] = k; }
	__syncthreads();
	if (tid == 0) {
		A.dig[r] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
		if (A.dig_kh) A.dig_kh[r] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
	}
}

// ---------------------------------------------------------------------------------------
// Wire format of cl->list (include/hao.h: hao_chain_hdr_t, hao_qmz_t, hao_unpack_hits).
// A chained hit is a pair (query minimizer, position on the target): self_offset and cnt (seed weight << 8 | span) are properties of the QUERY
// minimizer alone (anchor.cpp:1065-1076), and along a chain the target offset follows the query offset up to a small diagonal shift.  So the
// batch ships, per read, its minimizer table (self_offset, cnt: 8 bytes per minimizer, ~430 per 15 kb read) once, and per chained hit a code:
//     high nibble = (minimizers skipped since the previous hit of the chain) = dq - 1     (0 .. 14)
//     low nibble  = (target offset delta) - (self_offset delta) + 8                       (diagonal shift -8 .. 7)
// 0xff = the hit is in the batch's exception list (verbatim, with its minimizer index), keyed by the hit's index in the batch and sorted.
// More than nine hits in ten have the code 0x08 (next minimizer, same diagonal), so the codes do not travel as a byte per hit: the batch ships
// one BIT per hit (1 = this hit has a code byte), a rank directory (code bytes before every 64th hit) and the code bytes of the flagged hits
// (hao_pack_bits_kernel, hao_pack_codes_kernel below): ~0.3 bytes per chained hit across PCIe instead of 16.
// The first hit of a chain comes from the chain header (minimizer index, target offset).  ~1.3 bytes per chained hit across PCIe instead
// of 16; the consumer thread decodes straight into its Candidates_list.  One wave per chain, reading the hits where the chain kernels left
// them (chain descriptors): cl->list is never materialised in HBM on this path.  The minimizer index of a hit is recovered by a binary
// search of its self_offset in the read's table (a few hundred L1/L2-resident entries).
// ---------------------------------------------------------------------------------------
#define HAO_PACK_T 2048          // hits of cl->list per workgroup of the packer (8 per thread)
#define HAO_PACK_CMAX (HAO_PACK_T + 1 + 256)
struct hao_pack_args {
	hao_cdesc *cd; const hao_hit_t *hits, *ohits;      // (the header kernel leaves two flags in the descriptors' pad word for the packer)
	const uint64_t *mz_off; uint64_t rid_lo, mz0; const uint32_t *q_pos;      // per-read minimizer ranges (global offsets) and the batch's self_offset table
	hao_chain_hdr_t *hdr; uint8_t *bytes; hao_exc_t *exc; unsigned long long *exc_cnt; uint64_t exc_cap; uint32_t exc_every;
	const uint16_t *hq; const uint8_t *hcode;      // per seed hit: query minimizer index (seed kernel), wire code (chain_group_kernel); null: every hit's minimizer is searched
	uint32_t *blk_first; uint64_t n_blk;           // first chain of every HAO_PACK_T-hit piece of cl->list
	uint64_t *bits; uint32_t *cnt; uint64_t n_words_max;
};

// minimizer index of a hit: the table position of its self_offset (positions are strictly ascending in a read's table)
__device__ __forceinline__ uint32_t hao_pack_find_q(const uint32_t *qp, uint32_t nq, uint32_t self_offset)
{ uint32_t lo = 0, hi = nq; while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (qp[m] < self_offset) lo = m + 1; else hi = m; } return lo; }

// One LANE per chain: the chain header (the first hit: minimizer index, target offset) and, for every HAO_PACK_T-hit piece of cl->list whose first
// position the chain holds, the piece's first chain.  (The chains tile cl->list without gaps, in order.)  Every chain is independent here: the dependent
// loads of a chain (descriptor -> read's minimizer range -> first hit) overlap across 64 chains per wave instead of being paid once per chain and wave.
__global__ __launch_bounds__(256) void hao_pack_hdr_kernel(hao_pack_args A, const uint64_t *n_chains_dev)
{
	const uint64_t ci = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (ci >= *n_chains_dev) return;
	const hao_cdesc d = A.cd[ci];
	const hao_hit_t *src = hao_cd_src(d, A.hits, A.ohits);
	const uint64_t m0 = A.mz_off[A.rid_lo + d.r]; const uint32_t nq = (uint32_t)(A.mz_off[A.rid_lo + d.r + 1] - m0);
	const bool in_place = !(d.src & HAO_CD_OHITS);
	const hao_hit_t h0 = src[0];
	hao_chain_hdr_t H; H.n_hits = d.n; H.w0 = d.w0; H.offset = h0.offset;
	H.q0 = (A.hq && in_place && nq < 65535u) ? A.hq[d.src] : hao_pack_find_q(A.q_pos + (m0 - A.mz0), nq, h0.self_offset);
	A.hdr[ci] = H;
	// what the packer may use for this chain's hits - pad bit 1: their wire codes (bit 0 = the quick check wrote them; the 16-bit minimizer indices behind them
	// must not have saturated), bit 2: their minimizer indices
	const uint32_t fl = (A.hcode && (d.pad & 1u) && nq < 65535u) ? 2u : (A.hq && in_place && nq < 65535u) ? 4u : 0u;
	A.cd[ci].pad = (d.pad & 1u) | fl;
	for (uint64_t b = (d.dst + HAO_PACK_T - 1) / HAO_PACK_T; b * HAO_PACK_T < d.dst + d.n && b < A.n_blk; ++b) A.blk_first[b] = (uint32_t)ci;
}

// cl->list -> one code byte per hit + the bit stream (1 = the hit has a code byte) + the per-word counts of the rank directory, in ONE pass over the
// POSITIONS of cl->list: a workgroup takes HAO_PACK_T consecutive positions (thread t: 8 of them = one aligned 8-byte word of the code array), finds
// their chains in an LDS copy of the piece's chain table and gathers the codes the quick check left next to the seed hits (1 byte per hit; chains whose
// hits have no codes - tiny groups, the DP path - compute them from the hits).  Work is spread by position, not by chain: a 100 000-hit chain of a
// tandem array and a 1-hit chain cost the same per hit, and no wave waits on one chain's dependent loads.
__global__ __launch_bounds__(256) void hao_pack_flat_kernel(hao_pack_args A, const uint64_t *n_chains_dev, const uint64_t *n_cl_dev)
{
	__shared__ int32_t c_dst[HAO_PACK_CMAX]; __shared__ uint32_t c_src[HAO_PACK_CMAX]; __shared__ uint8_t c_fl[HAO_PACK_CMAX]; __shared__ int64_t s_dst0;      // starts relative to the piece (only its first chain can start before it: s_dst0)
	const uint64_t n_chains = *n_chains_dev, n_cl = *n_cl_dev, p_blk = (uint64_t)blockIdx.x * HAO_PACK_T; const uint32_t t = threadIdx.x;
	const uint64_t w = (p_blk + 8 * t) >> 6;
	if (p_blk >= n_cl) {      // past the batch's hits: the words up to the bound are zero (the rank scan runs over the bound)
		if ((t & 7) == 0 && w < A.n_words_max) { A.bits[w] = 0; A.cnt[w] = 0; }
		return;
	}
	// the piece's chains: [first, ...) while they start before its end (source index < 2^32: a batch has fewer seed hits; flags: 1 = in ohits, 2 = codes, 4 = minimizer indices)
	const uint64_t first = A.blk_first[blockIdx.x]; uint32_t nc = 0;
	for (uint32_t base = 0; base + 256 <= HAO_PACK_CMAX; base += 256) {
		const uint64_t ci = first + base + t; int more = 0;
		if (ci < n_chains) {
			const hao_cdesc d = A.cd[ci];
			if (d.dst < p_blk + HAO_PACK_T) {
				const int64_t rel = (int64_t)d.dst - (int64_t)p_blk;
				c_dst[base + t] = rel < 0 ? 0 : (int32_t)rel; c_src[base + t] = (uint32_t)d.src; c_fl[base + t] = (uint8_t)((d.src & HAO_CD_OHITS ? 1u : 0u) | (d.pad & 6u));
				if (base + t == 0) s_dst0 = rel;
				more = 1;
			}
		}
		const int got = __syncthreads_count(more);      // (chains are in position order: the ones inside the piece are a prefix of the round)
		nc += (uint32_t)got;
		if (got < 256) break;
	}
	__syncthreads();
	const int32_t p0 = 8 * (int32_t)t;
	uint32_t c; { uint32_t lo = 0, hi = nc; while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (c_dst[m] <= p0) lo = m; else hi = m; } c = lo; }
	// phase 1 (LDS only): chain and index inside the chain of each of the 8 positions; phase 2: the 8 code bytes - unconditional loads, all in flight together
	// (a position without a code reads byte 0 of the array and ignores it)
	uint64_t ik[8]; uint32_t sk[8], ck[8]; uint8_t fk[8], code[8]; uint32_t todo = 0;      // todo: positions whose code must be computed from the hits
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int32_t pr = p0 + k;
		while (c + 1 < nc && c_dst[c + 1] <= pr) ++c;
		ck[k] = c; sk[k] = c_src[c]; fk[k] = c_fl[c]; ik[k] = (uint64_t)((int64_t)pr - (c == 0 ? s_dst0 : (int64_t)c_dst[c]));
		if (p_blk + (uint64_t)pr >= n_cl) ik[k] = 0;      // filler past the last hit: reads as a chain start (code 0x08, no bit)
	}
	const uint8_t *hc = A.hcode ? A.hcode : A.bytes;      // (flag 2 is never set without codes)
	uint8_t raw[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) raw[k] = hc[(ik[k] > 0 && (fk[k] & 2u)) ? (uint64_t)sk[k] + ik[k] : 0];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		code[k] = 0x08;
		if (ik[k] > 0) { if (fk[k] & 2u) code[k] = raw[k]; else todo |= 1u << k; }
	}
	uint32_t escm = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const bool every = A.exc_every && ik[k] > 0 && ik[k] % A.exc_every == A.exc_every - 1;
		if (code[k] == 0xff || every) escm |= 1u << k;
	}
	// (the per-position arrays are indexed by compile-time constants only: a run-time index would move them to scratch memory)
	if (todo)      // chains without codes (tiny groups, the DP path's copies): from the hits themselves
#pragma unroll
		for (int k = 0; k < 8; ++k) if (todo & (1u << k)) {
			const uint64_t si = sk[k], i = ik[k]; const uint32_t fl = fk[k];
			const hao_hit_t *src = ((fl & 1u) ? A.ohits : A.hits) + si;
			const hao_hit_t h = src[i], ph = src[i - 1]; uint32_t q, pq;
			if (fl & 4u) { q = A.hq[si + i]; pq = A.hq[si + i - 1]; }
			else {
				const uint32_t rd = A.cd[first + ck[k]].r; const uint64_t m0 = A.mz_off[A.rid_lo + rd]; const uint32_t nq = (uint32_t)(A.mz_off[A.rid_lo + rd + 1] - m0);
				q = hao_pack_find_q(A.q_pos + (m0 - A.mz0), nq, h.self_offset); pq = hao_pack_find_q(A.q_pos + (m0 - A.mz0), nq, ph.self_offset);
			}
			const int64_t dq = (int64_t)q - (int64_t)pq, dd = ((int64_t)h.offset - (int64_t)ph.offset) - ((int64_t)h.self_offset - (int64_t)ph.self_offset);
			code[k] = (uint8_t)((dq - 1) << 4 | (dd + 8));
			if (dq < 1 || dq > 15 || dd < -8 || dd > 7) escm |= 1u << k;
		}
	if (escm)      // verbatim list (rare): the hit itself, with its minimizer index
#pragma unroll
		for (int k = 0; k < 8; ++k) if (escm & (1u << k)) {
			const uint64_t si = sk[k], i = ik[k]; const uint32_t fl = fk[k];
			code[k] = 0xff;
			const unsigned long long kx = atomicAdd(A.exc_cnt, 1ULL);
			if (kx < A.exc_cap) {      // past the capacity only the count matters: the host grows the list and packs again
				const hao_hit_t *src = ((fl & 1u) ? A.ohits : A.hits) + si;
				hao_exc_t e; e.index = p_blk + (uint64_t)(p0 + k); e.pad = 0; e.hit = src[i]; e.hit.w0 = A.cd[first + ck[k]].w0;
				if (fl & 6u) e.q = A.hq[si + i];
				else {
					const uint32_t rd = A.cd[first + ck[k]].r; const uint64_t m0 = A.mz_off[A.rid_lo + rd]; const uint32_t nq = (uint32_t)(A.mz_off[A.rid_lo + rd + 1] - m0);
					e.q = hao_pack_find_q(A.q_pos + (m0 - A.mz0), nq, e.hit.self_offset);
				}
				A.exc[kx] = e;
			}
		}
	uint64_t word = 0; uint32_t m8 = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) { word |= (uint64_t)code[k] << (8 * k); if (code[k] != 0x08) m8 |= 1u << k; }
	if (p_blk + 8 * t < n_cl) *(uint64_t*)(A.bytes + p_blk + 8 * t) = word;
	uint64_t bw = (uint64_t)m8 << ((t & 7) * 8);
	bw |= __shfl_xor(bw, 1); bw |= __shfl_xor(bw, 2); bw |= __shfl_xor(bw, 4);
	if ((t & 7) == 0 && w < A.n_words_max) { A.bits[w] = bw; A.cnt[w] = (uint32_t)__popcll(bw); }
}

// code bytes of the flagged hits, at their rank: thread t takes hits [8t, 8t + 8)
__global__ __launch_bounds__(256) void hao_pack_codes_kernel(const uint8_t *bytes, const uint64_t *n_dev, const uint64_t *bits, const uint32_t *rank, uint64_t n_words_max, uint8_t *codes,
		unsigned long long *n_codes)
{
	const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, n = *n_dev;
	if (t == 0) *n_codes = rank[n_words_max - 1];      // (the last word of the bound is empty: its exclusive prefix is the total)
	if (8 * t >= n) return;
	const uint64_t word = bits[t >> 3]; const int sh = (int)(t & 7) * 8;
	uint32_t m8 = (uint32_t)(word >> sh) & 0xffu;
	if (!m8) return;
	uint64_t at = rank[t >> 3] + (uint64_t)__popcll(word & ((1ULL << sh) - 1));
	const uint64_t v = *(const uint64_t*)(bytes + 8 * t);
	for (; m8; m8 &= m8 - 1) codes[at++] = (uint8_t)(v >> (8 * (__ffs((int)m8) - 1)));
}

// the batch's minimizer table for the consumer: (self_offset, cnt) per query minimizer, interleaved
__global__ void hao_qtab_kernel(const uint32_t *q_pos, const uint32_t *q_cnt, uint64_t n_mz, hao_qmz_t *out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_mz) { hao_qmz_t v; v.self_offset = q_pos[i]; v.cnt = q_cnt[i]; out[i] = v; }
}

// per read: first chain / first hit of the read in the batch (ranges of the headers and of the packed words)
__global__ void hao_read_ranges_kernel(const uint64_t *g_off, const uint64_t *ch_base, const uint64_t *cl_base, const uint64_t *mz_off, uint64_t rid_lo, uint64_t mz0, uint64_t n_sel,
		uint64_t *ch_off, uint64_t *cl_off, uint64_t *qm_off)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	const uint64_t g = g_off[r];
	ch_off[r] = ch_base[g]; cl_off[r] = cl_base[g]; qm_off[r] = mz_off[rid_lo + r] - mz0;
}

// ---------------------------------------------------------------------------------------
// Exact-overlap check right after chaining: exact_ec_check (ecovlp.cpp:2803-2808) as h_ec_lchain_fast_new applies it to
// every candidate of the final round (ecovlp.cpp:5103-5131): the query interval [x_pos_s, x_pos_e] and the target interval
// [y_pos_s, y_pos_e] (target strand coordinates; recover_UC_Read_sub_region, Process_Read.cpp:524-614) are "exact" iff they have the same
// length and the same characters - N sites included (an N equals only an N; the store keeps N as A plus a side list).
// One wave per overlap on the packed 2-bit reads already resident in HBM: 16 bases per lane and step, the reverse strand by reversing the
// 2-bit groups of the mirrored window and complementing.  flags[j] = 1 / 0 for overlap j of the batch's final ol->list order.
// ---------------------------------------------------------------------------------------
// 16 bases [pos, pos+16) of a packed read as one word, base `pos` in bits 31..30; bytes past the store (len/4+1 bytes) read as 0
__device__ __forceinline__ uint32_t hao_bases16(const uint8_t *rd, uint32_t nbytes, int64_t pos)
{
	uint64_t v = 0; const int64_t b = pos >> 2;
#pragma unroll
	for (int i = 0; i < 5; ++i) { const int64_t bi = b + i; v = v << 8 | ((bi >= 0 && bi < (int64_t)nbytes) ? rd[bi] : 0); }
	return (uint32_t)(v >> (8 - 2 * (pos & 3)));
}
struct hao_exact_args {
	const hao_ovlp_t *ol; uint64_t n_ol; uint64_t rid_base;      // x_id / y_id are global read ids; the packed store is indexed locally (single device: rid_base = 0)
	const uint8_t *packed; const uint64_t *pk_off; const uint32_t *len; const uint64_t *nsite_off; const uint32_t *nsite;      // nsite_off == nullptr: no read has N
	uint8_t *flags;
};
__global__ __launch_bounds__(256) void hao_exact_check_kernel(hao_exact_args A)
{
	const uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (j >= A.n_ol) return;
	const int lane = hao_lane();
	const hao_ovlp_t o = A.ol[j];
	const uint64_t q = o.x_id - A.rid_base, t = o.y_id - A.rid_base;
	const int64_t qs = o.x_pos_s, n = (int64_t)o.x_pos_e + 1 - qs, ts = o.y_pos_s, tn = (int64_t)o.y_pos_e + 1 - ts;
	if (n != tn) { if (lane == 0) A.flags[j] = 0; return; }
	const uint8_t *qd = A.packed + A.pk_off[q], *td = A.packed + A.pk_off[t];
	const uint32_t ql = A.len[q], tl = A.len[t], qb = ql / 4 + 1, tb = tl / 4 + 1; const bool rev = o.y_pos_strand != 0;
	bool diff = false;
	for (int64_t i0 = 0; i0 < n && !diff; i0 += 64 * 16) {
		const int64_t i = i0 + (int64_t)lane * 16; bool d = false;
		if (i < n) {
			const uint32_t a = hao_bases16(qd, qb, qs + i);
			uint32_t b;
			if (!rev) b = hao_bases16(td, tb, ts + i);
			else {      // strand-1 base k = complement of forward base tl - 1 - k: the window [ts+i, ts+i+16) mirrors to forward [tl-16-(ts+i), tl-(ts+i))
				uint32_t f = hao_bases16(td, tb, (int64_t)tl - 16 - (ts + i));
				f = __brev(f); f = ((f & 0x55555555u) << 1) | ((f >> 1) & 0x55555555u);      // reverse the order of the 2-bit groups
				b = ~f;
			}
			const int64_t rem = n - i; const uint32_t m = rem >= 16 ? 0xffffffffu : ~(0xffffffffu >> (2 * rem));
			d = ((a ^ b) & m) != 0;
		}
		diff = __any(d);
	}
	int ok = diff ? 0 : 1;
	if (ok && A.nsite_off && lane == 0) {      // N sites: the two intervals must carry N at exactly the same offsets (lists are ascending; rare, one lane)
		const uint64_t qa = A.nsite_off[q], qe = A.nsite_off[q + 1], ta = A.nsite_off[t], te = A.nsite_off[t + 1];
		if (qe > qa || te > ta) {
			uint64_t nq = 0, nt = 0;
			for (uint64_t x = qa; x < qe; ++x) { const int64_t p = A.nsite[x]; if (p >= qs && p < qs + n) ++nq; }
			for (uint64_t y = ta; y < te; ++y) { const int64_t p = rev ? (int64_t)tl - 1 - A.nsite[y] : (int64_t)A.nsite[y]; if (p >= ts && p < ts + n) ++nt; }
			if (nq != nt) ok = 0;
			for (uint64_t x = qa; x < qe && ok; ++x) {
				const int64_t p = A.nsite[x]; if (p < qs || p >= qs + n) continue;
				bool found = false;
				for (uint64_t y = ta; y < te && !found; ++y) { const int64_t pt = rev ? (int64_t)tl - 1 - A.nsite[y] : (int64_t)A.nsite[y]; found = pt - ts == p - qs; }
				if (!found) ok = 0;
			}
		}
	}
	ok = __shfl(ok, 0) && !diff;
	if (lane == 0) A.flags[j] = (uint8_t)ok;
}


// device -> mapped pinned host memory by a few workgroups of plain 16-byte loads / stores (alternative to the DMA engines for the delivery copy)
typedef uint32_t hao_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hao_d2h_kernel(const hao_v4u *src, hao_v4u *dst, uint64_t n16)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
