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
// Wire format of cl->list (include/hao.h, hao_chain_hdr_t / hao_unpack_hits).  A chain's hits are colinear and share their readID
// word, so a 16-byte k_mer_hit travels as ONE 32-bit word relative to its predecessor in the chain:
//     bits  0..12  self_offset - previous self_offset                       (0 .. 8191)
//     bits 13..19  (offset - previous offset) - (self_offset delta) + 64    (diagonal shift -64 .. 63)
//     bits 20..27  cnt & 0xff   (k-mer span)
//     bits 28..30  cnt >> 8     (seed weight 0 .. 7)
//     bit  31      0
// or, when any field does not fit:  bit 31 = 1, bits 0..30 = index of the verbatim k_mer_hit in the batch's exception list.
// The first hit of a chain is relative to the (offset, self_offset) stored in the chain header.  4 bytes per chained hit instead of 16
// across PCIe; the consumer thread decodes straight into its Candidates_list (one pass, no intermediate copy).
// One wave per chain; reads the hits where the chain kernels left them (chain descriptors), so cl->list is never materialised in HBM.
// ---------------------------------------------------------------------------------------
struct hao_pack_args {
	const hao_cdesc *cd; uint64_t n_chains; const hao_hit_t *hits, *ohits;
	hao_chain_hdr_t *hdr; uint32_t *words; hao_hit_t *exc; unsigned long long *exc_cnt; uint64_t exc_cap; uint32_t exc_every;
};

__global__ __launch_bounds__(256) void hao_pack_chains_kernel(hao_pack_args A, const uint64_t *n_chains_dev)
{
	const uint64_t ci = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (ci >= *n_chains_dev) return;      // launched over the host-side bound (3 chains per group)
	const int lane = hao_lane();
	const hao_cdesc d = A.cd[ci];
	const hao_hit_t *src = hao_cd_src(d, A.hits, A.ohits);
	if (lane == 0) { hao_chain_hdr_t h; h.n_hits = d.n; h.w0 = d.w0; h.offset = d.n ? src[0].offset : 0; h.self_offset = d.n ? src[0].self_offset : 0; A.hdr[ci] = h; }
	for (uint32_t b = 0; b < d.n; b += 64) {
		const uint32_t i = b + lane; const bool act = i < d.n;
		hao_hit_t h, p; uint32_t w = 0; bool esc = false;
		if (act) {
			h = src[i]; p = i ? src[i - 1] : h;
			const int64_t ds = (int64_t)h.self_offset - (int64_t)p.self_offset, dd = ((int64_t)h.offset - (int64_t)p.offset) - ds;
			esc = ds < 0 || ds > 8191 || dd < -64 || dd > 63 || (h.cnt >> 8) > 7 || (A.exc_every && i % A.exc_every == A.exc_every - 1);
			w = (uint32_t)ds | (uint32_t)(dd + 64) << 13 | (h.cnt & 0xffu) << 20 | (h.cnt >> 8) << 28;
		}
		const unsigned long long em = __ballot(act && esc);
		if (em) {
			unsigned long long base = 0;
			if (lane == 0) base = atomicAdd(A.exc_cnt, (unsigned long long)__popcll(em));
			base = (unsigned long long)hao_readlane_i64((int64_t)base, 0);
			if (act && esc) {
				const uint64_t k = base + __popcll(em & ((1ULL << lane) - 1));
				if (k < A.exc_cap) { h.w0 = d.w0; A.exc[k] = h; }      // past the capacity only the count matters: the host grows the list and packs again
				w = 0x80000000u | (uint32_t)(k & 0x7fffffffu);
			}
		}
		if (act) A.words[d.dst + i] = w;
	}
}

// per read: first chain / first hit of the read in the batch (ranges of the headers and of the packed words)
__global__ void hao_read_ranges_kernel(const uint64_t *g_off, const uint64_t *ch_base, const uint64_t *cl_base, uint64_t n_sel, uint64_t *ch_off, uint64_t *cl_off)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	const uint64_t g = g_off[r];
	ch_off[r] = ch_base[g]; cl_off[r] = cl_base[g];
}
