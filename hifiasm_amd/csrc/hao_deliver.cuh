// Result delivery kernels: per-read digests of a batch's results (integrity check across the boundary / full-size parity tests).
#pragma once
#include "hao_common.cuh"

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
