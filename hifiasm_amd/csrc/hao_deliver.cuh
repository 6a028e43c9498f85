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
// batch ships, per read, its minimizer table (self_offset, cnt: 8 bytes per minimizer, ~430 per 15 kb read) once, and per hit a code:
//     high nibble = (minimizers skipped since the previous hit of the chain) = dq - 1     (0 .. 14)
//     low nibble  = (target offset delta) - (self_offset delta) + 8                       (diagonal shift -8 .. 7)
// 0xff = the hit is in the batch's exception list (verbatim, with its minimizer index), keyed by the hit's POSITION and sorted.
// More than nine hits in ten have the code 0x08 (next minimizer, same diagonal), so the codes do not travel as a byte per hit: the batch ships
// one BIT per position (1 = a code byte exists), a rank directory (code bytes before every 64th position) and the code bytes of the flagged
// positions: ~0.3 bytes per chained hit across PCIe instead of 16.
//
// POSITIONS are indices into the batch's sorted SEED hits, not into the concatenated cl->list: > 99.9 % of the chains are contiguous runs of the
// seed hits (chain descriptors, hao_chain.cuh), and the quick check (chain_group_kernel) already computes every seed hit's code relative to its
// predecessor while it has both in registers.  With positions = seed-hit indices that byte array IS the code array of the wire format: nothing is
// gathered per chain (the per-chain gather was latency-bound: 3 ms per 745 M-hit batch), the bit stream is one coalesced pass over it, and a chain
// header carries the position of its first hit.  The 3.5 % of seed hits that are in no chain cost a bit each.  A group that went through the
// DP (its <= 3 chains were compacted into ohits[group start ..)) uses the same positions - the group's range is its own - and its codes are
// written over the quick check's by the DP kernels' tails (hao_chain.cuh: hao_code_chain_lane and the wave tail; verbatim hits of such chains are marked 0xfd so that
// the packer takes them from ohits / ohq).  The code byte at a chain's FIRST position is meaningless (the header describes that
// hit) and skipped by the decoder.  Exception entries carry the hit as the seed stage wrote it: the decoder sets its readID word from the header.
// ---------------------------------------------------------------------------------------
#define HAO_PACK_QCAP 1024
struct hao_pack_args {
	const hao_cdesc *cd; const hao_hit_t *hits, *ohits;
	const uint64_t *mz_off, *seg; uint64_t rid_lo, mz0, n_sel; const uint32_t *q_pos;      // per-read minimizer ranges (global offsets), seed-hit ranges, the batch's self_offset table
	hao_chain_hdr_t *hdr; uint8_t *bytes; hao_exc_t *exc; unsigned long long *exc_cnt; uint64_t exc_cap; uint32_t exc_every;
	const uint16_t *hq;      // per seed hit: query minimizer index (seed kernel)
	const uint16_t *ohq;                     // per ohits entry: query minimizer index (the DP tails, hao_chain.cuh)
};

// minimizer index of a hit: the table position of its self_offset (positions are strictly ascending in a read's table)
__device__ __forceinline__ uint32_t hao_pack_find_q(const uint32_t *qp, uint32_t nq, uint32_t self_offset)
{ uint32_t lo = 0, hi = nq; while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (qp[m] < self_offset) lo = m + 1; else hi = m; } return lo; }

// One LANE per chain: its header (hit count, readID word, first hit: minimizer index, target offset, position).  Chains are independent: the dependent
// loads of one (descriptor -> read's minimizer range -> first hit) overlap across the 64 chains of a wave.
__global__ __launch_bounds__(256) void hao_pack_hdr_kernel(hao_pack_args A, const uint64_t *n_chains_dev)
{
	const uint64_t ci = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (ci >= *n_chains_dev) return;
	const hao_cdesc d = A.cd[ci];
	const hao_hit_t h0 = hao_cd_src(d, A.hits, A.ohits)[0];
	const uint64_t pos = d.src & ~HAO_CD_OHITS;
	uint32_t q0 = A.hq ? ((d.src & HAO_CD_OHITS) ? (A.ohq ? A.ohq[pos] : 65535u) : A.hq[pos]) : 65535u;
	if (q0 == 65535u) { const uint64_t m0 = A.mz_off[A.rid_lo + d.r]; q0 = hao_pack_find_q(A.q_pos + (m0 - A.mz0), (uint32_t)(A.mz_off[A.rid_lo + d.r + 1] - m0), h0.self_offset); }
	hao_chain_hdr_t H; H.n_hits = d.n; H.w0 = d.w0; H.q0 = q0; H.offset = h0.offset; H.pos = pos;
	A.hdr[ci] = H;
}

// One byte per position -> bit stream + per-word counts of the rank directory; a 0xff met on the way (the quick check could not express the hit) becomes
// an entry of the verbatim list: the seed hit at the position and its minimizer index.  Thread t takes positions [8t, 8t + 8) (one 8-byte load); the
// eight threads of a 64-position word combine their flags.
// (The verbatim entries are NOT appended here - their number per word goes to ecnt[], a scan turns the counts into list positions and
// hao_pack_codes_kernel writes the entries in position order, in the pass that places the code bytes.  A repeat-rich 250 Mb batch has 14 M verbatim hits in 1.3 M waves: one atomic per wave on the list's
// counter - all on one address - made this kernel 11 ms instead of 0.3, and the list then needed a 24-byte-record merge sort by position, ~10 ms more.)
// (Round 6: four 8-byte chunks per thread, a grid's width apart, all four loads issued before the first is used - with one load per thread the kernel ran at 1.3 TB/s, the memory
// parallelism of 32 waves x 512 bytes per CU - and the two byte tests as SWAR masks compressed by a multiplication instead of a loop over the bytes.)
#define HAO_PACK_U 4
static_assert(HAO_CODE_EXC_OHITS == 0xfd, "hao_pack_bits_kernel tests the byte 0xfd");
__device__ __forceinline__ uint32_t hao_byte_flags(uint64_t hi_bits) { return (uint32_t)(((hi_bits >> 7) * 0x0102040810204080ULL) >> 56); }      // bit 7 of byte k -> bit k
__global__ __launch_bounds__(256) void hao_pack_bits_kernel(hao_pack_args A, uint64_t n, uint64_t n_words, uint64_t *bits, uint32_t *cnt, uint32_t *ecnt)
{
	const uint64_t T = (uint64_t)gridDim.x * 256, t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // (T is a multiple of 8: a word's eight threads stay adjacent lanes in every round)
	const uint64_t L7 = 0x7f7f7f7f7f7f7f7fULL, H8 = 0x8080808080808080ULL;
	uint64_t v[HAO_PACK_U];
#pragma unroll
	for (int u = 0; u < HAO_PACK_U; ++u) { const uint64_t t = t0 + (uint64_t)u * T; v[u] = 8 * t < n ? *(const uint64_t*)(A.bytes + 8 * t) : 0x0808080808080808ULL; }
#pragma unroll
	for (int u = 0; u < HAO_PACK_U; ++u) {
		const uint64_t t = t0 + (uint64_t)u * T, w = t >> 3;
		const uint64_t x = v[u] ^ 0x0808080808080808ULL, xf = ~v[u], xd = v[u] ^ 0xfdfdfdfdfdfdfdfdULL;      // a byte of x is zero iff the code is 0x08; of xf / xd iff it is 0xff / HAO_CODE_EXC_OHITS
		uint32_t m8 = hao_byte_flags((x | ((x & L7) + L7)) & H8);                                                  // bytes != 0x08
		uint32_t e8 = hao_byte_flags((~(((xf & L7) + L7) | xf | L7) | ~(((xd & L7) + L7) | xd | L7)) & H8);        // bytes == 0xff or == 0xfd
		const uint64_t left = 8 * t < n ? n - 8 * t : 0;      // positions of this chunk that exist
		if (left < 8) { const uint32_t keep = (1u << (uint32_t)left) - 1u; m8 &= keep; e8 &= keep; }
		// the thread's byte of the word goes out as a byte (adjacent lanes, adjacent bytes); the word's two counts travel in one register through three row_shl adds - lane 8 g of
		// a row ends with the sums of lanes 8 g .. 8 g + 7.  (Nine __shfl_xor - ds_bpermute_b32 at 24 cycles each, 216 per 8 bytes of a lane - held this kernel at 1.6 TB/s.)
		if (w < n_words) ((uint8_t*)bits)[t] = (uint8_t)m8;
		uint32_t pc = (uint32_t)__popc(m8) | (uint32_t)__popc(e8) << 16;
		pc += (uint32_t)hao_dpp<0x101, 0xf>(0, (int)pc); pc += (uint32_t)hao_dpp<0x102, 0xf>(0, (int)pc); pc += (uint32_t)hao_dpp<0x104, 0xf>(0, (int)pc);      // (no lane leaves early: DPP moves are wave-wide)
		if ((t & 7) == 0 && w < n_words) { cnt[w] = pc & 0xffffu; ecnt[w] = pc >> 16; }
	}
}

// code bytes of the flagged positions, at their rank: thread t takes positions [8t, 8t + 8).  The same pass writes the verbatim list at the
// positions erank[] (the scan of hao_pack_bits_kernel's counts) gives.
__global__ __launch_bounds__(256) void hao_pack_codes_kernel(hao_pack_args A, const uint8_t *bytes, uint64_t n, const uint64_t *bits, const uint32_t *rank, uint64_t n_words, uint8_t *codes,
		unsigned long long *n_codes, const uint32_t *erank)
{
	const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, w = t >> 3;
	if (t == 0) { *n_codes = rank[n_words]; *A.exc_cnt = erank[n_words]; }      // (exclusive prefixes over n_words + 1 counts: the last entries are the totals)
	const bool in = 8 * t < n;
	const uint64_t word = in ? bits[w] : 0; const int sh = (int)(t & 7) * 8;
	uint32_t m8 = (uint32_t)(word >> sh) & 0xffu, e8 = 0;
	uint64_t v = 0;
	if (m8) v = *(const uint64_t*)(bytes + 8 * t);      // (a word's bits beyond n are zero)
	if (m8) {
		uint64_t at = rank[w] + (uint64_t)__popcll(word & ((1ULL << sh) - 1));
		for (uint32_t m = m8; m; m &= m - 1) { const uint8_t b = (uint8_t)(v >> (8 * (__ffs((int)m) - 1))); codes[at++] = b >= HAO_CODE_EXC_OHITS ? (uint8_t)0xff : b; }      // (0xfd: the device-only flavour of "verbatim")
	}
	// every verbatim position is a flagged position: only the set bits of m8 can be 0xff / 0xfd
	for (uint32_t m = m8; m; m &= m - 1) { const int k = __ffs((int)m) - 1; const uint8_t b = (uint8_t)(v >> (8 * k)); if (b == 0xff || b == HAO_CODE_EXC_OHITS) e8 |= 1u << k; }
	uint64_t eword = (uint64_t)e8 << sh;
	eword |= __shfl_xor(eword, 1); eword |= __shfl_xor(eword, 2); eword |= __shfl_xor(eword, 4);      // (no lane has left before this)
	if (!e8) return;
	unsigned long long kx = (unsigned long long)erank[w] + (unsigned long long)__popcll(eword & ((1ULL << sh) - 1));
	for (uint32_t m = e8; m; m &= m - 1, ++kx) {
		if (kx >= A.exc_cap) continue;      // past the capacity only the count matters: the host grows the list and packs again
		const uint64_t p = 8 * t + (uint32_t)(__ffs((int)m) - 1);
		const bool oh = (uint8_t)(v >> (8 * (__ffs((int)m) - 1))) == HAO_CODE_EXC_OHITS;      // a hit of a chain the DP compacted: it (and its minimizer index) sit in ohits / ohq at the position
		hao_exc_t e; e.index = p; e.pad = 0; e.hit = oh ? A.ohits[p] : A.hits[p]; e.q = oh ? A.ohq[p] : (A.hq ? A.hq[p] : 65535u);
		if (e.q == 65535u) {      // the 16-bit index saturated (a read of > 65 534 minimizers): the read of the position, then its table
			uint64_t lo = 0, hi = A.n_sel; while (hi - lo > 1) { const uint64_t md = (lo + hi) >> 1; if (A.seg[md] <= p) lo = md; else hi = md; }
			const uint64_t m0 = A.mz_off[A.rid_lo + lo];
			e.q = hao_pack_find_q(A.q_pos + (m0 - A.mz0), (uint32_t)(A.mz_off[A.rid_lo + lo + 1] - m0), e.hit.self_offset);
		}
		A.exc[kx] = e;
	}
}

// the rank directory as it travels: one entry per 256 positions (every fourth word's; the decoder counts the bits of up to three words itself)
__global__ void hao_rank4_kernel(const uint32_t *rank, uint64_t n4, uint32_t *out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n4) out[i] = rank[4 * i];
}

// ---- fake cigars on the wire ----
// A fake cigar (gen_fake_cigar, Hash_Table.cpp:88-109) is a list of 8-byte (query site << 32 | diagonal shift) entries: (x_pos_s, 0), one entry wherever the
// chain's diagonal changes, and (x_pos_e, last shift) unless the last change sits there - ~11 entries per overlap of a HiFi pass, 2.66 GB of the 8.26 GB a
// configs[2] pass delivered, more than the chained hits themselves.  Sites ascend and shifts move by a base or two, so the wire carries 4 bytes per entry and
// none for the first one: bits 0 .. 19 = site - previous site, bits 20 .. 31 = zigzag(shift - previous shift), starting from the overlap's (x_pos_s, 0).  An
// overlap with a step that does not fit (or whose first entry is not (x_pos_s, 0)) travels raw, two words per entry; bit 63 of its offset says so.
// hao_unpack_cigar (include/hao.h) rebuilds the 8-byte entries.  Offsets count 32-bit words.
// (Written by chain_final_kernel, hao_chain.cuh, while it gathers the cigars into their final order: the packed position of an overlap's words follows from
// its entry offset and its index, so the wire form costs no pass of its own.)

// fill by a kernel: a big hipMemsetAsync travels through the DMA queues, where the previous batch's result copy is in flight (measured: the copy of a
// configs[2] batch then took 28 instead of 18 ms and was no longer hidden under the next batch's kernels)
typedef uint32_t hao_fill_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hao_fill16_kernel(hao_fill_v4 *p, uint64_t n16, uint32_t word)
{
	const hao_fill_v4 v = { word, word, word, word };
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = v;
}

// ol->list for the wire: 32 of the 48 bytes of every final overlap (hao_ovlp_wire_t; the receiver knows the read, x_pos_strand = 0 and align_length = 0)
__global__ __launch_bounds__(256) void hao_ol_wire_kernel(const hao_ovlp_t *ol, const uint64_t *n_dev, hao_ovlp_wire_t *out)
{
	const uint64_t n = *n_dev;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
		const hao_ovlp_t o = ol[i]; hao_ovlp_wire_t w;
		w.y = o.y_id | o.y_pos_strand << 31; w.x_pos_s = o.x_pos_s; w.x_pos_e = o.x_pos_e; w.y_pos_s = o.y_pos_s; w.y_pos_e = o.y_pos_e;
		w.shared_seed = o.shared_seed; w.non_homopolymer_errors = o.non_homopolymer_errors; w.fc_len = o.fc_len;
		out[i] = w;
	}
}

// the batch's minimizer table for the consumer: (self_offset, cnt) per query minimizer, interleaved
__global__ void hao_qtab_kernel(const uint32_t *q_pos, const uint32_t *q_cnt, uint64_t n_mz, hao_qmz_t *out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_mz) { hao_qmz_t v; v.self_offset = q_pos[i]; v.cnt = q_cnt[i]; out[i] = v; }
}

// the same in 4 bytes per minimizer (reads shorter than 65 536 bases, weights below 256: the host checked both; a value that does not fit raises *err all the same)
__global__ void hao_qtab16_kernel(const uint32_t *q_pos, const uint32_t *q_cnt, uint64_t n_mz, uint16_t *pos, uint16_t *cnt, int *err)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_mz) { const uint32_t p = q_pos[i], k = q_cnt[i]; pos[i] = (uint16_t)p; cnt[i] = (uint16_t)k; if (p > 0xffffu || k > 0xffffu) *err = 1; }
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


