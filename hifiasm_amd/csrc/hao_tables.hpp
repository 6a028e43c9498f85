// Host orchestration: ha_ft_gen / ha_pt_gen equivalents (part of libhao.so).
#pragma once
#include <chrono>
#include "hao_comm.hpp"
#include "hao_pipeline.hpp"
#include "hao_index.cuh"
#include "hao_host.hpp"

// first index with keys[i] >= target[t] (sorted keys); one thread per target
__global__ void hao_lower_bound_kernel(const uint64_t *keys, uint64_t n, const uint64_t *targets, int nt, uint64_t *out)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nt) return;
	uint64_t lo = 0, hi = n, x = targets[t];
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (keys[m] < x) lo = m + 1; else hi = m; }
	out[t] = lo;
}
// same on the low 12 bits (keys sorted by sub-table)
__global__ void hao_lower_bound_low12_kernel(const uint64_t *keys, uint64_t n, const uint64_t *targets, int nt, uint64_t *out)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nt) return;
	uint64_t lo = 0, hi = n, x = targets[t];
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if ((keys[m] & 4095) < x) lo = m + 1; else hi = m; }
	out[t] = lo;
}
// owner bits of a minimizer hash (top 16) + identity permutation; gather through a sorted permutation
__global__ void hao_owner_key_kernel(const uint64_t *x, uint64_t n, uint32_t *key, uint32_t *idx)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { key[i] = (uint32_t)(x[i] >> 48); idx[i] = (uint32_t)i; }
}
__global__ void hao_gather2_kernel(const uint32_t *idx, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *oa, uint64_t *ob)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const uint32_t j = idx[i]; oa[i] = a[j]; ob[i] = b[j]; }
}
__global__ void hao_add_const_kernel(uint64_t *v, uint64_t n, uint64_t add)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] += add;
}
__global__ void hao_add_u32_kernel(uint32_t *v, uint64_t n, uint32_t add)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] += add;
}
__global__ void hao_adjdiff_kernel(const uint64_t *off, uint64_t n, uint64_t *cnt)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cnt[i] = off[i + 1] - off[i];
}

// run-count index for reads [lo, hi); scalar flag = read has N (or force_scalar_all)
static int hao_prepare_runs(hao_ctx *c, uint64_t lo, uint64_t hi, bool force_scalar_all, std::vector<uint32_t> &slist)
{
	const uint64_t n_sel = hi - lo;
	std::vector<uint64_t> tile_off(n_sel + 1); std::vector<uint8_t> flag(n_sel);
	slist.clear();
	for (uint64_t r = 0; r < n_sel; ++r) {
		tile_off[r] = r == 0 ? 0 : tile_off[r - 1] + (c->h_len[lo + r - 1] + HAO_SK_TILE - 1) / HAO_SK_TILE + 1;
		bool hasn = c->has_n && c->h_nsite_off[lo + r + 1] > c->h_nsite_off[lo + r];
		flag[r] = (hasn || force_scalar_all) ? 1 : 0;
		if (flag[r]) slist.push_back((uint32_t)r);
	}
	tile_off[n_sel] = tile_off[n_sel - 1] + (c->h_len[hi - 1] + HAO_SK_TILE - 1) / HAO_SK_TILE + 1;
	HIP_TRY(c->d_tile_off.reserve(n_sel + 1)); HIP_TRY(c->d_tile_ord.reserve(tile_off[n_sel] + 1)); HIP_TRY(c->d_n_runs.reserve(n_sel + 1));
	HIP_TRY(c->d_scalar_flag.reserve(n_sel + 1)); HIP_TRY(c->d_scalar_list.reserve(slist.size() + 1));
	HIP_TRY(hipMemcpyAsync(c->d_tile_off.p, tile_off.data(), (n_sel + 1) * 8, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_scalar_flag.p, flag.data(), n_sel, hipMemcpyHostToDevice, c->stream));
	if (!slist.empty()) HIP_TRY(hipMemcpyAsync(c->d_scalar_list.p, slist.data(), slist.size() * 4, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(hpc_index_kernel, dim3((unsigned)((n_sel + 3) / 4)), dim3(256), 0, c->stream, c->d_packed.p, c->d_pk_off.p, c->d_len.p,
					   c->d_tile_off.p, c->d_tile_ord.p, c->d_n_runs.p, lo, n_sel, c->opt.hpc, (uint8_t*)nullptr);
	HAO_CHECK_LAUNCH();
	HIP_TRY(hipStreamSynchronize(c->stream));   // host vectors go out of scope
	return HAO_OK;
}

// prior homozygous coverage of the peak finder: total bases of ALL reads / --hg-size (htab.cpp:1156,1254: ct->bs / hg_size), -1 without one
static int hao_prior_hom(const hao_ctx *c)
{
	if (c->opt.hg_size <= 0) return -1;
	uint64_t bs = 0; for (uint32_t l : c->h_len_all) bs += l;
	return (int)(bs / (uint64_t)c->opt.hg_size);
}

// Sharded mode: the shards must be contiguous read ranges in rank order (the index build relies on "source-rank order = global read order",
// the Bloom replay on "rank order = insertion order").  One tiny all-gather at the start of EVERY sharded ha_ft_gen / ha_pt_gen: each rank sees
// every rank's (n_reads, rid_base, n_total) and evaluates the same predicate on the same data, so all ranks pass or fail together; the collective
// also carries local_rc, the status of whatever local work preceded it.
static int hao_shard_layout_check(hao_ctx *c, hao_comm &cm, int local_rc)
{
	const uint64_t mine[3] = { c->n_reads, c->rid_base, c->n_total }; std::vector<uint64_t> all;
	if (int rc = hao_comm_allgather_u64n(c, cm, mine, 3, all, local_rc)) return rc;
	uint64_t b = 0; bool ok = true;
	for (int r = 0; r < cm.world; ++r) { ok = ok && all[3 * r + 1] == b && all[3 * r + 2] == all[2]; b += all[3 * r]; }
	if (!ok || b != all[2]) { hao_set_err(c, "shards are not contiguous read ranges in rank order"); return HAO_EINVAL; }
	return HAO_OK;
}
// status agreement without payload (before a bulk exchange whose buffers were just allocated)
static int hao_comm_agree(hao_ctx *c, hao_comm &cm, int local_rc) { std::vector<uint64_t> t; return hao_comm_allgather_u64(c, cm, 0, t, local_rc); }

struct NotSentinel { __host__ __device__ bool operator()(const uint64_t &h) const { return h != UINT64_MAX; } };
struct RunHead { const uint64_t *k; __host__ __device__ uint64_t operator()(uint64_t i) const { return (i == 0 || k[i] != k[i - 1]) ? 1 : 0; } };
struct RunHead32 { const uint64_t *k; __host__ __device__ uint32_t operator()(uint64_t i) const { return (i == 0 || k[i] != k[i - 1]) ? 1u : 0u; } };

// Run-length encoding of a sorted array.  rocprim::run_length_encode takes its size as `unsigned int` (it would silently encode n mod 2^32 items: the k-mer
// occurrences of BASELINE configs[2] are 5.6 G), so longer inputs go through what it is built on - reduce_by_key over a constant 1 - whose size is a size_t.
template<typename Key>
static int hao_rle(hao_ctx *c, const Key *sorted, uint64_t n, Key *ukeys, uint32_t *ucnt, uint64_t *d_n_runs)
{
	size_t tb = 0;
	if (n <= 0xffffffffULL) {
		HIP_TRY(rocprim::run_length_encode(nullptr, tb, sorted, (unsigned int)n, ukeys, ucnt, d_n_runs, c->stream)); HIP_TRY(hao_tmp(c, tb));
		HIP_TRY(rocprim::run_length_encode(c->d_tmp.p, tb, sorted, (unsigned int)n, ukeys, ucnt, d_n_runs, c->stream));
	} else {
		auto ones = rocprim::make_constant_iterator<uint32_t>(1);
		HIP_TRY(rocprim::reduce_by_key(nullptr, tb, sorted, ones, (size_t)n, ukeys, ucnt, d_n_runs, rocprim::plus<uint32_t>(), rocprim::equal_to<Key>(), c->stream)); HIP_TRY(hao_tmp(c, tb));
		HIP_TRY(rocprim::reduce_by_key(c->d_tmp.p, tb, sorted, ones, (size_t)n, ukeys, ucnt, d_n_runs, rocprim::plus<uint32_t>(), rocprim::equal_to<Key>(), c->stream));
	}
	return HAO_OK;
}
// hipMemsetAsync in pieces below 2^31 bytes (the fill kernel's size arithmetic is 32-bit on some ROCm versions)
static int hao_memset_big(hao_ctx *c, void *p, int v, uint64_t bytes)
{
	for (uint64_t o = 0; o < bytes; o += 1ULL << 30) HIP_TRY(hipMemsetAsync((char*)p + o, v, (size_t)std::min<uint64_t>(1ULL << 30, bytes - o), c->stream));
	return HAO_OK;
}

// sort keys, run-length encode, histogram.  in: d_keys[n] (destroyed). out: unique keys / counts in c->d_u_keys / d_u_cnt, n_unique.
struct hao_rle_out { uint64_t n_unique; };
static int hao_sort_rle_hist(hao_ctx *c, uint64_t *d_keys, uint64_t *d_keys_alt, uint64_t n, DevBuf<uint64_t> &ukeys, DevBuf<uint32_t> &ucnt, uint64_t *n_unique, int64_t hist[HAO_N_COUNTS], uint64_t **sorted_out, uint32_t bias = 0)
{
	*n_unique = 0; memset(hist, 0, sizeof(int64_t) * HAO_N_COUNTS); *sorted_out = d_keys;
	if (n == 0) return HAO_OK;
	size_t tb = 0;
	rocprim::double_buffer<uint64_t> db(d_keys, d_keys_alt);
	HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, db, n, 0, 64, c->stream));
	HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::radix_sort_keys(c->d_tmp.p, tb, db, n, 0, 64, c->stream));
	uint64_t *sorted = db.current(); *sorted_out = sorted;
	HIP_TRY(c->d_cursor.reserve(2));
	uint64_t n_runs = 0;
	{	// number of distinct keys = number of run heads: lets the RLE outputs be sized exactly (matters at 10^9 k-mers)
		auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RunHead{sorted});
		tb = 0;
		HIP_TRY(rocprim::reduce(nullptr, tb, heads, (uint64_t*)c->d_cursor.p, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
		HIP_TRY(hao_tmp(c, tb));
		HIP_TRY(rocprim::reduce(c->d_tmp.p, tb, heads, (uint64_t*)c->d_cursor.p, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
		HIP_TRY(hipMemcpyAsync(&n_runs, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	HIP_TRY(ukeys.reserve(n_runs + 1)); HIP_TRY(ucnt.reserve(n_runs + 1));
	if (int rc = hao_rle(c, sorted, n, ukeys.p, ucnt.p, (uint64_t*)c->d_cursor.p)) return rc;
	HIP_TRY(hipMemcpyAsync(n_unique, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (bias && *n_unique) { hipLaunchKernelGGL(hao_add_u32_kernel, dim3((unsigned)((*n_unique + 255) / 256)), dim3(256), 0, c->stream, ucnt.p, *n_unique, bias); HAO_CHECK_LAUNCH(); }   // table entries start at `bias`
	DevBuf<unsigned long long> dh; HIP_TRY(dh.reserve(HAO_N_COUNTS));
	HIP_TRY(hipMemsetAsync(dh.p, 0, HAO_N_COUNTS * 8, c->stream));
	unsigned nb = (unsigned)std::min<uint64_t>((*n_unique + 255) / 256, 2048);
	if (nb) hipLaunchKernelGGL(hao_count_hist_kernel, dim3(nb), dim3(256), 0, c->stream, ucnt.p, *n_unique, dh.p);
	HAO_CHECK_LAUNCH();
	HIP_TRY(hipMemcpyAsync(hist, dh.p, HAO_N_COUNTS * 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	dh.release();
	return HAO_OK;
}

// keep runs with lo <= min(cnt,4095) <= hi -> (keys, start, cnt) ; start may be null
static int hao_keep_runs(hao_ctx *c, const uint64_t *ukeys, const uint32_t *ucnt, uint64_t n_unique, int lo, int hi,
						 DevBuf<uint64_t> &keys, DevBuf<uint64_t> *start, DevBuf<uint32_t> &cnt, uint64_t *n_kept, uint64_t *n_pos)
{
	*n_kept = 0; if (n_pos) *n_pos = 0;
	if (n_unique == 0) return HAO_OK;
	DevBuf<uint64_t> &flag = c->w_flag, &kpos = c->w_kpos, &ustart = c->w_ustart;
	HIP_TRY(flag.reserve(n_unique + 1)); HIP_TRY(kpos.reserve(n_unique + 1)); HIP_TRY(ustart.reserve(n_unique + 1));
	hipLaunchKernelGGL(hao_range_flag_kernel, dim3((unsigned)((n_unique + 256) / 256)), dim3(256), 0, c->stream, ucnt, n_unique, lo, hi, flag.p);
	HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, flag.p, kpos.p, n_unique + 1)) return rc;
	{ auto it = rocprim::make_transform_iterator(ucnt, U32ToU64()); if (int rc = hao_excl_scan_u64(c, it, ustart.p, n_unique)) return rc; }
	HIP_TRY(hipMemcpyAsync(n_kept, kpos.p + n_unique, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(keys.reserve(*n_kept + 1)); HIP_TRY(cnt.reserve(*n_kept + 1)); if (start) HIP_TRY(start->reserve(*n_kept + 1));
	hipLaunchKernelGGL(hao_keep_scatter_kernel, dim3((unsigned)((n_unique + 255) / 256)), dim3(256), 0, c->stream, ukeys, ucnt, ustart.p, flag.p, kpos.p, n_unique,
					   keys.p, start ? start->p : nullptr, cnt.p);
	HAO_CHECK_LAUNCH();
	if (n_pos && *n_kept) {   // sum of kept counts
		size_t tb = 0; auto it = rocprim::make_transform_iterator(cnt.p, U32ToU64());
		HIP_TRY(rocprim::reduce(nullptr, tb, it, (uint64_t*)c->d_cursor.p, (uint64_t)0, *n_kept, rocprim::plus<uint64_t>(), c->stream));
		HIP_TRY(hao_tmp(c, tb));
		HIP_TRY(rocprim::reduce(c->d_tmp.p, tb, it, (uint64_t*)c->d_cursor.p, (uint64_t)0, *n_kept, rocprim::plus<uint64_t>(), c->stream));
		HIP_TRY(hipMemcpyAsync(n_pos, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HAO_OK;
}

static int hao_build_bucket(hao_ctx *c, const uint64_t *keys, uint64_t n, int bits, DevBuf<uint32_t> &bucket)
{
	uint32_t nb = 1u << bits;
	HIP_TRY(bucket.reserve(nb + 2));
	hipLaunchKernelGGL(hao_bucket_kernel, dim3((nb + 256) / 256), dim3(256), 0, c->stream, keys, n, 64 - bits, nb, bucket.p);
	HAO_CHECK_LAUNCH();
	return HAO_OK;
}

// the filter table's hash view (hao_sketch.cuh: hao_ft_dev) from the sorted device arrays: 2^(hbits - 2) buckets of two slots for ~0.4 - 0.75 keys per bucket
static int hao_ft_build_hash(hao_ctx *c, uint64_t n)
{
	int hb = 15; while (hb < 32 && (1ULL << (hb - 2)) * 3 < n * 4) ++hb;      // buckets >= 4/3 n (hbits <= 32: the lookups index the bitmap with 32 bits; beyond 0.8 G keys the
	                                                                             // buckets fill up and more of them fall back to the sorted array - slower, still exact)
	c->ft_hbits = hb;
	const uint64_t nbk = 1ULL << (hb - 2), nw = (1ULL << hb) / 32;
	HIP_TRY(c->d_ft_hbit.reserve(nw + 1)); HIP_TRY(c->d_ft_hslot.reserve(2 * nbk + 2));
	HIP_TRY(hipMemsetAsync(c->d_ft_hbit.p, 0, nw * 4, c->stream));
	for (uint64_t o = 0; o < 2 * nbk * 8; o += 1ULL << 30) HIP_TRY(hipMemsetAsync((char*)c->d_ft_hslot.p + o, 0, std::min<uint64_t>(1ULL << 30, 2 * nbk * 8 - o), c->stream));
	if (n) { hipLaunchKernelGGL(hao_ft_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_ft_keys.p, c->d_ft_vals.p, n, hb, c->d_ft_hbit.p, c->d_ft_hslot.p); HAO_CHECK_LAUNCH(); }
	return HAO_OK;
}

// HAO_DBG_BLOOM: self-checks of the replay's intermediate arrays (sortedness, key / value pairing, run totals) with wall times, on stderr
struct BlkInv { const uint32_t *k; __host__ __device__ uint64_t operator()(uint64_t i) const { return (i > 0 && k[i] < k[i - 1]) ? 1 : 0; } };
struct BlkPair { const uint32_t *k; const uint64_t *v; int xb; __host__ __device__ uint64_t operator()(uint64_t i) const {
	const uint64_t h = v[i]; const uint32_t b = h == UINT64_MAX ? 1u << (12 + xb) : (uint32_t)((h & 4095) << xb | ((h >> 12) & ((1ULL << xb) - 1))); return b != k[i] ? 1 : 0; } };
struct MaxU32 { __host__ __device__ uint64_t operator()(const uint64_t &a, const uint64_t &b) const { return a > b ? a : b; } };
template<typename It, typename Op> static uint64_t hao_dbg_reduce(hao_ctx *c, It it, uint64_t n, Op op)
{
	uint64_t r = 0; size_t tb = 0; (void)c->d_cursor.reserve(2);
	(void)rocprim::reduce(nullptr, tb, it, (uint64_t*)c->d_cursor.p, (uint64_t)0, n, op, c->stream); (void)hao_tmp(c, tb);
	(void)rocprim::reduce(c->d_tmp.p, tb, it, (uint64_t*)c->d_cursor.p, (uint64_t)0, n, op, c->stream);
	(void)hipMemcpy(&r, c->d_cursor.p, 8, hipMemcpyDeviceToHost);
	return r;
}

// Bloom replay (hao_index.cuh): in[n] = k-mer hashes in insertion order (sentinels allowed) -> the occurrences that reach the count
// table, compacted; *out / *out_alt = the two buffers (of in / alt) to hand to the counting step.
static int hao_bloom_filter(hao_ctx *c, uint64_t *in, uint64_t *alt, uint64_t n, uint64_t **out, uint64_t **out_alt, uint64_t *n_out)
{
	const int xb = c->opt.bf_shift - 21;                   // log2 of the 512-bit blocks per sub-table
	*out = in; *out_alt = alt; *n_out = 0;
	if (n == 0) return HAO_OK;
	DevBuf<uint32_t> blk, blk2; DevBuf<uint8_t> flag;
	const bool dbg = c->sw.bloom;      // (HAO_DBG_PRINT=bloom)
	auto now_ = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_ = now_();
	auto lap = [&](const char *what) { if (dbg) { (void)hipStreamSynchronize(c->stream); const double t1 = now_(); fprintf(stderr, "[bloom] %-14s %8.3f s  (%s)\n", what, t1 - t_, hipGetErrorString(hipGetLastError())); fflush(stderr); t_ = now_(); } };
	if (dbg) fprintf(stderr, "[bloom] n = %llu, xb = %d\n", (unsigned long long)n, xb);
	HIP_TRY(blk.reserve(n + 1)); HIP_TRY(blk2.reserve(n + 1)); HIP_TRY(flag.reserve(n + 1));
	hipLaunchKernelGGL(hao_bf_block_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, in, n, xb, blk.p);
	HAO_CHECK_LAUNCH();
	lap("alloc+blockid");
	size_t tb = 0; rocprim::double_buffer<uint32_t> dk(blk.p, blk2.p); rocprim::double_buffer<uint64_t> dv(in, alt);
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, n, 0, 13 + xb, c->stream)); HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, dk, dv, n, 0, 13 + xb, c->stream));      // stable: insertion order inside a block
	lap("sort_pairs");
	if (dbg) {
		const uint64_t inv = hao_dbg_reduce(c, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), BlkInv{dk.current()}), n, rocprim::plus<uint64_t>());
		const uint64_t mis = hao_dbg_reduce(c, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), BlkPair{dk.current(), dv.current(), xb}), n, rocprim::plus<uint64_t>());
		fprintf(stderr, "[bloom] after the sort: %llu inversions, %llu (key, value) pairs that do not belong together\n", (unsigned long long)inv, (unsigned long long)mis);
		lap("checks");
	}
	{	// runs of equal block id -> (block, length, start); one lane replays one run
		const uint64_t max_runs = std::min<uint64_t>(n, (1ULL << (12 + xb)) + 1);
		DevBuf<uint32_t> rk, rl; DevBuf<uint64_t> rs; uint64_t n_runs = 0;
		HIP_TRY(rk.reserve(max_runs + 1)); HIP_TRY(rl.reserve(max_runs + 1)); HIP_TRY(rs.reserve(max_runs + 2)); HIP_TRY(c->d_cursor.reserve(2));
		if (int rc = hao_memset_big(c, flag.p, 0, n)) return rc;
		if (int rc = hao_rle(c, (const uint32_t*)dk.current(), n, rk.p, rl.p, (uint64_t*)c->d_cursor.p)) return rc;
		HIP_TRY(hipMemcpyAsync(&n_runs, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		lap("rle");
		auto it = rocprim::make_transform_iterator(rl.p, U32ToU64());
		if (dbg) {
			const uint64_t sum = hao_dbg_reduce(c, it, n_runs, rocprim::plus<uint64_t>()), mx = hao_dbg_reduce(c, it, n_runs, MaxU32());
			fprintf(stderr, "[bloom] runs: %llu (bound %llu), lengths sum to %llu (n = %llu), longest %llu\n", (unsigned long long)n_runs, (unsigned long long)max_runs, (unsigned long long)sum, (unsigned long long)n, (unsigned long long)mx);
			if (n_runs > max_runs) { hao_set_err(c, "bloom replay: more runs than blocks"); return HAO_EUNSUPP; }
		}
		if (int rc = hao_excl_scan_u64(c, it, rs.p, n_runs)) return rc;
		hipLaunchKernelGGL(hao_bf_replay_kernel, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, c->stream, rk.p, rl.p, rs.p, n_runs, dv.current(), xb, flag.p);
		HAO_CHECK_LAUNCH();
		HIP_TRY(hipStreamSynchronize(c->stream));
		lap("replay");
		rk.release(); rl.release(); rs.release();
	}
	HIP_TRY(c->d_cursor.reserve(2));
	tb = 0;
	HIP_TRY(rocprim::select(nullptr, tb, dv.current(), flag.p, dv.alternate(), (uint64_t*)c->d_cursor.p, n, c->stream)); HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::select(c->d_tmp.p, tb, dv.current(), flag.p, dv.alternate(), (uint64_t*)c->d_cursor.p, n, c->stream));
	HIP_TRY(hipMemcpyAsync(n_out, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	*out = dv.alternate(); *out_alt = dv.current();
	lap("select");
	if (dbg) fprintf(stderr, "[bloom] %llu of %llu occurrences reach the count table\n", (unsigned long long)*n_out, (unsigned long long)n);
	blk.release(); blk2.release(); flag.release();
	return HAO_OK;
}

// ---------------------------------------------------------------------------------------
// ha_ft_gen (htab.cpp:1136-1169): all HPC k-mers -> [Bloom replay at -f > 0, hao_index.cuh] -> counts -> histogram -> peaks ->
// keep count >= cutoff -> filter table. + ha_opt_update_cov (CommandLines.cpp:411-418).
// ---------------------------------------------------------------------------------------
// ha_ft_gen in hash-range PASSES.  One pass holds every k-mer occurrence of the local reads at once: two buffers of 8 bytes per base, the largest allocations of the
// whole engine (configs[3]'s share of one of eight GPUs: 15 Gbases -> 2 x 120 GB).  The reference never does (htab.cpp:707-882 counts pipeline batches into 4096
// sub-tables, :147-151).  With P passes, pass j holds only the occurrences whose hash lies in the j-th P-th of every owner's range (exact counting: ranges of the
// hash itself, so the passes' run lists concatenate into the sorted list one pass gives; through the Bloom filter: ranges of the SUB-TABLE index - hash bits 0 - 11 -
// because a filter's state belongs to its sub-table and the replay needs a sub-table's occurrences in insertion order): the hashes are computed read chunk by read
// chunk into a 2 GB scratch and the pass's share is appended, in order, to the pass buffer.  Peak = 2 x 8 B x bases / P + the scratch + 12 B per distinct k-mer.
struct FtPassPred {      // occurrences of this pass: key (the hash, or its sub-table index) inside one of the W inclusive ranges bnd[2 d] .. bnd[2 d + 1]
	const uint64_t *bnd; int W, low12;
	__host__ __device__ bool operator()(const uint64_t &h) const {
		if (h == UINT64_MAX) return false;
		const uint64_t key = low12 ? (h & 4095) : h;
		for (int d = 0; d < W; ++d) if (key >= bnd[2 * d] && key <= bnd[2 * d + 1]) return true;
		return false;
	}
};
#define HAO_FT_BYTES_PER_SLOT 20.0              // two 8-byte occurrence buffers + 25 % (the sort's scratch, the run lists); hifiasm_amd/memplan.py uses the same figures
#define HAO_FT_BYTES_PER_SLOT_SHARDED 46.0
#define HAO_FT_RUN_BYTES_PER_SLOT 3.0
#define HAO_FT_BYTES_PER_SLOT_BLOOM 10.0        // on top, counting through the blocked Bloom filter (-f >= 21, the reference's default): hao_bloom_filter holds two 4-byte block ids and a flag per occurrence while both occurrence buffers live (+ its run table)
#define HAO_FT_CHUNK_SLOTS (1ULL << 28)      // k-mer slots hashed per chunk of reads in pass mode (2 GB of scratch, twice)
// passes needed so that the two occurrence buffers (+ 25 % for the sort's scratch and the run lists) fit into the free device memory; HAO_FT_PASSES forces a number (tests)
// (sharded: a pass keeps its two buffers while the receive buffer and its sort twin of about the same size exist: four buffers of a P-th, DevBuf slack included)
static uint64_t hao_ft_pass_count(hao_ctx *c, uint64_t n_slots, bool sharded, bool bloom)
{
	if (c->sw.ft_passes > 0) return (uint64_t)c->sw.ft_passes;
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 1; }
	const double per_slot = (sharded ? HAO_FT_BYTES_PER_SLOT_SHARDED : HAO_FT_BYTES_PER_SLOT) + (bloom ? HAO_FT_BYTES_PER_SLOT_BLOOM : 0.0);
	const double need = per_slot * (double)n_slots + (double)(1ULL << 30), have = 0.9 * (double)fr;
	// a pass's two occurrence buffers hold at most 2^32 slots (32 GB) each even when everything would fit at once: allocating (and first touching) a pair of 60 GB
	// buffers costs more than hashing the reads a second time - configs[2] (7.5 G slots): 1.07 s in one pass (3 - 6 s on a box where the allocation stalls), 0.65 s in
	// two, 0.80 in three (profiles/r06/ft_passes.txt)
	const uint64_t p_size = (n_slots + (1ULL << 32) - 1) >> 32;
	if (need <= have) return std::max<uint64_t>(1, std::min<uint64_t>(64, p_size));
	// what the chunk scratch and the run lists of ALL passes leave (12 bytes per distinct k-mer, once more while a pass's runs are appended; one distinct k-mer per ~7
	// occurrences is allowed for: 1 / coverage + error rate x k = 0.076 at 40x and 0.1 % - an exact count of noisy reads at this scale wants the Bloom filter, as in the reference)
	const double rest = have - 2.0 * 8.0 * (double)HAO_FT_CHUNK_SLOTS - (double)(2ULL << 30) - HAO_FT_RUN_BYTES_PER_SLOT * (double)n_slots;
	if (rest <= 0) return 64;
	return (uint64_t)std::min<double>(64.0, std::max<double>((double)p_size, std::ceil(per_slot * (double)n_slots / rest)));
}

static int hao_ft_run(hao_ctx *c)
{
	const uint64_t n = c->n_reads; const int k = c->opt.k;
	c->has_ft = false; c->h_ft_keys.clear(); c->h_ft_vals.clear();
	std::vector<uint32_t> slist;
	DevBuf<uint64_t> slots, chunks, kmer_off, kh_chunk_off, kh, kh2;
	uint64_t n_slots = 0, n_chunks = 0, n_real = 0;
	DevBuf<uint64_t> ukeys; DevBuf<uint32_t> ucnt; uint64_t n_unique = 0, *sorted = nullptr;
	const bool sharded = c->comm && c->comm->active();
	const bool bloom = c->opt.bf_shift >= 21;                  // ha_ct_init / yak_bf_init: a filter needs n_shift > pre and >= 2^9 bits per sub-table (htab.cpp:83,153), else exact counting
	uint64_t n_cnt = 0; uint64_t *cnt_in = nullptr, *cnt_alt = nullptr; uint32_t bias = 0;
	uint64_t P = 1, pass_cap = 0;      // hash-range passes (1: every occurrence at once); occurrences a pass buffer holds
	std::vector<uint64_t> h_kmer_off, h_chunk_off;      // pass mode: the per-read slot / workgroup offsets on the host (chunk boundaries)
	DevBuf<uint64_t> ch_tmp, ch_sel, d_bnd;             // pass mode: one chunk's hashes, the chunk's share of the pass, the pass's ranges
	// the hash kernels over reads [a, b) into out[kmer_off[r] - base] (base = kmer_off[a]: the chunk's first slot)
	auto hash_reads = [&](uint64_t a, uint64_t b, uint64_t slot_base, uint64_t chunk_base, uint64_t n_ch, uint64_t n_sl, uint64_t *out, uint64_t *scalar_real) -> int {
		if (scalar_real) *scalar_real = 0;
		const auto s0 = std::lower_bound(slist.begin(), slist.end(), (uint32_t)a), s1 = std::lower_bound(slist.begin(), slist.end(), (uint32_t)b);
		if (s0 != s1) {    // slots of N reads are upper bounds: sentinel-fill, count the real ones
			if (int rc = hao_memset_big(c, out, 0xff, n_sl * 8)) return rc;
			HIP_TRY(c->d_cursor.reserve(4)); HIP_TRY(hipMemsetAsync(c->d_cursor.p + 2, 0, 8, c->stream));
			const uint32_t ns = (uint32_t)(s1 - s0);
			hipLaunchKernelGGL(kmer_hash_scalar_kernel, dim3((ns + 63) / 64), dim3(64), 0, c->stream, c->d_packed.p, c->d_pk_off.p, c->d_len.p,
							   c->d_nsite_off.p, c->d_nsite.p, c->d_scalar_list.p + (s0 - slist.begin()), ns, kmer_off.p, (uint64_t)0, k, c->opt.hpc, out - slot_base, c->d_cursor.p + 2);
			HAO_CHECK_LAUNCH();
			if (scalar_real) { HIP_TRY(hipMemcpyAsync(scalar_real, c->d_cursor.p + 2, 8, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
		}
		if (n_ch) {
			hao_kh_args a_;
			a_.packed = c->d_packed.p; a_.pk_off = c->d_pk_off.p; a_.len = c->d_len.p; a_.tile_off = c->d_tile_off.p; a_.tile_ord = c->d_tile_ord.p; a_.n_runs = c->d_n_runs.p;
			a_.chunk_off = kh_chunk_off.p; a_.scalar_flag = c->d_scalar_flag.p; a_.kmer_off = kmer_off.p; a_.rid_lo = 0; a_.n_sel = n; a_.k = k; a_.hpc = c->opt.hpc; a_.out = out - slot_base; a_.ch0 = chunk_base;
			hipLaunchKernelGGL(kmer_hash_chunk_kernel, dim3((unsigned)n_ch), dim3(256), hao_kh_smem_bytes(k), c->stream, a_);
			HAO_CHECK_LAUNCH();
		}
		return HAO_OK;
	};
	// everything up to the first exchange is local: in sharded mode its status travels with the first collective (ranks fail together)
	auto local_index = [&]() -> int {
		if (n == 0) { hao_set_err(c, "no reads"); return HAO_EINVAL; }
		if (c->n_bases >= (1ULL << 32) * 16) { hao_set_err(c, "read set too large for one device pass"); return HAO_EUNSUPP; }
		if (int rc = hao_prepare_runs(c, 0, n, false, slist)) return rc;
		HIP_TRY(slots.reserve(n + 2)); HIP_TRY(chunks.reserve(n + 2)); HIP_TRY(kmer_off.reserve(n + 2)); HIP_TRY(kh_chunk_off.reserve(n + 2));
		hipLaunchKernelGGL(hao_kmer_slots_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream, c->d_n_runs.p, c->d_scalar_flag.p, c->d_len.p, (uint64_t)0, n, k, slots.p, chunks.p);
		HAO_CHECK_LAUNCH();
		if (int rc = hao_excl_scan_u64(c, slots.p, kmer_off.p, n + 1)) return rc;
		if (int rc = hao_excl_scan_u64(c, chunks.p, kh_chunk_off.p, n + 1)) return rc;
		HIP_TRY(hipMemcpyAsync(&n_slots, kmer_off.p + n, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpyAsync(&n_chunks, kh_chunk_off.p + n, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->timer.mark("ft_index");
		if (bloom && c->opt.bf_shift - 21 + 12 > 31) { hao_set_err(c, "bf_shift > 40 is not supported"); return HAO_EUNSUPP; }
		return HAO_OK;
	};
	const int index_rc = local_index();
	if (index_rc && !sharded) return index_rc;
	P = index_rc ? 1 : hao_ft_pass_count(c, n_slots, sharded, bloom);
	if (sharded) {      // every rank runs the same number of passes (the exchanges are collective): the largest any rank needs; the status of the local work travels along
		hao_comm &cm = *c->comm;
		if (int rc = hao_shard_layout_check(c, cm, index_rc)) return rc;
		std::vector<uint64_t> all; if (int rc = hao_comm_allgather_u64(c, cm, P, all, 0)) return rc;
		for (uint64_t v : all) P = std::max(P, v);
	}
	c->ft_passes_used = (int)P;
	// one pass's occurrences into kh[0 .. n_slots) (P == 1: every occurrence, in its slot, sentinels where a read with N has fewer k-mers than bases)
	auto local_hashes = [&](uint64_t pass) -> int {
		if (P == 1) {
			HIP_TRY(kh.reserve(n_slots + 1)); HIP_TRY(kh2.reserve(n_slots + 1));
			n_real = n_slots;
			uint64_t scalar_real = 0;
			if (int rc = hash_reads(0, n, 0, 0, n_chunks, n_slots, kh.p, &scalar_real)) return rc;
			if (!slist.empty()) { uint64_t scalar_slots = 0; for (uint32_t r : slist) scalar_slots += c->h_len[r]; n_real = n_slots - scalar_slots + scalar_real; }
		} else {
			if (pass == 0) {
				h_kmer_off.resize(n + 1); h_chunk_off.resize(n + 1);
				HIP_TRY(hipMemcpyAsync(h_kmer_off.data(), kmer_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipMemcpyAsync(h_chunk_off.data(), kh_chunk_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipStreamSynchronize(c->stream));
				const uint64_t all_slots = h_kmer_off[n];
				pass_cap = std::min<uint64_t>(all_slots, all_slots / P + all_slots / (4 * P) + (1ULL << 20));      // (hashes are uniform: a P-th + 25 %; a tiny input simply gets room for everything)
				uint64_t biggest = 0; for (uint64_t r = 0; r < n; ++r) biggest = std::max(biggest, h_kmer_off[r + 1] - h_kmer_off[r]);
				const uint64_t chs = std::min<uint64_t>(all_slots, std::max<uint64_t>(c->sw.ft_chunk_slots ? (uint64_t)c->sw.ft_chunk_slots : HAO_FT_CHUNK_SLOTS, biggest));
				HIP_TRY(ch_tmp.reserve_exact(chs + 1)); HIP_TRY(ch_sel.reserve_exact(chs + 1)); HIP_TRY(kh.reserve_exact(pass_cap + 1)); HIP_TRY(kh2.reserve_exact(pass_cap + 1));
				HIP_TRY(d_bnd.reserve(2 * 4096 + 2)); HIP_TRY(c->d_cursor.reserve(4));
			}
			// the pass's ranges, one per owner (a rank's share of the hash space / of the 4096 sub-tables), as the partition below cuts them
			const int W = sharded ? c->comm->world : 1; std::vector<uint64_t> bnd(2 * (size_t)W);
			for (int d = 0; d < W; ++d) {
				unsigned __int128 lo, hi;      // the owner's keys [lo, hi)
				if (!bloom) { lo = d == 0 ? 0 : (((unsigned __int128)d << 64) / (unsigned)W); hi = d + 1 == W ? ((unsigned __int128)1 << 64) : (((unsigned __int128)(d + 1) << 64) / (unsigned)W); }
				else { lo = ((uint64_t)d * 4096 + W - 1) / W; hi = d + 1 == W ? 4096 : ((uint64_t)(d + 1) * 4096 + W - 1) / W; }
				const unsigned __int128 span = hi - lo, a_ = lo + span * pass / P, b_ = lo + span * (pass + 1) / P;
				if (a_ == b_) { bnd[2 * d] = 1; bnd[2 * d + 1] = 0; } else { bnd[2 * d] = (uint64_t)a_; bnd[2 * d + 1] = (uint64_t)(b_ - 1); }
			}
			HIP_TRY(hipMemcpyAsync(d_bnd.p, bnd.data(), bnd.size() * 8, hipMemcpyHostToDevice, c->stream));
			const FtPassPred pred{d_bnd.p, W, bloom ? 1 : 0};
			uint64_t cnt = 0; const uint64_t chs = ch_tmp.cap - 1;
			for (uint64_t a = 0; a < n; ) {
				uint64_t b = a + 1; while (b < n && h_kmer_off[b + 1] - h_kmer_off[a] <= chs) ++b;      // reads [a, b): at most chs slots (one read always fits)
				const uint64_t sl = h_kmer_off[b] - h_kmer_off[a], nch = h_chunk_off[b] - h_chunk_off[a];
				if (sl) {
					if (int rc = hash_reads(a, b, h_kmer_off[a], h_chunk_off[a], nch, sl, ch_tmp.p, nullptr)) return rc;
					size_t tb = 0; uint64_t sel = 0;
					HIP_TRY(rocprim::select(nullptr, tb, ch_tmp.p, ch_sel.p, (uint64_t*)c->d_cursor.p, (size_t)sl, pred, c->stream)); HIP_TRY(hao_tmp(c, tb));
					HIP_TRY(rocprim::select(c->d_tmp.p, tb, ch_tmp.p, ch_sel.p, (uint64_t*)c->d_cursor.p, (size_t)sl, pred, c->stream));
					HIP_TRY(hipMemcpyAsync(&sel, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
					HIP_TRY(hipStreamSynchronize(c->stream));
					if (cnt + sel > pass_cap) { hao_set_err(c, "ha_ft_gen: a hash-range pass holds more k-mer occurrences than its buffer (set HAO_FT_PASSES higher)"); return HAO_ENOMEM; }
					if (sel) HIP_TRY(hipMemcpyAsync(kh.p + cnt, ch_sel.p, sel * 8, hipMemcpyDeviceToDevice, c->stream));
					cnt += sel;
				}
				a = b;
			}
			HIP_TRY(hipStreamSynchronize(c->stream));
			n_slots = cnt; n_real = cnt;      // (no sentinels in a pass's stream)
		}
		c->timer.mark("ft_hash");
		n_cnt = n_slots; cnt_in = kh.p; cnt_alt = kh2.p; bias = 0;
		if (bloom) {
			if (!sharded) { if (int rc = hao_bloom_filter(c, kh.p, kh2.p, n_slots, &cnt_in, &cnt_alt, &n_cnt)) return rc; }
			bias = 1;                                                 // the entry is created with count 1, then incremented (htab.cpp:201-205)
			c->timer.mark("ft_bloom");
		}
		return HAO_OK;
	};
	DevBuf<uint64_t> all_keys; DevBuf<uint32_t> all_cnt; uint64_t all_unique = 0; int64_t all_hist[HAO_N_COUNTS];      // pass mode: the passes' run lists, one after the other
	memset(all_hist, 0, sizeof(all_hist));
	for (uint64_t pass = 0; pass < P; ++pass) {
	const int hash_rc = local_hashes(pass);
	if (hash_rc && !sharded) return hash_rc;
	if (!sharded) {
		// sentinels (0xff..ff) sort to the end: only the first n_real entries are real k-mers
		if (int rc = hao_sort_rle_hist(c, cnt_in, cnt_alt, n_cnt, ukeys, ucnt, &n_unique, c->ft_hist, &sorted, bias)) return rc;
		if (!bloom && n_real < n_slots && n_unique) { --n_unique; c->ft_hist[std::min<uint64_t>(n_slots - n_real, HAO_MAX_COUNT)] -= 1; }   // drop the sentinel run (the Bloom replay already dropped it)
	} else {
		// hash-range partition: sort local hashes, cut at i * 2^64 / world, all-to-all-v, count the owned range.
		// Local phases are lambdas whose status travels with the next collective (hao_comm.hpp): ranks fail together, nobody hangs.
		hao_comm &cm = *c->comm; const int W = cm.world;
		uint64_t *loc = kh.p; uint64_t n_loc = n_real;
		std::vector<uint64_t> tg(W), cut(W + 1, 0), scnt(W, 0), sdisp(W, 0), rcnt;
		DevBuf<uint64_t> dt, dc, rv, rv2;
		auto local_partition = [&]() -> int {
			if (hash_rc) return hash_rc;
			HIP_TRY(dt.reserve(W + 1)); HIP_TRY(dc.reserve(W + 1));
			if (!bloom) {
				if (n_slots) {
					size_t tb = 0; rocprim::double_buffer<uint64_t> db(kh.p, kh2.p);
					HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, db, n_slots, 0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
					HIP_TRY(rocprim::radix_sort_keys(c->d_tmp.p, tb, db, n_slots, 0, 64, c->stream));
					loc = db.current();
				}
				for (int d = 0; d < W; ++d) tg[d] = d == 0 ? 0 : (uint64_t)(((unsigned __int128)d << 64) / (unsigned)W);
				HIP_TRY(hipMemcpyAsync(dt.p, tg.data(), 8 * W, hipMemcpyHostToDevice, c->stream));
				hipLaunchKernelGGL(hao_lower_bound_kernel, dim3((W + 63) / 64), dim3(64), 0, c->stream, loc, n_real, dt.p, W, dc.p);
				HAO_CHECK_LAUNCH();
			} else {
				// the filter's blocks (and every copy of a k-mer) are determined by the LOW hash bits: partition by sub-table (low 12 bits), with a
				// STABLE sort on those bits only, so that each piece stays in (read, position) order; pieces arrive in rank order = global read order
				if (n_slots) {
					HIP_TRY(c->d_cursor.reserve(4)); size_t tb = 0;
					HIP_TRY(rocprim::select(nullptr, tb, kh.p, kh2.p, (uint64_t*)c->d_cursor.p, n_slots, NotSentinel(), c->stream)); HIP_TRY(hao_tmp(c, tb));
					HIP_TRY(rocprim::select(c->d_tmp.p, tb, kh.p, kh2.p, (uint64_t*)c->d_cursor.p, n_slots, NotSentinel(), c->stream));
					HIP_TRY(hipMemcpyAsync(&n_loc, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
					HIP_TRY(hipStreamSynchronize(c->stream));
					tb = 0; rocprim::double_buffer<uint64_t> db(kh2.p, kh.p);
					HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, db, n_loc, 0, 12, c->stream)); HIP_TRY(hao_tmp(c, tb));
					HIP_TRY(rocprim::radix_sort_keys(c->d_tmp.p, tb, db, n_loc, 0, 12, c->stream));
					loc = db.current();
				}
				for (int d = 0; d < W; ++d) tg[d] = ((uint64_t)d * 4096 + W - 1) / W;      // first sub-table of rank d
				HIP_TRY(hipMemcpyAsync(dt.p, tg.data(), 8 * W, hipMemcpyHostToDevice, c->stream));
				hipLaunchKernelGGL(hao_lower_bound_low12_kernel, dim3((W + 63) / 64), dim3(64), 0, c->stream, loc, n_loc, dt.p, W, dc.p);
				HAO_CHECK_LAUNCH();
			}
			HIP_TRY(hipMemcpyAsync(cut.data(), dc.p, 8 * W, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			cut[W] = n_loc;
			for (int d = 0; d < W; ++d) { sdisp[d] = cut[d]; scnt[d] = cut[d + 1] - cut[d]; }
			return HAO_OK;
		};
		if (int rc = hao_comm_exchange_counts(c, cm, scnt, rcnt, local_partition())) return rc;
		uint64_t n_recv = 0; for (int d = 0; d < W; ++d) n_recv += rcnt[d];
		// peak memory: the hash buffers are the largest allocations of the whole engine (8 B per base each, divided by the number of passes).  Only the sorted local
		// piece and the receive buffer exist during the exchange; the second receive-side buffer (sort scratch) is allocated after the local piece is gone.
		auto local_recv_bufs = [&]() -> int { if (P == 1) { if (loc == kh.p) kh2.release(); else kh.release(); } HIP_TRY(rv.reserve(n_recv + 1)); return HAO_OK; };
		if (int rc = hao_comm_agree(c, cm, local_recv_bufs())) return rc;
		if (int rc = hao_comm_alltoallv_u64(c, cm, loc, scnt, sdisp, rv.p, rcnt)) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->timer.mark("ft_exchange");
		auto local_count = [&]() -> int {
			if (P == 1) { kh.release(); kh2.release(); }      // (pass mode keeps its two pass buffers for the next pass)
			HIP_TRY(rv2.reserve(n_recv + 1));
			uint64_t *ci = rv.p, *ca = rv2.p, n_ci = n_recv;
			if (bloom) { if (int rc = hao_bloom_filter(c, rv.p, rv2.p, n_recv, &ci, &ca, &n_ci)) return rc; }
			return hao_sort_rle_hist(c, ci, ca, n_ci, ukeys, ucnt, &n_unique, c->ft_hist, &sorted, bias);
		};
		if (int rc = hao_comm_allreduce_i64(c, cm, c->ft_hist, HAO_N_COUNTS, local_count())) return rc;
		rv.release(); rv2.release(); dt.release(); dc.release();
	}
	if (P > 1) {      // this pass's runs behind the earlier ones
		if (all_unique + n_unique + 1 > all_keys.cap) {
			const uint64_t want = std::max<uint64_t>((all_unique + n_unique) * (pass + 1 < P ? 2 : 1) + 64, (n_unique + 64) * (P - pass));
			DevBuf<uint64_t> nk_; DevBuf<uint32_t> nc_; HIP_TRY(nk_.reserve_exact(want)); HIP_TRY(nc_.reserve_exact(want));
			if (all_unique) { HIP_TRY(hipMemcpyAsync(nk_.p, all_keys.p, all_unique * 8, hipMemcpyDeviceToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(nc_.p, all_cnt.p, all_unique * 4, hipMemcpyDeviceToDevice, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
			std::swap(all_keys, nk_); std::swap(all_cnt, nc_); nk_.release(); nc_.release();
		}
		if (n_unique) { HIP_TRY(hipMemcpyAsync(all_keys.p + all_unique, ukeys.p, n_unique * 8, hipMemcpyDeviceToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(all_cnt.p + all_unique, ucnt.p, n_unique * 4, hipMemcpyDeviceToDevice, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
		all_unique += n_unique;
		for (int i = 0; i < HAO_N_COUNTS; ++i) all_hist[i] += c->ft_hist[i];
	}
	}      // passes
	if (P > 1) {
		kh.release(); kh2.release(); ch_tmp.release(); ch_sel.release(); ukeys.release(); ucnt.release();
		if (bloom && all_unique) {      // passes own sub-tables, not hash ranges: the concatenation is not sorted by key yet
			DevBuf<uint64_t> k2; DevBuf<uint32_t> c2; HIP_TRY(k2.reserve_exact(all_unique + 1)); HIP_TRY(c2.reserve_exact(all_unique + 1));
			size_t tb = 0; rocprim::double_buffer<uint64_t> dk(all_keys.p, k2.p); rocprim::double_buffer<uint32_t> dv(all_cnt.p, c2.p);
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, all_unique, 0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
			HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, dk, dv, all_unique, 0, 64, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (dk.current() != all_keys.p) std::swap(all_keys, k2);
			if (dv.current() != all_cnt.p) std::swap(all_cnt, c2);
			k2.release(); c2.release();
		}
		std::swap(ukeys, all_keys); std::swap(ucnt, all_cnt); n_unique = all_unique;
		memcpy(c->ft_hist, all_hist, sizeof(all_hist));
	}
	c->timer.mark("ft_count");
	kh.release(); kh2.release();
	c->ft_peak_hom = hao_find_peaks(c->ft_hist, HAO_N_COUNTS, c->opt.min_hist_cnt, &c->ft_peak_het, hao_prior_hom(c));
	int cutoff = (int)(c->ft_peak_hom * c->opt.high_factor);                 // htab.cpp:1160
	if (cutoff > HAO_MAX_COUNT - 1) cutoff = HAO_MAX_COUNT - 1;
	c->ft_cutoff = cutoff;
	DevBuf<uint32_t> kcnt; uint64_t n_kept = 0;
	const int keep_rc = hao_keep_runs(c, ukeys.p, ucnt.p, n_unique, cutoff, HAO_MAX_COUNT, c->d_ft_keys, nullptr, kcnt, &n_kept, nullptr);
	if (keep_rc && !sharded) return keep_rc;
	ukeys.release(); ucnt.release(); c->w_flag.release(); c->w_kpos.release(); c->w_ustart.release();      // k-mer sized scratch: do not keep it
	if (sharded) {   // every rank kept its hash range: concatenation in rank order is the globally sorted table
		hao_comm &cm = *c->comm; std::vector<uint64_t> cnts;
		if (int rc = hao_comm_allgather_u64(c, cm, n_kept, cnts, keep_rc)) return rc;
		uint64_t tot = 0; for (uint64_t v : cnts) tot += v;
		DevBuf<uint64_t> gk; DevBuf<uint32_t> gc;
		auto local_bufs = [&]() -> int { HIP_TRY(gk.reserve(tot + 1)); HIP_TRY(gc.reserve(tot + 1)); return HAO_OK; };
		if (int rc = hao_comm_agree(c, cm, local_bufs())) return rc;
		if (int rc = hao_comm_allgatherv(c, cm, c->d_ft_keys.p, n_kept, 8, gk.p, cnts)) return rc;
		if (int rc = hao_comm_allgatherv(c, cm, kcnt.p, n_kept, 4, gc.p, cnts)) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (bloom && tot) {      // ranks own sub-tables, not hash ranges: the concatenation is not sorted by key yet
			DevBuf<uint64_t> gk2; DevBuf<uint32_t> gc2; HIP_TRY(gk2.reserve(tot + 1)); HIP_TRY(gc2.reserve(tot + 1));
			size_t tb = 0; rocprim::double_buffer<uint64_t> dk(gk.p, gk2.p); rocprim::double_buffer<uint32_t> dv(gc.p, gc2.p);
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, tot, 0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
			HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, dk, dv, tot, 0, 64, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (dk.current() != gk.p) std::swap(gk, gk2);
			if (dv.current() != gc.p) std::swap(gc, gc2);
			gk2.release(); gc2.release();
		}
		std::swap(c->d_ft_keys, gk); std::swap(kcnt, gc); gk.release(); gc.release();
		n_kept = tot;
	}
	// map values (gen_hh, htab.cpp:1038-1062) as ha_ft_cnt returns them (htab.cpp:1064-1070)
	int max_cnt = c->opt.max_kmer_cnt;
	if (max_cnt > HAO_MAX_COUNT - 1) max_cnt = HAO_MAX_COUNT - 1;
	if (max_cnt > INT16_MAX - 1) max_cnt = INT16_MAX - 1;
	c->h_ft_keys.resize(n_kept); c->h_ft_vals.resize(n_kept);
	std::vector<uint32_t> hc(n_kept);
	if (n_kept) {
		HIP_TRY(hipMemcpy(c->h_ft_keys.data(), c->d_ft_keys.p, n_kept * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hc.data(), kcnt.p, n_kept * 4, hipMemcpyDeviceToHost));
	}
	for (uint64_t i = 0; i < n_kept; ++i) c->h_ft_vals[i] = (int)hc[i] > max_cnt ? INT32_MAX : (int32_t)hc[i];
	HIP_TRY(c->d_ft_vals.reserve(n_kept + 1)); HIP_TRY(c->d_ft_keys.reserve(n_kept + 1));
	if (n_kept) HIP_TRY(hipMemcpyAsync(c->d_ft_vals.p, c->h_ft_vals.data(), n_kept * 4, hipMemcpyHostToDevice, c->stream));
	if (int rc = hao_build_bucket(c, c->d_ft_keys.p, n_kept, 16, c->d_ft_bucket)) return rc;
	if (int rc = hao_ft_build_hash(c, n_kept)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	kcnt.release(); slots.release(); chunks.release(); kmer_off.release(); kh_chunk_off.release();
	c->has_ft = true;
	{	// ha_opt_update_cov
		int mx = (int)(c->ft_peak_hom * c->opt.high_factor + .499);
		c->hom_cov = c->ft_peak_hom;
		if (c->max_n_chain < mx) c->max_n_chain = mx;
	}
	c->timer.mark("ft_table");
	return HAO_OK;
}

// (hash, read-order index) pairs in stable hash order: x[] -> sx[], iota -> oi2[] (oi[] is scratch).  sort40: 5 radix passes over hash bits 24 .. 63, then the
// fix-up of the 40-bit runs that hold more than one key (hao_index.cuh); its counters (runs listed, scratch used, scratch overflow) land in c->peek_h[16 .. 18]
// once the stream has been synchronised - the caller sorts again with sort40 = false if peek_h[16] > hao_sort40_cap(m) or peek_h[18] != 0.
static inline uint64_t hao_sort40_cap(uint64_t m) { return std::max<uint64_t>(1 << 16, m >> 10); }      // dirty runs listed (expected: m^2 / 2^41 runs of ~2 keys)
static int hao_index_sort(hao_ctx *c, const uint64_t *x, uint64_t *sx, uint32_t *oi, uint32_t *oi2, uint64_t m, bool sort40)
{
	hipLaunchKernelGGL(hao_iota_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, oi, m); HAO_CHECK_LAUNCH();
	const int b0 = sort40 ? HAO_SORT40_LOWBITS : 0;
	size_t tb = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, x, sx, oi, oi2, m, b0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, x, sx, oi, oi2, m, b0, 64, c->stream));
	if (sort40) {
		const uint64_t cap = hao_sort40_cap(m), tcap = 64 * cap;
		HIP_TRY(c->w_s40_list.reserve(cap)); HIP_TRY(c->w_s40_x.reserve(tcap)); HIP_TRY(c->w_s40_o.reserve(tcap)); HIP_TRY(c->w_s40_cnt.reserve(4));
		HIP_TRY(hipMemsetAsync(c->w_s40_cnt.p, 0, 32, c->stream));
		hipLaunchKernelGGL(hao_sort40_mark_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, sx, m, c->w_s40_list.p, c->w_s40_cnt.p, cap); HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL(hao_sort40_fix_kernel, dim3((unsigned)cap), dim3(64), 0, c->stream, sx, oi2, m, c->w_s40_list.p, c->w_s40_cnt.p, cap, c->w_s40_x.p, c->w_s40_o.p, tcap); HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)c->w_s40_cnt.p, 3, c->peek_d + 16); HAO_CHECK_LAUNCH();
	}
	return HAO_OK;
}

// ---------------------------------------------------------------------------------------
// ha_pt_gen (htab.cpp:1232-1287): sketch all reads (once - the reference does it twice, pass A to
// count and pass B to insert), stable sort by hash, run-length -> histogram -> peaks -> keep keys
// occurring 2..4094 times (2..cutoff without a filter table).
// ---------------------------------------------------------------------------------------
static int hao_pt_run(hao_ctx *c)
{
	const uint64_t n = c->n_reads;
	c->has_pt = false; c->h_ix_valid = false; c->h_ix_mz_off.clear();
	const bool sharded = c->comm && c->comm->active();
	// local: sketch every read; in sharded mode the status travels with the first collective (ranks fail together, hao_comm.hpp)
	auto local_sketch = [&]() -> int {
		if (n == 0) { hao_set_err(c, "no reads"); return HAO_EINVAL; }
		if (int rc = hao_sketch_run(c, 0, n, c->has_ft, c->opt.sample_dist, 1)) return rc;
		// the read-ordered minimizers of the LOCAL reads stay for the query side (the reference re-sketches every query read; same result)
		std::swap(c->d_ix_mz_x, c->d_mz_x); std::swap(c->d_ix_mz_info, c->d_mz_info); std::swap(c->d_ix_mz_off, c->d_mz_off);
		// the two buffer sets alternate between "index" and "next sketch": give the idle twin its size now, not in the middle of the next pass
		HIP_TRY(c->d_mz_x.reserve_exact(c->d_ix_mz_x.cap)); HIP_TRY(c->d_mz_info.reserve_exact(c->d_ix_mz_info.cap)); HIP_TRY(c->d_mz_off.reserve_exact(c->d_ix_mz_off.cap));
		c->ix_n_mz = c->sk_total; c->sk_n = 0;
		return HAO_OK;
	};
	const int sk_rc = local_sketch();
	if (sk_rc && !sharded) return sk_rc;
	DevBuf<uint64_t> &ukeys = c->w_ukeys; DevBuf<uint32_t> &ucnt = c->w_ucnt; uint64_t n_unique = 0;
	memset(c->pt_hist, 0, sizeof(c->pt_hist));
	auto rle_hist = [&](const uint64_t *sorted, uint64_t cnt) -> int {       // -> ukeys/ucnt/n_unique, c->pt_hist (local)
		n_unique = 0;
		if (!cnt) return HAO_OK;
		HIP_TRY(ukeys.reserve(cnt + 1)); HIP_TRY(ucnt.reserve(cnt + 1)); HIP_TRY(c->d_cursor.reserve(2));
		if (int rc = hao_rle(c, sorted, cnt, ukeys.p, ucnt.p, (uint64_t*)c->d_cursor.p)) return rc;
		HIP_TRY(hipMemcpyAsync(&n_unique, c->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		DevBuf<unsigned long long> &dh = c->w_hist; HIP_TRY(dh.reserve(HAO_N_COUNTS)); HIP_TRY(hipMemsetAsync(dh.p, 0, HAO_N_COUNTS * 8, c->stream));
		unsigned nb = (unsigned)std::min<uint64_t>((n_unique + 255) / 256, 2048);
		if (nb) hipLaunchKernelGGL(hao_count_hist_kernel, dim3(nb), dim3(256), 0, c->stream, ucnt.p, n_unique, dh.p);
		HAO_CHECK_LAUNCH();
		HIP_TRY(hipMemcpyAsync(c->pt_hist, dh.p, HAO_N_COUNTS * 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		return HAO_OK;
	};
	auto peaks_and_range = [&](int *hi_out) {
		int het = -1;
		c->hom_cov = hao_find_peaks(c->pt_hist, HAO_N_COUNTS, c->opt.min_hist_cnt, &het, hao_prior_hom(c)); c->het_cov = het;
		int hi;
		if (c->has_ft) hi = HAO_MAX_COUNT - 1;                                   // htab.cpp:1266-1269
		else { hi = (int)(c->hom_cov * c->opt.high_factor); if (hi > HAO_MAX_COUNT - 1) hi = HAO_MAX_COUNT - 1; }   // :1258-1262
		*hi_out = hi;
	};
	c->lk_valid = false;
	if (!sharded) {
		// hash order through a (hash, read-order index) sort: 12 bytes per element and pass instead of 16, and the index is what lets every
		// minimizer learn its own lookup result at build time (hao_index.cuh)
		const uint64_t m = c->ix_n_mz;
		if (m >= (1ULL << 32)) { hao_set_err(c, "more than 2^32 minimizers on one device"); return HAO_EUNSUPP; }
		const uint64_t pad = c->ix_pad = c->sw.ix_pad;      // (tests: the index's position records start `pad` entries into their buffer, so list starts exceed 2^32 on a small read set)
		HIP_TRY(c->d_ix_sx.reserve(m + 1)); HIP_TRY(c->d_ix_sinfo.reserve(m + pad + 8)); HIP_TRY(c->w_oi.reserve(m + 1)); HIP_TRY(c->w_oi2.reserve(m + 1));      // (sinfo + 8: the merge kernel reads 32 bytes at a time, up to three records past the last list)
		HIP_TRY(c->w_runid.reserve(m + 1)); HIP_TRY(c->d_ix_lk.reserve(m + 1));
		// big inputs: 5 passes over hash bits 24 .. 63 + the fix-up of the 40-bit runs that hold two keys (hao_index_sort).  Small ones sort all 64 bits (rocprim's
		// bit-range sort mis-sorts inputs of 5 k - 200 k elements on this ROCm, tests/test_gpu_rocprim.py); so does a retry when the fix-up's scratch ran out.
		bool sort40 = m >= c->sw.sort40_min;
		for (;;) {
			if (m) { if (int rc = hao_index_sort(c, c->d_ix_mz_x.p, c->d_ix_sx.p, c->w_oi.p, c->w_oi2.p, m, sort40)) return rc; }
			c->ix_n_sorted = m;
			c->timer.mark("pt_sort");
			if (int rc = rle_hist(c->d_ix_sx.p, m)) return rc;      // (synchronises the stream)
			if (m && sort40) {
				c->s40_runs = c->peek_h[16];
				if (c->peek_h[16] > hao_sort40_cap(m) || c->peek_h[18]) { sort40 = false; continue; }      // more two-key runs than the fix-up was sized for: all 64 bits after all
			}
			break;
		}
		c->timer.mark("pt_count");
		int hi; peaks_and_range(&hi);
		if (int rc = hao_keep_runs(c, ukeys.p, ucnt.p, n_unique, 2, hi, c->d_ix_keys, &c->d_ix_start, c->d_ix_cnt, &c->ix_n_keys, &c->ix_n_pos)) return rc;
		if (m) {      // run id of every sorted position (inclusive count of run heads), then one gather + scatter pass (c->w_ustart = run starts, left by hao_keep_runs)
			auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RunHead32{c->d_ix_sx.p});
			size_t tb = 0;
			HIP_TRY(rocprim::inclusive_scan(nullptr, tb, heads, c->w_runid.p, m, rocprim::plus<uint32_t>(), c->stream)); HIP_TRY(hao_tmp(c, tb));
			HIP_TRY(rocprim::inclusive_scan(c->d_tmp.p, tb, heads, c->w_runid.p, m, rocprim::plus<uint32_t>(), c->stream));
			if (m >= c->sw.sort40_min) {      // big index: gather, one radix pass on the top 8 bits of the read-order index, windowed scatter (hao_index.cuh)
				int nb = 1; while ((1ULL << nb) < m) ++nb;
				const int b0 = nb > 8 ? nb - 8 : 0;
				HIP_TRY(c->w_lkv2.reserve(m + 1));
				hipLaunchKernelGGL(hao_index_gather_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, m, c->w_oi2.p, c->w_runid.p, ucnt.p, c->w_ustart.p, 2, hi,
								   c->d_ix_mz_info.p, c->d_ix_sinfo.p + pad, c->d_ix_sx.p, pad);      // (the sorted hashes are dead: their array takes the results in hash order)
				HAO_CHECK_LAUNCH();
				size_t tb2 = 0;
				HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, c->w_oi2.p, c->w_oi.p, c->d_ix_sx.p, c->w_lkv2.p, m, b0, nb, c->stream)); HIP_TRY(hao_tmp(c, tb2));
				HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb2, c->w_oi2.p, c->w_oi.p, c->d_ix_sx.p, c->w_lkv2.p, m, b0, nb, c->stream));
				hipLaunchKernelGGL(hao_scatter_u64_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, c->w_lkv2.p, c->w_oi.p, m, c->d_ix_lk.p);
				HAO_CHECK_LAUNCH();
			} else {
				hipLaunchKernelGGL(hao_index_finish_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, m, c->w_oi2.p, c->w_runid.p, ucnt.p, c->w_ustart.p, 2, hi,
								   c->d_ix_mz_info.p, c->d_ix_sinfo.p + pad, c->d_ix_lk.p, pad);
				HAO_CHECK_LAUNCH();
			}
			if (pad && c->ix_n_keys) { hipLaunchKernelGGL(hao_add_const_kernel, dim3((unsigned)((c->ix_n_keys + 255) / 256)), dim3(256), 0, c->stream, c->d_ix_start.p, c->ix_n_keys, pad); HAO_CHECK_LAUNCH(); }
		}
		c->lk_valid = true;
		c->timer.mark("pt_lookup");
	} else {
		// Sharded build (SURVEY 2 C1 + 8e layout i): every rank owns the hash range [r, r+1) * 2^64 / world.
		//   local stable grouping by owner -> all-to-all-v of (x, info) by range -> stable sort of the received pieces (source-rank order =
		//   global read order, so per-key lists come out in (rid,pos) order) -> count / histogram (all-reduced) / peaks / keep ->
		//   all-gather of the sorted position records (8 B each) and of the key tables: concatenation in rank order is the global index;
		//   the owner also answers the lookup of every minimizer it received ((list start, count) of its run: 8 bytes back through the reverse all-to-all-v),
		//   so the query pass of a sharded engine unpacks its lookups exactly like a single device does - no per-batch search of the key table.
		// Local phases are lambdas whose status travels with the next collective: ranks fail together.
		hao_comm &cm = *c->comm; const int W = cm.world;
		if (int rc = hao_shard_layout_check(c, cm, sk_rc)) return rc;
		const uint64_t ml = c->ix_n_mz;
		DevBuf<uint64_t> lsx, lsi, dt, dc, rx, ri, px, pi, lkr, lkl; DevBuf<uint32_t> ai, ai2; const uint32_t *lperm = nullptr;
		std::vector<uint64_t> tg(W), cut(W + 1, 0), scnt(W, 0), sdisp(W, 0), rcnt;
		// owner of a hash = ((x >> 48) * W) >> 16: ranges of the top 16 bits, so the local pass only groups by those bits - a stable 2-pass radix sort of
		// (owner bits, index) and one gather instead of 8 passes over the 16-byte records; pieces stay in read order, the owner sorts what it receives.
		// (begin_bit = 0 on a separate 16-bit key: rocprim 4.2's radix sort mis-sorts small inputs when begin_bit > 0; tests/test_gpu_rocprim.py pins that.)
		auto local_group = [&]() -> int {
			HIP_TRY(lsx.reserve(ml + 1)); HIP_TRY(lsi.reserve(ml + 1));
			if (ml) {
				DevBuf<uint32_t> &ok = c->w_ok, &ok2 = c->w_ok2, &oi = c->w_oi, &oi2 = c->w_oi2;      // persistent scratch: no allocation inside the pass
				HIP_TRY(ok.reserve(ml + 1)); HIP_TRY(ok2.reserve(ml + 1)); HIP_TRY(oi.reserve(ml + 1)); HIP_TRY(oi2.reserve(ml + 1));
				hipLaunchKernelGGL(hao_owner_key_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, c->stream, c->d_ix_mz_x.p, ml, ok.p, oi.p);
				HAO_CHECK_LAUNCH();
				size_t tb = 0; rocprim::double_buffer<uint32_t> dk(ok.p, ok2.p), dv(oi.p, oi2.p);
				HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, ml, 0, 16, c->stream)); HIP_TRY(hao_tmp(c, tb));
				HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, dk, dv, ml, 0, 16, c->stream));
				hipLaunchKernelGGL(hao_gather2_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, c->stream, dv.current(), c->d_ix_mz_x.p, c->d_ix_mz_info.p, ml, lsx.p, lsi.p);
				HAO_CHECK_LAUNCH();
				lperm = dv.current();      // (c->w_oi or c->w_oi2: untouched until the lookup results come back in this order)
			}
			for (int d = 0; d < W; ++d) tg[d] = (((uint64_t)d * 65536 + W - 1) / W) << 48;      // first hash owned by rank d (low 48 bits zero: comparing whole keys orders by the top 16 bits)
			HIP_TRY(dt.reserve(W + 1)); HIP_TRY(dc.reserve(W + 1));
			HIP_TRY(hipMemcpyAsync(dt.p, tg.data(), 8 * W, hipMemcpyHostToDevice, c->stream));
			hipLaunchKernelGGL(hao_lower_bound_kernel, dim3((W + 63) / 64), dim3(64), 0, c->stream, lsx.p, ml, dt.p, W, dc.p);
			HAO_CHECK_LAUNCH();
			HIP_TRY(hipMemcpyAsync(cut.data(), dc.p, 8 * W, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			cut[W] = ml;
			for (int d = 0; d < W; ++d) { sdisp[d] = cut[d]; scnt[d] = cut[d + 1] - cut[d]; }
			return HAO_OK;
		};
		if (int rc = hao_comm_exchange_counts(c, cm, scnt, rcnt, local_group())) return rc;
		uint64_t n_recv = 0; for (int d = 0; d < W; ++d) n_recv += rcnt[d];
		auto local_recv_bufs = [&]() -> int { HIP_TRY(rx.reserve(n_recv + 1)); HIP_TRY(ri.reserve(n_recv + 1)); HIP_TRY(px.reserve(n_recv + 1)); HIP_TRY(pi.reserve(n_recv + 1)); return HAO_OK; };
		if (int rc = hao_comm_agree(c, cm, local_recv_bufs())) return rc;
		if (int rc = hao_comm_alltoallv2_u64(c, cm, lsx.p, lsi.p, scnt, sdisp, rx.p, ri.p, rcnt)) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->timer.mark("pt_alltoall");
		auto local_count = [&]() -> int {
			if (n_recv >= (1ULL << 32)) { hao_set_err(c, "more than 2^32 minimizers in one hash partition"); return HAO_EUNSUPP; }
			// (hash, arrival index) sort, as on a single device: the arrival index gathers the records and addresses the answers
			HIP_TRY(ai.reserve(n_recv + 1)); HIP_TRY(ai2.reserve(n_recv + 1));
			if (n_recv) {
				hipLaunchKernelGGL(hao_iota_kernel, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, c->stream, ai.p, n_recv); HAO_CHECK_LAUNCH();
				size_t tb = 0;
				HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, rx.p, px.p, ai.p, ai2.p, n_recv, 0, 64, c->stream)); HIP_TRY(hao_tmp(c, tb));
				HIP_TRY(rocprim::radix_sort_pairs(c->d_tmp.p, tb, rx.p, px.p, ai.p, ai2.p, n_recv, 0, 64, c->stream));
			}
			c->timer.mark("pt_sort");
			return rle_hist(px.p, n_recv);
		};
		if (int rc = hao_comm_allreduce_i64(c, cm, c->pt_hist, HAO_N_COUNTS, local_count())) return rc;
		c->timer.mark("pt_count");
		int hi; peaks_and_range(&hi);
		DevBuf<uint64_t> pk, pst; DevBuf<uint32_t> pc; uint64_t nk_p = 0, np_p = 0;
		const int keep_rc = hao_keep_runs(c, ukeys.p, ucnt.p, n_unique, 2, hi, pk, &pst, pc, &nk_p, &np_p);
		// global layout
		std::vector<uint64_t> part(W), nks(W), nps(W), trip;
		{ const uint64_t mine[3] = { n_recv, nk_p, np_p };
		  if (int rc = hao_comm_allgather_u64n(c, cm, mine, 3, trip, keep_rc)) return rc;
		  for (int r = 0; r < W; ++r) { part[r] = trip[3 * r]; nks[r] = trip[3 * r + 1]; nps[r] = trip[3 * r + 2]; } }
		uint64_t m = 0, base = 0, nk = 0, np = 0;
		for (int r = 0; r < W; ++r) { if (r < cm.rank) base += part[r]; m += part[r]; nk += nks[r]; np += nps[r]; }
		// (no limit on m: list starts are 48-bit everywhere - lk[], d_ix_start, the seed kernels' staged words; only a PARTITION is limited to 2^32 records, by its
		// sort's 32-bit arrival index, and a partition of that size - ~60 bytes per record while it is built - would not fit a device's memory anyway)
		const uint64_t pad = c->ix_pad = c->sw.ix_pad; base += pad;
		{	// records into hash order + the answer to every received minimizer (hao_index.cuh), then the answers go home: reverse all-to-all-v, 8 bytes per minimizer
			auto local_lk = [&]() -> int {
				HIP_TRY(lkr.reserve(n_recv + 1)); HIP_TRY(lkl.reserve(ml + 1)); HIP_TRY(c->w_runid.reserve(n_recv + 1)); HIP_TRY(c->d_ix_lk.reserve(ml + 1));
				if (n_recv) {
					auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RunHead32{px.p});
					size_t tb = 0;
					HIP_TRY(rocprim::inclusive_scan(nullptr, tb, heads, c->w_runid.p, n_recv, rocprim::plus<uint32_t>(), c->stream)); HIP_TRY(hao_tmp(c, tb));
					HIP_TRY(rocprim::inclusive_scan(c->d_tmp.p, tb, heads, c->w_runid.p, n_recv, rocprim::plus<uint32_t>(), c->stream));
					hipLaunchKernelGGL(hao_index_finish_kernel, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, c->stream, n_recv, ai2.p, c->w_runid.p, ucnt.p, c->w_ustart.p, 2, hi,
									   ri.p, pi.p, lkr.p, base);
					HAO_CHECK_LAUNCH();
				}
				return HAO_OK;
			};
			if (int rc = hao_comm_agree(c, cm, local_lk())) return rc;
			std::vector<uint64_t> rdisp(W, 0), back;
			for (int d = 1; d < W; ++d) rdisp[d] = rdisp[d - 1] + rcnt[d - 1];
			if (int rc = hao_comm_alltoallv_u64(c, cm, lkr.p, rcnt, rdisp, lkl.p, scnt)) return rc;      // what came from rank d goes back to rank d, in the order it came
			if (ml) { hipLaunchKernelGGL(hao_scatter_u64_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, c->stream, lkl.p, lperm, ml, c->d_ix_lk.p); HAO_CHECK_LAUNCH(); }
			HIP_TRY(hipStreamSynchronize(c->stream));
			c->lk_valid = true;
			c->timer.mark("pt_lookup");
		}
		// ONE all-gather for the whole index: every rank's slot = [position records | keys | list starts | counts], padded to the largest partition
		uint64_t maxm = 0, maxk = 0; for (int r = 0; r < W; ++r) { maxm = std::max(maxm, part[r]); maxk = std::max(maxk, nks[r]); }
		const size_t o_key = (size_t)maxm * 8, o_st = o_key + (size_t)maxk * 8, o_cnt = o_st + (size_t)maxk * 8, slot = (o_cnt + (size_t)maxk * 4 + 15) & ~(size_t)15;
		auto local_slot = [&]() -> int {
			if (nk_p) { hipLaunchKernelGGL(hao_add_const_kernel, dim3((unsigned)((nk_p + 255) / 256)), dim3(256), 0, c->stream, pst.p, nk_p, base); HAO_CHECK_LAUNCH(); }
			HIP_TRY(c->d_ix_sinfo.reserve(m + pad + 8));      // (+ 8: slack for the merge kernel's 32-byte reads; the sorted hashes themselves are not needed once the key table exists: only the 8-byte position records travel)
			HIP_TRY(c->d_ix_keys.reserve(nk + 1)); HIP_TRY(c->d_ix_start.reserve(nk + 1)); HIP_TRY(c->d_ix_cnt.reserve(nk + 1));
			HIP_TRY(cm.ag_tmp.reserve(slot * W + 16));
			char *mine = cm.ag_tmp.p + slot * cm.rank;
			if (n_recv) HIP_TRY(hipMemcpyAsync(mine, pi.p, n_recv * 8, hipMemcpyDeviceToDevice, c->stream));
			if (nk_p) {
				HIP_TRY(hipMemcpyAsync(mine + o_key, pk.p, nk_p * 8, hipMemcpyDeviceToDevice, c->stream));
				HIP_TRY(hipMemcpyAsync(mine + o_st, pst.p, nk_p * 8, hipMemcpyDeviceToDevice, c->stream));
				HIP_TRY(hipMemcpyAsync(mine + o_cnt, pc.p, nk_p * 4, hipMemcpyDeviceToDevice, c->stream));
			}
			return HAO_OK;
		};
		if (int rc = hao_comm_agree(c, cm, local_slot())) return rc;
		if (int rc = hao_comm_allgather_fixed(c, cm, cm.ag_tmp.p, slot)) return rc;
		uint64_t dm = pad, dk = 0;
		for (int r = 0; r < W; ++r) {
			const char *sr = cm.ag_tmp.p + slot * r;
			if (part[r]) HIP_TRY(hipMemcpyAsync(c->d_ix_sinfo.p + dm, sr, part[r] * 8, hipMemcpyDeviceToDevice, c->stream));
			if (nks[r]) {
				HIP_TRY(hipMemcpyAsync(c->d_ix_keys.p + dk, sr + o_key, nks[r] * 8, hipMemcpyDeviceToDevice, c->stream));
				HIP_TRY(hipMemcpyAsync(c->d_ix_start.p + dk, sr + o_st, nks[r] * 8, hipMemcpyDeviceToDevice, c->stream));
				HIP_TRY(hipMemcpyAsync(c->d_ix_cnt.p + dk, sr + o_cnt, nks[r] * 4, hipMemcpyDeviceToDevice, c->stream));
			}
			dm += part[r]; dk += nks[r];
		}
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->ix_n_sorted = m; c->ix_n_keys = nk; c->ix_n_pos = np;
		lsx.release(); lsi.release(); rx.release(); ri.release(); px.release(); pi.release(); lkr.release(); lkl.release(); ai.release(); ai2.release(); pk.release(); pst.release(); pc.release(); dt.release(); dc.release();
		c->timer.mark("pt_allgather");
	}
	int bits = 16; while ((1ULL << bits) < c->ix_n_keys / 2 && bits < 26) ++bits;
	if (int rc = hao_build_bucket(c, c->d_ix_keys.p, c->ix_n_keys, bits, c->d_ix_bucket)) return rc;
	c->ix_bucket_bits = bits;
	// host copy of the per-read minimizer offsets (batch sizing): taken here, not lazily inside a batch - batch contexts attached to this engine read it
	c->h_ix_mz_off.resize(n + 1);
	HIP_TRY(hipMemcpyAsync(c->h_ix_mz_off.data(), c->d_ix_mz_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (!c->has_ft) { int mx = (int)(c->hom_cov * c->opt.high_factor + .499); if (c->max_n_chain < mx) c->max_n_chain = mx; }   // Assembly.cpp:1011-1012
	c->has_pt = true;
	c->timer.mark("pt_table");
	return HAO_OK;
}

static hao_pt_dev hao_pt_view(hao_ctx *c)
{
	hao_pt_dev p; p.keys = c->d_ix_keys.p; p.start = c->d_ix_start.p; p.cnt = c->d_ix_cnt.p; p.bucket = c->d_ix_bucket.p; p.sinfo = c->d_ix_sinfo.p;
	p.n_keys = c->ix_n_keys; p.bshift = 64 - c->ix_bucket_bits;
	return p;
}

// host-readable copy of the index in the canonical (keys, CSR offsets, positions) form
static int hao_pt_download(hao_ctx *c)
{
	if (c->h_ix_valid) return HAO_OK;
	const uint64_t nk = c->ix_n_keys;
	std::vector<uint64_t> start(nk); std::vector<uint32_t> cnt(nk); std::vector<uint64_t> sinfo(c->ix_n_sorted);
	c->h_ix_keys.resize(nk); c->h_ix_off.resize(nk + 1); c->h_ix_pos.resize(c->ix_n_pos);
	if (nk) {
		HIP_TRY(hipMemcpy(c->h_ix_keys.data(), c->d_ix_keys.p, nk * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(start.data(), c->d_ix_start.p, nk * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(cnt.data(), c->d_ix_cnt.p, nk * 4, hipMemcpyDeviceToHost));
	}
	if (c->ix_n_sorted) HIP_TRY(hipMemcpy(sinfo.data(), c->d_ix_sinfo.p + c->ix_pad, c->ix_n_sorted * 8, hipMemcpyDeviceToHost));
	uint64_t o = 0;
	for (uint64_t i = 0; i < nk; ++i) { c->h_ix_off[i] = o; memcpy(c->h_ix_pos.data() + o, sinfo.data() + (start[i] - c->ix_pad), (size_t)cnt[i] * 8); o += cnt[i]; }
	c->h_ix_off[nk] = o;
	c->h_ix_valid = true;
	return HAO_OK;
}
