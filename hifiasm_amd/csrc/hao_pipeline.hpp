// Host orchestration of the device pipeline stages (part of libhao.so).
#pragma once
#include "hao_ctx.hpp"
#include "hao_sketch.cuh"
#include "hao_sketch3.cuh"

static void hao_batch_free(hao_ctx *c);
static void hao_release_all(hao_ctx *c);

struct U32ToU64 { __host__ __device__ uint64_t operator()(uint32_t v) const { return v; } };

__global__ void hao_chunk_count_kernel(const uint32_t *n_runs, const uint8_t *scalar_flag, uint64_t n_sel, int k, int chunk, uint64_t *cnt)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_sel) return;
	if (r == n_sel) { cnt[r] = 0; return; }
	uint64_t nm = n_runs[r] >= (uint32_t)k ? n_runs[r] - k + 1 : 0;
	cnt[r] = scalar_flag[r] ? 1 : (nm == 0 ? 1 : (nm + chunk - 1) / chunk);
}

// read of every unit (the unit kernel's one load instead of a search in chunk_off)
__global__ void hao_unit_rid_kernel(const uint64_t *chunk_off, uint64_t n_sel, uint32_t *unit_rid)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_sel) return;
	for (uint64_t u = chunk_off[r]; u < chunk_off[r + 1]; ++u) unit_rid[u] = (uint32_t)r;
}

static hao_ft_dev hao_ft_view(hao_ctx *c)
{
	hao_ft_dev f; f.keys = c->d_ft_keys.p; f.vals = c->d_ft_vals.p; f.bucket = c->d_ft_bucket.p; f.n = c->h_ft_keys.size();
	f.hbit = c->d_ft_hbit.p; f.hslot = (const ulonglong2*)c->d_ft_hslot.p; f.hbits = c->ft_hbits;
	return f;
}

// Sketch reads [lo, hi) -> c->d_mz_x / d_mz_info / d_mz_off (device). stamp_rid: write the read id into info.rid
// (index-time call htab.cpp:691) or 0 (query-time call anchor.cpp:1003).
static int hao_sketch_run(hao_ctx *c, uint64_t lo, uint64_t hi, int use_ft, int sample_dist, int stamp_rid)
{
	const uint64_t n_sel = hi - lo; const int k = c->opt.k, w = c->opt.w;
	const bool unit_variant = w == 51 && k == 51;     // default parameters: one wave per unit of 1024 window ordinals
	const bool wave_variant = w == 51 && k + 7 <= 64;     // other k at the default w: the round-2 kernel, one workgroup per chunk
	const int chunk = unit_variant ? hao_sk3<51, 51>::MW : (wave_variant ? hao_sk2<51>::CHUNK : HAO_SK_CHUNK);
	c->sk_lo = lo; c->sk_n = n_sel; c->sk_total = 0;
	HIP_TRY(c->d_mz_off.reserve(n_sel + 2));
	if (n_sel == 0) { HIP_TRY(hipMemsetAsync(c->d_mz_off.p, 0, 8, c->stream)); return HAO_OK; }
	// host: tile offsets, scalar-path flags
	std::vector<uint64_t> tile_off(n_sel + 1); std::vector<uint8_t> flag(n_sel); std::vector<uint32_t> slist; uint64_t nb = 0;
	const bool even_k = (k & 1) == 0;
	for (uint64_t r = 0; r < n_sel; ++r) {
		uint32_t L = c->h_len[lo + r];
		tile_off[r] = r == 0 ? 0 : tile_off[r - 1] + (c->h_len[lo + r - 1] + HAO_SK_TILE - 1) / HAO_SK_TILE + 1;
		bool hasn = c->has_n && c->h_nsite_off[lo + r + 1] > c->h_nsite_off[lo + r];
		flag[r] = (hasn || even_k) ? 1 : 0;
		if (flag[r]) slist.push_back((uint32_t)r);
		nb += L;
	}
	tile_off[n_sel] = tile_off[n_sel - 1] + (c->h_len[hi - 1] + HAO_SK_TILE - 1) / HAO_SK_TILE + 1;
	HIP_TRY(c->d_tile_off.reserve(n_sel + 1)); HIP_TRY(c->d_tile_ord.reserve(tile_off[n_sel] + 1)); HIP_TRY(c->d_n_runs.reserve(n_sel + 1));
	HIP_TRY(c->d_tot_l.reserve(n_sel + 1)); HIP_TRY(c->d_scalar_flag.reserve(n_sel + 1)); HIP_TRY(c->d_scalar_list.reserve(slist.size() + 1));
	HIP_TRY(c->d_chunk_off.reserve(n_sel + 2)); HIP_TRY(c->d_chunk_cnt64.reserve(n_sel + 2)); HIP_TRY(c->d_cursor.reserve(2)); HIP_TRY(c->d_err.reserve(2));
	HIP_TRY(hipMemcpyAsync(c->d_tile_off.p, tile_off.data(), (n_sel + 1) * 8, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_scalar_flag.p, flag.data(), n_sel, hipMemcpyHostToDevice, c->stream));
	if (!slist.empty()) HIP_TRY(hipMemcpyAsync(c->d_scalar_list.p, slist.data(), slist.size() * 4, hipMemcpyHostToDevice, c->stream));
	c->timer.mark("sk_h2d");
	hipLaunchKernelGGL(hpc_index_kernel, dim3((unsigned)((n_sel + 3) / 4)), dim3(256), 0, c->stream, c->d_packed.p, c->d_pk_off.p, c->d_len.p,
					   c->d_tile_off.p, c->d_tile_ord.p, c->d_n_runs.p, lo, n_sel, c->opt.hpc, unit_variant ? c->d_scalar_flag.p : (uint8_t*)nullptr);
	HAO_CHECK_LAUNCH();
	hipLaunchKernelGGL(hao_chunk_count_kernel, dim3((unsigned)((n_sel + 256) / 256)), dim3(256), 0, c->stream, c->d_n_runs.p, c->d_scalar_flag.p, n_sel, k, chunk, c->d_chunk_cnt64.p);
	HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, c->d_chunk_cnt64.p, c->d_chunk_off.p, n_sel + 1)) return rc;
	HIP_TRY(hipMemcpyAsync(c->d_tot_l.p, c->d_n_runs.p, n_sel * 4, hipMemcpyDeviceToDevice, c->stream));
	uint64_t n_chunks = 0;
	HIP_TRY(hipMemcpyAsync(&n_chunks, c->d_chunk_off.p + n_sel, 8, hipMemcpyDeviceToHost, c->stream));
	if (unit_variant) HIP_TRY(hipMemcpyAsync(flag.data(), c->d_scalar_flag.p, n_sel, hipMemcpyDeviceToHost, c->stream));      // + the reads hpc_index_kernel left to the scalar kernel
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (unit_variant) {
		const size_t n0 = slist.size(); slist.clear();
		for (uint64_t r = 0; r < n_sel; ++r) if (flag[r]) slist.push_back((uint32_t)r);
		if (slist.size() != n0) { HIP_TRY(c->d_scalar_list.reserve(slist.size() + 1)); HIP_TRY(hipMemcpyAsync(c->d_scalar_list.p, slist.data(), slist.size() * 4, hipMemcpyHostToDevice, c->stream)); }
	}
	c->timer.mark("sk_index");
	HIP_TRY(c->d_chunk_base.reserve(n_chunks + 1)); HIP_TRY(c->d_chunk_cnt.reserve(n_chunks + 1)); HIP_TRY(c->d_chunk_dst.reserve(n_chunks + 2));
	if (unit_variant) {
		HIP_TRY(c->d_unit_rid.reserve(n_chunks + 1));
		hipLaunchKernelGGL(hao_unit_rid_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, c->stream, c->d_chunk_off.p, n_sel, c->d_unit_rid.p);
		HAO_CHECK_LAUNCH();
	}
	if (!slist.empty()) { HIP_TRY(c->d_ring.reserve(slist.size() * 256 * sizeof(hao_cand))); HIP_TRY(c->d_ringord.reserve(slist.size() * 256)); HIP_TRY(c->d_cnt_ws.reserve(slist.size() + 1)); }
	const size_t smem = hao_sk_smem_bytes(w, k);
	// pool: every candidate of every chunk (bound nb / 6, grown on overflow); gathered / final lists: an estimate well above the usual one minimizer per
	// ~35 bases, checked on the device (the exact total is only read back at the end)
	const uint64_t pool_static = unit_variant ? n_chunks * SK3_SLOT : 0;
	uint64_t cap = pool_static + nb / (unit_variant ? 32 : 6) + 65536, gcap = std::min(nb / 6 + 65536, nb / 16 + 65536), total = 0;
	if (c->sw.sk_gcap >= 0) gcap = (uint64_t)c->sw.sk_gcap;          // force the overflow / retry path (tests)
	hao_scalar_args sa;
	for (int attempt = 0; ; ++attempt) {
		HIP_TRY(c->d_pool_x.reserve(cap)); HIP_TRY(c->d_pool_info.reserve(cap)); HIP_TRY(c->d_pool_ord.reserve(cap));
		HIP_TRY(hipMemsetAsync(c->d_cursor.p, 0, 8, c->stream)); HIP_TRY(hipMemsetAsync(c->d_err.p, 0, 4, c->stream));
		HIP_TRY(hipMemsetAsync(c->d_chunk_cnt.p, 0, (n_chunks + 1) * 4, c->stream));
		if (!slist.empty()) {
			sa.packed = c->d_packed.p; sa.pk_off = c->d_pk_off.p; sa.len = c->d_len.p; sa.nsite_off = c->has_n ? c->d_nsite_off.p : nullptr; sa.nsite = c->has_n ? c->d_nsite.p : nullptr;
			sa.chunk_off = c->d_chunk_off.p; sa.scalar_flag = c->d_scalar_flag.p; sa.scalar_list = c->d_scalar_list.p; sa.n_scalar = (uint32_t)slist.size();
			sa.rid_lo = lo; sa.k = k; sa.w = w; sa.hpc = c->opt.hpc; sa.use_ft = use_ft; sa.ft = hao_ft_view(c);
			sa.ring_ws = (hao_cand*)c->d_ring.p; sa.ringord_ws = c->d_ringord.p;
			sa.pool_x = c->d_pool_x.p; sa.pool_info = c->d_pool_info.p; sa.pool_ord = c->d_pool_ord.p; sa.pool_cursor = c->d_cursor.p; sa.pool_cap = cap;
			sa.chunk_base = c->d_chunk_base.p; sa.chunk_cnt = c->d_chunk_cnt.p; sa.tot_l = c->d_tot_l.p; sa.err = c->d_err.p; sa.pass = 0; sa.cnt_ws = c->d_cnt_ws.p; sa.pool_static = pool_static;
			hipLaunchKernelGGL(sketch_scalar_kernel, dim3((unsigned)((slist.size() + 63) / 64)), dim3(64), 0, c->stream, sa);
			HAO_CHECK_LAUNCH();
		}
		hao_sk_args a;
		a.packed = c->d_packed.p; a.pk_off = c->d_pk_off.p; a.len = c->d_len.p; a.tile_off = c->d_tile_off.p; a.tile_ord = c->d_tile_ord.p; a.n_runs = c->d_n_runs.p;
		a.chunk_off = c->d_chunk_off.p; a.scalar_flag = c->d_scalar_flag.p; a.rid_lo = lo; a.n_sel = n_sel; a.k = k; a.w = w; a.hpc = c->opt.hpc; a.ft = hao_ft_view(c);
		a.pool_x = c->d_pool_x.p; a.pool_info = c->d_pool_info.p; a.pool_ord = c->d_pool_ord.p; a.pool_cursor = c->d_cursor.p; a.pool_cap = cap;
		a.chunk_base = c->d_chunk_base.p; a.chunk_cnt = c->d_chunk_cnt.p; a.err = c->d_err.p; a.pool_static = pool_static; a.unit_rid = c->d_unit_rid.p; a.n_units = n_chunks;
		if (unit_variant) {
			if (use_ft && a.ft.n > 0) hipLaunchKernelGGL((sketch_unit_kernel<true, 51, 51>), dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, c->stream, a);
			else hipLaunchKernelGGL((sketch_unit_kernel<false, 51, 51>), dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, c->stream, a);
		} else if (wave_variant) {
			if (use_ft && a.ft.n > 0) hipLaunchKernelGGL((sketch_chunk_wave_kernel<true, 51>), dim3((unsigned)n_chunks), dim3(256), 0, c->stream, a);
			else hipLaunchKernelGGL((sketch_chunk_wave_kernel<false, 51>), dim3((unsigned)n_chunks), dim3(256), 0, c->stream, a);
		} else if (use_ft && a.ft.n > 0) hipLaunchKernelGGL(sketch_chunk_kernel<true>, dim3((unsigned)n_chunks), dim3(HAO_SK_THREADS), smem, c->stream, a);
		else hipLaunchKernelGGL(sketch_chunk_kernel<false>, dim3((unsigned)n_chunks), dim3(HAO_SK_THREADS), smem, c->stream, a);
		HAO_CHECK_LAUNCH();
		c->timer.mark("sk_chunks");
		{	// chunk_dst = exclusive scan of chunk_cnt (u32 -> u64)
			auto it = rocprim::make_transform_iterator(c->d_chunk_cnt.p, U32ToU64());
			if (int rc = hao_excl_scan_u64(c, it, c->d_chunk_dst.p, n_chunks + 1)) return rc;   // chunk_cnt[n_chunks] is 0 (memset)
		}
		if (!slist.empty()) { sa.pass = 1; hipLaunchKernelGGL(sketch_scalar_kernel, dim3((unsigned)((slist.size() + 63) / 64)), dim3(64), 0, c->stream, sa); HAO_CHECK_LAUNCH(); }
		// everything downstream is sized by gcap, so the exact total is read back only once, at the end
		const bool thin = use_ft && a.ft.n > 0 && sample_dist > w;      // (an empty filter table: no minimizer has a count, mz1_select_mz_h keeps everything)
		HIP_TRY(c->d_g_off.reserve(n_sel + 2)); HIP_TRY(c->d_mz_x.reserve(gcap + 1)); HIP_TRY(c->d_mz_info.reserve(gcap + 1));
		hipLaunchKernelGGL(sketch_read_off_kernel, dim3((unsigned)((n_sel + 256) / 256)), dim3(256), 0, c->stream, c->d_chunk_off.p, c->d_chunk_dst.p, n_sel, c->d_g_off.p);
		HAO_CHECK_LAUNCH();
		if (!thin) {      // pool -> final lists in one pass
			HIP_TRY(hipMemcpyAsync(c->d_mz_off.p, c->d_g_off.p, (n_sel + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
			hipLaunchKernelGGL(sketch_gather_finish_kernel, dim3((unsigned)((n_sel + 3) / 4)), dim3(256), 0, c->stream, c->d_pool_x.p, c->d_pool_info.p, c->d_chunk_base.p, c->d_chunk_cnt.p,
							   c->d_chunk_dst.p, c->d_chunk_off.p, lo + c->rid_base, n_sel, stamp_rid, c->d_mz_x.p, c->d_mz_info.p, gcap, c->d_err.p);
			HAO_CHECK_LAUNCH();
			c->timer.mark("sk_gather");
		} else {
			HIP_TRY(c->d_g_x.reserve(gcap + 1)); HIP_TRY(c->d_g_info.reserve(gcap + 1)); HIP_TRY(c->d_g_ord.reserve(gcap + 1));
			hipLaunchKernelGGL(sketch_gather_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, c->stream, c->d_pool_x.p, c->d_pool_info.p, c->d_pool_ord.p,
							   c->d_chunk_base.p, c->d_chunk_cnt.p, c->d_chunk_dst.p, n_chunks, c->d_g_x.p, c->d_g_info.p, c->d_g_ord.p, gcap, c->d_err.p);
			HAO_CHECK_LAUNCH();
			c->timer.mark("sk_gather");
			HIP_TRY(c->d_new_n.reserve(n_sel + 2)); HIP_TRY(hipMemsetAsync(c->d_new_n.p + n_sel, 0, 4, c->stream));
			{	// the wave-parallel thinning (hao_select2.cuh), one wave per read: reads of up to 512 candidates (30 KB of LDS per wave), then the longer ones
				hipLaunchKernelGGL((sketch_select2_kernel<HAO_S2_CAP_SMALL, 64>), dim3((unsigned)n_sel), dim3(64), 0, c->stream, c->d_g_x.p, c->d_g_info.p, c->d_g_ord.p, c->d_g_off.p,
								   c->d_len.p, c->d_tot_l.p, lo, n_sel, sample_dist, c->opt.rewin, k, c->d_new_n.p, c->d_err.p);
				HAO_CHECK_LAUNCH();
				hipLaunchKernelGGL((sketch_select2_kernel<HAO_S2_CAP, 256>), dim3((unsigned)n_sel), dim3(256), 0, c->stream, c->d_g_x.p, c->d_g_info.p, c->d_g_ord.p, c->d_g_off.p,
								   c->d_len.p, c->d_tot_l.p, lo, n_sel, sample_dist, c->opt.rewin, k, c->d_new_n.p, c->d_err.p);
			}
			HAO_CHECK_LAUNCH();
			auto it = rocprim::make_transform_iterator(c->d_new_n.p, U32ToU64());
			if (int rc = hao_excl_scan_u64(c, it, c->d_mz_off.p, n_sel + 1)) return rc;
			c->timer.mark("sk_select");
			hipLaunchKernelGGL(sketch_finish_kernel, dim3((unsigned)((n_sel + 3) / 4)), dim3(256), 0, c->stream, c->d_g_x.p, c->d_g_info.p, c->d_g_off.p, c->d_mz_off.p, lo + c->rid_base, n_sel, stamp_rid,
							   c->d_mz_x.p, c->d_mz_info.p, c->d_err.p);
			HAO_CHECK_LAUNCH();
		}
		int err = 0;
		HIP_TRY(hipMemcpyAsync(&err, c->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpyAsync(&total, c->d_mz_off.p + n_sel, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (!err) break;
		if (attempt >= 3) { hao_set_err(c, "minimizer pool overflow"); return HAO_ENOMEM; }
		cap = pool_static + (attempt == 0 ? nb / 2 + 65536 : nb + 65536); gcap = cap - pool_static;       // (the kernels above ran on truncated buffers: everything is redone)
	}
	c->sk_total = total;
	c->timer.mark("sk_finish");
	return HAO_OK;
}
