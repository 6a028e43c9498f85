// remaining C ABI entry points + resource release
#pragma once

static void hao_batch_free(hao_ctx *c) { if (c->batch) { c->batch->release(); delete c->batch; c->batch = nullptr; } }

static void hao_release_all(hao_ctx *c)
{
	c->d_packed.release(); c->d_pk_off.release(); c->d_len.release(); c->d_len_all.release(); c->d_nsite_off.release(); c->d_nsite.release();
	c->d_ft_keys.release(); c->d_ft_vals.release(); c->d_ft_bucket.release(); c->d_ft_hbit.release(); c->d_ft_hslot.release();
	c->d_tile_off.release(); c->d_tile_ord.release(); c->d_n_runs.release(); c->d_tot_l.release(); c->d_chunk_off.release(); c->d_chunk_cnt64.release();
	c->d_scalar_flag.release(); c->d_scalar_list.release(); c->d_pool_x.release(); c->d_pool_info.release(); c->d_pool_ord.release(); c->d_cursor.release(); c->d_err.release();
	c->d_chunk_base.release(); c->d_chunk_dst.release(); c->d_chunk_cnt.release(); c->d_g_x.release(); c->d_g_info.release(); c->d_g_ord.release(); c->d_g_off.release();
	c->d_new_n.release(); c->d_new_n64.release(); c->d_mz_x.release(); c->d_mz_info.release(); c->d_mz_off.release(); c->d_tmp.release(); c->d_ring.release(); c->d_ringord.release(); c->d_cnt_ws.release();
	c->w_ukeys.release(); c->w_flag.release(); c->w_kpos.release(); c->w_ustart.release(); c->w_ucnt.release(); c->w_hist.release(); c->w_ok.release(); c->w_ok2.release(); c->w_oi.release(); c->w_oi2.release();
	c->w_lkv2.release(); c->w_s40_list.release(); c->w_s40_o.release(); c->w_s40_x.release(); c->w_s40_cnt.release(); c->d_ix_lk.release(); c->w_runid.release(); c->d_ix_mz_x.release(); c->d_ix_mz_info.release(); c->d_ix_mz_off.release(); c->d_ix_sx.release(); c->d_ix_sinfo.release();
	c->d_ix_keys.release(); c->d_ix_start.release(); c->d_ix_cnt.release(); c->d_ix_bucket.release();
	c->al_task.release(); c->al_k1.release(); c->al_k2.release(); c->al_path.release(); c->al_i1.release(); c->al_order.release(); c->al_sel.release(); c->al_res.release(); c->al_tres.release(); c->al_want.release(); c->al_cig.release();
}

#include <atomic>
#include <new>
#include <stdexcept>
#include <thread>
extern "C" {

int hao_ft_gen(hao_ctx *c, int32_t *hom_cov)
{
	if (!c) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_ft_gen"); ++c->index_gen;
	HIP_TRY(hipSetDevice(c->device));
	c->timer.begin(c->stream);
	int rc = hao_ft_run(c);
	if (rc != HAO_OK) return rc;
	c->timer.mark("ft_release");      // (hao_ft_run's buffers are gone: host_ft_release = what freeing them took)
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->timer.collect(c->stage_ms);
	if (hom_cov) *hom_cov = c->ft_peak_hom;
	return HAO_OK;
}

int32_t hao_ft_cnt(hao_ctx *c, uint64_t y)
{
	if (!c || !c->has_ft) return 0;
	int64_t i = hao_bsearch(c->h_ft_keys.data(), c->h_ft_keys.size(), y);
	return i < 0 ? 0 : c->h_ft_vals[i];
}

int hao_ft_table(hao_ctx *c, uint64_t *n, const uint64_t **keys, const int32_t **vals)
{
	if (!c || !c->has_ft) return HAO_EINVAL;
	*n = c->h_ft_keys.size(); *keys = c->h_ft_keys.data(); *vals = c->h_ft_vals.data();
	return HAO_OK;
}

int hao_hist(hao_ctx *c, int which, int64_t cnt[4096])
{
	if (!c) return HAO_EINVAL;
	memcpy(cnt, which == 0 ? c->ft_hist : c->pt_hist, sizeof(int64_t) * HAO_N_COUNTS);
	return HAO_OK;
}

int hao_stats(hao_ctx *c, int64_t out[8])
{
	if (!c) return HAO_EINVAL;
	uint32_t h, l; hao_occ_thresholds(c->hom_cov, &h, &l);
	out[0] = c->ft_peak_hom; out[1] = c->ft_peak_het; out[2] = c->ft_cutoff; out[3] = c->max_n_chain;
	out[4] = c->hom_cov; out[5] = c->het_cov; out[6] = h; out[7] = l;
	return HAO_OK;
}

int hao_ft_passes(hao_ctx *c) { return c ? c->ft_passes_used : HAO_EINVAL; }

int hao_pt_gen(hao_ctx *c, int32_t *hom_cov, int32_t *het_cov)
{
	if (!c) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_pt_gen"); ++c->index_gen;
	HIP_TRY(hipSetDevice(c->device));
	c->timer.begin(c->stream);
	int rc = hao_pt_run(c);
	if (rc != HAO_OK) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->timer.collect(c->stage_ms);
	if (hom_cov) *hom_cov = c->hom_cov;
	if (het_cov) *het_cov = c->het_cov;
	return HAO_OK;
}

int hao_pt_get(hao_ctx *c, uint64_t hash, const uint64_t **pos, int32_t *n)
{
	if (!c || !c->has_pt) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_pt_download(c)) return rc;
	int64_t i = hao_bsearch(c->h_ix_keys.data(), c->h_ix_keys.size(), hash);
	if (i < 0) { *pos = nullptr; *n = 0; return HAO_OK; }
	*pos = c->h_ix_pos.data() + c->h_ix_off[i]; *n = (int32_t)(c->h_ix_off[i + 1] - c->h_ix_off[i]);
	return HAO_OK;
}

int hao_pass_default(hao_ctx *c, hao_pass_t *p)
{
	if (!c || !p) return HAO_EINVAL;
	if (int rc = hao_view_refresh(c)) return rc;
	memset(p, 0, sizeof(*p));
	p->bw_thres = c->opt.is_ont ? 0.05 : 0.02; p->max_n_chain = c->max_n_chain;
	hao_occ_thresholds(c->hom_cov, &p->high_occ, &p->low_occ);
	p->apend_be = 1; p->is_accurate = 1; p->gen_off = 1; p->mcopy_num = 3; p->mcopy_rate = 0.7; p->chain_cutoff = 2; p->mcopy_khit_cut = 32; p->ocv_w = 3072;
	return HAO_OK;
}

int hao_overlap_batch(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi)
{
	hao_pass_t ps;
	if (int rc = hao_pass_default(c, &ps)) return rc;
	return hao_overlap_batch_ex(c, rid_lo, rid_hi, &ps);
}

int hao_overlap_batch_ex(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, const hao_pass_t *pass)
{
	if (c) { if (int rc = hao_view_refresh(c)) return rc; }
	if (!c || !pass || rid_lo > rid_hi || rid_hi > c->n_reads) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	c->timer.begin(c->stream);
	int rc = hao_overlap_run(c, rid_lo, rid_hi, *pass);
	if (rc != HAO_OK) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->timer.collect(c->stage_ms);
	return HAO_OK;
}

int hao_overlap_batch_async(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, const hao_pass_t *pass, uint32_t parts, int *slot)
{
	if (c) { if (int rc = hao_view_refresh(c)) return rc; }
	if (!c || rid_lo > rid_hi || rid_hi > c->n_reads || !(parts & (HAO_DELIVER_OL | HAO_DELIVER_CL | HAO_DELIVER_EXACT))) return HAO_EINVAL;
	hao_pass_t ps;
	if (!pass) { if (int rc = hao_pass_default(c, &ps)) return rc; pass = &ps; }
	HIP_TRY(hipSetDevice(c->device));
	c->timer.begin(c->stream);
	int rc = hao_overlap_run(c, rid_lo, rid_hi, *pass, parts, slot);
	if (rc != HAO_OK) return rc;
	c->timer.collect(c->stage_ms);      // (the batch's kernels are complete: hao_overlap_run ends with the read-back of its totals; only the copy is still running)
	return HAO_OK;
}

int hao_deliver_wait(hao_ctx *c, int slot, hao_delivery_t *out)
{
	if (!c || !out || slot < 0 || slot > 1 || !c->batch || !c->batch->dl_ready) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	hao_ctx::Batch &B = *c->batch;
	if (B.dl_pending[slot]) { HIP_TRY(hipEventSynchronize(B.ev_done[slot])); B.dl_pending[slot] = false; float ms = 0; if (hipEventElapsedTime(&ms, B.ev_ready[slot], B.ev_done[slot]) == hipSuccess) B.dl[slot].copy_ms = ms;
		// the rate of the copy itself: a big batch that crossed at less than 40 GB/s (a good arena gives 50 - 56 beside the next batch's kernels) gets its arena allocated again
		// before the slot's next batch, with a timed copy into every NUMA node (batches of 256 MB and more: smaller ones pay per-copy overheads that say nothing about the arena) (once per slot; round 6 saw arenas that passed the probe at allocation and then copied at 30 GB/s for the whole run)
		float cms = 0; const double mb_ = (double)B.dl[slot].bytes / 1e6;
		if (hipEventElapsedTime(&cms, B.ev_cstart[slot], B.ev_done[slot]) == hipSuccess && cms > 0 && B.arena_retry[slot] < 1 && ((mb_ >= 256.0 && mb_ / cms < 40.0) || (c->sw.arena_probe && mb_ > 0))) {
			B.arena_bad[slot] = true; ++B.arena_retry[slot];
			fprintf(stderr, "[hao] delivery arena %d: a batch of %.0f MB was copied at %.1f GB/s; the arena is allocated again for the slot's next batch, every NUMA node tried\n", slot, mb_, mb_ / cms);
		}
	}
	if (B.dl[slot].n_ol && B.dl[slot].fc_off) ((uint64_t*)B.dl[slot].fc_off)[B.dl[slot].n_ol] = B.dl[slot].n_fc;      // end of the last cigar: a host-side word next to the region the copy wrote, set once the copy has landed
	*out = B.dl[slot];
	return HAO_OK;
}

// pure function of a delivered view (any thread): the wire bytes of read rid back into k_mer_hits (hao_deliver.cuh describes the format)
uint64_t hao_unpack_hits(const hao_delivery_t *d, uint64_t rid, hao_hit_t *out, uint64_t cap)
{
	if (!d || rid < d->rid_lo || rid >= d->rid_lo + d->n_reads || !d->cl_off) return 0;
	const uint64_t r = rid - d->rid_lo, h0 = d->cl_off[r], nh = d->cl_off[r + 1] - h0;
	if (nh > cap || !out) return nh;
	const hao_qmz_t *qt = d->qmz ? d->qmz + d->qm_off[r] : nullptr;
	const uint16_t *qp16 = d->qmz ? nullptr : d->qmz_pos + d->qm_off[r]; const uint16_t *qc8 = d->qmz ? nullptr : d->qmz_cnt + d->qm_off[r];      // (the packed tables)
	auto q_self = [&](uint32_t q_) -> uint32_t { return qt ? qt[q_].self_offset : (uint32_t)qp16[q_]; };
	auto q_cnt = [&](uint32_t q_) -> uint32_t { return qt ? qt[q_].cnt : (uint32_t)qc8[q_]; };
	uint64_t k = 0;
	for (uint64_t ci = d->ch_off[r]; ci < d->ch_off[r + 1]; ++ci) {
		const hao_chain_hdr_t &H = d->chains[ci];
		uint32_t q = H.q0, off = H.offset;
		const uint64_t g = H.pos;      // position of the chain's first hit; the code bytes of its later hits start at rank(g + 1) (the byte at g itself, if any, is not the chain's)
		uint64_t cp = 0;
		if (H.n_hits > 1) {      // the directory has an entry per 256 positions: + the bits of the words between it and the position
			const uint64_t wp = (g + 1) >> 6;
			cp = d->cl_rank[wp >> 2];
			for (uint64_t w = wp & ~3ULL; w < wp; ++w) cp += (uint64_t)__builtin_popcountll(d->cl_bits[w]);
			cp += (uint64_t)__builtin_popcountll(d->cl_bits[wp] & ((1ULL << ((g + 1) & 63)) - 1));
		}
		for (uint32_t i = 0; i < H.n_hits; ++i) {
			hao_hit_t &o = out[k + i];
			if (i) {
				const uint64_t gi = g + i; uint8_t w = 0x08;      // no code byte: the read's next minimizer, same diagonal
				if (d->cl_bits[gi >> 6] >> (gi & 63) & 1) w = d->cl_codes[cp++];
				if (w == 0xff) {      // verbatim: binary search of the position in the sorted exception list
					uint64_t lo = 0, hi = d->n_exc;
					while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (d->cl_exc[m].index < gi) lo = m + 1; else hi = m; }
					o = d->cl_exc[lo].hit; o.w0 = H.w0; q = d->cl_exc[lo].q; off = o.offset;
					continue;
				}
				const uint32_t qn = q + (w >> 4) + 1; off = (uint32_t)((int64_t)off + (int64_t)(q_self(qn) - q_self(q)) + (int64_t)(w & 15) - 8); q = qn;
			}
			o.w0 = H.w0; o.offset = off; o.self_offset = q_self(q); o.cnt = q_cnt(q);
		}
		k += H.n_hits;
	}
	return k;
}

// ol->list of a delivered read back into hao_ovlp_t records (the wire drops x_id, x_pos_strand and align_length: the read, 0 and 0 - hao_ol_wire_kernel)
uint64_t hao_unpack_overlaps(const hao_delivery_t *d, uint64_t rid, hao_ovlp_t *out, uint64_t cap)
{
	if (!d || rid < d->rid_lo || rid >= d->rid_lo + d->n_reads || !d->ol_off || !d->ol) return 0;
	const uint64_t r = rid - d->rid_lo, o0 = d->ol_off[r], n = d->ol_off[r + 1] - o0;
	if (n > cap || !out) return n;
	for (uint64_t i = 0; i < n; ++i) {
		const hao_ovlp_wire_t &w = d->ol[o0 + i]; hao_ovlp_t &o = out[i];
		o.x_id = (uint32_t)rid; o.x_pos_s = w.x_pos_s; o.x_pos_e = w.x_pos_e; o.x_pos_strand = 0;
		o.y_id = w.y & 0x7fffffffu; o.y_pos_s = w.y_pos_s; o.y_pos_e = w.y_pos_e; o.y_pos_strand = w.y >> 31;
		o.shared_seed = w.shared_seed; o.align_length = 0; o.non_homopolymer_errors = w.non_homopolymer_errors; o.fc_len = w.fc_len;
	}
	return n;
}

// the fake cigar of delivered overlap j back into 8-byte entries (hao_deliver.cuh: "fake cigars on the wire")
uint32_t hao_unpack_cigar(const hao_delivery_t *d, uint64_t j, uint64_t *out, uint32_t cap)
{
	if (!d || !d->ol || !d->fc_off || j >= d->n_ol) return 0;
	const uint32_t fl = d->ol[j].fc_len;
	if (fl > cap || !out || fl == 0) return fl;
	const uint64_t o = d->fc_off[j]; const uint32_t *w = d->fc + (o & ~HAO_FC_RAW);
	if (o & HAO_FC_RAW) { for (uint32_t k = 0; k < fl; ++k) out[k] = (uint64_t)w[2 * k + 1] << 32 | w[2 * k]; return fl; }
	uint32_t site = d->ol[j].x_pos_s; int64_t sh = 0;
	out[0] = (uint64_t)site << 32;
	for (uint32_t k = 1; k < fl; ++k) {
		const uint32_t x = w[k - 1], z = x >> 20;
		site += x & 0xfffffu; sh += (int64_t)(z >> 1) ^ -(int64_t)(z & 1);
		out[k] = (uint64_t)site << 32 | (sh < 0 ? ((uint32_t)(-sh) << 1 | 1u) : (uint32_t)sh << 1);
	}
	return fl;
}

// hao_batch_digest's value for every read of a DELIVERED batch, computed on the host from the bytes in the pinned arena: ol->list, the fake cigars and
// cl->list decoded out of the wire format by hao_unpack_hits.  A pure function of the view; the reads are spread over n_threads host threads.
int hao_delivery_digest(const hao_delivery_t *d, uint64_t *out, int n_threads)
{
	if (!d || (!out && d->n_reads)) return HAO_EINVAL;
	if (d->n_reads && (!d->ol_off || !d->cl_off)) return HAO_EINVAL;      // needs HAO_DELIVER_OL | HAO_DELIVER_CL
	const uint64_t n = d->n_reads;
	if (n_threads < 1) n_threads = 1;
	if ((uint64_t)n_threads > n) n_threads = (int)std::max<uint64_t>(1, n);
	std::atomic<uint64_t> next(0); std::atomic<int> bad(0);
	auto work = [&]() {
		std::vector<hao_hit_t> buf; std::vector<uint64_t> fcb; std::vector<hao_ovlp_t> olb;
		for (;;) {
			const uint64_t b0 = next.fetch_add(64); if (b0 >= n) break;
			for (uint64_t r = b0; r < std::min(n, b0 + 64); ++r) {
				const uint64_t o0 = d->ol_off[r], o1 = d->ol_off[r + 1], nh = d->cl_off[r + 1] - d->cl_off[r];
				if (buf.size() < nh) buf.resize(nh + nh / 4 + 64);
				if (hao_unpack_hits(d, d->rid_lo + r, buf.data(), nh) != nh) { bad = 1; out[r] = 0; continue; }
				if (olb.size() < o1 - o0) olb.resize(o1 - o0 + 64);
				if (hao_unpack_overlaps(d, d->rid_lo + r, olb.data(), o1 - o0) != o1 - o0) { bad = 1; out[r] = 0; continue; }
				uint64_t s = 0; const uint64_t *w = (const uint64_t*)olb.data();
				for (uint64_t i = 0; i < (o1 - o0) * 6; ++i) s += hao_dg_term(1, i, w[i]);
				{	uint64_t fi = 0;      // the fake cigars of the read's overlaps, entry by entry, through the decoder
					for (uint64_t j = o0; j < o1; ++j) {
						const uint32_t fl = d->ol[j].fc_len; if (fcb.size() < fl) fcb.resize(fl);
						if (hao_unpack_cigar(d, j, fcb.data(), fl) != fl) bad = 1;
						for (uint32_t k = 0; k < fl; ++k) s += hao_dg_term(2, fi++, fcb[k]);
					} }
				w = (const uint64_t*)buf.data();
				for (uint64_t i = 0; i < nh * 2; ++i) s += hao_dg_term(3, i, w[i]);
				out[r] = s;
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
	work();
	for (auto &t : th) t.join();
	return bad ? HAO_EINVAL : HAO_OK;
}

int hao_index_save(hao_ctx *c, const char *prefix, int32_t number_of_round, const char *const *names)
{
	if (!c || !prefix) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_index_save");
	HIP_TRY(hipSetDevice(c->device));
	return hao_index_save_impl(c, prefix, number_of_round, names);
}

int hao_next_slot(hao_ctx *c, int *slot)
{
	if (!c || !slot) return HAO_EINVAL;
	*slot = c->batch ? (int)(c->batch->dl_seq & 1) : 0;
	return HAO_OK;
}

int hao_index_load(hao_ctx *c, const char *prefix, int32_t *number_of_round)
{
	if (!c || !prefix) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_index_load");
	HIP_TRY(hipSetDevice(c->device));
	try { return hao_index_load_impl(c, prefix, number_of_round); }      // (the loader sizes host vectors from file fields - checked against the file's length, but an exception must not cross the C boundary)
	catch (const std::bad_alloc &) { c->has_ft = false; c->has_pt = false; c->lk_valid = false; hao_set_err(c, "hao_index_load: out of host memory"); return HAO_ENOMEM; }
	catch (const std::exception &e) { c->has_ft = false; c->has_pt = false; c->lk_valid = false; hao_set_err(c, std::string("hao_index_load: ") + e.what()); return HAO_EINVAL; }
}

int hao_exact_check(hao_ctx *c)
{
	if (!c || !c->batch || !c->batch->valid) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_exact_run(c)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HAO_OK;
}

int hao_window_ed_grid(hao_ctx *c, uint32_t window, uint32_t thre, uint64_t *n_tasks)
{
	if (!c || !n_tasks || !c->batch || !c->batch->valid) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_ed_grid_run(c, window, thre, n_tasks)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HAO_OK;
}

int hao_fetch_ed_grid(hao_ctx *c, hao_ed_task_t *tasks, hao_ed_result_t *res, uint64_t cap)
{
	if (!c) return HAO_EINVAL;
	if (!c->batch || !c->batch->valid || (c->al_grid_n == 0 && cap)) { hao_set_err(c, "hao_fetch_ed_grid: no pairs of hao_window_ed_grid are resident (a new batch or another window-alignment call has reused the scratch)"); return HAO_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	const uint64_t n = std::min<uint64_t>(cap, c->al_grid_n);
	if (n && tasks) HIP_TRY(hipMemcpyAsync(tasks, c->al_task.p, n * sizeof(hao_ed_task_t), hipMemcpyDeviceToHost, c->stream));
	if (n && res) HIP_TRY(hipMemcpyAsync(res, c->al_res.p, n * sizeof(hao_ed_result_t), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HAO_OK;
}

int hao_fetch_exact(hao_ctx *c, uint64_t rid, const uint8_t **flags, uint64_t *n)
{
	if (!c || !flags || !n || !c->batch || !c->batch->valid || rid < c->batch->lo || rid >= c->batch->lo + c->batch->n) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_exact_check(c)) return rc;
	if (int rc = hao_batch_download(c)) return rc;
	hao_ctx::Batch &B = *c->batch;
	if (B.h_exact.size() != B.n_ol + 1) { B.h_exact.assign(B.n_ol + 1, 0); if (B.n_ol) HIP_TRY(hipMemcpy(B.h_exact.data(), B.O().exact.p, B.n_ol, hipMemcpyDeviceToHost)); }
	const uint64_t r = rid - B.lo, s_ = B.h_fin_off[r], e_ = B.h_fin_off[r + 1];
	*flags = B.h_exact.data() + s_; *n = e_ - s_;
	return HAO_OK;
}

int hao_fetch_seed_hits(hao_ctx *c, uint64_t rid, const hao_hit_t **hits, uint64_t *n)
{
	if (!c || !c->batch || !c->batch->valid || rid < c->batch->lo || rid >= c->batch->lo + c->batch->n) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_batch_download(c)) return rc;
	hao_ctx::Batch &B = *c->batch; uint64_t r = rid - B.lo;
	*hits = B.h_hits.data() + B.h_seg[r]; *n = B.h_seg[r + 1] - B.h_seg[r];
	return HAO_OK;
}

int hao_fetch_overlaps(hao_ctx *c, uint64_t rid, const hao_ovlp_t **ol, uint64_t *n_ol, const uint64_t **fc, const uint64_t **fc_off, const hao_hit_t **cl, uint64_t *n_cl)
{
	if (!c || !c->batch || !c->batch->valid || rid < c->batch->lo || rid >= c->batch->lo + c->batch->n) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_batch_download(c)) return rc;
	hao_ctx::Batch &B = *c->batch; uint64_t r = rid - B.lo, s = B.h_fin_off[r], e = B.h_fin_off[r + 1];
	*ol = B.h_ol.data() + s; *n_ol = e - s;
	*fc = B.h_fc.data() + B.h_fc_out_off[s]; *fc_off = B.h_fc_out_off.data() + s;     // absolute offsets; entry i spans [fc_off[i]-fc_off[0], fc_off[i+1]-fc_off[0]) of *fc
	*cl = B.h_cl.data() + B.h_cl_off[r]; *n_cl = B.h_cl_off[r + 1] - B.h_cl_off[r];
	return HAO_OK;
}

int hao_batch_totals(hao_ctx *c, uint64_t out[8])
{
	if (!c || !c->batch || !c->batch->valid) return HAO_EINVAL;
	hao_ctx::Batch &B = *c->batch;
	memset(out, 0, 8 * sizeof(uint64_t));
	out[0] = B.n_ol; out[1] = B.n_cl; out[2] = B.n_anchor; out[3] = B.n_groups; out[4] = B.n_mz; out[5] = B.n_chains; out[6] = B.n_generic; out[7] = B.n_generic_hits;
	return HAO_OK;
}

int hao_batch_seed_path(hao_ctx *c, uint64_t out[4])
{
	if (!c || !out || !c->batch || !c->batch->valid) return HAO_EINVAL;
	hao_ctx::Batch &B = *c->batch;
	out[0] = B.seed_path; out[1] = B.seed_left[0]; out[2] = B.seed_left[1]; out[3] = B.seed_left[2];
	return HAO_OK;
}

int hao_batch_digest(hao_ctx *c, uint64_t *out, uint64_t *out_kh)
{
	if (!c || !out || !c->batch || !c->batch->valid) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	hao_ctx::Batch &B = *c->batch; const uint64_t n = B.n;
	if (n == 0) return HAO_OK;
	if (int rc = hao_batch_materialize_cl(c)) return rc;
	DevBuf<uint64_t> d; HIP_TRY(d.reserve(2 * n + 2));
	hao_digest_args a;
	a.fin_off = B.O().fin_off.p; a.fcf_off = B.fcf_off.p; a.g_off = B.g_off.p; a.cl_base = B.cl_base.p; a.seg = B.seg.p;
	a.ol = (const uint64_t*)B.O().ol_out.p; a.fc = B.O().fc_out.p; a.cl = (const uint64_t*)B.cl.p; a.hits = (const uint64_t*)B.hits.p;
	a.n_sel = n; a.dig = d.p; a.dig_kh = out_kh ? d.p + n : nullptr;
	hipLaunchKernelGGL(hao_digest_kernel, dim3((unsigned)n), dim3(256), 0, c->stream, a);
	HAO_CHECK_LAUNCH();
	HIP_TRY(hipMemcpyAsync(out, d.p, n * 8, hipMemcpyDeviceToHost, c->stream));
	if (out_kh) HIP_TRY(hipMemcpyAsync(out_kh, d.p + n, n * 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	d.release();
	return HAO_OK;
}

// Self-test of the two ways to group minimizer records by their 16 owner bits (hao_pt_run, sharded): out[0] = order violations of
// rocprim::radix_sort_pairs(begin_bit = 48, end_bit = 64) on (hash, index) pairs, out[1] = violations of the path the engine uses (separate 16-bit
// key, begin_bit = 0, then a gather).  A "violation" = a position whose owner bits decrease, or equal owner bits with a decreasing index (the
// sort must be stable).  tests/test_gpu_rocprim.py pins out[1] == 0 and logs out[0].
__global__ void hao_selftest_fill_kernel(uint64_t n, uint64_t *x, uint64_t *v)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { x[i] = hao_hash64(i * 0x9E3779B97F4A7C15ULL + 12345); v[i] = i; }
}
// Self-test of the index sort on 40 hash bits (hao_index_sort): n keys built so that many 40-bit runs hold up to four distinct keys, interleaved in input
// order; sorted both ways on the device.  out = { positions where the two results differ, two-key runs the fix-up listed, scratch elements used, overflow flag }
__global__ void hao_selftest_s40_fill_kernel(uint64_t n, uint64_t *x)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const uint64_t g = i % (n / 5 + 1); x[i] = (hao_hash64(g + 99) & ~((1ULL << HAO_SORT40_LOWBITS) - 1)) | ((g & 255) ? 0x155ULL : ((hao_hash64(i) >> 7) & 3) * 0x2a5b7ULL); }      // one 40-bit group in 256 holds up to four keys, interleaved in input order
}
__global__ void hao_selftest_s40_cmp_kernel(const uint64_t *a, const uint64_t *b, const uint32_t *ia, const uint32_t *ib, uint64_t n, unsigned long long *bad)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && (a[i] != b[i] || ia[i] != ib[i])) atomicAdd(bad, 1ULL);
}
int hao_selftest_sortbits(hao_ctx *c, uint64_t n, uint64_t out[4])
{
	if (!c || !out || n < 16 || n >= (1ULL << 32)) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	DevBuf<uint64_t> x, s1, s2; DevBuf<uint32_t> o, o1, o2; DevBuf<unsigned long long> bad;
	HIP_TRY(x.reserve(n)); HIP_TRY(s1.reserve(n)); HIP_TRY(s2.reserve(n)); HIP_TRY(o.reserve(n)); HIP_TRY(o1.reserve(n)); HIP_TRY(o2.reserve(n)); HIP_TRY(bad.reserve(1));
	HIP_TRY(hipMemsetAsync(bad.p, 0, 8, c->stream));
	hipLaunchKernelGGL(hao_selftest_s40_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, x.p); HAO_CHECK_LAUNCH();
	int rc = hao_index_sort(c, x.p, s1.p, o.p, o1.p, n, false);
	if (!rc) rc = hao_index_sort(c, x.p, s2.p, o.p, o2.p, n, true);
	if (!rc) {
		hipLaunchKernelGGL(hao_selftest_s40_cmp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, s1.p, s2.p, o1.p, o2.p, n, bad.p);
		unsigned long long hb = 0;
		if (hipMemcpyAsync(&hb, bad.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = HAO_ENODEV;
		out[0] = hb; out[1] = c->peek_h[16]; out[2] = c->peek_h[17]; out[3] = c->peek_h[18];
	}
	x.release(); s1.release(); s2.release(); o.release(); o1.release(); o2.release(); bad.release();
	return rc;
}

int hao_selftest_rocprim(uint64_t n, uint64_t out[2])
{
	hao_ctx tmp_ctx; hao_ctx *c = &tmp_ctx;      // only for the error-string macros
	if (!out || n == 0 || n >= (1ULL << 32)) return HAO_EINVAL;
	out[0] = out[1] = 0;
	hipStream_t st = nullptr;
	DevBuf<uint64_t> x, v, x2, v2, gx, gv; DevBuf<uint32_t> ok, ok2, oi, oi2; DevBuf<unsigned char> tmp;
	HIP_TRY(x.reserve(n)); HIP_TRY(v.reserve(n)); HIP_TRY(x2.reserve(n)); HIP_TRY(v2.reserve(n)); HIP_TRY(gx.reserve(n)); HIP_TRY(gv.reserve(n));
	HIP_TRY(ok.reserve(n)); HIP_TRY(ok2.reserve(n)); HIP_TRY(oi.reserve(n)); HIP_TRY(oi2.reserve(n));
	hipLaunchKernelGGL(hao_selftest_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x.p, v.p);
	HIP_TRY(hipGetLastError());
	size_t tb = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, x.p, x2.p, v.p, v2.p, n, 48, 64, st)); HIP_TRY(tmp.reserve(tb + 256));
	HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, x.p, x2.p, v.p, v2.p, n, 48, 64, st));
	hipLaunchKernelGGL(hao_owner_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x.p, n, ok.p, oi.p);
	HIP_TRY(hipGetLastError());
	rocprim::double_buffer<uint32_t> dk(ok.p, ok2.p), dv(oi.p, oi2.p); tb = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, n, 0, 16, st)); HIP_TRY(tmp.reserve(tb + 256));
	HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, dk, dv, n, 0, 16, st));
	hipLaunchKernelGGL(hao_gather2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dv.current(), x.p, v.p, n, gx.p, gv.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	std::vector<uint64_t> hx(n), hv(n);
	for (int which = 0; which < 2; ++which) {
		HIP_TRY(hipMemcpy(hx.data(), which ? gx.p : x2.p, n * 8, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(hv.data(), which ? gv.p : v2.p, n * 8, hipMemcpyDeviceToHost));
		uint64_t bad = 0;
		for (uint64_t i = 1; i < n; ++i) { const uint64_t a = hx[i - 1] >> 48, b = hx[i] >> 48; if (a > b || (a == b && hv[i - 1] >= hv[i])) ++bad; }
		for (uint64_t i = 0; i < n; ++i) if (hv[i] >= n || hx[i] != hao_hash64(hv[i] * 0x9E3779B97F4A7C15ULL + 12345)) ++bad;      // a permutation of the input pairs
		out[which] = bad;
	}
	x.release(); v.release(); x2.release(); v2.release(); gx.release(); gv.release(); ok.release(); ok2.release(); oi.release(); oi2.release(); tmp.release();
	return HAO_OK;
}

// Self-test of the paths that see more than 2^32 items (the k-mer occurrences of a 250 Mb genome at 30x are 5.6 G): n u32 keys, key[i] = i / 8, written by
// the grid-stride launch shape the Bloom replay uses and run-length encoded by hao_rle.  out = { runs, sum of the run lengths, runs whose length is not 8 }:
// n / 8, n and 0 for n a multiple of 8.  (A launch of more than 2^32 work-items and rocprim::run_length_encode's `unsigned int size` both silently process
// n mod 2^32 items.)
__global__ void hao_selftest_fill32_kernel(uint64_t n, uint32_t *k)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) k[i] = (uint32_t)(i >> 3);
}
struct hao_not8 { __host__ __device__ uint64_t operator()(const uint32_t &v) const { return v != 8; } };
int hao_selftest_big(uint64_t n, uint64_t out[3])
{
	hao_ctx tmp_ctx; hao_ctx *c = &tmp_ctx;
	if (!out || n == 0 || (n >> 3) >= (1ULL << 32)) return HAO_EINVAL;
	out[0] = out[1] = out[2] = 0;
	DevBuf<uint32_t> k, uk, uc;
	HIP_TRY(k.reserve_exact(n)); HIP_TRY(uk.reserve_exact((n >> 3) + 2)); HIP_TRY(uc.reserve_exact((n >> 3) + 2)); HIP_TRY(c->d_cursor.reserve(2));
	hipLaunchKernelGGL(hao_selftest_fill32_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, n, k.p);
	HIP_TRY(hipGetLastError());
	if (int rc = hao_rle(c, (const uint32_t*)k.p, n, uk.p, uc.p, (uint64_t*)c->d_cursor.p)) return rc;
	HIP_TRY(hipMemcpy(&out[0], c->d_cursor.p, 8, hipMemcpyDeviceToHost));
	if (out[0] <= (n >> 3) + 1) {
		out[1] = hao_dbg_reduce(c, rocprim::make_transform_iterator(uc.p, U32ToU64()), out[0], rocprim::plus<uint64_t>());
		out[2] = hao_dbg_reduce(c, rocprim::make_transform_iterator(uc.p, hao_not8()), out[0], rocprim::plus<uint64_t>());
	}
	HIP_TRY(hipDeviceSynchronize());
	k.release(); uk.release(); uc.release(); c->d_cursor.release(); c->d_tmp.release();
	return HAO_OK;
}

int hao_pt_table(hao_ctx *c, uint64_t *n_keys, const uint64_t **keys, const uint64_t **off, const uint64_t **pos, uint64_t *n_pos)
{
	if (!c || !c->has_pt) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = hao_pt_download(c)) return rc;
	*n_keys = c->h_ix_keys.size(); *keys = c->h_ix_keys.data(); *off = c->h_ix_off.data(); *pos = c->h_ix_pos.data(); *n_pos = c->h_ix_pos.size();
	return HAO_OK;
}
}


int hao_ovlp_bin_read(const char *path, uint64_t *n_reads, uint8_t **flags, uint64_t **off, hao_ma_hit_t **hits)
{
	if (!path || !n_reads || !flags || !off || !hits) return HAO_EINVAL;
	try { return hao_ovlp_bin_read_impl(path, n_reads, flags, off, hits); }
	catch (const std::bad_alloc &) { return HAO_ENOMEM; }
	catch (const std::exception &) { return HAO_EINVAL; }
}
int hao_ovlp_bin_write(const char *path, uint64_t n_reads, const uint8_t *flags, const uint64_t *off, const hao_ma_hit_t *hits)
{ return hao_ovlp_bin_write_impl(path, n_reads, flags, off, hits); }
