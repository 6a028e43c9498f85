// remaining C ABI entry points + resource release
#pragma once

static void hao_batch_free(hao_ctx *c) { (void)c; }

static void hao_release_all(hao_ctx *c)
{
	c->d_packed.release(); c->d_pk_off.release(); c->d_len.release(); c->d_nsite_off.release(); c->d_nsite.release();
	c->d_ft_keys.release(); c->d_ft_vals.release(); c->d_ft_bucket.release();
	c->d_tile_off.release(); c->d_tile_ord.release(); c->d_n_runs.release(); c->d_tot_l.release(); c->d_chunk_off.release(); c->d_chunk_cnt64.release();
	c->d_scalar_flag.release(); c->d_scalar_list.release(); c->d_pool_x.release(); c->d_pool_info.release(); c->d_pool_ord.release(); c->d_cursor.release(); c->d_err.release();
	c->d_chunk_base.release(); c->d_chunk_dst.release(); c->d_chunk_cnt.release(); c->d_g_x.release(); c->d_g_info.release(); c->d_g_ord.release(); c->d_g_off.release();
	c->d_new_n.release(); c->d_new_n64.release(); c->d_mz_x.release(); c->d_mz_info.release(); c->d_mz_off.release(); c->d_tmp.release(); c->d_ring.release(); c->d_ringord.release(); c->d_cnt_ws.release();
	c->d_ix_mz_x.release(); c->d_ix_mz_info.release(); c->d_ix_mz_off.release(); c->d_ix_sx.release(); c->d_ix_sinfo.release();
	c->d_ix_keys.release(); c->d_ix_start.release(); c->d_ix_cnt.release(); c->d_ix_bucket.release();
}

extern "C" {
#ifndef HAO_HAVE_FT
int hao_ft_gen(hao_ctx *c, int32_t *hom_cov) { hao_set_err(c, "not implemented"); return HAO_EINVAL; }
int32_t hao_ft_cnt(hao_ctx *c, uint64_t y) { return 0; }
int hao_ft_table(hao_ctx *c, uint64_t *n, const uint64_t **keys, const int32_t **vals) { return HAO_EINVAL; }
int hao_hist(hao_ctx *c, int which, int64_t cnt[4096]) { return HAO_EINVAL; }
int hao_stats(hao_ctx *c, int64_t out[8]) { return HAO_EINVAL; }
#endif
#ifndef HAO_HAVE_PT
int hao_pt_gen(hao_ctx *c, int32_t *hom_cov, int32_t *het_cov) { hao_set_err(c, "not implemented"); return HAO_EINVAL; }
int hao_pt_get(hao_ctx *c, uint64_t hash, const uint64_t **pos, int32_t *n) { return HAO_EINVAL; }
int hao_pt_table(hao_ctx *c, uint64_t *n_keys, const uint64_t **keys, const uint64_t **off, const uint64_t **pos, uint64_t *n_pos) { return HAO_EINVAL; }
#endif
#ifndef HAO_HAVE_QUERY
int hao_overlap_batch(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi) { hao_set_err(c, "not implemented"); return HAO_EINVAL; }
int hao_fetch_seed_hits(hao_ctx *c, uint64_t rid, const hao_hit_t **hits, uint64_t *n) { return HAO_EINVAL; }
int hao_fetch_overlaps(hao_ctx *c, uint64_t rid, const hao_ovlp_t **ol, uint64_t *n_ol, const uint64_t **fc, const uint64_t **fc_off, const hao_hit_t **cl, uint64_t *n_cl) { return HAO_EINVAL; }
int hao_batch_totals(hao_ctx *c, uint64_t out[8]) { return HAO_EINVAL; }
#endif
}
