// Drop-in shim: the definitions a hifiasm maintainer adds so that the three seam functions
// (ha_ft_gen htab.cpp:1136, ha_pt_gen htab.cpp:1232, h_ec_lchain anchor.cpp:2302) and the table
// accessors (ha_ft_cnt :1064, ha_pt_get :518, ha_pt_cnt :540, ha_ft_destroy, ha_pt_destroy) are
// served by libhao.so (include/hao.h).  Everything else of hifiasm links unchanged.
//
// This file is INTEGRATION code: it includes the reference's own headers and is compiled only
// where the reference sources exist (oracle/Makefile target `hao-hifiasm`, output under
// oracle/_ref/); it is not part of libhao.so.  The reference's definitions of the same symbols in
// htab.o / anchor.o are demoted to weak with objcopy, so no reference file is edited.
//
// Restrictions of this shim (exit(1) with a message, like the reference does on bad input): no trio/hp mode.  -f (Bloom filter, default 37) and
// --hg-size are forwarded.  h_ec_lchain is served from the streaming delivery path of libhao.so: the kt_for workers (kthread.cpp:31-53) consume
// batch i from a pinned host arena while batch i + 1 computes; at most two batches are ever resident on the host.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>
#include <time.h>
#include <vector>
#include "kseq.h"
#include "CommandLines.h"
#include "Process_Read.h"
#include "Hash_Table.h"
#include "htab.h"
#include "hao.h"

KSEQ_INIT(gzFile, gzread)
void ha_compress_qual(uint8_t* dest, char* src, uint64_t src_l, uint64_t bitn, uint64_t sc_off);   // Process_Read.cpp:888

static hao_ctx *g_hao = NULL;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static uint64_t g_index_gen = 0;          // bumped by every ha_pt_gen
static bool g_reads_uploaded = false;

// Streaming server behind h_ec_lchain.  Reads are cut into batches of g_bsz; a batch lives in one of the engine's two delivery slots.  kt_for hands
// out mostly increasing read ids, so the first worker that needs batch k + 1 computes it (hao_overlap_batch_async: ~ms) while the other workers keep
// decoding reads of batch k out of its arena; computing k + 1 recycles the slot of k - 1 once its readers have left.  A late request for an evicted
// batch (work stealing) simply computes it again.  A producer thread computes batch k + 1 as soon as the first worker touches batch k (look-ahead).
// Everything below is guarded by g_mu; the arenas themselves are read outside the lock under a
// per-slot reader count.
struct slot_t { int64_t batch = -1; bool ready = false; int readers = 0; hao_delivery_t view; };
static slot_t g_slot[2];
static hao_pass_t g_ps; static bool g_ps_valid = false; static uint64_t g_ps_gen = 0;
static int64_t g_computing = -1;                 // batch some worker is computing right now (the engine runs one batch at a time)
static uint64_t g_bsz = 0;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static uint64_t g_n_reads = 0; static int64_t g_prefetch = -1; static bool g_producer_on = false, g_lookahead = true;
static int64_t g_hi_batch = -1;                  // highest batch a worker has asked for in the current pass: only that one triggers a look-ahead

static void die(const char *msg) { fprintf(stderr, "[hao-shim] ERROR: %s%s%s\n", msg, g_hao ? ": " : "", g_hao ? hao_last_error(g_hao) : ""); exit(1); }
#define CK(x) do { if ((x) != 0) die(#x); } while (0)

static void ensure_ctx(const hifiasm_opt_t *o)
{
	if (g_hao) return;
	hao_opt_t p; hao_opt_default(&p);
	p.k = o->k_mer_length; p.w = o->mz_win; p.hpc = !(o->flag & HA_F_NO_HPC); p.sample_dist = o->mz_sample_dist; p.rewin = o->mz_rewin;
	p.min_hist_cnt = o->min_hist_kmer_cnt; p.max_kmer_cnt = o->max_kmer_cnt; p.max_n_chain = o->max_n_chain; p.high_factor = o->high_factor; p.is_ont = o->is_ont;
	p.bf_shift = o->bf_shift;         // the device path replays the reference's Bloom filter exactly (-f37 default included)
	p.hg_size = o->hg_size;            // prior for the peak finder (htab.cpp:1156,1254)
	if (hao_create(0, &p, &g_hao) != 0) { fprintf(stderr, "[hao-shim] ERROR: no HIP device (the device path has no CPU fallback)\n"); exit(1); }
}

// quality filter of the reader (flt_quals, htab.cpp:548-562)
static int quals_ok(const char *q, uint64_t l, uint64_t off, int64_t cut)
{
	int64_t mn = l * cut, tot = 0; uint64_t k;
	for (k = 0; k < l && tot < mn; ++k) tot += (uint8_t)q[k] - off;
	return tot >= mn;
}

// The reader half of mz1_worker_count step 0 (htab.cpp:757-806): pass 0 lengths+names, pass 1 sequences.
static void load_reads(const hifiasm_opt_t *o, All_reads *rs)
{
	for (int pass = 0; pass < 2; ++pass) {
		uint64_t n_seq = 0;
		if (pass == 0) init_All_reads(rs); else malloc_All_reads(rs);
		for (int fi = 0; fi < o->num_reads; ++fi) {
			gzFile fp = gzopen(o->read_file_names[fi], "r");
			if (!fp) continue;
			kseq_t *ks = kseq_init(fp);
			while (kseq_read(ks) >= 0) {
				int ada = o->adapterLen, l = (int)ks->seq.l - ada - ada;
				if (l <= 0 || l < o->rl_cut) continue;
				if (o->is_sc && o->sc_cut > 0 && !quals_ok(ks->qual.s + ada, l, 33, o->sc_cut)) continue;
				if (n_seq >= 1u << 28) die("this implementation supports no more than 2^28 reads");
				if (pass == 0) ha_insert_read_len(rs, l, ks->name.l);
				else {
					int i, n_N = 0;
					for (i = 0; i < l; ++i) if (seq_nt4_table[(uint8_t)ks->seq.s[i + ada]] >= 4) ++n_N;
					ha_compress_base(Get_READ(*rs, n_seq), ks->seq.s + ada, l, &rs->N_site[n_seq], n_N);
					memcpy(&rs->name[rs->name_index[n_seq]], ks->name.s, ks->name.l);
					if (rs->rsc) ha_compress_qual(Get_QUAL(*rs, n_seq), ks->qual.s + ada, l, sc_bn, 33);
				}
				++n_seq;
			}
			kseq_destroy(ks); gzclose(fp);
		}
	}
}

// All_reads (Process_Read.h:115-146) -> the flat arrays hao_set_reads takes
static void upload_reads(All_reads *rs)
{
	uint64_t n = rs->total_reads, i, nb = 0, nn = 0;
	std::vector<uint64_t> pk_off(n + 1), ns_off(n + 1); std::vector<uint32_t> len(n), ns;
	for (i = 0; i < n; ++i) { pk_off[i] = nb; nb += rs->read_length[i] / 4 + 1; ns_off[i] = nn; if (rs->N_site[i]) nn += rs->N_site[i][0]; len[i] = (uint32_t)rs->read_length[i]; }
	pk_off[n] = nb; ns_off[n] = nn;
	std::vector<uint8_t> packed(nb + 1); ns.resize(nn + 1);
	for (i = 0; i < n; ++i) {
		uint64_t bytes = rs->read_length[i] / 4 + 1, used = (rs->read_length[i] + 3) / 4;   // the last byte is uninitialised in the reference when len % 4 == 0
		memset(&packed[pk_off[i]], 0, bytes); memcpy(&packed[pk_off[i]], rs->read_sperate[i], used);
		if (rs->N_site[i]) for (uint64_t j = 1; j <= rs->N_site[i][0]; ++j) ns[ns_off[i] + j - 1] = (uint32_t)rs->N_site[i][j];
	}
	CK(hao_set_reads(g_hao, packed.data(), pk_off.data(), len.data(), n, nn ? ns_off.data() : NULL, nn ? ns.data() : NULL));
	g_reads_uploaded = true;
}

// Between passes the main thread rewrites the read store and rebuilds the tables on the same hao_ctx the look-ahead thread computes batches on: before any
// such call, cancel the look-ahead request nobody has started, invalidate the pass (the producer does not start another batch) and wait until the batch in
// flight, if any, has landed and every reader has left its slot.  (kt_for has returned by then, so no worker is inside h_ec_lchain; a look-ahead started
// by the last reads of the pass may still be running.)
static void quiesce_server()
{
	pthread_mutex_lock(&g_mu);
	g_prefetch = -1; g_ps_valid = false;
	while (g_computing >= 0 || g_slot[0].readers || g_slot[1].readers) pthread_cond_wait(&g_cv, &g_mu);
	for (int x = 0; x < 2; ++x) { g_slot[x].batch = -1; g_slot[x].ready = false; }
	pthread_mutex_unlock(&g_mu);
}

void *ha_ft_gen(const hifiasm_opt_t *asm_opt, All_reads *rs, int *hom_cov, int is_hp_mode, int read_from_store)
{
	if (is_hp_mode) die("hp mode is not replaced by the device path");
	ensure_ctx(asm_opt);
	quiesce_server();
	if (!read_from_store) load_reads(asm_opt, rs);
	upload_reads(rs);
	int32_t hc = -1;
	CK(hao_ft_gen(g_hao, &hc));
	if (hom_cov) *hom_cov = hc;
	uint64_t n; const uint64_t *k; const int32_t *v; CK(hao_ft_table(g_hao, &n, &k, &v));
	fprintf(stderr, "[M::%s] (device) peak_hom: %d; filtered out %ld k-mers\n", __func__, hc, (long)n);
	return (void*)g_hao;
}

ha_pt_t *ha_pt_gen(const hifiasm_opt_t *asm_opt, const void *flt_tab, int read_from_store, int is_hp_mode, All_reads *rs, int *hom_cov, int *het_cov)
{
	if (is_hp_mode) die("hp mode is not replaced by the device path");
	ensure_ctx(asm_opt);
	quiesce_server();
	if (!read_from_store && rs->total_reads == 0) load_reads(asm_opt, rs);
	if (read_from_store || !g_reads_uploaded || flt_tab == 0) upload_reads(rs);     // reads were rewritten by the previous round
	int32_t hc = -1, ht = -1;
	CK(hao_pt_gen(g_hao, &hc, &ht));
	if (hom_cov) *hom_cov = hc;
	if (het_cov) *het_cov = ht;
	uint64_t nk, np; const uint64_t *k, *o, *p; (void)k; (void)o; (void)p;
	pthread_mutex_lock(&g_mu); ++g_index_gen; g_ps_valid = false; pthread_mutex_unlock(&g_mu);
	(void)nk; (void)np;
	fprintf(stderr, "[M::%s] (device) peak_hom: %d; peak_het: %d\n", __func__, hc, ht);
	return (ha_pt_t*)g_hao;
}

int32_t ha_ft_cnt(const void *hh, uint64_t y) { return hh ? hao_ft_cnt(g_hao, y) : 0; }
void ha_ft_destroy(void *h) { (void)h; }
void ha_pt_destroy(ha_pt_t *h) { (void)h; }
const ha_idxpos_t *ha_pt_get(const ha_pt_t *h, uint64_t hash, int *n)
{
	const uint64_t *pos = NULL; int32_t m = 0; (void)h;
	CK(hao_pt_get(g_hao, hash, &pos, &m));
	*n = m;
	return (const ha_idxpos_t*)pos;       // same 8-byte bit layout (htab.h:20-22)
}
const int ha_pt_cnt(const ha_pt_t *h, uint64_t hash) { int n; ha_pt_get(h, hash, &n); return n; }

static_assert(sizeof(k_mer_hit) == sizeof(hao_hit_t), "k_mer_hit layout");
// the slot that holds batch k and is ready, or -1 (g_mu held)
static int find_slot(int64_t k) { for (int x = 0; x < 2; ++x) if (g_slot[x].batch == k && g_slot[x].ready) return x; return -1; }

static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double g_wait_s = 0, g_compute_s = 0; static uint64_t g_n_wait = 0, g_n_batches = 0, g_n_prefetched = 0;

// Compute batch k of the current pass into the engine's next delivery slot (g_mu held on entry and exit; released around the device work).  The engine
// says which slot it will write (hao_next_slot): that arena may still be read by stragglers of the batch it holds - wait for them first.
static void compute_batch(int64_t k)
{
	g_computing = k;
	int nx = 0; CK(hao_next_slot(g_hao, &nx));
	while (g_slot[nx].readers) pthread_cond_wait(&g_cv, &g_mu);
	g_slot[nx].batch = -1; g_slot[nx].ready = false;
	const hao_pass_t ps = g_ps; const uint64_t n_reads = g_n_reads;
	pthread_mutex_unlock(&g_mu);
	const double t0 = now_s();
	const uint64_t lo = (uint64_t)k * g_bsz, hi = lo + g_bsz < n_reads ? lo + g_bsz : n_reads; int got = -1; hao_delivery_t view;
	// Every caller gets ol->list AND cl->list.  Opt-in (HAO_SHIM_FINAL_OL_ONLY=1): the final round (worker_hap_dc_ec_gen_new_idx, ecovlp.cpp:3948-3972, recognised by
	// its bw_thres = 0.001 - the reference offers no other marker) reads ol->list only - h_ec_lchain_fast_new and push_ff_ovlp never look at cl->list - so its
	// chained hits can stay on the device.  Not the default: a caller that passes 0.001 and does read cl->list would silently see an empty list.
	static const bool final_ol_only = getenv("HAO_SHIM_FINAL_OL_ONLY") != nullptr;
	const uint32_t parts = HAO_DELIVER_OL | (final_ol_only && ps.bw_thres == 0.001 ? 0u : (uint32_t)HAO_DELIVER_CL);
	CK(hao_overlap_batch_async(g_hao, lo, hi, &ps, parts, &got));
	CK(hao_deliver_wait(g_hao, got, &view));
	pthread_mutex_lock(&g_mu);
	if (got != nx) die("delivery slot order");
	g_slot[got].batch = k; g_slot[got].ready = true; g_slot[got].view = view;
	g_compute_s += now_s() - t0; ++g_n_batches;
	g_computing = -1;
	pthread_cond_broadcast(&g_cv);
}

static void *producer_main(void *)
{
	pthread_mutex_lock(&g_mu);
	for (;;) {
		while (g_prefetch < 0 || g_computing >= 0 || !g_ps_valid) pthread_cond_wait(&g_cv, &g_mu);
		const int64_t k = g_prefetch; g_prefetch = -1;
		if (find_slot(k) >= 0 || (uint64_t)k * g_bsz >= g_n_reads) continue;      // a worker got there first / past the end
		compute_batch(k); ++g_n_prefetched;
	}
	return NULL;
}

__attribute__((destructor)) static void shim_report()
{
	if (getenv("HAO_SHIM_STATS") && g_n_batches)
		fprintf(stderr, "[hao-shim] %llu batches of %llu reads (%llu by the look-ahead thread), %.1f ms each on the device path; workers waited for a batch %llu times, %.2f ms per batch in total\n",
				(unsigned long long)g_n_batches, (unsigned long long)g_bsz, (unsigned long long)g_n_prefetched, 1e3 * g_compute_s / g_n_batches, (unsigned long long)g_n_wait, 1e3 * g_wait_s / g_n_batches);
}

void h_ec_lchain(ha_abuf_t *ab, uint32_t rid, char* rs, uint64_t rl, uint64_t mz_w, uint64_t mz_k, All_reads *rref, overlap_region_alloc *overlap_list, Candidates_list *cl, double bw_thres,
				 int max_n_chain, int apend_be, kvec_t_u8_warp* k_flag, kvec_t_u64_warp* dbg_ct, st_mt_t *sp, uint32_t *high_occ, uint32_t *low_occ, uint32_t is_accurate, uint32_t gen_off,
				 int64_t mcopy_num, double mcopy_rate, uint32_t chain_cutoff, uint32_t mcopy_khit_cut, uint64_t ocv_w)
{
	(void)ab; (void)rs; (void)rl; (void)mz_w; (void)mz_k; (void)k_flag; (void)dbg_ct; (void)sp;
	hao_pass_t ps; memset(&ps, 0, sizeof(ps));
	ps.bw_thres = bw_thres; ps.max_n_chain = max_n_chain; ps.high_occ = high_occ ? *high_occ : UINT32_MAX; ps.low_occ = low_occ ? *low_occ : 0;
	ps.apend_be = apend_be; ps.is_accurate = is_accurate; ps.gen_off = gen_off; ps.mcopy_num = (int32_t)mcopy_num; ps.mcopy_rate = mcopy_rate;
	ps.chain_cutoff = chain_cutoff; ps.mcopy_khit_cut = mcopy_khit_cut; ps.ocv_w = ocv_w;
	const uint64_t n_reads = rref->total_reads;
	pthread_mutex_lock(&g_mu);
	if (!g_bsz) { const char *e = getenv("HAO_SHIM_BATCH"); g_bsz = e ? strtoull(e, 0, 10) : 4096; if (!g_bsz) g_bsz = 4096; g_lookahead = getenv("HAO_SHIM_NO_LOOKAHEAD") == NULL; }
	// a new pass (other arguments - e.g. the final round's bw_thres = 0.001, ecovlp.cpp:3957 - or a rebuilt index): drain the readers, forget the slots
	while (!g_ps_valid || g_ps_gen != g_index_gen || memcmp(&g_ps, &ps, sizeof(ps)) != 0) {
		if (g_computing >= 0 || g_slot[0].readers || g_slot[1].readers) { pthread_cond_wait(&g_cv, &g_mu); continue; }
		g_prefetch = -1;      // (a look-ahead request of the old pass that nobody has started)
		g_ps = ps; g_ps_gen = g_index_gen; g_ps_valid = true; g_hi_batch = -1;
		for (int x = 0; x < 2; ++x) { g_slot[x].batch = -1; g_slot[x].ready = false; }
	}
	const int64_t k = (int64_t)(rid / g_bsz);
	g_n_reads = n_reads;
	int sl; bool waited = false; const double tw0 = now_s();
	while ((sl = find_slot(k)) < 0) {
		waited = true;
		if (g_computing >= 0) { pthread_cond_wait(&g_cv, &g_mu); continue; }       // somebody (a worker or the look-ahead thread) is computing - maybe this very batch: wait and look again
		compute_batch(k);                                                           // (drops and retakes g_mu around the device work)
	}
	if (waited) { g_wait_s += now_s() - tw0; ++g_n_wait; }
	// look-ahead: kt_for hands out ascending read ids, so batch k + 1 is needed as soon as the workers are through with k - the first reader of k asks
	// the producer thread for it now, and it computes + lands in the other arena while k is being decoded (without it every worker would finish k and
	// then all of them would sit through the whole compute + copy of k + 1)
	// Only the front of the pass looks ahead: a straggler that had to recompute an evicted batch j must not ask for j + 1 (already consumed) - that would
	// evict a live batch and leave a stray computation running when kt_for returns.
	const bool front = k >= g_hi_batch; if (front) g_hi_batch = k;
	if (g_lookahead && front && (uint64_t)(k + 1) * g_bsz < n_reads && find_slot(k + 1) < 0 && g_computing != k + 1 && g_prefetch != k + 1) {
		g_prefetch = k + 1;
		if (!g_producer_on) { g_producer_on = true; pthread_t th; if (pthread_create(&th, NULL, producer_main, NULL) != 0) die("pthread_create"); pthread_detach(th); }
		pthread_cond_broadcast(&g_cv);
	}
	++g_slot[sl].readers;
	const hao_delivery_t d = g_slot[sl].view;
	pthread_mutex_unlock(&g_mu);
	// ---- decode out of the arena (no lock: the reader count keeps the slot alive) ----
	// ol->list: cleared, then one zero-initialised slot per overlap (kv_pushp_ol, Hash_Table.h:251-258); scalar fields as
	// push_ovlp_chain_qgen sets them (Hash_Table.cpp:1752-1780); f_cigar through the slot's own buffer
	const uint64_t r = rid - d.rid_lo;
	clear_overlap_region_alloc(overlap_list);
	static thread_local std::vector<hao_ovlp_t> olb;      // the wire's 32-byte records back into hao_ovlp_t
	const uint64_t n_ol = d.ol_off[r + 1] - d.ol_off[r];
	if (olb.size() < n_ol) olb.resize(n_ol + 64);
	if (hao_unpack_overlaps(&d, rid, olb.data(), n_ol) != n_ol) die("hao_unpack_overlaps");
	for (uint64_t i = d.ol_off[r]; i < d.ol_off[r + 1]; ++i) {
		const hao_ovlp_t &o = olb[i - d.ol_off[r]]; overlap_region *z;
		kv_pushp_ol(overlap_region, (*overlap_list), &z);
		z->x_id = o.x_id; z->x_pos_s = o.x_pos_s; z->x_pos_e = o.x_pos_e; z->x_pos_strand = o.x_pos_strand;
		z->y_id = o.y_id; z->y_pos_s = o.y_pos_s; z->y_pos_e = o.y_pos_e; z->y_pos_strand = o.y_pos_strand;
		z->shared_seed = o.shared_seed; z->align_length = 0; z->is_match = 0; z->non_homopolymer_errors = o.non_homopolymer_errors; z->strong = 0; z->overlapLen = 0;
		resize_fake_cigar(&z->f_cigar, o.fc_len, NULL);
		if (hao_unpack_cigar(&d, i, (uint64_t*)z->f_cigar.buffer, o.fc_len) != o.fc_len) die("hao_unpack_cigar");      // (the wire carries 4 bytes per cigar entry)
		z->f_cigar.length = o.fc_len;
	}
	// cl->list: the wire bytes decode straight into the caller's list (k_mer_hit and hao_hit_t share their layout, Hash_Table.h:116-120)
	const uint64_t n_cl = d.cl_off ? d.cl_off[r + 1] - d.cl_off[r] : 0;      // (no cl_off: a batch delivered without its chained hits - the final round)
	clear_Candidates_list(cl);
	if ((uint64_t)cl->size < n_cl + 1) { cl->size = n_cl + 1; REALLOC(cl->list, cl->size); }
	if (n_cl && hao_unpack_hits(&d, rid, (hao_hit_t*)cl->list, n_cl) != n_cl) die("hao_unpack_hits");
	cl->length = n_cl;
	pthread_mutex_lock(&g_mu);
	if (--g_slot[sl].readers == 0) pthread_cond_broadcast(&g_cv);
	pthread_mutex_unlock(&g_mu);
}
