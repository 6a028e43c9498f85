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
// Restrictions of this shim (exit(1) with a message, like the reference does on bad input):
// -f0 only (exact counting), no trio/hp mode, whole pass results are kept in host memory
// (fine for the plumbing configuration; a production shim streams batches).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>
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

struct pass_cache_t {
	uint64_t gen = 0; bool valid = false; hao_pass_t ps;
	std::vector<uint64_t> ol_off, cl_off, fc_off_base; std::vector<hao_ovlp_t> ol; std::vector<hao_hit_t> cl; std::vector<uint64_t> fc, fc_off;
};
static pass_cache_t g_pass;

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

void *ha_ft_gen(const hifiasm_opt_t *asm_opt, All_reads *rs, int *hom_cov, int is_hp_mode, int read_from_store)
{
	if (is_hp_mode) die("hp mode is not replaced by the device path");
	ensure_ctx(asm_opt);
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
	if (!read_from_store && rs->total_reads == 0) load_reads(asm_opt, rs);
	if (read_from_store || !g_reads_uploaded || flt_tab == 0) upload_reads(rs);     // reads were rewritten by the previous round
	int32_t hc = -1, ht = -1;
	CK(hao_pt_gen(g_hao, &hc, &ht));
	if (hom_cov) *hom_cov = hc;
	if (het_cov) *het_cov = ht;
	uint64_t nk, np; const uint64_t *k, *o, *p; (void)k; (void)o; (void)p;
	pthread_mutex_lock(&g_mu); ++g_index_gen; g_pass.valid = false; pthread_mutex_unlock(&g_mu);
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

// one all-reads pass on the device, results kept on the host until the index changes
static void run_pass(const hao_pass_t *ps, uint64_t n_reads)
{
	pass_cache_t &P = g_pass;
	P.ol_off.assign(n_reads + 1, 0); P.cl_off.assign(n_reads + 1, 0); P.ol.clear(); P.cl.clear(); P.fc.clear(); P.fc_off.clear();
	const uint64_t B = 2048;
	for (uint64_t lo = 0; lo < n_reads; lo += B) {
		uint64_t hi = lo + B < n_reads ? lo + B : n_reads;
		CK(hao_overlap_batch_ex(g_hao, lo, hi, ps));
		for (uint64_t r = lo; r < hi; ++r) {
			const hao_ovlp_t *ol; const uint64_t *fc, *fo; const hao_hit_t *cl; uint64_t n_ol, n_cl;
			CK(hao_fetch_overlaps(g_hao, r, &ol, &n_ol, &fc, &fo, &cl, &n_cl));
			P.ol_off[r] = P.ol.size(); P.cl_off[r] = P.cl.size();
			for (uint64_t i = 0; i < n_ol; ++i) { P.fc_off.push_back(P.fc.size()); P.fc.insert(P.fc.end(), fc + (fo[i] - fo[0]), fc + (fo[i] - fo[0]) + ol[i].fc_len); }
			P.ol.insert(P.ol.end(), ol, ol + n_ol); P.cl.insert(P.cl.end(), cl, cl + n_cl);
		}
	}
	P.ol_off[n_reads] = P.ol.size(); P.cl_off[n_reads] = P.cl.size(); P.fc_off.push_back(P.fc.size());
	P.ps = *ps; P.gen = g_index_gen; P.valid = true;
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
	pthread_mutex_lock(&g_mu);
	if (!g_pass.valid || g_pass.gen != g_index_gen || memcmp(&g_pass.ps, &ps, sizeof(ps)) != 0) run_pass(&ps, rref->total_reads);
	pthread_mutex_unlock(&g_mu);
	const pass_cache_t &P = g_pass;
	// ol->list: cleared, then one zero-initialised slot per overlap (kv_pushp_ol, Hash_Table.h:251-258); scalar fields as
	// push_ovlp_chain_qgen sets them (Hash_Table.cpp:1752-1780); f_cigar through the slot's own buffer
	clear_overlap_region_alloc(overlap_list);
	for (uint64_t i = P.ol_off[rid]; i < P.ol_off[rid + 1]; ++i) {
		const hao_ovlp_t &s = P.ol[i]; overlap_region *z;
		kv_pushp_ol(overlap_region, (*overlap_list), &z);
		z->x_id = s.x_id; z->x_pos_s = s.x_pos_s; z->x_pos_e = s.x_pos_e; z->x_pos_strand = s.x_pos_strand;
		z->y_id = s.y_id; z->y_pos_s = s.y_pos_s; z->y_pos_e = s.y_pos_e; z->y_pos_strand = s.y_pos_strand;
		z->shared_seed = s.shared_seed; z->align_length = 0; z->is_match = 0; z->non_homopolymer_errors = s.non_homopolymer_errors; z->strong = 0; z->overlapLen = 0;
		resize_fake_cigar(&z->f_cigar, s.fc_len, NULL);
		memcpy(z->f_cigar.buffer, &P.fc[P.fc_off[i]], sizeof(uint64_t) * s.fc_len); z->f_cigar.length = s.fc_len;
	}
	// cl->list
	uint64_t n_cl = P.cl_off[rid + 1] - P.cl_off[rid];
	clear_Candidates_list(cl);
	if ((uint64_t)cl->size < n_cl + 1) { cl->size = n_cl + 1; REALLOC(cl->list, cl->size); }
	memcpy(cl->list, &P.cl[P.cl_off[rid]], n_cl * sizeof(k_mer_hit));
	cl->length = n_cl;
}
