// TEST INFRASTRUCTURE (oracle/_ref): wrapper translation unit around the
// UNMODIFIED reference htab.cpp.  The reference text is compiled from where it
// lies (-I/root/reference; nothing is copied into this repository); we only append
// enumeration helpers because yak_ft_t / yak_pt_t / ha_pt_s are file-static
// types there (htab.cpp:122-124, 299-314, 1036) and the dumps need to walk them.
#include "htab.cpp"

#include <vector>
#include <algorithm>

// Enumerate the high-count filter table (yak_ft_t, htab.cpp:1036-1070).
// Returns parallel arrays sorted by key. val is the raw int16 map value.
extern "C" uint64_t refdump_ft(void *flt_tab, uint64_t **keys_out, int32_t **vals_out)
{
	yak_ft_t *h = (yak_ft_t*)flt_tab;
	std::vector<std::pair<uint64_t,int32_t> > v;
	if (h) {
		for (khint_t k = 0; k < kh_end(h); ++k)
			if (kh_exist(h, k)) v.push_back(std::make_pair((uint64_t)kh_key(h, k), (int32_t)kh_val(h, k)));
	}
	std::sort(v.begin(), v.end());
	uint64_t n = v.size();
	*keys_out = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
	*vals_out = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
	for (uint64_t i = 0; i < n; ++i) (*keys_out)[i] = v[i].first, (*vals_out)[i] = v[i].second;
	return n;
}

// Enumerate the position index (ha_pt_t, htab.cpp:303-314): for every key the
// full 64-bit hash, the occurrence count and the ordered ha_idxpos_t list.
// keys sorted ascending; off[] is the CSR offset array (n_keys+1); pos[] raw 8-byte records.
extern "C" uint64_t refdump_pt(ha_pt_t *pt, uint64_t **keys_out, uint64_t **off_out, uint64_t **pos_out, uint64_t *n_pos_out)
{
	struct ent { uint64_t key; const ha_idxpos_t *a; uint32_t n; };
	std::vector<ent> v;
	uint64_t tot = 0;
	for (int i = 0; i < 1<<pt->pre; ++i) {
		ha_pt1_t *g = &pt->h[i];
		for (khint_t k = 0; k < kh_end(g->h); ++k) {
			if (!kh_exist(g->h, k)) continue;
			ent e;
			e.key = (kh_key(g->h, k) >> YAK_COUNTER_BITS << pt->pre) | (uint64_t)i;
			e.n = kh_key(g->h, k) & YAK_MAX_COUNT;
			e.a = &g->a[kh_val(g->h, k)];
			tot += e.n;
			v.push_back(e);
		}
	}
	std::sort(v.begin(), v.end(), [](const ent &a, const ent &b){ return a.key < b.key; });
	uint64_t n = v.size();
	*keys_out = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
	*off_out = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
	*pos_out = (uint64_t*)malloc(sizeof(uint64_t) * (tot + 1));
	uint64_t o = 0;
	for (uint64_t i = 0; i < n; ++i) {
		(*keys_out)[i] = v[i].key; (*off_out)[i] = o;
		memcpy(*pos_out + o, v[i].a, sizeof(uint64_t) * v[i].n);
		o += v[i].n;
	}
	(*off_out)[n] = o;
	*n_pos_out = tot;
	return n;
}

// All-k-mer count histogram + peaks of the ha_ft_gen counting pass, re-run here
// because ha_ft_gen (htab.cpp:1136-1169) frees the count table before returning.
// Mirrors exactly the calls ha_ft_gen makes (same flags, same arguments).
extern "C" void refdump_ft_hist(const hifiasm_opt_t *o, All_reads *rs, int64_t cnt[YAK_N_COUNTS], int *peak_hom, int *peak_het, uint64_t *n_distinct)
{
	ha_ct_t *h = ha_count(o, HAF_COUNT_ALL|HAF_RS_READ, !(o->flag&HA_F_NO_HPC), o->k_mer_length, o->mz_win, NULL, NULL, rs, NULL, 1, NULL, 0);
	ha_ct_hist(h, cnt, o->thread_num);
	*peak_hom = ha_analyze_count(YAK_N_COUNTS, o->min_hist_kmer_cnt, o->hg_size>0?(h->bs/o->hg_size):(-1), cnt, peak_het);
	*n_distinct = h->tot;
	ha_ct_destroy(h);
}

// Minimizer-count histogram of ha_pt_gen's pass A (htab.cpp:1249-1256), re-run.
extern "C" void refdump_pt_hist(const hifiasm_opt_t *o, const void *flt_tab, All_reads *rs, int64_t cnt[YAK_N_COUNTS], uint64_t *n_distinct)
{
	ha_ct_t *ct = ha_count(o, HAF_COUNT_EXACT|HAF_RS_READ, !(o->flag&HA_F_NO_HPC), o->k_mer_length, o->mz_win, NULL, flt_tab, rs, NULL, 1, NULL, 0);
	ha_ct_hist(ct, cnt, o->thread_num);
	*n_distinct = ct->tot;
	ha_ct_destroy(ct);
}
