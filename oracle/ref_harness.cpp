// TEST INFRASTRUCTURE (oracle/_ref): driver that links the UNMODIFIED reference
// objects (chhylp123/hifiasm 0.25.0-r726, compiled from /root/reference where
// they lie) and (a) dumps every intermediate of the overlap hot path so the C
// restatement in oracle/hao_oracle.c and the HIP path can be pinned against
// the real thing, (b) times the reference CPU path for bench.py's
// cpu_baseline {"kind":"reference"}.
//
// Call sequence = what ha_assemble does up to the first all-reads pass
// (Assembly.cpp:2083-2084 ha_ft_gen + ha_opt_update_cov; Assembly.cpp:1007-1008
// ha_pt_gen; ecovlp.cpp:3234-3274 worker_hap_ec up to and including h_ec_lchain).
//
// usage: ref_harness [--ont] [-t N] [-k K] [-w W] [-f BLOOM_BITS] [--no-hpc] [--hg-size N] [--bw X] [-N MAX_N_CHAIN] [--rl-cut N] [--sc-cut N] [--dump PREFIX] [--time] [--nodump-hits]
//                    [--reads-list FILE] [--no-tables] [--digest] reads.fa
//   --reads-list FILE  per-read dumps (minimizers, seed hits, ol / fc / cl) only for the read ids listed in FILE (text, one per line);
//                      the *_off arrays then have one entry per LISTED read (+1), in list order
//   --no-tables        skip the ft / pt table dumps and the hf = NULL minimizer dump (large read sets)
//   --digest           PREFIX.dig.u64 = per read [digest of (ol, fc, cl), digest of the seed hits], all reads, all threads; the digest is
//                      hao_batch_digest's (include/hao.h): a position-salted sum of mixed 64-bit words, so the device computes the same
//                      value with a parallel reduction
//   --ed-tasks FILE    f3: FILE = uint32[n][10] tasks (p_rid, p_pos, p_len, p_rev, t_rid, t_pos, t_len, t_rev, thre, abs_diag); PREFIX.ed.i32 = int32[n][2]
//                      (err, pe) of the reference's ed_band_cal_semi_64_w_absent_diag (Levenshtein_distance.h:3727) on the strings
//                      recover_UC_Read_sub_region (Process_Read.cpp:524) builds for those intervals
//   --edg-tasks FILE   f3, second variant: same task records (abs_diag unused); PREFIX.edg.i32 = int32[n][6] (err, ps, pe, ts, te, cigar entries) and
//                      PREFIX.edg_cig.u16 = the cigars, concatenated, of ed_band_cal_global_64_w_trace (Levenshtein_distance.h:3370) on a cleared bit_extz_t
//   --eds-tasks FILE   the same for ed_band_cal_semi_64_w_absent_diag_trace (:3778; abs_diag is used): PREFIX.eds.i32, PREFIX.eds_cig.u16
//   --ed1-tasks / --ed2-tasks FILE   the same for ed_band_cal_extension_64_0_w_trace / _1_w_trace (:3512, :3620): PREFIX.ed1.* / PREFIX.ed2.* (the bit_extz_t fields as
//                      the call leaves them, also without an alignment)
//   --load-index PFX   f4: skip ha_ft_gen / ha_pt_gen and the read parser: the tables and the read store come from PFX.pt_flt (+ .bin, .paf.bin) through
//                      the reference's own load_pt_index (htab.cpp:1432); every dump then describes what a stock hifiasm sees after loading that index
//   -N X / --rl-cut N / --sc-cut N   handed to the reference's own option parser unchanged (CommandLines.cpp:891, :1003-1005): the floor of max_n_chain
//                      (ha_opt_update_cov raises it to hom_cov * high_factor, :411-418) and the --ont reader's length / quality cuts (htab.cpp:763-764; the
//                      random workloads of tests/simt_fuzz.py have reads shorter than the default 1000)
//   --bw X             bw_thres of the pass (default 0.02 / 0.05 --ont; the final round uses 0.001, ecovlp.cpp:3957)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <thread>
#include "CommandLines.h"
#include "Process_Read.h"
#include "Hash_Table.h"
#include "htab.h"
#include "kthread.h"
#include "Levenshtein_distance.h"

#define HA_KMER_GOOD_RATIO 0.333   // ecovlp.cpp / anchor.cpp:11
#define COV_W 3072                 // ecovlp.cpp:16

// ad-hoc prototypes exactly as the reference declares them (ecovlp.cpp:110; anchor.cpp:987)
void h_ec_lchain(ha_abuf_t *ab, uint32_t rid, char* rs, uint64_t rl, uint64_t mz_w, uint64_t mz_k, All_reads *rref, overlap_region_alloc *overlap_list, Candidates_list *cl, double bw_thres,
				 int max_n_chain, int apend_be, kvec_t_u8_warp* k_flag, kvec_t_u64_warp* dbg_ct, st_mt_t *sp, uint32_t *high_occ, uint32_t *low_occ, uint32_t is_accurate, uint32_t gen_off, int64_t mcopy_num, double mcopy_rate, uint32_t chain_cutoff, uint32_t mcopy_khit_cut, uint64_t ocv_w);
void minimizers_qgen0(ha_abuf_t *ab, char* rs, int64_t rl, uint64_t mz_w, uint64_t mz_k, Candidates_list *cl, kvec_t_u8_warp* k_flag,
				 void *ha_flt_tab, ha_pt_t *ha_idx, All_reads* rdb, kvec_t_u64_warp* dbg_ct, st_mt_t *sp, uint32_t *high_occ, uint32_t *low_occ);

extern "C" uint64_t refdump_ft(void *flt_tab, uint64_t **keys_out, int32_t **vals_out);
extern "C" uint64_t refdump_pt(ha_pt_t *pt, uint64_t **keys_out, uint64_t **off_out, uint64_t **pos_out, uint64_t *n_pos_out);
extern "C" void refdump_ft_hist(const hifiasm_opt_t *o, All_reads *rs, int64_t cnt[4096], int *peak_hom, int *peak_het, uint64_t *n_distinct);
extern "C" void refdump_pt_hist(const hifiasm_opt_t *o, const void *flt_tab, All_reads *rs, int64_t cnt[4096], uint64_t *n_distinct);

static void wr(const std::string &prefix, const char *name, const void *p, size_t nbytes)
{
	std::string fn = prefix + "." + name;
	FILE *fp = fopen(fn.c_str(), "wb");
	if (!fp) { fprintf(stderr, "cannot write %s\n", fn.c_str()); exit(1); }
	if (nbytes) fwrite(p, 1, nbytes, fp);
	fclose(fp);
}

typedef struct {
	UC_Read ur; ha_abuf_t *ab; overlap_region_alloc ol; Candidates_list cl; st_mt_t sp;
	uint64_t n_ovlp, n_hits;
} tbuf_t;

typedef struct { tbuf_t *b; double bw; uint32_t high_occ, low_occ; uint64_t *dig; } pass_t;

// ---- per-read result digest (same definition as hao_batch_digest, include/hao.h) ----
static inline uint64_t dg_mix(uint64_t z) { z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL; z ^= z >> 27; z *= 0x94d049bb133111ebULL; z ^= z >> 31; return z; }
static inline uint64_t dg_term(uint64_t stream, uint64_t i, uint64_t w) { return dg_mix(w + 0x9E3779B97F4A7C15ULL * (i + 1) + stream * 0xD6E8FEB86659FD93ULL); }
static void ol_words(const overlap_region *r, uint64_t q[6])
{
	uint32_t f[12] = { r->x_id, r->x_pos_s, r->x_pos_e, r->x_pos_strand, r->y_id, r->y_pos_s, r->y_pos_e, r->y_pos_strand,
					   (uint32_t)r->shared_seed, r->align_length, r->non_homopolymer_errors, r->f_cigar.length };
	for (int t = 0; t < 6; ++t) q[t] = (uint64_t)f[2 * t] | (uint64_t)f[2 * t + 1] << 32;
}
static uint64_t digest_result(const overlap_region_alloc *ol, const Candidates_list *cl)
{
	uint64_t d = 0, nfc = 0;
	for (uint64_t j = 0; j < ol->length; ++j) {
		uint64_t q[6]; ol_words(&ol->list[j], q);
		for (int t = 0; t < 6; ++t) d += dg_term(1, j * 6 + t, q[t]);
		for (uint32_t c = 0; c < ol->list[j].f_cigar.length; ++c) d += dg_term(2, nfc++, ol->list[j].f_cigar.buffer[c]);
	}
	for (uint64_t j = 0; j < (uint64_t)cl->length; ++j) { uint64_t q[2]; memcpy(q, &cl->list[j], 16); d += dg_term(3, 2 * j, q[0]) + dg_term(3, 2 * j + 1, q[1]); }
	return d;
}
static uint64_t digest_hits(const Candidates_list *cl)
{
	uint64_t d = 0;
	for (uint64_t j = 0; j < (uint64_t)cl->length; ++j) { uint64_t q[2]; memcpy(q, &cl->list[j], 16); d += dg_term(4, 2 * j, q[0]) + dg_term(4, 2 * j + 1, q[1]); }
	return d;
}

static void worker_digest(void *data, long i, int tid)
{
	pass_t *p = (pass_t*)data; tbuf_t *b = &p->b[tid];
	uint32_t high_occ = p->high_occ, low_occ = p->low_occ;
	recover_UC_Read(&b->ur, &R_INF, i);
	minimizers_qgen0(b->ab, b->ur.seq, b->ur.length, asm_opt.mz_win, asm_opt.k_mer_length, &b->cl, NULL, ha_flt_tab, ha_idx, &R_INF, NULL, &b->sp, &high_occ, &low_occ);
	p->dig[2 * i + 1] = digest_hits(&b->cl);
	high_occ = p->high_occ; low_occ = p->low_occ;
	h_ec_lchain(b->ab, i, b->ur.seq, b->ur.length, asm_opt.mz_win, asm_opt.k_mer_length, &R_INF, &b->ol, &b->cl, p->bw, asm_opt.max_n_chain, 1, NULL, NULL, &b->sp, &high_occ, &low_occ, 1, 1, 3, 0.7, 2, 32, COV_W);
	p->dig[2 * i] = digest_result(&b->ol, &b->cl);
}

static void worker_pass(void *data, long i, int tid)
{
	pass_t *p = (pass_t*)data; tbuf_t *b = &p->b[tid];
	uint32_t high_occ = p->high_occ, low_occ = p->low_occ;
	recover_UC_Read(&b->ur, &R_INF, i);
	h_ec_lchain(b->ab, i, b->ur.seq, b->ur.length, asm_opt.mz_win, asm_opt.k_mer_length, &R_INF, &b->ol, &b->cl, p->bw, asm_opt.max_n_chain, 1, NULL, NULL, &b->sp, &high_occ, &low_occ, 1, 1, 3, 0.7, 2, 32, COV_W);
	b->n_ovlp += b->ol.length; b->n_hits += b->cl.length;
}

static tbuf_t *tbuf_init(int n)
{
	tbuf_t *b = (tbuf_t*)calloc(n, sizeof(tbuf_t));
	for (int i = 0; i < n; ++i) {
		init_UC_Read(&b[i].ur); b[i].ab = ha_abuf_init();
		init_overlap_region_alloc(&b[i].ol); init_Candidates_list(&b[i].cl);
	}
	return b;
}

int main(int argc, char *argv[])
{
	int no_tables_hist = 0, ft_tables = 0;
	int n_thread = 1, is_ont = 0, do_time = 0, dump_hits = 1, k = -1, w = -1, bf_shift = 0, no_hpc = 0, no_tables = 0, do_digest = 0; const char *fa = 0, *list_fn = 0, *hg = 0, *opt_N = 0, *rl_cut = 0, *sc_cut = 0, *ed_fn = 0, *edg_fn = 0, *eds_fn = 0, *ed1_fn = 0, *ed2_fn = 0, *load_pfx = 0, *save_pfx = 0; std::string prefix;
	double bw_arg = -1;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--ont")) is_ont = 1;
		else if (!strcmp(argv[i], "-t")) n_thread = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-k")) k = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-w")) w = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-f")) bf_shift = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--dump")) prefix = argv[++i];
		else if (!strcmp(argv[i], "--time")) do_time = 1;
		else if (!strcmp(argv[i], "--nodump-hits")) dump_hits = 0;
		else if (!strcmp(argv[i], "--no-hpc")) no_hpc = 1;
		else if (!strcmp(argv[i], "--hg-size")) hg = argv[++i];
		else if (!strcmp(argv[i], "--bw")) bw_arg = atof(argv[++i]);
		else if (!strcmp(argv[i], "-N")) opt_N = argv[++i];
		else if (!strcmp(argv[i], "--rl-cut")) rl_cut = argv[++i];
		else if (!strcmp(argv[i], "--sc-cut")) sc_cut = argv[++i];
		else if (!strcmp(argv[i], "--reads-list")) list_fn = argv[++i];
		else if (!strcmp(argv[i], "--no-tables")) no_tables = 1;
		else if (!strcmp(argv[i], "--ft-tables")) ft_tables = 1;      // with --no-tables: still dump the all-k-mer histogram and the filter table (not the position index)
		else if (!strcmp(argv[i], "--digest")) do_digest = 1;
		else if (!strcmp(argv[i], "--ed-tasks")) ed_fn = argv[++i];
		else if (!strcmp(argv[i], "--edg-tasks")) edg_fn = argv[++i];
		else if (!strcmp(argv[i], "--eds-tasks")) eds_fn = argv[++i];
		else if (!strcmp(argv[i], "--ed1-tasks")) ed1_fn = argv[++i];
		else if (!strcmp(argv[i], "--ed2-tasks")) ed2_fn = argv[++i];
		else if (!strcmp(argv[i], "--load-index")) load_pfx = argv[++i];
		else if (!strcmp(argv[i], "--save-index")) save_pfx = argv[++i];      // write_pt_index (htab.cpp:1367) after ha_pt_gen: <prefix>.pt_flt, .pt_flt.bin, .pt_flt.paf.bin
		else fa = argv[i];
	}
	if (!fa) { fprintf(stderr, "usage: ref_harness [--ont] [-t N] [-k K] [-w W] [-f BLOOM_BITS] [--dump PREFIX] [--time] reads.fa\n"); return 1; }
	// the reference's own option parser; -f0 (exact counting) unless -f is given (hifiasm's own default is -f37: 16 GB of Bloom filter)
	std::vector<std::string> av; char tb[32], kb[32], wb[32], fb[32];
	av.push_back("hifiasm"); av.push_back("-o"); av.push_back(prefix.empty()? std::string("/tmp/ref_harness_out") : prefix);
	snprintf(tb, 32, "%d", n_thread); av.push_back("-t"); av.push_back(tb); snprintf(fb, 32, "-f%d", bf_shift); av.push_back(fb);
	if (k > 0) { snprintf(kb, 32, "%d", k); av.push_back("-k"); av.push_back(kb); }
	if (w > 0) { snprintf(wb, 32, "%d", w); av.push_back("-w"); av.push_back(wb); }
	if (is_ont) av.push_back("--ont");
	if (hg) { av.push_back("--hg-size"); av.push_back(hg); }
	if (opt_N) { av.push_back("-N"); av.push_back(opt_N); }
	if (rl_cut) { av.push_back("--rl-cut"); av.push_back(rl_cut); }
	if (sc_cut) { av.push_back("--sc-cut"); av.push_back(sc_cut); }
	av.push_back(fa);
	std::vector<char*> avp; for (size_t i = 0; i < av.size(); ++i) avp.push_back((char*)av[i].c_str());
	yak_reset_realtime();
	init_opt(&asm_opt);
	if (!CommandLine_process((int)avp.size(), &avp[0], &asm_opt)) return 1;
	if (no_hpc) asm_opt.flag |= HA_F_NO_HPC;      // what the reference's own --no-hpc style switches set (CommandLines.h); every hot-path call reads this flag

	int hom_cov_ft = -1, hom_cov = -1, het_cov = -1;
	double t0 = yak_realtime();
	double t_ft = 0, t_pt = 0;
	if (load_pfx) {
		if (!load_pt_index(&ha_flt_tab, &ha_idx, &R_INF, &asm_opt, (char*)load_pfx)) { fprintf(stderr, "load_pt_index(%s) failed\n", load_pfx); return 1; }
		hom_cov = asm_opt.hom_cov; het_cov = asm_opt.het_cov; hom_cov_ft = -1; no_tables_hist = 1;
	} else {
		ha_flt_tab = ha_ft_gen(&asm_opt, &R_INF, &hom_cov_ft, 0, 0);
		ha_opt_update_cov(&asm_opt, hom_cov_ft);
		t_ft = yak_realtime() - t0; t0 = yak_realtime();
		ha_idx = ha_pt_gen(&asm_opt, ha_flt_tab, 0, 0, &R_INF, &hom_cov, &het_cov);
		asm_opt.hom_cov = hom_cov; asm_opt.het_cov = het_cov;
		t_pt = yak_realtime() - t0;
		if (save_pfx && !write_pt_index(ha_flt_tab, ha_idx, &R_INF, &asm_opt, (char*)save_pfx)) { fprintf(stderr, "write_pt_index(%s) failed\n", save_pfx); return 1; }
	}

	uint64_t n_reads = R_INF.total_reads;
	uint32_t high_occ = asm_opt.hom_cov * (2.0 - HA_KMER_GOOD_RATIO);   // ecovlp.cpp:3237
	uint32_t low_occ = asm_opt.hom_cov * HA_KMER_GOOD_RATIO;            // ecovlp.cpp:3238
	double bw = is_ont ? 0.05 : 0.02;                                     // ecovlp.cpp:3274
	if (bw_arg >= 0) bw = bw_arg;

	if (do_time) {
		tbuf_t *b = tbuf_init(n_thread); pass_t p; p.b = b; p.bw = bw; p.high_occ = high_occ; p.low_occ = low_occ;
		t0 = yak_realtime();
		kt_for(n_thread, worker_pass, &p, n_reads);
		double t_pass = yak_realtime() - t0; uint64_t n_ovlp = 0, n_hits = 0;
		for (int i = 0; i < n_thread; ++i) n_ovlp += b[i].n_ovlp, n_hits += b[i].n_hits;
		printf("{\"n_reads\": %lu, \"threads\": %d, \"t_ft_gen\": %.4f, \"t_pt_gen\": %.4f, \"t_pass\": %.4f, \"overlaps\": %lu, \"chained_hits\": %lu, \"overlaps_per_sec\": %.1f, \"hom_cov\": %d, \"het_cov\": %d, \"max_n_chain\": %d}\n",
			   (unsigned long)n_reads, n_thread, t_ft, t_pt, t_pass, (unsigned long)n_ovlp, (unsigned long)n_hits, n_ovlp / (t_pt + t_pass), hom_cov, het_cov, asm_opt.max_n_chain);
		fflush(stdout);
	}
	if (prefix.empty()) return 0;
	if (do_digest) {
		std::vector<uint64_t> dig(2 * n_reads, 0);
		tbuf_t *b = tbuf_init(n_thread); pass_t p; p.b = b; p.bw = bw; p.high_occ = high_occ; p.low_occ = low_occ; p.dig = dig.data();
		kt_for(n_thread, worker_digest, &p, n_reads);
		wr(prefix, "dig.u64", dig.data(), 8 * dig.size());
	}
	if (ed_fn) {
		FILE *fp = fopen(ed_fn, "rb"); if (!fp) { fprintf(stderr, "cannot read %s\n", ed_fn); return 1; }
		std::vector<uint32_t> tk; uint32_t rec[10];
		while (fread(rec, 4, 10, fp) == 10) tk.insert(tk.end(), rec, rec + 10);
		fclose(fp);
		std::vector<int32_t> res; std::vector<char> ps, ts; bit_extz_t ez; init_bit_extz_t(&ez, 63);
		for (size_t i = 0; i + 10 <= tk.size(); i += 10) {
			const uint32_t *t = &tk[i];
			ps.resize(t[2] + 8); ts.resize(t[6] + 8);
			recover_UC_Read_sub_region(ps.data(), t[1], t[2], (uint8_t)t[3], &R_INF, t[0]);
			recover_UC_Read_sub_region(ts.data(), t[5], t[6], (uint8_t)t[7], &R_INF, t[4]);
			if (2 * t[8] + 1 <= 64) ed_band_cal_semi_64_w_absent_diag(ps.data(), (int32_t)t[2], ts.data(), (int32_t)t[6], (int32_t)t[8], (int32_t)t[9], &ez);
			else if (2 * t[8] + 1 <= 128) ed_band_cal_semi_128_w_absent_diag(ps.data(), (int32_t)t[2], ts.data(), (int32_t)t[6], (int32_t)t[8], (int32_t)t[9], &ez);      // (bands of 65 .. 127 diagonals: HA_ED_INIT(128))
			else { int32_t nw_ = (int32_t)((2 * t[8] + 1 + 63) >> 6); ed_band_cal_semi_infi_w_absent_diag(ps.data(), (int32_t)t[2], ts.data(), (int32_t)t[6], (int32_t)t[8], (int32_t)t[9], &nw_, &ez); }      // (wider bands: nword as cal_exz_infi computes it, Correct.cpp:14511)
			res.push_back(ez.err); res.push_back(ez.pe);
		}
		wr(prefix, "ed.i32", res.data(), 4 * res.size());
		if (do_time) {      // the same calls on n_thread threads (own strings and bit_extz_t each; a task = recover both strings + the function, as Correct.cpp:3897 does per candidate)
			const size_t nt = tk.size() / 10; const int T = n_thread > 0 ? n_thread : 1;
			const double t_ed0 = yak_realtime();
			std::vector<std::thread> th; std::vector<int64_t> sink(T, 0);
			for (int w_ = 0; w_ < T; ++w_) th.emplace_back([&, w_]() {
				std::vector<char> ps_, ts_; bit_extz_t ez_; init_bit_extz_t(&ez_, 63); int64_t acc = 0;
				for (size_t q = nt * w_ / T; q < nt * (w_ + 1) / T; ++q) {
					const uint32_t *t = &tk[10 * q];
					ps_.resize(t[2] + 8); ts_.resize(t[6] + 8);
					recover_UC_Read_sub_region(ps_.data(), t[1], t[2], (uint8_t)t[3], &R_INF, t[0]);
					recover_UC_Read_sub_region(ts_.data(), t[5], t[6], (uint8_t)t[7], &R_INF, t[4]);
					if (2 * t[8] + 1 <= 64) ed_band_cal_semi_64_w_absent_diag(ps_.data(), (int32_t)t[2], ts_.data(), (int32_t)t[6], (int32_t)t[8], (int32_t)t[9], &ez_);
					else ed_band_cal_semi_128_w_absent_diag(ps_.data(), (int32_t)t[2], ts_.data(), (int32_t)t[6], (int32_t)t[8], (int32_t)t[9], &ez_);
					acc += ez_.err;
				}
				sink[w_] = acc;
			});
			for (auto &x : th) x.join();
			const double dt_ed = yak_realtime() - t_ed0;
			printf("{\"ed_pairs\": %zu, \"ed_threads\": %d, \"ed_seconds\": %.4f, \"ed_pairs_per_sec\": %.1f, \"ed_checksum\": %lld}\n", nt, T, dt_ed, nt / (dt_ed > 0 ? dt_ed : 1e-9), (long long)sink[0]);
		}
	}
	for (int tm = 0; tm < 4; ++tm) {      // 0 global, 1 semi-global, 2 / 3 forward / backward extension
		const char *tfn = tm == 0 ? edg_fn : tm == 1 ? eds_fn : tm == 2 ? ed1_fn : ed2_fn; if (!tfn) continue;
		const char *tag = tm == 0 ? "edg" : tm == 1 ? "eds" : tm == 2 ? "ed1" : "ed2";
		FILE *fp = fopen(tfn, "rb"); if (!fp) { fprintf(stderr, "cannot read %s\n", tfn); return 1; }
		std::vector<uint32_t> tk; uint32_t rec[10];
		while (fread(rec, 4, 10, fp) == 10) tk.insert(tk.end(), rec, rec + 10);
		fclose(fp);
		std::vector<int32_t> res; std::vector<uint16_t> cg; std::vector<char> ps, ts; bit_extz_t ez; init_bit_extz_t(&ez, 63);
		for (size_t i = 0; i + 10 <= tk.size(); i += 10) {
			const uint32_t *t = &tk[i];
			ps.resize(t[2] + 8); ts.resize(t[6] + 8);
			recover_UC_Read_sub_region(ps.data(), t[1], t[2], (uint8_t)t[3], &R_INF, t[0]);
			recover_UC_Read_sub_region(ts.data(), t[5], t[6], (uint8_t)t[7], &R_INF, t[4]);
			clear_align(ez); ez.pe = ez.te = -1; ez.cigar.n = 0;
			const int32_t pn_ = (int32_t)t[2], tn_ = (int32_t)t[6], th_ = (int32_t)t[8];
			if (2 * th_ + 1 <= 64) {
				if (tm == 1) ed_band_cal_semi_64_w_absent_diag_trace(ps.data(), pn_, ts.data(), tn_, th_, (int32_t)t[9], &ez);
				else if (tm == 2) ed_band_cal_extension_64_0_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
				else if (tm == 3) ed_band_cal_extension_64_1_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
				else ed_band_cal_global_64_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
			} else if (2 * th_ + 1 > 128) {      // more than two words: the *_infi_* functions with nword = ceil((2 thre + 1) / 64), the choice of cal_exz_infi, Correct.cpp:14556-14565
				int32_t nw_ = (2 * th_ + 1 + 63) >> 6;
				if (tm == 1) ed_band_cal_semi_infi_w_absent_diag_trace(ps.data(), pn_, ts.data(), tn_, th_, (int32_t)t[9], &nw_, &ez);
				else if (tm == 2) ed_band_cal_extension_infi_0_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &nw_, &ez);
				else if (tm == 3) ed_band_cal_extension_infi_1_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &nw_, &ez);
				else ed_band_cal_global_infi_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &nw_, &ez);
			} else {      // two words (HA_ED_INIT(128), Levenshtein_distance.h:2129): the choice of cal_exz_global, Correct.cpp:15482-15494
				if (tm == 1) ed_band_cal_semi_128_w_absent_diag_trace(ps.data(), pn_, ts.data(), tn_, th_, (int32_t)t[9], &ez);
				else if (tm == 2) ed_band_cal_extension_128_0_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
				else if (tm == 3) ed_band_cal_extension_128_1_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
				else ed_band_cal_global_128_w_trace(ps.data(), pn_, ts.data(), tn_, th_, &ez);
			}
			const bool ok = is_align(ez);
			if (tm >= 2) { res.push_back(ok ? ez.err : INT32_MAX); res.push_back(ez.ps); res.push_back(ez.pe); res.push_back(ez.ts); res.push_back(ez.te); }
			else { res.push_back(ok ? ez.err : INT32_MAX); res.push_back(ok ? ez.ps : (tm ? -1 : 0)); res.push_back(ok ? ez.pe : -1); res.push_back(ok ? ez.ts : 0); res.push_back(ok || tm ? (int32_t)t[6] - 1 : -1); }
			res.push_back(ok ? (int32_t)ez.cigar.n : 0);
			if (ok) cg.insert(cg.end(), ez.cigar.a, ez.cigar.a + ez.cigar.n);
		}
		wr(prefix, (std::string(tag) + ".i32").c_str(), res.data(), 4 * res.size());
		cg.push_back(0);
		wr(prefix, (std::string(tag) + "_cig.u16").c_str(), cg.data(), 2 * (cg.size() - 1));
	}
	std::vector<uint64_t> sel;      // reads of the per-read dumps
	if (list_fn) {
		FILE *fp = fopen(list_fn, "r"); unsigned long v;
		if (!fp) { fprintf(stderr, "cannot read %s\n", list_fn); return 1; }
		while (fscanf(fp, "%lu", &v) == 1) if (v < n_reads) sel.push_back(v);
		fclose(fp);
	} else for (uint64_t i = 0; i < n_reads; ++i) sel.push_back(i);
	const uint64_t n_sel = sel.size();

	// ---------------- dumps ----------------
	{ // read lengths
		wr(prefix, "rlen.u64", R_INF.read_length, sizeof(uint64_t) * n_reads);
	}
	int64_t ft_hist[4096], pt_hist[4096]; int ft_peak_hom, ft_peak_het; uint64_t ft_distinct, pt_distinct;
	memset(ft_hist, 0, sizeof(ft_hist)); ft_peak_hom = hom_cov_ft; ft_peak_het = -1; ft_distinct = 0;
	if ((!no_tables || ft_tables) && !no_tables_hist) refdump_ft_hist(&asm_opt, &R_INF, ft_hist, &ft_peak_hom, &ft_peak_het, &ft_distinct);      // (recounts every k-mer: skipped for large sets)
	memset(pt_hist, 0, sizeof(pt_hist)); pt_distinct = 0;
	if (!no_tables_hist) refdump_pt_hist(&asm_opt, ha_flt_tab, &R_INF, pt_hist, &pt_distinct);
	wr(prefix, "ft_hist.i64", ft_hist, sizeof(ft_hist));
	wr(prefix, "pt_hist.i64", pt_hist, sizeof(pt_hist));
	uint64_t n_ft = 0, n_ptk = 0, n_ptp = 0;
	if (!no_tables || ft_tables) {
		uint64_t *keys; int32_t *vals;
		n_ft = refdump_ft(ha_flt_tab, &keys, &vals);
		wr(prefix, "ft_keys.u64", keys, sizeof(uint64_t) * n_ft);
		wr(prefix, "ft_vals.i32", vals, sizeof(int32_t) * n_ft);
		free(keys); free(vals);
	}
	if (!no_tables) {
		uint64_t *keys, *off, *pos;
		n_ptk = refdump_pt(ha_idx, &keys, &off, &pos, &n_ptp);
		wr(prefix, "pt_keys.u64", keys, sizeof(uint64_t) * n_ptk);
		wr(prefix, "pt_off.u64", off, sizeof(uint64_t) * (n_ptk + 1));
		wr(prefix, "pt_pos.u64", pos, sizeof(uint64_t) * n_ptp);
		free(keys); free(off); free(pos);
	}
	// per-read minimizers, index-time call (htab.cpp:691): rid = read id, hf = ha_flt_tab
	{
		UC_Read ur; init_UC_Read(&ur); ha_mz1_v mz = {0,0,0}; st_mt_t mt = {0,0,0};
		std::vector<uint64_t> off(n_sel + 1, 0), rec;
		for (uint64_t ii = 0; ii < n_sel; ++ii) {
			const uint64_t i = sel[ii];
			recover_UC_Read(&ur, &R_INF, i); mz.n = 0;
			mz1_ha_sketch(ur.seq, ur.length, asm_opt.mz_win, asm_opt.k_mer_length, i, !(asm_opt.flag & HA_F_NO_HPC), &mz, ha_flt_tab, asm_opt.mz_sample_dist, 0, 0, NULL, -1, asm_opt.dp_min_len, asm_opt.dp_e, &mt, asm_opt.mz_rewin, 0, NULL);
			for (uint32_t j = 0; j < mz.n; ++j) { uint64_t q[2]; memcpy(q, &mz.a[j], 16); rec.push_back(q[0]); rec.push_back(q[1]); }
			off[ii + 1] = rec.size() / 2;
		}
		wr(prefix, "mz_off.u64", &off[0], sizeof(uint64_t) * off.size());
		wr(prefix, "mz.u64", rec.data(), sizeof(uint64_t) * rec.size());
		// same with hf = NULL and sample_dist = 0 (pure window minimizers, no count order / thinning)
		rec.clear();
		for (uint64_t ii = 0; ii < n_sel && !no_tables; ++ii) {
			const uint64_t i = sel[ii];
			recover_UC_Read(&ur, &R_INF, i); mz.n = 0;
			mz1_ha_sketch(ur.seq, ur.length, asm_opt.mz_win, asm_opt.k_mer_length, i, !(asm_opt.flag & HA_F_NO_HPC), &mz, NULL, 0, 0, 0, NULL, -1, asm_opt.dp_min_len, asm_opt.dp_e, &mt, asm_opt.mz_rewin, 0, NULL);
			for (uint32_t j = 0; j < mz.n; ++j) { uint64_t q[2]; memcpy(q, &mz.a[j], 16); rec.push_back(q[0]); rec.push_back(q[1]); }
			off[ii + 1] = rec.size() / 2;
		}
		wr(prefix, "mz0_off.u64", &off[0], sizeof(uint64_t) * off.size());
		wr(prefix, "mz0.u64", rec.data(), sizeof(uint64_t) * rec.size());
		free(mz.a); free(mt.a); destory_UC_Read(&ur);
	}
	uint64_t tot_ol = 0, tot_cl = 0, tot_kh = 0;
	{ // per-read seed hits before chaining, and (ol, cl) after h_ec_lchain
		tbuf_t *b = tbuf_init(1);
		std::vector<uint64_t> kh_off(n_sel + 1, 0), ol_off(n_sel + 1, 0), cl_off(n_sel + 1, 0), fc_off(1, 0), fc;
		std::vector<uint32_t> kh, ol, cl; std::vector<uint8_t> ex; UC_Read tu; init_UC_Read(&tu);
		for (uint64_t ii = 0; ii < n_sel; ++ii) {
			const uint64_t i = sel[ii];
			uint32_t ho = high_occ, lo = low_occ;
			recover_UC_Read(&b->ur, &R_INF, i);
			if (dump_hits) {
				minimizers_qgen0(b->ab, b->ur.seq, b->ur.length, asm_opt.mz_win, asm_opt.k_mer_length, &b->cl, NULL, ha_flt_tab, ha_idx, &R_INF, NULL, &b->sp, &ho, &lo);
				for (uint64_t j = 0; j < (uint64_t)b->cl.length; ++j) { uint32_t q[4]; memcpy(q, &b->cl.list[j], 16); kh.insert(kh.end(), q, q + 4); }
			}
			kh_off[ii + 1] = kh.size() / 4;
			ho = high_occ; lo = low_occ;
			h_ec_lchain(b->ab, i, b->ur.seq, b->ur.length, asm_opt.mz_win, asm_opt.k_mer_length, &R_INF, &b->ol, &b->cl, bw, asm_opt.max_n_chain, 1, NULL, NULL, &b->sp, &ho, &lo, 1, 1, 3, 0.7, 2, 32, COV_W);
			for (uint64_t j = 0; j < b->ol.length; ++j) {
				overlap_region *r = &b->ol.list[j];
				uint32_t q[12] = { r->x_id, r->x_pos_s, r->x_pos_e, r->x_pos_strand, r->y_id, r->y_pos_s, r->y_pos_e, r->y_pos_strand,
								   (uint32_t)r->shared_seed, r->align_length, r->non_homopolymer_errors, r->f_cigar.length };
				ol.insert(ol.end(), q, q + 12);
				{	// the final round's exact-overlap check of this candidate: exact_ec_check (ecovlp.cpp:2803-2808, a file-local inline: equal lengths + memcmp)
					// on the strings h_ec_lchain_fast_new builds (ecovlp.cpp:5124-5131): the whole query read, the target interval on its strand
					int64_t bq0 = r->x_pos_s, bq1 = (int64_t)r->x_pos_e + 1, bt0 = r->y_pos_s, bt1 = (int64_t)r->y_pos_e + 1; uint8_t e = 0;
					if (bq1 - bq0 == bt1 - bt0) {
						if (bt1 - bt0 + 8 > (int64_t)tu.size) { tu.size = bt1 - bt0 + 8; tu.seq = (char*)realloc(tu.seq, tu.size); }      // (resize_UC_Read, Correct.h:1341)
						recover_UC_Read_sub_region(tu.seq, bt0, bt1 - bt0, r->y_pos_strand, &R_INF, r->y_id);
						e = memcmp(b->ur.seq + bq0, tu.seq, bq1 - bq0) == 0;
					}
					ex.push_back(e);
				}
				for (uint32_t c = 0; c < r->f_cigar.length; ++c) fc.push_back(r->f_cigar.buffer[c]);
				fc_off.push_back(fc.size());
			}
			ol_off[ii + 1] = ol.size() / 12;
			if (dump_hits)
				for (uint64_t j = 0; j < (uint64_t)b->cl.length; ++j) { uint32_t q[4]; memcpy(q, &b->cl.list[j], 16); cl.insert(cl.end(), q, q + 4); }
			cl_off[ii + 1] = cl.size() / 4;
			tot_cl += b->cl.length;
		}
		tot_ol = ol.size() / 12; tot_kh = kh.size() / 4;
		wr(prefix, "sel.u64", sel.data(), 8 * sel.size());
		wr(prefix, "kh_off.u64", &kh_off[0], 8 * kh_off.size()); wr(prefix, "kh.u32", kh.data(), 4 * kh.size());
		wr(prefix, "ol_off.u64", &ol_off[0], 8 * ol_off.size()); wr(prefix, "ol.u32", ol.data(), 4 * ol.size()); wr(prefix, "ex.u8", ex.data(), ex.size());
		wr(prefix, "fc_off.u64", &fc_off[0], 8 * fc_off.size()); wr(prefix, "fc.u64", fc.data(), 8 * fc.size());
		wr(prefix, "cl_off.u64", &cl_off[0], 8 * cl_off.size()); wr(prefix, "cl.u32", cl.data(), 4 * cl.size());
	}
	int64_t meta[24]; memset(meta, 0, sizeof(meta));
	meta[0] = n_reads; meta[1] = asm_opt.k_mer_length; meta[2] = asm_opt.mz_win; meta[3] = hom_cov_ft;
	meta[4] = ft_peak_hom; meta[5] = ft_peak_het; meta[6] = asm_opt.max_n_chain; meta[7] = hom_cov; meta[8] = het_cov;
	meta[9] = high_occ; meta[10] = low_occ; meta[11] = n_ft; meta[12] = n_ptk; meta[13] = n_ptp; meta[14] = tot_ol;
	meta[15] = tot_cl; meta[16] = tot_kh; meta[17] = ft_distinct; meta[18] = pt_distinct; meta[19] = is_ont;
	meta[20] = asm_opt.mz_sample_dist; meta[21] = asm_opt.mz_rewin; meta[22] = asm_opt.max_kmer_cnt;
	meta[23] = (int64_t)(asm_opt.high_factor * 1000);
	wr(prefix, "meta.i64", meta, sizeof(meta));
	fprintf(stderr, "[ref_harness] reads=%lu ft=%lu pt_keys=%lu pt_pos=%lu overlaps=%lu chained_hits=%lu seed_hits=%lu hom_cov=%d het_cov=%d max_n_chain=%d\n",
			(unsigned long)n_reads, (unsigned long)n_ft, (unsigned long)n_ptk, (unsigned long)n_ptp, (unsigned long)tot_ol, (unsigned long)tot_cl, (unsigned long)tot_kh, hom_cov, het_cov, asm_opt.max_n_chain);
	return 0;
}
