/* TEST ORACLE - NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the hifiasm (0.25.0-r726) candidate-
 * overlap hot path: HPC k-mer hashing -> exact k-mer counts / histogram / peaks ->
 * high-count filter table -> (count,hash) minimizer sketch -> count+position index
 * -> seed hits -> per-target linear chaining -> per-read chain selection.
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (hifiasm_amd/, include/hao.h) never
 * links, imports or executes anything under oracle/.
 *
 * Parity pinning: this restatement is checked against the real reference
 * (oracle/_ref/ref_harness = the unmodified reference sources compiled where they
 * lie) in this container, and against golden dumps of that harness committed under
 * tests/golden/ (see tests/golden/make_golden.py).
 */
#ifndef HAO_ORACLE_H
#define HAO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ha_mz1_t (htab.h:13-18): info = rid:28 | pos:27 | rev:1 | span:8, LSB first */
typedef struct { uint64_t x, info; } hao_or_mz_t;
/* k_mer_hit (Hash_Table.h:116-120): w0 = readID:31 | strand:1 */
typedef struct { uint32_t w0, offset, self_offset, cnt; } hao_or_hit_t;
/* the overlap_region fields h_ec_lchain defines (Hash_Table.h:78-106), in the
 * order ref_harness dumps them */
typedef struct {
	uint32_t x_id, x_pos_s, x_pos_e, x_pos_strand, y_id, y_pos_s, y_pos_e, y_pos_strand;
	uint32_t shared_seed, align_length, non_homopolymer_errors, fc_len;
} hao_or_ovlp_t;

typedef struct {
	int k, w, hpc;            /* -k, -w (CommandLines.cpp:259,263), !HA_F_NO_HPC */
	int sample_dist, rewin;   /* mz_sample_dist=500, mz_rewin=1000 (CommandLines.cpp:266,268) */
	int min_hist_cnt;         /* min_hist_kmer_cnt=5 */
	int max_kmer_cnt;         /* 2000 (CommandLines.cpp:270) */
	double high_factor;       /* 5.0 (CommandLines.cpp:271) */
	int max_n_chain;          /* 100 (CommandLines.cpp:276); raised by ha_opt_update_cov */
	int is_ont;               /* bw_thres 0.05 instead of 0.02 (ecovlp.cpp:3274) */
	int bf_shift;             /* -f: log2 of the Bloom filter bits in front of the k-mer count table (CommandLines.cpp:269 default 37); 0 = exact counting */
	double bw_thres;          /* bw_thres of h_ec_lchain; 0 = the EC-round value 0.02 / 0.05 --ont (ecovlp.cpp:3274); the final round passes 0.001 (:3957) */
	long long hg_size;        /* --hg-size (CommandLines.cpp:331,959), -1 = unset: prior for the peak finder (htab.cpp:1156,1254) */
} hao_or_opt_t;

typedef struct hao_or_ctx hao_or_ctx;

void hao_or_opt_default(hao_or_opt_t *o);
uint64_t hao_or_hash64(uint64_t key);

/* codes: concatenated base codes 0..3 (>=4 = N), off[n_reads+1] */
hao_or_ctx *hao_or_create(const uint8_t *codes, const uint64_t *off, uint64_t n_reads, const hao_or_opt_t *opt);
void hao_or_destroy(hao_or_ctx *c);

/* ha_ft_gen + ha_opt_update_cov (htab.cpp:1136-1169, CommandLines.cpp:411-418); returns peak_hom */
int hao_or_ft_gen(hao_or_ctx *c);
/* ha_pt_gen (htab.cpp:1232-1287) + the asm_opt.hom_cov/het_cov update of Assembly.cpp:1007-1008; returns peak_hom */
int hao_or_pt_gen(hao_or_ctx *c);

/* accessors (pointers stay owned by the ctx) */
const int64_t *hao_or_ft_hist(const hao_or_ctx *c);
const int64_t *hao_or_pt_hist(const hao_or_ctx *c);
uint64_t hao_or_ft_table(const hao_or_ctx *c, const uint64_t **keys, const int32_t **vals);
uint64_t hao_or_pt_table(const hao_or_ctx *c, const uint64_t **keys, const uint64_t **off, const uint64_t **pos, uint64_t *n_pos);
/* out[0]=ft peak_hom out[1]=ft peak_het out[2]=ft cutoff out[3]=max_n_chain out[4]=hom_cov out[5]=het_cov out[6]=high_occ out[7]=low_occ */
void hao_or_stats(const hao_or_ctx *c, int64_t out[8]);
int32_t hao_or_ft_cnt(const hao_or_ctx *c, uint64_t y);

/* all-k-mer hashes of one read (htab.cpp:608-645); out must hold len entries; returns count */
int64_t hao_or_kmer_hashes(const uint8_t *codes, int64_t len, int k, int hpc, uint64_t *out);

/* mz1_ha_sketch (sketch.cpp:454-579). use_ft=0 -> hf=NULL. Returns n; *out is a ctx-owned
 * scratch buffer valid until the next call. */
int64_t hao_or_sketch(hao_or_ctx *c, uint64_t rid, int use_ft, int sample_dist, const hao_or_mz_t **out);
/* same on caller-provided codes (kernel-level tests) */
/* test support: as hao_or_sketch(use_ft = 1), plus the candidate list mz1_select_mz_h (sketch.cpp:247) received: hash, filter-table count, position, k-mer ordinal; tot_l */
int64_t hao_or_sketch_pre(hao_or_ctx *c, uint64_t rid, int sample_dist, const hao_or_mz_t **out, int64_t *pre_n, const uint64_t **x, const uint32_t **cnt, const uint32_t **pos,
		const uint64_t **ord, int64_t *tot_l);
int64_t hao_or_sketch_seq(hao_or_ctx *c, const uint8_t *codes, int64_t len, uint32_t rid, int use_ft, int sample_dist, const hao_or_mz_t **out);

/* minimizers_qgen0 (anchor.cpp:987-1081): sorted seed hits of one read, before chaining */
int64_t hao_or_seed_hits(hao_or_ctx *c, uint64_t rid, const hao_or_hit_t **out);

/* h_ec_lchain (anchor.cpp:2302-2315) with the ecovlp.cpp:3274 arguments.
 * Returns ol length; outputs are ctx-owned scratch valid until the next call. */
int64_t hao_or_lchain(hao_or_ctx *c, uint64_t rid, const hao_or_ovlp_t **ol, const uint64_t **fc, const uint64_t **fc_off,
					  const hao_or_hit_t **cl, int64_t *cl_n);

/* exact_ec_check (ecovlp.cpp:2803-2808) as h_ec_lchain_fast_new applies it to every overlap h_ec_lchain returned (ecovlp.cpp:5103-5131): out[j] = 1 iff
 * the query interval [x_pos_s, x_pos_e] equals, character by character (N only equals N), the target interval [y_pos_s, y_pos_e] taken on strand
 * y_pos_strand (recover_UC_Read_sub_region, Process_Read.cpp:524-614).  ol = n overlaps as hao_or_lchain returned them; out holds n bytes. */
void hao_or_exact(const hao_or_ctx *c, const hao_or_ovlp_t *ol, int64_t n, uint8_t *out);

/* ed_band_cal_semi_64_w_absent_diag (Levenshtein_distance.h:3727-3776; ed_core_64 :3116-3125): banded Myers bit-vector edit distance of the text
 * (read t_rid, [t_pos, t_pos + t_len) on strand t_rev) against the pattern (read p_rid, [p_pos, p_pos + p_len) on strand p_rev), threshold thre,
 * abs_diag leading diagonals absent.  task = 10 uint32 in that order (+ thre, abs_diag); out[0] = err (INT32_MAX: none), out[1] = pe (-1: none). */
void hao_or_window_ed(const hao_or_ctx *c, const uint32_t *task, int64_t n, int32_t *out);
/* f3, global alignment with traceback (ed_band_cal_global_64_w_trace + gen_trace, Levenshtein_distance.h:3370,903): out[n][6] = err, ps, pe, ts, te, cigar
 * entries; cigar q at cig + q * cap */
void hao_or_window_trace(const hao_or_ctx *c, const uint32_t *task, int64_t n, int mode /* 0 global, 1 / 2 forward / backward extension (:3512, :3620), 3 semi-global with absent diagonals (:3778) */, int32_t *out, uint16_t *cig, int64_t cap);

/* ha_analyze_count (hist.cpp:74-157) with m_peak_hom <= 0 (hg_size unset) / with the prior m_peak_hom (adj_m_peak_hom, hist.cpp:46-72) */
int hao_or_analyze_count(int n_cnt, int start_cnt, const int64_t *cnt, int *peak_het);
int hao_or_analyze_count_m(int n_cnt, int start_cnt, int m_peak_hom, const int64_t *cnt, int *peak_het);

#ifdef __cplusplus
}
#endif
#endif
