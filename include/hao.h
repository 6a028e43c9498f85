/* hao.h - C ABI of the MI355X-native all-vs-all overlap engine ("hao" = hifiasm-amd
 * overlap).  Plain pointers and sizes only; every entry point names the reference
 * interface (chhylp123/hifiasm 0.25.0-r726, file:line) it stands in for.
 *
 * The reference has no plugin/FFI layer: its seam is three externally linked C++
 * functions reached from per-read worker threads,
 *     ha_ft_gen   (htab.h:79,  htab.cpp:1136)   k-mer count -> high-count filter table
 *     ha_pt_gen   (htab.h:86,  htab.cpp:1232)   minimizer count + position index
 *     h_ec_lchain (anchor.cpp:2302, declared ad hoc at ecovlp.cpp:110)
 *                                               per-read seeds -> chains -> overlap list
 * plus the accessors ha_ft_cnt (htab.h:80), ha_pt_get / ha_pt_cnt (htab.h:88-90) and the
 * finer per-read seam mz1_ha_sketch (htab.h:122).  A GPU wants batches, so this ABI is
 * "precompute, then serve": hao_overlap_batch() runs the whole path for a range of
 * query reads on the device and hao_fetch_*() serve the per-read results that a
 * drop-in h_ec_lchain shim copies into the caller's overlap_region_alloc /
 * Candidates_list (INTEGRATION.md shows that shim).
 *
 * All functions return 0 on success, a negative HAO_E* code otherwise (the reference
 * itself exits on error; a shim maps non-zero to exit(1)).  There is NO CPU fallback:
 * without a HIP device hao_create fails.
 */
#ifndef HAO_H
#define HAO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAO_OK          0
#define HAO_ENODEV     -1   /* no HIP device / HIP runtime error */
#define HAO_EINVAL     -2   /* bad argument or call order */
#define HAO_ENOMEM     -3
#define HAO_EUNSUPP    -4   /* input outside what the device path implements (fails loudly, never falls back) */

typedef struct hao_ctx hao_ctx;

/* Mirrors the fields of hifiasm_opt_t (CommandLines.h:35-173) the hot path reads. */
typedef struct {
	int32_t k;             /* k_mer_length   (CommandLines.cpp:259) default 51 */
	int32_t w;             /* mz_win         (CommandLines.cpp:263) default 51 */
	int32_t hpc;           /* !(flag & HA_F_NO_HPC), default 1 */
	int32_t sample_dist;   /* mz_sample_dist (CommandLines.cpp:268) default 500 */
	int32_t rewin;         /* mz_rewin       (CommandLines.cpp:266) default 1000 */
	int32_t min_hist_cnt;  /* min_hist_kmer_cnt, default 5 */
	int32_t max_kmer_cnt;  /* (CommandLines.cpp:270) default 2000 */
	int32_t max_n_chain;   /* (CommandLines.cpp:276) default 100; raised like ha_opt_update_cov */
	double  high_factor;   /* (CommandLines.cpp:271) default 5.0 */
	int32_t is_ont;        /* --ont: bw_thres 0.05 instead of 0.02 (ecovlp.cpp:3274) */
	int32_t bf_shift;      /* -f (CommandLines.cpp:269, reference default 37): log2 of the Bloom filter bits in front of the k-mer count table of
	                        * ha_ft_gen (htab.cpp:99-116,140-160,196-206); 0 = exact counting. hao_opt_default sets 0; a shim passes asm_opt.bf_shift */
	int64_t hg_size;       /* --hg-size (CommandLines.cpp:331,959; default -1): when > 0 the peak finder is given the prior
	                        * homozygous coverage total_bases / hg_size (htab.cpp:1156,1254; adj_m_peak_hom, hist.cpp:46-72) */
} hao_opt_t;

/* ha_mz1_t (htab.h:13-18): info = rid:28 | pos:27 | rev:1 | span:8 (LSB first).
 * ha_idxpos_t (htab.h:20-22) has the same bit layout without x. */
typedef struct { uint64_t x, info; } hao_mz_t;
/* k_mer_hit (Hash_Table.h:116-120): w0 = readID:31 | strand:1 */
typedef struct { uint32_t w0, offset, self_offset, cnt; } hao_hit_t;
/* the scalar fields of overlap_region (Hash_Table.h:78-106) that h_ec_lchain defines */
typedef struct {
	uint32_t x_id, x_pos_s, x_pos_e, x_pos_strand;
	uint32_t y_id, y_pos_s, y_pos_e, y_pos_strand;
	int32_t  shared_seed;
	uint32_t align_length;            /* zero on return (anchor.cpp:2098) */
	uint32_t non_homopolymer_errors;  /* index of the chain's first hit in the read's hit list */
	uint32_t fc_len;                  /* Fake_Cigar.length */
} hao_ovlp_t;

void hao_opt_default(hao_opt_t *o);                       /* init_opt, CommandLines.cpp:243-380 */
int  hao_create(int device, const hao_opt_t *opt, hao_ctx **out);
void hao_destroy(hao_ctx *c);
const char *hao_last_error(const hao_ctx *c);

/* Read store hand-over.  Layout = the reference's All_reads (Process_Read.h:115-146):
 * packed = concatenation of read_sperate[i] (len/4+1 bytes per read, 4 bases/byte, first base
 * in bits 7..6; ha_compress_base, Process_Read.cpp:792-850), pk_off[n+1] byte offsets,
 * len[n] = read_length[], nsite_off[n+1]/nsite[] = flattened N_site lists (may be NULL).
 * Host pointers; copied to HBM once. */
int hao_set_reads(hao_ctx *c, const uint8_t *packed, const uint64_t *pk_off, const uint32_t *len, uint64_t n_reads,
				  const uint64_t *nsite_off, const uint32_t *nsite);

/* ---- sharded mode: one process per GPU, reads partitioned by query read ----
 * The reference shards its tables over threads (4096 sub-tables, htab.cpp:147-151,594-606); across GPUs the engine
 * shards READS: rank r holds reads [rid_base, rid_base+n_local) of n_total.  After hao_set_reads (local reads) call
 * hao_set_shard with the lengths of ALL reads (read_length[], replicated: 4 B/read), and hao_dist_init with an id
 * produced once by hao_dist_unique_id and broadcast by the launcher (ncclGetUniqueId / ncclCommInitRank).
 * hao_ft_gen then counts k-mers by hash range (all-to-all-v of 8-byte hashes, 32 KB histogram all-reduce,
 * all-gather-v of the small filter table); hao_pt_gen all-gathers the 16-byte minimizer records so every rank holds
 * the whole index; hao_overlap_batch needs no communication.  Read ids in all results are GLOBAL; the batch range of
 * hao_overlap_batch / hao_fetch_* stays LOCAL (0 .. n_local). */
int hao_set_shard(hao_ctx *c, uint64_t rid_base, uint64_t n_total, const uint32_t *all_len);
int hao_dist_unique_id(uint8_t id[128]);
int hao_dist_init(hao_ctx *c, const uint8_t id[128], int rank, int world);
/* loopback exchange backend: `world` engines in ONE process (one host thread each) on one GPU; test-only substitute for RCCL */
void *hao_loop_create(int world);
void hao_loop_destroy(void *grp);
int hao_dist_init_loopback(hao_ctx *c, void *grp, int rank);

/* ha_ft_gen (htab.cpp:1136-1169), exact (-f0) or through the reference's blocked Bloom filter (opt.bf_shift > 12, bit-exact incl. its false
 * positives: see hao_tables.hpp), + ha_opt_update_cov (CommandLines.cpp:411-418). */
int hao_ft_gen(hao_ctx *c, int32_t *hom_cov);
/* ha_pt_gen (htab.cpp:1232-1287) + the asm_opt.hom_cov/het_cov update of Assembly.cpp:1007-1008.
 * The index and the per-read minimizers stay resident in HBM. */
int hao_pt_gen(hao_ctx *c, int32_t *hom_cov, int32_t *het_cov);

/* host-side views (what non-default CPU consumers of ha_flt_tab / ha_idx need) */
int32_t hao_ft_cnt(hao_ctx *c, uint64_t y);                                   /* ha_ft_cnt htab.cpp:1064 */
int hao_pt_get(hao_ctx *c, uint64_t hash, const uint64_t **pos, int32_t *n);  /* ha_pt_get htab.cpp:518  */
int hao_ft_table(hao_ctx *c, uint64_t *n, const uint64_t **keys, const int32_t **vals);
int hao_pt_table(hao_ctx *c, uint64_t *n_keys, const uint64_t **keys, const uint64_t **off, const uint64_t **pos, uint64_t *n_pos);
int hao_hist(hao_ctx *c, int which /*0 = all k-mers (ft), 1 = minimizers (pt)*/, int64_t cnt[4096]);
/* out[0..7] = ft peak_hom, ft peak_het, ft cutoff, max_n_chain, hom_cov, het_cov, high_occ, low_occ */
int hao_stats(hao_ctx *c, int64_t out[8]);
/* hash-range passes the last hao_ft_gen counted in (1: every k-mer occurrence of the local reads at once, 2 x 8 bytes per base; more when that does not fit the
   free device memory, or as HAO_FT_PASSES says: htab.cpp:707-882 never holds all occurrences either); < 0: error */
int hao_ft_passes(hao_ctx *c);

/* mz1_ha_sketch (sketch.cpp:454-579) for reads [rid_lo, rid_hi): results stay on the device;
 * use_ft = 0 passes hf = NULL; sample_dist <= w disables the high-count thinning (sketch.cpp:575).
 * hao_fetch_sketch copies one read's list (rid field = read id, as at index time htab.cpp:691). */
int hao_sketch_batch(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, int use_ft, int sample_dist);
int hao_fetch_sketch(hao_ctx *c, uint64_t rid, const hao_mz_t **mz, uint64_t *n);

/* The per-pass arguments of h_ec_lchain (anchor.cpp:2302). hao_pass_default fills them as
 * worker_hap_ec does (ecovlp.cpp:3237-3238,3274): bw_thres 0.02 (0.05 --ont), max_n_chain and
 * high/low_occ from the current coverage peaks, mcopy 3 / 0.7 / 32, chain_cutoff 2, ocv_w 3072.
 * The final-round caller (ecovlp.cpp:3957) differs only in bw_thres = 0.001. */
typedef struct {
	double   bw_thres;
	int32_t  max_n_chain;
	uint32_t high_occ, low_occ;
	int32_t  apend_be, is_accurate, gen_off;   /* must be 1, 1, 1 (every hot-path call site) */
	int32_t  mcopy_num;                        /* <= 3 */
	double   mcopy_rate;
	uint32_t chain_cutoff, mcopy_khit_cut;
	uint64_t ocv_w;
} hao_pass_t;
int hao_pass_default(hao_ctx *c, hao_pass_t *p);

/* h_ec_lchain (anchor.cpp:2302-2315) for query reads [rid_lo, rid_hi); hao_overlap_batch uses
 * hao_pass_default. Results stay in HBM and are served per read by hao_fetch_*. */
int hao_overlap_batch(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi);
int hao_overlap_batch_ex(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, const hao_pass_t *pass);
/* seed hits before chaining (cl->list after minimizers_qgen0, anchor.cpp:987-1081) */
int hao_fetch_seed_hits(hao_ctx *c, uint64_t rid, const hao_hit_t **hits, uint64_t *n);
/* ol->list[0..n_ol), their fake cigars (fc_off[n_ol+1] into fc) and cl->list[0..n_cl) */
int hao_fetch_overlaps(hao_ctx *c, uint64_t rid, const hao_ovlp_t **ol, uint64_t *n_ol, const uint64_t **fc, const uint64_t **fc_off,
					   const hao_hit_t **cl, uint64_t *n_cl);
/* totals of the last batch: out[0] = overlaps (sum ol->length), out[1] = chained hits, out[2] = seed hits,
 * out[3] = chain groups, out[4] = minimizers of the query reads */
int hao_batch_totals(hao_ctx *c, uint64_t out[8]);
/* which kernels carried the seed stage (minimizers_qgen0, anchor.cpp:987-1081) of the last batch - a measurement aid, the results do not depend on it:
 * out[0] = first launch: 2 the list-major kernel (hao_query5.cuh), 1 unused (round 5's one-wave merge kernel), 0 the table kernels for every read;
 * out[1] = reads that launch left to the table kernels, out[2] / out[3] = reads whose bins overflowed the 512- / the 1024-slot table */
int hao_batch_seed_path(hao_ctx *c, uint64_t out[4]);

/* A second (third ...) batch context over the same reads and index: own stream, scratch and result buffers, nothing else.  Batches on different
 * contexts are independent, so one host thread per context keeps two batches in flight on the device: the seed stage of one (memory-bound) runs under
 * the chain stage of the other (instruction-bound) - the all-reads pass of configs[2] takes 128 instead of 149 ms with two contexts.  Every batch,
 * fetch, delivery and digest call works on a view; calls that change reads or index (hao_set_reads, hao_ft_gen, hao_pt_gen, ...) belong to the owner
 * and return HAO_EINVAL on a view.  A view follows the owner's rebuilds by itself (it takes the owner's buffers again at its next batch); as with the
 * owner's own batches, no batch may run while the owner rebuilds, and views are destroyed (hao_destroy) before their owner. */
int hao_attach(hao_ctx *owner, hao_ctx **view);

/* ---- streaming result delivery: results of batch i cross PCIe while batch i + 1 computes ----
 * h_ec_lchain hands ol->list and cl->list back to a per-read caller (anchor.cpp:2302; consumed by gen_hc_r_alin_ea, ecovlp.cpp:3288).  A batch's
 * results are ~190 KB per 15 kb read, almost all of it cl->list (16 bytes per chained hit), more than PCIe can carry at the rate the device produces
 * them.  The delivery path therefore (a) ships cl->list in a wire format of ~0.3 bytes per hit: a chained hit is (query minimizer, target offset);
 * self_offset and cnt belong to the query minimizer (anchor.cpp:1065-1076) and travel once per read in its minimizer table; a hit whose
 * chain simply moves on to the read's next minimizer on the same diagonal (> 90 % of them) is a 0 in the batch's bit stream, any other hit a 1
 * plus one code byte - minimizers skipped since the previous hit of the chain (high nibble) and diagonal shift + 8 (low nibble), 0xff = look the
 * hit up in the (sorted) exception list.  Bits, codes and exceptions are addressed by POSITION = index among the batch's sorted seed hits (a chain is
 * a contiguous run of positions, its header says where it starts; the code at a chain's first position is not the chain's and is skipped; positions
 * in no chain cost their bit); the consumer thread decodes straight into its Candidates_list (hao_unpack_hits); (b) copies into one of two pinned host
 * arenas on copy streams, under the next batch's kernels.  hao_overlap_batch_async returns when the batch's kernels are done and its copy is
 * queued; hao_deliver_wait blocks until the copy has landed and describes the arena.  A slot's arena (and the device buffers behind it) is reused by
 * the second-next async batch: at most two batches are in flight, and the caller must be done with batch i before it starts batch i + 2.  The views
 * are read-only and may be read by any number of threads; hao_unpack_hits is a pure function of the view. */
#define HAO_DELIVER_OL 1u      /* ol->list + fake cigars */
#define HAO_DELIVER_CL 2u      /* cl->list (wire format) */
#define HAO_DELIVER_EXACT 4u   /* one byte per overlap: the exact-overlap check of the final round (hao_exact_check) */
/* one overlap of ol->list on the wire, 32 bytes: hao_ovlp_t without what the receiver knows (x_id = the read, x_pos_strand = 0, align_length = 0; y = y_id | y_pos_strand << 31) - hao_unpack_overlaps */
typedef struct { uint32_t y, x_pos_s, x_pos_e, y_pos_s, y_pos_e; int32_t shared_seed; uint32_t non_homopolymer_errors, fc_len; } hao_ovlp_wire_t;
typedef struct { uint32_t n_hits, w0, q0, offset; uint64_t pos; } hao_chain_hdr_t;   /* one chain of cl->list: hit count, the readID word its hits share, first hit: minimizer index in the read, target offset, position; hit i of the chain has position pos + i */
typedef struct { uint32_t self_offset, cnt; } hao_qmz_t;                  /* one query minimizer: k_mer_hit::self_offset and ::cnt of every hit it seeds */
typedef struct { uint64_t index; uint32_t q, pad; hao_hit_t hit; } hao_exc_t;   /* verbatim hit: its position, its minimizer index, the hit (its readID word is the seed stage's: the decoder writes the chain's) */
typedef struct {
	uint64_t rid_lo, n_reads, n_ol, n_fc, n_chains, n_cl, n_exc, n_codes, n_pos, bytes;   /* n_pos = positions of the batch (its seed hits); bytes = what crossed PCIe for this batch */
	const uint64_t *ol_off;          /* [n_reads + 1]: ol->list of read r = ol[ol_off[r] .. ol_off[r + 1]) */
	const hao_ovlp_wire_t *ol;       /* [n_ol]: read them with hao_unpack_overlaps */
	const uint64_t *fc_off;          /* [n_ol + 1]: the fake cigar of overlap j starts at 32-bit word fc_off[j] & ~HAO_FC_RAW of fc[]; read it with hao_unpack_cigar */
	const uint32_t *fc;              /* [n_fc] words: 4 bytes per cigar entry after an overlap's first (site step | zigzag(shift step) << 20); raw overlaps (bit 63 of their offset): 2 words per entry */
	const uint64_t *ch_off, *cl_off, *qm_off; /* [n_reads + 1]: chains / hits / minimizers of read r = chains[ch_off[r] ..), hits cl_off[r] .. of the batch, qmz[qm_off[r] ..) */
	const hao_chain_hdr_t *chains;
	const uint64_t *cl_bits;         /* bit p (word p / 64, bit p % 64) = position p has a code byte */
	const uint32_t *cl_rank;         /* [n_pos / 256 + 1]: code bytes before position 256 i */
	const uint8_t *cl_codes;         /* [n_codes] code bytes, in position order */
	const hao_qmz_t *qmz;            /* minimizer tables of the batch's reads; NULL when they travel packed (qmz_pos / qmz_cnt below) */
	const hao_exc_t *cl_exc;         /* [n_exc] sorted by position */
	const uint8_t *exact;            /* [n_ol] with HAO_DELIVER_EXACT, else NULL */
	double copy_ms;                  /* from "batch computed" to "copy landed" (includes waiting behind the previous batch's copy); filled by hao_deliver_wait */
	const uint16_t *qmz_pos;         /* packed minimizer tables (4 instead of 8 bytes per minimizer): self_offset and cnt of minimizer qm_off[r] + q in two 16-bit arrays.  The engine packs when */
	const uint16_t *qmz_cnt;         /* every read is shorter than 65 536 bases and the pass's seed weights (anchor.cpp:160-173; cnt = weight << 8 | span) are below 256; else both are NULL and qmz is set */
} hao_delivery_t;
#define HAO_FC_RAW (1ULL << 63)
/* The fake cigar (Fake_Cigar, Hash_Table.h:54-59; gen_fake_cigar, Hash_Table.cpp:88-109) of overlap j of a delivered batch as its ol[j].fc_len 8-byte entries
 * (site << 32 | shift code), rebuilt from the packed words: a pure function of the view.  Returns the entry count (nothing is written when it exceeds cap). */
uint32_t hao_unpack_cigar(const hao_delivery_t *d, uint64_t j, uint64_t *out, uint32_t cap);
/* ol->list of read rid (a read of the delivered batch) as hao_ovlp_t records in out[cap]: overlaps ol_off[r] .. ol_off[r + 1) of the batch; returns their number (nothing is
 * written if cap is too small).  A pure function of the view. */
uint64_t hao_unpack_overlaps(const hao_delivery_t *d, uint64_t rid, hao_ovlp_t *out, uint64_t cap);
int hao_overlap_batch_async(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, const hao_pass_t *pass /* NULL: hao_pass_default */, uint32_t parts, int *slot);
/* The delivery slot (0 / 1) the NEXT hao_overlap_batch_async of this context will write: the caller must have stopped reading that arena before it
 * starts the batch (the engine alternates the two slots; asking it keeps that policy out of the caller). */
int hao_next_slot(hao_ctx *c, int *slot);
int hao_deliver_wait(hao_ctx *c, int slot, hao_delivery_t *out);
/* cl->list of read rid (a read of the delivered batch) decoded into out[cap]; returns the number of hits (nothing is written if cap is too small) */
uint64_t hao_unpack_hits(const hao_delivery_t *d, uint64_t rid, hao_hit_t *out, uint64_t cap);

/* Exact-overlap check right after chaining: exact_ec_check (ecovlp.cpp:2803-2808) as the final round applies it to every
 * candidate h_ec_lchain returns (h_ec_lchain_fast_new, ecovlp.cpp:5103-5131; also gen_hc_r_alin_ea's pre-pass :2847-2856): flag = 1 iff the query
 * interval [x_pos_s, x_pos_e] and the target interval [y_pos_s, y_pos_e] on strand y_pos_strand (recover_UC_Read_sub_region,
 * Process_Read.cpp:524-614) have equal length and equal characters, N sites included.  Runs on the packed reads resident in HBM, one wave per
 * overlap, for the final ol->list of the last batch; single-device mode only (a sharded engine holds only its own reads' bases: HAO_EUNSUPP).
 * hao_fetch_exact serves one read's flags (aligned with hao_fetch_overlaps' ol). */
int hao_exact_check(hao_ctx *c);
int hao_fetch_exact(hao_ctx *c, uint64_t rid, const uint8_t **flags, uint64_t *n);

/* Windowed bit-vector edit distance: ed_band_cal_semi_64_w_absent_diag (Levenshtein_distance.h:3727-3776) for a batch of independent
 * (pattern, text) pairs taken from the reads resident in HBM - the call Correct.cpp:3897,4092,4156 makes per 775-base query window and candidate:
 * pattern = the padded target region [p_pos, p_pos + p_len) of read p_rid on strand p_rev, text = the query window [t_pos, t_pos + t_len) of read
 * t_rid on strand t_rev, thre = error threshold, abs_diag = leading diagonals missing because the pattern was clipped at the start of its read.
 * Band width: 2 thre + 1 diagonals in one 64-bit word (thre <= 31), in two (thre 32 .. 63: the reference's ed_band_cal_*_128_* functions, generated
 * from the same text by HA_ED_INIT(128), :1287-2129, and chosen by band width, cal_exz_global Correct.cpp:15482-15494), or in nword = 3 / 4 words
 * (thre 64 .. 95 / 96 .. 127: the reference's ed_band_cal_*_infi_* functions, :2134-3100, which cal_exz_infi calls with nword = ceil((2 thre + 1) / 64),
 * Correct.cpp:14508-14565); thre > HAO_ED_MAX_THRE: HAO_EINVAL.  With a band of two or more words p_len - t_len + abs_diag <= 64 nword is required
 * (beyond it the reference's final scan indexes past its band words).  out[i].err = edit distance or INT32_MAX (no alignment within thre, the reference's clear_align state), out[i].pe = end of the
 * alignment on the pattern or -1; ps = -1, ts = 0, te = t_len - 1 are constants of the call.  One lane per pair; single-device mode (the bases of
 * both reads must be local).  This is the data-parallel core of the window alignment; window placement and retries stay with the caller. */
#define HAO_ED_MAX_THRE 127     /* widest band: 255 diagonals in four 64-bit words */
typedef struct { uint32_t p_rid, p_pos, p_len, p_rev, t_rid, t_pos, t_len, t_rev, thre, abs_diag; } hao_ed_task_t;
typedef struct { int32_t err, pe; } hao_ed_result_t;
int hao_window_ed_batch(hao_ctx *c, const hao_ed_task_t *tasks, uint64_t n_tasks, hao_ed_result_t *out);

/* f3 without a host in the loop (round 5): the window / candidate pairs of the LAST batch (hao_overlap_batch[_ex]; results resident) are generated on the device from the
 * batch's final ol->list on the reference's fixed window grid - windows of `window` query bases (WINDOW = 375, Hash_Table.h:9; Correct.cpp:5645, 5993) starting at
 * multiples of `window`, one pair per overlap and grid window it covers, the window clipped to the overlap at its ends, the pattern = the target interval on the
 * overlap's diagonal padded by thre on both sides and clipped at the read ends with abs_diag = the bases clipped at its start (the operands Correct.cpp:3897 hands to
 * ed_band_cal_semi_64_w_absent_diag, without the fake-cigar shift) - in text order (query read, grid window, position in ol->list), and the distance-only window
 * alignment (hao_window_ed_batch's kernels) runs over them where they lie.  Tasks and results stay in device memory; *n_tasks = their number.
 * hao_fetch_ed_grid copies the first `cap` of them out (either pointer may be NULL).  One threshold per call (thre <= HAO_ED_MAX_THRE); single-device mode. */
int hao_window_ed_grid(hao_ctx *c, uint32_t window, uint32_t thre, uint64_t *n_tasks);
int hao_fetch_ed_grid(hao_ctx *c, hao_ed_task_t *tasks, hao_ed_result_t *res, uint64_t cap);

/* Second variant: global alignment inside the band WITH traceback - ed_band_cal_global_64_w_trace (Levenshtein_distance.h:3370-3442) on a
 * cleared bit_extz_t followed by gen_trace(ez, thre, 1) (:903-985), the call cal_exz_global / Correct.cpp:14537 make once a window's end points are fixed.
 * Same task records (abs_diag is ignored); both strings are consumed entirely, so |p_len - t_len| <= thre or there is no alignment.  Per task: err
 * (INT32_MAX = none within thre; then pe = te = -1 and no cigar), ps = ts = 0, pe = p_len - 1, te = t_len - 1, and the cigar in push_trace's encoding
 * (uint16: op << 14 | len; op 0 match, 1 mismatch, 2 more pattern, 3 more text) at cigars + i * cigar_cap; n_cigar entries exist, those past cigar_cap
 * are not written (an alignment within thre has at most 2 thre + 3 entries for strings shorter than 16 383: 257 for the widest band).  Band widths as for
 * hao_window_ed_batch (thre <= HAO_ED_MAX_THRE). */
typedef struct { int32_t err, ps, pe, ts, te, n_cigar; } hao_trace_result_t;
#define HAO_ALIGN_GLOBAL 0      /* ed_band_cal_global_64_w_trace */
#define HAO_ALIGN_EXT_FWD 1     /* ed_band_cal_extension_64_0_w_trace (:3512-3618): both strings start together, the alignment ends where the pattern or the text runs out (the
                                 * longer one is first cut to the other's length + thre): pe / te come out of the sweep.  Without an alignment: err INT32_MAX, pe = te = -1 */
#define HAO_ALIGN_EXT_BWD 2     /* ed_band_cal_extension_64_1_w_trace (:3620-3735): both strings END together; ps / ts come out (INT32_MAX without an alignment), pe = p_len - 1,
                                 * te = t_len - 1.  In both extension modes an alignment whose sweep was later abandoned (running error > 3 thre) keeps err and coordinates
                                 * but has no cigar (n_cigar = 0), as in the reference */
#define HAO_ALIGN_SEMI 3        /* ed_band_cal_semi_64_w_absent_diag_trace (Levenshtein_distance.h:3778-3848): the traced twin of hao_window_ed_batch - the text is consumed, the
                                 * pattern starts and ends inside the band: ps comes out of the walk, ts = 0, te = t_len - 1 (also without an alignment), abs_diag is used.
                                 * The band must cover the pattern: 0 <= p_len - t_len + abs_diag <= 2 thre and t_len > abs_diag (else HAO_EINVAL: the reference's
                                 * traceback would index its column words out of range).  (Numbering: the modes of Correct.cpp:14536-14545.) */
int hao_window_trace_batch(hao_ctx *c, int mode, const hao_ed_task_t *tasks, uint64_t n_tasks, hao_trace_result_t *out, uint16_t *cigars, uint32_t cigar_cap);

/* On-disk formats: the filter table, the position index and the read store in the reference's own resume format, so a GPU-built
 * index can be handed to a stock hifiasm (load_pt_index, htab.cpp:1432-1550, called from Assembly.cpp:2078):
 *   <prefix>.pt_flt  (write_pt_index, htab.cpp:1367-1430),  <prefix>.pt_flt.bin  (write_All_reads, Process_Read.cpp:69-125 - the layout of *.ec.bin),
 *   <prefix>.pt_flt.paf.bin  (empty overlap lists).
 * names[i] = read names or NULL ("r<i>"); number_of_round must equal the loader's -r (default 3: it exits otherwise, htab.cpp:1501-1505).
 * Needs hao_ft_gen + hao_pt_gen; single-device mode. */
int hao_index_save(hao_ctx *c, const char *prefix, int32_t number_of_round, const char *const *names);
/* The reader of the same files = load_pt_index (htab.cpp:1432-1550) for the engine: an index written by a stock hifiasm (write_pt_index) or by
 * hao_index_save becomes the engine's read store (replaces hao_set_reads), filter table and position index (replace hao_ft_gen / hao_pt_gen), with the
 * file's hom_cov / het_cov / max_n_chain; *number_of_round = the value stored in the file.  The read-ordered minimizers of the query side are
 * sketched here with the loaded filter table (the reference re-sketches every query read).  The file's k must equal the engine's; w, HPC and the
 * other options are the caller's to match, as with the reference.  Histograms are not in the file: hao_hist returns zeros afterwards. */
int hao_index_load(hao_ctx *c, const char *prefix, int32_t *number_of_round);

/* <prefix>.ovlp.source.bin / .ovlp.reverse.bin (write_ma_hit_ts / load_ma_hit_ts, Overlaps.cpp:23328-23469): the per-read lists of ma_hit_t the reference builds from the
 * overlap regions AFTER alignment and correction (not a product of this path: the engine serves h_ec_lchain, the reference's own code fills and writes these lists - the
 * drop-in run's files are byte-identical, tests/test_gpu_dropin.py).  Reader and writer of the format, for tools on either side of the path: a record is the field-by-field
 * image write_ma produces (42 bytes: qns u64, qe tn ts te u32, el no_l_indel u8, ml rev bl del as u32 each - the bit fields widened), a read contributes
 * (is_fully_corrected u8, is_abnormal u8, length u32, its records), the file starts with the read count as int64.  Host code, no device, no context.
 *   hao_ovlp_bin_read: *flags = 2 bytes per read, *off = n_reads + 1 record offsets, *hits = the records; all three malloc'ed (free() them); HAO_EINVAL on a damaged file
 *   hao_ovlp_bin_write: the inverse; the output of a read is byte-identical to what was read */
typedef struct { uint64_t qns; uint32_t qe, tn, ts, te; uint32_t ml, rev, bl, del; uint8_t el, no_l_indel, pad[6]; } hao_ma_hit_t;      /* 48 bytes in memory */
int hao_ovlp_bin_read(const char *path, uint64_t *n_reads, uint8_t **flags, uint64_t **off, hao_ma_hit_t **hits);
int hao_ovlp_bin_write(const char *path, uint64_t n_reads, const uint8_t *flags, const uint64_t *off, const hao_ma_hit_t *hits);

/* Per-read digests of the last batch's results, computed on the device (one workgroup per read) and copied to out[n] / out_kh[n]
 * (n = reads of the batch; out_kh may be NULL):
 *   out[r]    = sum of term(1, i, w) over the 64-bit words of ol->list (6 per overlap_region: the 12 u32 fields of hao_ovlp_t)
 *             + sum of term(2, i, w) over the read's fake cigars in ol order + sum of term(3, i, w) over cl->list (2 words per k_mer_hit)
 *   out_kh[r] = sum of term(4, i, w) over the seed hits before chaining (hao_fetch_seed_hits)
 *   term(s, i, w) = mix64(w + 0x9E3779B97F4A7C15 * (i + 1) + s * 0xD6E8FEB86659FD93)  mod 2^64,  mix64 = splitmix64's finaliser.
 * An end-to-end integrity check for consumers on the far side of the PCIe boundary, and the way the full-size parity tests compare
 * EVERY read of a 500 000-read pass with the reference (oracle/ref_harness.cpp --digest computes the same value from the reference's
 * own overlap_region / Candidates_list after h_ec_lchain, anchor.cpp:2302). */
int hao_batch_digest(hao_ctx *c, uint64_t *out, uint64_t *out_kh);
/* The same out[r] for every read of a DELIVERED batch (needs HAO_DELIVER_OL | HAO_DELIVER_CL), computed on the host from what landed in the pinned arena:
 * ol->list, the fake cigars, and cl->list decoded out of the wire format by hao_unpack_hits.  A pure function of the view (any thread, while the slot is
 * not being rewritten); the reads are spread over n_threads host threads.  What crossed PCIe can thus be compared, read by read, with the device's own
 * digest or with the reference's (tests/test_gpu_fullgold.py does it for all 500 000 reads of configs[2]; bench.py for the batches it delivers). */
int hao_delivery_digest(const hao_delivery_t *d, uint64_t *out, int n_threads);

/* Device self-test of the record grouping used by the sharded index build (pins a rocPRIM bit-range sort behaviour, see hao_capi_rest.hpp):
 * out[0] = order violations of the begin_bit = 48 sort, out[1] = of the path the engine uses (must be 0). */
int hao_selftest_rocprim(uint64_t n, uint64_t out[2]);
/* Self-test of the code paths that handle more than 2^32 items (grid-stride launches, run-length encoding through reduce_by_key): n u32 keys i / 8;
 * out = { runs, sum of run lengths, runs of a length other than 8 } - n / 8, n, 0 for n a multiple of 8 (tests/test_gpu_rocprim.py). */
int hao_selftest_big(uint64_t n, uint64_t out[3]);

/* Self-test of the index sort on 40 of the 64 hash bits + fix-up of the runs that hold several keys (hao_index.cuh; what hao_pt_gen runs on more than 2^23
 * minimizers): n synthetic keys with many such runs, sorted that way and by the stable 64-bit sort; out = { positions where the two results differ (must
 * be 0), runs the fix-up rewrote (must be > 0 for the test to mean anything), scratch elements used, scratch overflow flag }. */
int hao_selftest_sortbits(hao_ctx *c, uint64_t n, uint64_t out[4]);

/* per-stage device time of the last call in milliseconds (HIP events on the engine's stream);
 * names[i] points to static strings. Returns the number of stages. */
int hao_stage_times(hao_ctx *c, const char **names, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif
