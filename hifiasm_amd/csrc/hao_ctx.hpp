// Engine context: device buffers, stream, stage timers (host side of libhao.so).
#pragma once
#include <time.h>
#include <rocprim/rocprim.hpp>
#include <map>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "hao_common.cuh"

struct hao_ctx;
static void hao_set_err(hao_ctx *c, const std::string &m);

// grow-only device buffer
template<typename T> struct DevBuf {
	T *p = nullptr; size_t cap = 0; bool borrowed = false;      // borrowed: a read-only view of another engine's buffer (hao_attach) - never freed or grown here
	hipError_t reserve(size_t n) {
		if (n <= cap) return hipSuccess;
		if (borrowed) return hipErrorInvalidValue;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		size_t want = n + n / 8 + 64;
		hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
		if (e == hipSuccess) cap = want;
		return e;
	}
	// allocate exactly n elements if smaller (twin of a ping-pong pair: avoids a late first allocation inside a timed pass)
	hipError_t reserve_exact(size_t n) {
		if (n <= cap) return hipSuccess;
		if (borrowed) return hipErrorInvalidValue;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
		if (e == hipSuccess) cap = n;
		return e;
	}
	void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
	void borrow(const DevBuf<T> &o) { release(); p = o.p; cap = o.cap; borrowed = o.p != nullptr; }
};

static bool hao_dbg_sync = false;      // HAO_DBG_PRINT=sync (hao_switches::load): wait after every stage and say its name - localises a device fault
struct StageTimer {
	std::vector<std::string> names; std::vector<hipEvent_t> ev; std::vector<double> host_t; hipStream_t st = nullptr;
	static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
	void begin(hipStream_t s) { st = s; names.clear(); host_t.clear(); mark("__begin"); }
	void mark(const char *name) {
		size_t i = names.size();
		if (i >= ev.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev.push_back(e); }
		names.push_back(name); host_t.push_back(now()); (void)hipEventRecord(ev[i], st);
		if (hao_dbg_sync) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[stage] %s: %s\n", name, hipGetErrorString(e)); fflush(stderr); }
	}
	// after stream sync: ms between consecutive marks, labelled by the later mark
	void collect(std::vector<std::pair<std::string, float> > &out) {
		out.clear();
		for (size_t i = 1; i < names.size(); ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]); out.push_back(std::make_pair(names[i], ms)); }
		// the HOST's wall clock between the same marks for ha_ft_gen's stages ("host_ft_..."): its wall time is allocation, scratch and per-read host loops, not kernels
		for (size_t i = 1; i < names.size(); ++i) if (names[i].compare(0, 3, "ft_") == 0) out.push_back(std::make_pair("host_" + names[i], (float)((host_t[i] - host_t[i - 1]) * 1e3)));
	}
	~StageTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
};

// run-time switches: read from the environment ONCE, in hao_create.  Eight variables:
//   HAO_SEED_LDS, HAO_SEED_LDS_RATIO, HAO_SEED_MERGE_MAXN   which reads / batches the list-major seed kernel takes (below)
//   HAO_FT_PASSES                                           ha_ft_gen's hash-range passes (0: what the free device memory asks for)
//   HAO_ARENA_NUMA                                          placement of the pinned delivery arenas
//   HAO_DBG_PRINT=seed,qc,dp,sel,dl,bloom,sync              timers / counters on stderr (sync: wait after every stage and say its name - localises a device fault)
//   HAO_DBG_FORCE=seq_chain,dp_seqtail,dp_nospec,dp_serial,seq_prune,noql      send every read / group down one of the engine's FALLBACK paths (tests: each has to give the default path's bytes)
//   HAO_DBG_TEST=ix_pad=N,sort40_min=N,sk_gcap=N,exc_cap=N,fc_raw_every=N,exc_every=N,qmz_raw=1,arena_probe=1,ft_chunk_slots=N      capacities and thresholds shrunk so that small inputs reach the overflow / big-index code
struct hao_switches {
	bool seedphase = false, qcphase = false, dp_stats = false, selphase = false, dltime = false, bloom = false;                 // HAO_DBG_PRINT
	bool seq_chain = false, dp_seqtail = false, dp_nospec = false, dp_serial = false, seq_prune = false, seed_noql = false;     // HAO_DBG_FORCE
	long long sk_gcap = -1, exc_cap = -1, ft_chunk_slots = 0; unsigned long long ix_pad = 0, sort40_min = 1ULL << 23; int fc_raw_every = 0, exc_every = 0; bool qmz_raw = false, arena_probe = false;      // HAO_DBG_TEST
	int arena_numa = 3, seed_merge_maxn = 24000, seed_lds_ratio = 120, ft_passes = 0, seed_lds = 1;
	static bool in_list(const char *v, const char *name) {      // name is an element of the comma-separated list v
		const size_t n = strlen(name);
		for (const char *p = v; p && *p; ) { const char *e = strchr(p, ','); const size_t l = e ? (size_t)(e - p) : strlen(p); if (l == n && strncmp(p, name, n) == 0) return true; p = e ? e + 1 : nullptr; }
		return false;
	}
	static bool in_kv(const char *v, const char *name, unsigned long long &out) {      // "name=123" is an element of v
		const size_t n = strlen(name);
		for (const char *p = v; p && *p; ) { const char *e = strchr(p, ','); if (strncmp(p, name, n) == 0 && p[n] == '=') { out = strtoull(p + n + 1, nullptr, 10); return true; } p = e ? e + 1 : nullptr; }
		return false;
	}
	void load() {
		if (const char *v = getenv("HAO_DBG_PRINT")) {
			seedphase = in_list(v, "seed"); qcphase = in_list(v, "qc"); dp_stats = in_list(v, "dp"); selphase = in_list(v, "sel"); dltime = in_list(v, "dl"); bloom = in_list(v, "bloom"); hao_dbg_sync = in_list(v, "sync");
		}
		if (const char *v = getenv("HAO_DBG_FORCE")) {
			seq_chain = in_list(v, "seq_chain");      // every group through the complete sequential chaining routine (one lane per group)
			dp_seqtail = in_list(v, "dp_seqtail");    // groups the quick check rejects: the DP's tail walked sequentially
			dp_nospec = in_list(v, "dp_nospec");      // the DP without speculation
			dp_serial = in_list(v, "dp_serial");      // DP and pack kernels on the engine's stream (no side streams)
			seq_prune = in_list(v, "seq_prune");      // the selection's pruning by one lane
			seed_noql = in_list(v, "noql");           // the seed kernels' generic per-minimizer tables (the path of batches with a read of more than HAO_QTAB_CAP minimizers)
		}
		if (const char *v = getenv("HAO_DBG_TEST")) {
			unsigned long long x;
			if (in_kv(v, "ix_pad", x)) ix_pad = x;                        // unused position records in front of the index (list starts beyond 2^32 on a small read set)
			if (in_kv(v, "sort40_min", x)) sort40_min = x;                // the big-index path (40-bit sort + fix-up, gather, windowed scatter) from this many minimizers on (default 2^23: rocprim's bit-range sort, tests/test_gpu_rocprim.py)
			if (in_kv(v, "sk_gcap", x)) sk_gcap = (long long)x;           // capacity of the sketch's minimizer pool (the grow-and-rerun path)
			if (in_kv(v, "exc_cap", x)) exc_cap = (long long)x;           // capacity of the wire format's verbatim-hit list (the grow-and-repack path)
			if (in_kv(v, "fc_raw_every", x)) fc_raw_every = (int)x;       // every n-th overlap's fake cigar travels raw (the fallback of the packed wire form)
			if (in_kv(v, "exc_every", x)) exc_every = (int)x;             // every n-th hit of a chain travels verbatim (the exception list)
			if (in_kv(v, "qmz_raw", x)) qmz_raw = x != 0;                   // the delivered minimizer tables in their 8-byte form whatever the read lengths (the form of batches with a read of 65 536 bases or more)
			if (in_kv(v, "arena_probe", x)) arena_probe = x != 0;           // the delivery arenas' placement probe (hao_deliver_enqueue) whatever their size and rate
			if (in_kv(v, "ft_chunk_slots", x)) ft_chunk_slots = (long long)x;      // k-mer slots hashed per chunk of reads in ha_ft_gen's pass mode
		}
		if (const char *e = getenv("HAO_SEED_LDS")) seed_lds = atoi(e) ? 1 : 0;      // 0 = the table kernels (hao_query.cuh, hao_query3.cuh) for every read instead of the list-major kernel (hao_query5.cuh): the tests run them on every scenario - they carry repeat-rich batches and the reads the list-major kernel leaves
		if (const char *e = getenv("HAO_SEED_LDS_RATIO")) seed_lds_ratio = std::max(0, atoi(e));      // (per cent) batches with more seed hits per (query minimizer x coverage peak) than this take the table kernels: reads across repeat families (hao_batch.hpp); tests force either side
		if (const char *e = getenv("HAO_SEED_MERGE_MAXN")) seed_merge_maxn = std::max(0, atoi(e));      // reads with more seed hits than this are left to the table kernels by the list-major kernel (repeat families: hundreds of targets per read)
		if (const char *e = getenv("HAO_FT_PASSES")) ft_passes = std::max(0, atoi(e));
		if (const char *e = getenv("HAO_ARENA_NUMA")) arena_numa = atoi(e);      // 0: plain hipHostMalloc, 1: thread policy "prefer the GPU's node", 3 (default): "bind to it", then 1 if that fails - and, when the pages still are elsewhere, mmap + mbind + hipHostRegister; 2: 3 + hipHostMallocNumaUser; 4: always mmap + mbind + hipHostRegister (tests)
	}
};

// A few words the host needs from the device in the middle of a batch (sizes for the next allocation): written by a one-wave kernel into pinned,
// device-mapped host memory instead of a hipMemcpy.  A D2H memcpy - however small - queues on the device-to-host DMA engine, behind the bulk copy of the
// previous batch's results that the delivery path has in flight there: the "asynchronous" delivery would serialise with the compute it should hide under.
static __global__ void hao_peek_kernel(const unsigned long long *src, int n, unsigned long long *dst)
{ if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x]; }

struct hao_ctx {
	int device = 0; hao_opt_t opt; std::string err; hipStream_t stream = nullptr; hao_switches sw;
	// hao_attach: a second batch context over this engine's reads and index (own stream, scratch, results); index_gen counts the owner's rebuilds
	hao_ctx *owner = nullptr; uint64_t index_gen = 0, attached_gen = 0;
	unsigned long long *peek_h = nullptr, *peek_d = nullptr;      // 512 words of mapped pinned memory
	// ---- read store (HBM) ----
	uint64_t n_reads = 0, n_bases = 0, n_pk_bytes = 0; bool has_n = false; uint32_t max_len = 0;
	DevBuf<uint8_t> d_packed; DevBuf<uint64_t> d_pk_off; DevBuf<uint32_t> d_len; DevBuf<uint64_t> d_nsite_off; DevBuf<uint32_t> d_nsite;
	std::vector<uint32_t> h_len; std::vector<uint64_t> h_nsite_off;
	// sharded mode: this engine holds reads [rid_base, rid_base + n_reads) of n_total; lengths of ALL reads are replicated
	uint64_t rid_base = 0, n_total = 0; uint32_t max_len_all = 0; int n_cu = 256; DevBuf<uint32_t> d_len_all; std::vector<uint32_t> h_len_all; struct hao_comm *comm = nullptr;
	// ---- filter table ----
	bool has_ft = false; int ft_peak_hom = -1, ft_peak_het = -1, ft_cutoff = 0, ft_passes_used = 1; int64_t ft_hist[HAO_N_COUNTS];
	std::vector<uint64_t> h_ft_keys; std::vector<int32_t> h_ft_vals;
	DevBuf<uint64_t> d_ft_keys; DevBuf<int32_t> d_ft_vals; DevBuf<uint32_t> d_ft_bucket;
	DevBuf<uint32_t> d_ft_hbit; DevBuf<unsigned long long> d_ft_hslot; int ft_hbits = 0;      // the filter table's hash view for the device lookups (hao_sketch.cuh: hao_ft_dev)
	int max_n_chain = 100, hom_cov = -1, het_cov = -1;
	// ---- sketch workspace / results ----
	uint64_t sk_lo = 0, sk_n = 0, sk_total = 0; bool sk_is_index = false;
	DevBuf<uint64_t> d_tile_off; DevBuf<uint32_t> d_tile_ord, d_n_runs, d_tot_l; DevBuf<uint64_t> d_chunk_off, d_chunk_cnt64;
	DevBuf<uint8_t> d_scalar_flag; DevBuf<uint32_t> d_scalar_list, d_unit_rid;
	DevBuf<uint64_t> d_pool_x, d_pool_info; DevBuf<uint32_t> d_pool_ord; DevBuf<unsigned long long> d_cursor; DevBuf<int> d_err;
	DevBuf<uint64_t> d_chunk_base, d_chunk_dst; DevBuf<uint32_t> d_chunk_cnt;
	DevBuf<uint64_t> d_g_x, d_g_info; DevBuf<uint32_t> d_g_ord; DevBuf<uint64_t> d_g_off; DevBuf<uint32_t> d_new_n; DevBuf<uint64_t> d_new_n64;
	DevBuf<uint64_t> d_mz_x, d_mz_info, d_mz_off;          // final per-read minimizers of the last sketch_batch
	DevBuf<unsigned char> d_tmp; DevBuf<unsigned char> d_ring; DevBuf<uint32_t> d_ringord, d_cnt_ws;
	std::vector<uint64_t> h_mz_off; std::vector<hao_mz_t> h_mz_fetch;
	// ---- index (pt) ----
	bool has_pt = false; int64_t pt_hist[HAO_N_COUNTS];
	uint64_t ix_n_mz = 0, ix_n_sorted = 0, ix_n_keys = 0, ix_n_pos = 0, ix_pad = 0; int ix_bucket_bits = 16;   // ix_n_mz: local read-ordered records; ix_n_sorted: records in the (replicated) index
	DevBuf<uint64_t> d_ix_mz_x, d_ix_mz_info, d_ix_mz_off;  // all reads' minimizers in read order (query side reuses them)
	DevBuf<uint64_t> d_ix_sx, d_ix_sinfo;                    // sorted by hash (stable)
	DevBuf<uint64_t> d_ix_lk; bool lk_valid = false; DevBuf<uint32_t> w_runid;   // per minimizer (read order): list start | count << 48 of its key (single-device build)
	DevBuf<uint64_t> d_ix_keys, d_ix_start; DevBuf<uint32_t> d_ix_cnt; DevBuf<uint32_t> d_ix_bucket;
	DevBuf<uint64_t> w_ukeys, w_flag, w_kpos, w_ustart; DevBuf<uint32_t> w_ucnt; DevBuf<unsigned long long> w_hist; DevBuf<uint32_t> w_ok, w_ok2, w_oi, w_oi2;   // persistent scratch of the index build
	DevBuf<uint32_t> w_s40_list, w_s40_o; DevBuf<uint64_t> w_s40_x, w_lkv2; DevBuf<unsigned long long> w_s40_cnt; uint64_t s40_runs = 0;      // fix-up of the 40-bit index sort (hao_index.cuh)
	std::vector<uint64_t> h_ix_keys, h_ix_off, h_ix_pos, h_ix_mz_off; bool h_ix_valid = false;
	// ---- query batch ----
	// f3 (hao_align.cuh): scratch of the window-alignment batches, kept between calls (a hipMalloc / hipFree pair per buffer and call cost more than the kernels)
	uint64_t al_grid_n = 0;      // pairs hao_window_ed_grid left in al_task / al_res
	DevBuf<hao_ed_task_t> al_task; DevBuf<uint64_t> al_k1, al_k2, al_path; DevBuf<uint32_t> al_i1, al_order, al_sel; DevBuf<hao_ed_result_t> al_res; DevBuf<hao_trace_result_t> al_tres;
	DevBuf<uint8_t> al_want; DevBuf<uint16_t> al_cig;
	struct Batch;
	Batch *batch = nullptr;
	StageTimer timer; std::vector<std::pair<std::string, float> > stage_ms;
};

static void hao_set_err(hao_ctx *c, const std::string &m) { if (c) c->err = m; }

#define HAO_CHECK_LAUNCH() HIP_TRY(hipGetLastError())

// sharded mode (the engine holds a slice of the read store)?  A view (hao_attach) has no communicator of its own: its owner's decides.
static bool hao_comm_is_active(const struct hao_comm *cm);
static inline bool hao_is_sharded(const hao_ctx *c) { const hao_ctx *o = c->owner ? c->owner : c; return o->comm && hao_comm_is_active(o->comm); }

// scratch for rocprim calls
static inline hipError_t hao_tmp(hao_ctx *c, size_t bytes) { return c->d_tmp.reserve(bytes + 256); }

template<typename In, typename Out>
static int hao_excl_scan_u64(hao_ctx *c, In in, Out out, size_t n)
{
	size_t tb = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tb, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
	HIP_TRY(hao_tmp(c, tb));
	HIP_TRY(rocprim::exclusive_scan(c->d_tmp.p, tb, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
	return HAO_OK;
}

// An attached batch context (hao_attach) reads the owner's read store and index through borrowed pointers; they are taken again whenever the owner has
// rebuilt something since (the owner must not be rebuilding while a view runs a batch: same rule as for the owner's own batches).
static int hao_view_refresh(hao_ctx *c)
{
	hao_ctx *o = c->owner;
	if (!o || c->attached_gen == o->index_gen) return HAO_OK;
	c->opt = o->opt;
	c->n_reads = o->n_reads; c->n_bases = o->n_bases; c->n_pk_bytes = o->n_pk_bytes; c->has_n = o->has_n; c->max_len = o->max_len;
	c->rid_base = o->rid_base; c->n_total = o->n_total; c->max_len_all = o->max_len_all; c->n_cu = o->n_cu; c->max_n_chain = o->max_n_chain; c->hom_cov = o->hom_cov; c->het_cov = o->het_cov;
	c->has_pt = o->has_pt; c->ix_n_mz = o->ix_n_mz; c->ix_n_sorted = o->ix_n_sorted; c->ix_n_keys = o->ix_n_keys; c->ix_n_pos = o->ix_n_pos; c->ix_pad = o->ix_pad; c->ix_bucket_bits = o->ix_bucket_bits; c->lk_valid = o->lk_valid;
	c->h_len = o->h_len; c->h_nsite_off = o->h_nsite_off; c->h_len_all = o->h_len_all; c->h_ix_mz_off = o->h_ix_mz_off;      // (empty: copied from the device on first use)
	c->d_packed.borrow(o->d_packed); c->d_pk_off.borrow(o->d_pk_off); c->d_len.borrow(o->d_len); c->d_len_all.borrow(o->d_len_all); c->d_nsite_off.borrow(o->d_nsite_off); c->d_nsite.borrow(o->d_nsite);
	c->d_ix_mz_x.borrow(o->d_ix_mz_x); c->d_ix_mz_info.borrow(o->d_ix_mz_info); c->d_ix_mz_off.borrow(o->d_ix_mz_off); c->d_ix_sinfo.borrow(o->d_ix_sinfo); c->d_ix_lk.borrow(o->d_ix_lk);
	c->d_ix_keys.borrow(o->d_ix_keys); c->d_ix_start.borrow(o->d_ix_start); c->d_ix_cnt.borrow(o->d_ix_cnt); c->d_ix_bucket.borrow(o->d_ix_bucket);
	c->attached_gen = o->index_gen;
	return HAO_OK;
}
