// libhao.so: C ABI (include/hao.h) + host orchestration of the HIP kernels.
// Single translation unit; the kernel files are headers.
#include <algorithm>
#include <cmath>
#include <cstring>
#include "hao_ctx.hpp"
#include "hao_sketch.cuh"
#include "hao_select2.cuh"
#include "hao_host.hpp"
#include "hao_index.cuh"
#include "hao_query.cuh"
#include "hao_chain.cuh"
#include "hao_deliver.cuh"
#include "hao_comm.hpp"
#include "hao_pipeline.hpp"
#include "hao_tables.hpp"
#include "hao_batch.hpp"
#include "hao_files.hpp"

extern "C" {

void hao_opt_default(hao_opt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->k = 51; o->w = 51; o->hpc = 1; o->sample_dist = 500; o->rewin = 1000; o->min_hist_cnt = 5;
	o->max_kmer_cnt = 2000; o->max_n_chain = 100; o->high_factor = 5.0; o->is_ont = 0; o->hg_size = -1;
}

int hao_create(int device, const hao_opt_t *opt, hao_ctx **out)
{
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return HAO_ENODEV;   // no CPU fallback, by design
	if (hipSetDevice(device) != hipSuccess) return HAO_ENODEV;
	if (!opt || opt->k <= 0 || opt->k > 63 || opt->w <= 0 || opt->w >= 256) return HAO_EINVAL;
	hao_ctx *c = new hao_ctx();
	c->device = device; c->opt = *opt; c->max_n_chain = opt->max_n_chain; c->sw.load();
	{ int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu; }      // (persistent kernels: one workgroup per CU)
	{	// HIP multiplexes streams onto a few hardware queues: with several copy streams in flight the engine's stream can end up behind a bulk copy in its
		// queue (measured: every batch started ~5 ms late with four copy streams), which is why the delivery path uses ONE copy stream by default.
		// (The engine's streams in the highest priority class instead - own queues - measured worse: the copy then starves.)
		int lo_ = 0, hi_ = 0; (void)hipDeviceGetStreamPriorityRange(&lo_, &hi_);
		if (hipStreamCreateWithPriority(&c->stream, hipStreamDefault, lo_) != hipSuccess) { delete c; return HAO_ENODEV; }
	}
	if (hipHostMalloc((void**)&c->peek_h, 512 * 8, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->peek_d, c->peek_h, 0) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return HAO_ENOMEM; }
	memset(c->ft_hist, 0, sizeof(c->ft_hist)); memset(c->pt_hist, 0, sizeof(c->pt_hist));
	*out = c;
	return HAO_OK;
}

void hao_destroy(hao_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	hao_batch_free(c);
	if (c->comm) { if (c->comm->nccl) ncclCommDestroy(c->comm->nccl); c->comm->release(); delete c->comm; c->comm = nullptr; }
	// DevBuf members are released explicitly (no destructors: the struct is POD-ish on purpose)
	hao_release_all(c);
	(void)hipStreamDestroy(c->stream);
	if (c->peek_h) (void)hipHostFree(c->peek_h);
	delete c;
}

const char *hao_last_error(const hao_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

#define HAO_NOT_ON_VIEW(c, what) do { if ((c)->owner) { hao_set_err((c), what ": not on an attached batch context (hao_attach)"); return HAO_EINVAL; } } while (0)

int hao_attach(hao_ctx *o, hao_ctx **out)
{
	if (!o || !out) return HAO_EINVAL;
	*out = nullptr;
	if (o->owner) { hao_set_err(o, "hao_attach: attach to the owning engine, not to another view"); return HAO_EINVAL; }
	hao_ctx *c = nullptr;
	if (int rc = hao_create(o->device, &o->opt, &c)) return rc;
	c->owner = o; c->attached_gen = ~0ULL;      // the first batch takes the owner's current reads and index (hao_view_refresh)
	*out = c;
	return HAO_OK;
}

int hao_set_reads(hao_ctx *c, const uint8_t *packed, const uint64_t *pk_off, const uint32_t *len, uint64_t n_reads,
				  const uint64_t *nsite_off, const uint32_t *nsite)
{
	if (!c || !packed || !pk_off || !len) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_set_reads"); ++c->index_gen;
	if (n_reads >= (1ULL << 28)) { hao_set_err(c, "more than 2^28 reads (htab.cpp:765)"); return HAO_EUNSUPP; }
	HIP_TRY(hipSetDevice(c->device));
	c->n_reads = n_reads; c->n_pk_bytes = pk_off[n_reads]; c->max_len = 0;
	c->h_len.assign(len, len + n_reads);
	c->n_bases = 0;
	for (uint64_t i = 0; i < n_reads; ++i) {
		if (len[i] >= (1u << 27)) { hao_set_err(c, "read longer than 2^27 (htab.h:13-18)"); return HAO_EUNSUPP; }
		c->n_bases += len[i];
	}
	HIP_TRY(c->d_packed.reserve(c->n_pk_bytes + 16)); HIP_TRY(c->d_pk_off.reserve(n_reads + 1)); HIP_TRY(c->d_len.reserve(n_reads + 1));
	HIP_TRY(hipMemcpyAsync(c->d_packed.p, packed, c->n_pk_bytes, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemsetAsync(c->d_packed.p + c->n_pk_bytes, 0, 16, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_pk_off.p, pk_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_len.p, len, n_reads * 4, hipMemcpyHostToDevice, c->stream));
	c->has_n = nsite_off && nsite && nsite_off[n_reads] > 0;
	c->h_nsite_off.clear();
	if (c->has_n) {
		c->h_nsite_off.assign(nsite_off, nsite_off + n_reads + 1);
		HIP_TRY(c->d_nsite_off.reserve(n_reads + 1)); HIP_TRY(c->d_nsite.reserve(nsite_off[n_reads] + 1));
		HIP_TRY(hipMemcpyAsync(c->d_nsite_off.p, nsite_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->d_nsite.p, nsite, nsite_off[n_reads] * 4, hipMemcpyHostToDevice, c->stream));
	}
	// unsharded default: this engine owns every read
	c->rid_base = 0; c->n_total = n_reads; c->h_len_all = c->h_len;
	c->max_len_all = 0; for (uint64_t i = 0; i < n_reads; ++i) c->max_len_all = std::max(c->max_len_all, len[i]);
	HIP_TRY(c->d_len_all.reserve(n_reads + 1));
	HIP_TRY(hipMemcpyAsync(c->d_len_all.p, len, n_reads * 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->has_ft = false; c->has_pt = false; c->h_ix_valid = false; c->sk_n = 0;
	return HAO_OK;
}

int hao_set_shard(hao_ctx *c, uint64_t rid_base, uint64_t n_total, const uint32_t *all_len)
{
	if (!c || !all_len || rid_base + c->n_reads > n_total) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_set_shard"); ++c->index_gen;
	if (n_total >= (1ULL << 28)) { hao_set_err(c, "more than 2^28 reads (htab.cpp:765)"); return HAO_EUNSUPP; }
	for (uint64_t i = 0; i < c->n_reads; ++i) if (all_len[rid_base + i] != c->h_len[i]) { hao_set_err(c, "all_len disagrees with the local read lengths"); return HAO_EINVAL; }
	HIP_TRY(hipSetDevice(c->device));
	c->rid_base = rid_base; c->n_total = n_total; c->h_len_all.assign(all_len, all_len + n_total); c->max_len = 0;
	c->max_len_all = 0; for (uint64_t i = 0; i < n_total; ++i) c->max_len_all = std::max(c->max_len_all, all_len[i]);
	HIP_TRY(c->d_len_all.reserve(n_total + 1));
	HIP_TRY(hipMemcpy(c->d_len_all.p, all_len, n_total * 4, hipMemcpyHostToDevice));
	c->has_ft = false; c->has_pt = false; c->h_ix_valid = false;
	return HAO_OK;
}

int hao_dist_unique_id(uint8_t id[128])
{
	ncclUniqueId u; if (ncclGetUniqueId(&u) != ncclSuccess) return HAO_ENODEV;
	static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId");
	memset(id, 0, 128); memcpy(id, &u, sizeof(u));
	return HAO_OK;
}

int hao_dist_init(hao_ctx *c, const uint8_t id[128], int rank, int world)
{
	if (!c || !id || rank < 0 || rank >= world) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_dist_init");
	HIP_TRY(hipSetDevice(c->device));
	if (!c->comm) c->comm = new hao_comm();
	ncclUniqueId u; memcpy(&u, id, sizeof(u));
	c->comm->rank = rank; c->comm->world = world; c->comm->loop = nullptr;
	{	// libhao.so is compiled against /opt/rocm's rccl.h but binds at run time to whichever librccl.so.1 the process loaded first (torch bundles
		// its own): the calls used here (unique id, comm init, all-gather, all-reduce, broadcast, send / recv) are stable across 2.x, a different major is not
		int rt = 0; NCCL_TRY(ncclGetVersion(&rt));
		const int rt_major = rt >= 10000 ? rt / 10000 : rt / 1000;
		if (rt_major != NCCL_MAJOR) { hao_set_err(c, "RCCL major version " + std::to_string(rt_major) + " at run time, " + std::to_string(NCCL_MAJOR) + " at compile time"); return HAO_EUNSUPP; }
	}
	NCCL_TRY(ncclCommInitRank(&c->comm->nccl, world, u, rank));
	return HAO_OK;
}

// loopback group: `world` engines inside one process (one host thread each) on one GPU - for single-GPU tests of the sharded path
void *hao_loop_create(int world)
{
	hao_loop_group *g = new hao_loop_group();
	g->world = world; pthread_barrier_init(&g->bar, nullptr, world); g->ptr.assign(world, nullptr); g->ptr2.assign(world, nullptr); g->cnt.resize(world); g->hostv.resize(world);
	return g;
}
void hao_loop_destroy(void *grp) { if (grp) { hao_loop_group *g = (hao_loop_group*)grp; pthread_barrier_destroy(&g->bar); delete g; } }
int hao_dist_init_loopback(hao_ctx *c, void *grp, int rank)
{
	if (!c || !grp) return HAO_EINVAL;
	hao_loop_group *g = (hao_loop_group*)grp;
	if (rank < 0 || rank >= g->world) return HAO_EINVAL;
	if (!c->comm) c->comm = new hao_comm();
	c->comm->rank = rank; c->comm->world = g->world; c->comm->loop = g; c->comm->nccl = nullptr;
	return HAO_OK;
}

int hao_sketch_batch(hao_ctx *c, uint64_t rid_lo, uint64_t rid_hi, int use_ft, int sample_dist)
{
	if (!c || rid_lo > rid_hi || rid_hi > c->n_reads) return HAO_EINVAL;
	HAO_NOT_ON_VIEW(c, "hao_sketch_batch");
	HIP_TRY(hipSetDevice(c->device));
	c->timer.begin(c->stream);
	int rc = hao_sketch_run(c, rid_lo, rid_hi, use_ft && c->has_ft, sample_dist, 1);
	if (rc != HAO_OK) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->timer.collect(c->stage_ms);
	c->h_mz_off.resize(c->sk_n + 1);
	HIP_TRY(hipMemcpy(c->h_mz_off.data(), c->d_mz_off.p, (c->sk_n + 1) * 8, hipMemcpyDeviceToHost));
	return HAO_OK;
}

int hao_fetch_sketch(hao_ctx *c, uint64_t rid, const hao_mz_t **mz, uint64_t *n)
{
	if (!c || rid < c->sk_lo || rid >= c->sk_lo + c->sk_n || c->h_mz_off.size() != c->sk_n + 1) return HAO_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	uint64_t s = c->h_mz_off[rid - c->sk_lo], e = c->h_mz_off[rid - c->sk_lo + 1], m = e - s;
	c->h_mz_fetch.resize(m + 1);
	std::vector<uint64_t> x(m + 1), info(m + 1);
	if (m) {
		HIP_TRY(hipMemcpy(x.data(), c->d_mz_x.p + s, m * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(info.data(), c->d_mz_info.p + s, m * 8, hipMemcpyDeviceToHost));
	}
	for (uint64_t i = 0; i < m; ++i) { c->h_mz_fetch[i].x = x[i]; c->h_mz_fetch[i].info = info[i]; }
	*mz = c->h_mz_fetch.data(); *n = m;
	return HAO_OK;
}

int hao_stage_times(hao_ctx *c, const char **names, float *ms, int cap)
{
	if (!c) return 0;
	int n = 0;
	for (auto &p : c->stage_ms) { if (n >= cap) break; names[n] = p.first.c_str(); ms[n] = p.second; ++n; }
	return n;
}

} // extern "C"

#include "hao_capi_rest.hpp"
