// Host orchestration of one query batch: h_ec_lchain for reads [lo, hi) (part of libhao.so).
#pragma once
#include <sys/mman.h>
#include "hao_tables.hpp"
#include "hao_query.cuh"
#include "hao_query3.cuh"
#include "hao_query5.cuh"
#include "hao_grid.cuh"
#include "hao_chain.cuh"

struct hao_ctx::Batch {
	uint64_t n_generic = 0, n_generic_hits = 0, seed_path = 0, seed_left[3] = {0, 0, 0}; DevBuf<unsigned long long> stats, dbgbuf;
	uint64_t lo = 0, n = 0, mz0 = 0, n_mz = 0, n_anchor = 0, n_groups = 0, n_chains = 0, n_cl = 0, n_fc_raw = 0, n_ol = 0, n_fc = 0, n_fcw = 0;
	bool valid = false, host_valid = false;
	DevBuf<uint64_t> s_start, s_pk, a_off, seg, g_cnt, g_off, g_start, ch_base, cl_base, fc_base, fcs, fc_raw, ol_fc_off, cc_off, cc, fc_final, fcf_off;
	DevBuf<uint64_t> nch64;
	DevBuf<uint64_t> g_tmp, cls_cc, cls_co; DevBuf<hao_gent> glist; DevBuf<uint8_t> g_cls; DevBuf<uint32_t> slow, ovf_list; hipStream_t side[HAO_NCLS]; hipEvent_t ev_qc[HAO_NCLS], ev_dp[HAO_NCLS]; bool side_ready = false; hipEvent_t ev_pk0 = nullptr, ev_pk1 = nullptr; DevBuf<unsigned char> pk_tmp;
	DevBuf<uint32_t> q_pos, q_cnt, s_n, g_read, wgt, nch, nout, perm, n_final, fclen;
	DevBuf<hao_hit_t> hits, ohits, cl;
	DevBuf<int32_t> f, ii, p, key_sc, tm; DevBuf<int64_t> t; DevBuf<uint64_t> key_xs; DevBuf<uint32_t> key_al, key_tmp;
	DevBuf<hao_chain_rec> rec; DevBuf<hao_ovlp_t> ol; DevBuf<hao_cdesc> cd; bool cl_valid = false;
	DevBuf<uint16_t> hq, ohq; DevBuf<uint8_t> hcode;      // delivery path: query minimizer index / wire code of every seed hit (seed kernel, chain_group_kernel)
	DevBuf<uint32_t> pk_cnt, pk_ecnt, pk_erank; uint64_t n_codes = 0;      // one code byte per chained hit (device only) before it is split into bits + code bytes
	// Results of a batch that leave the device.  Two sets (+ two pinned host arenas): while the copy stream drains the set of batch i, batch i + 1
	// computes into the other one (hao_overlap_batch_async).  The blocking API keeps using the current set.
	struct OutSet {
		DevBuf<hao_ovlp_t> ol_out; DevBuf<hao_ovlp_wire_t> ol_wire; DevBuf<uint64_t> fin_off, fc_out, fc_out_off, ch_off, cl_off, qm_off, fcw_off; DevBuf<uint32_t> fcw;      // (fcw*: the fake cigars as they travel, hao_deliver.cuh)
		//      // ol->list in final order, per-read offsets, fake cigars
		DevBuf<hao_chain_hdr_t> hdr; DevBuf<uint64_t> bits; DevBuf<uint32_t> rank, rank4; DevBuf<uint8_t> codes; DevBuf<hao_exc_t> exc; DevBuf<hao_qmz_t> qmz; DevBuf<uint16_t> qmz_pos, qmz_cnt; bool qmz16 = false;   // cl->list in the wire format (hao_deliver.cuh)
		DevBuf<uint8_t> exact;                                                                        // exact-overlap flags of ol_out
		void release() { fcw_off.release(); fcw.release(); ol_out.release(); ol_wire.release(); fin_off.release(); fc_out.release(); fc_out_off.release(); ch_off.release(); cl_off.release(); qm_off.release(); hdr.release(); bits.release(); rank.release(); rank4.release(); codes.release(); exc.release(); qmz.release(); qmz_pos.release(); qmz_cnt.release(); exact.release(); }
	} out[2];
	int cur = 0;
	OutSet &O() { return out[cur]; }
	// delivery state: pinned host arenas, copy stream, per-slot completion events
	unsigned char *arena[2] = { nullptr, nullptr }; size_t arena_cap[2] = { 0, 0 }; bool arena_reg[2] = { false, false };      // arena_reg: mmap + mbind + hipHostRegister (hao_arena_alloc)
	void arena_free(int x) { if (!arena[x]) return; if (arena_reg[x]) { (void)hipHostUnregister(arena[x]); (void)munmap(arena[x], arena_cap[x]); } else (void)hipHostFree(arena[x]); arena[x] = nullptr; arena_cap[x] = 0; arena_reg[x] = false; } hipStream_t copy_stream = nullptr; hipEvent_t ev_ready[2], ev_done[2], ev_cstart[2]; bool arena_bad[2] = { false, false }; int arena_retry[2] = { 0, 0 }, arena_node = -1;      /* arena_node: the NUMA node a probe found best (a box of round 6 reported the GPU on node 0 and copied at 30 GB/s into node 0, 56 into node 1) */ bool dl_ready = false, dl_pending[2] = { false, false };
	uint32_t wgt_hi = 0xffffffffu, wgt_lo = 0xffffffffu, wgt_max = 0xffffffffu;      // (wgt_max: the largest k_mer_hit::cnt the pass's weight table can give)
	double t_evsync = 0, t_enq = 0, t_alloc = 0, t_s1 = 0, t_s2 = 0, t_s3 = 0, t_run = 0, t_pre = 0; uint64_t t_n = 0, t_nrun = 0;      // host-side time spent in the delivery plumbing (HAO_DBG_PRINT=dl)
	hao_delivery_t dl[2]; uint64_t dl_seq = 0, n_exc = 0; uint32_t dl_parts = 0; bool exact_valid = false; std::vector<uint8_t> h_exact;
	// host copies for fetch
	std::vector<uint64_t> h_seg, h_fin_off, h_cl_off, h_fc_out_off; std::vector<hao_hit_t> h_hits, h_cl; std::vector<hao_ovlp_t> h_ol; std::vector<uint64_t> h_fc;
	std::vector<uint64_t> fetch_fc_off, h_cco;
	void release() {
		pk_tmp.release(); if (ev_pk0) { (void)hipEventDestroy(ev_pk0); (void)hipEventDestroy(ev_pk1); ev_pk0 = ev_pk1 = nullptr; }
		if (side_ready) { for (int x = 0; x < HAO_NCLS; ++x) { (void)hipStreamDestroy(side[x]); (void)hipEventDestroy(ev_qc[x]); (void)hipEventDestroy(ev_dp[x]); } side_ready = false; }
		s_start.release(); s_pk.release(); a_off.release(); seg.release(); g_cnt.release(); g_off.release(); g_start.release(); ch_base.release(); cl_base.release();
		fc_base.release(); fcs.release(); fc_raw.release(); ol_fc_off.release(); cc_off.release(); cc.release(); fc_final.release(); fcf_off.release();
		nch64.release(); g_tmp.release(); cls_cc.release(); cls_co.release(); glist.release(); g_cls.release(); slow.release(); ovf_list.release(); q_pos.release(); q_cnt.release(); s_n.release(); g_read.release(); wgt.release(); nch.release(); nout.release(); perm.release(); n_final.release(); fclen.release();
		tm.release(); key_sc.release(); key_xs.release(); key_al.release(); key_tmp.release(); hits.release(); ohits.release(); cl.release(); f.release(); ii.release(); p.release(); t.release(); rec.release(); ol.release(); cd.release(); pk_cnt.release(); pk_ecnt.release(); pk_erank.release(); hq.release(); ohq.release(); hcode.release(); out[0].release(); out[1].release();
		if (dl_ready) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); for (int x = 0; x < 2; ++x) { (void)hipEventDestroy(ev_ready[x]); (void)hipEventDestroy(ev_done[x]); (void)hipEventDestroy(ev_cstart[x]); arena_free(x); } dl_ready = false; }
	}
};

// coverage windows per read of the selection's pruning scan (anchor.cpp:1966-2055: ocv_w-sized windows over the query): len / ocv_w + 2
__global__ void hao_cc_count_kernel(const uint32_t *len, uint64_t rid0, uint64_t n, uint64_t ocv_w, uint64_t *out)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n) return;
	out[r] = r < n ? len[rid0 + r] / ocv_w + 2 : 0;
}

__global__ void hao_fclen_kernel(const hao_chain_rec *rec, const uint32_t *nch, uint64_t n_groups, uint64_t *out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;     // chain slot g*3+c
	if (i > n_groups * HAO_MCOPY_MAX) return;
	if (i == n_groups * HAO_MCOPY_MAX) { out[i] = 0; return; }
	uint64_t g = i / HAO_MCOPY_MAX; uint32_t c = (uint32_t)(i % HAO_MCOPY_MAX);
	out[i] = c < nch[g] ? rec[i].fc_len : 0;
}

#include <chrono>
#include <sys/syscall.h>
#include <unistd.h>
// The delivery arenas should live on the NUMA node the GPU hangs off: on a two-socket host a pinned buffer on the far socket costs the DMA ~40 % of its
// rate (measured: 29-35 GB/s instead of 51-56).  The pages of a hipHostMalloc are placed by the calling thread's memory policy, so the allocation is
// bracketed by set_mempolicy(MPOL_PREFERRED, gpu node) / MPOL_DEFAULT (raw syscalls: no libnuma in the image; failures - seccomp, no sysfs - are ignored).
static int hao_gpu_numa_node(int device)
{
	char bus[64] = {0};
	if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return -1;
	for (char *p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
	char path[160]; snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
	FILE *fp = fopen(path, "r"); int node = -1;
	if (fp) { if (fscanf(fp, "%d", &node) != 1) node = -1; fclose(fp); }
	return node;
}
// The calling thread's NUMA memory policy around one allocation.  The caller may be a thread of a host application (the shim inside hifiasm) that runs under a policy
// of its own (numactl --interleave ...): the policy found is saved and put back, never reset to the default.
struct hao_mempolicy_guard {
	int old_mode = 0; unsigned long old_mask[16]; bool saved = false, applied = false;
	// mode: 1 = MPOL_PREFERRED, 2 = MPOL_BIND
	hao_mempolicy_guard(int node, int mode) {
		memset(old_mask, 0, sizeof(old_mask));
		if (node < 0 || node >= 1024) return;
		saved = syscall(SYS_get_mempolicy, &old_mode, old_mask, 1024UL, nullptr, 0UL) == 0;
		if (!saved) return;      // cannot restore what cannot be read: leave the policy alone
		unsigned long mask[16]; memset(mask, 0, sizeof(mask)); mask[node / 64] |= 1UL << (node % 64);
		applied = syscall(SYS_set_mempolicy, mode, mask, 1024UL) == 0;
	}
	~hao_mempolicy_guard() {
		if (!applied) return;
		bool any = false; for (int i = 0; i < 16; ++i) any |= old_mask[i] != 0;
		(void)syscall(SYS_set_mempolicy, old_mode, any ? old_mask : (unsigned long*)nullptr, any ? 1024UL : 0UL);
	}
};
// how many of 32 sampled pages of [p, p + bytes) lie on `node` (move_pages with no target nodes only reports); -1: cannot tell
static int hao_pages_on_node(const void *p, size_t bytes, int node)
{
	const long ps = sysconf(_SC_PAGESIZE); if (ps <= 0 || bytes < (size_t)ps) return -1;
	void *pg[32]; int st[32]; const size_t np = bytes / (size_t)ps;
	for (int i = 0; i < 32; ++i) { pg[i] = (void*)(((uintptr_t)p + (np - 1) * (size_t)i / 31 * (size_t)ps) & ~(uintptr_t)(ps - 1)); st[i] = -1; }
	if (syscall(SYS_move_pages, 0, 32UL, pg, (const int*)nullptr, st, 0) != 0) return -1;
	int on = 0; for (int i = 0; i < 32; ++i) on += st[i] == node;
	return on;
}
// A pinned host buffer whose pages are ON `node`, whatever the allocator of hipHostMalloc does: anonymous mapping, mbind(MPOL_BIND) before the first touch (the kernel
// then reclaims that node's page cache instead of falling over to the far socket), touched, registered with the runtime.  nullptr when any step fails.
static unsigned char *hao_arena_alloc_bound(size_t bytes, int node)
{
	if (node < 0 || node >= 1024) return nullptr;
	// (2 MB-aligned and advised as huge pages: what round 6's slow arenas had in common was not their node - a fresh mapping on the SAME node copied at 56 GB/s where the
	// hipHostMalloc'ed one gave 30 - which leaves the page size the DMA translates through.  The caller passes a multiple of 2 MB.)
	const size_t HP = (size_t)2 << 20;
	void *m0 = mmap(nullptr, bytes + HP, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (m0 == MAP_FAILED) return nullptr;
	void *m = (void*)(((uintptr_t)m0 + HP - 1) & ~(uintptr_t)(HP - 1));
	if (m != m0) (void)munmap(m0, (size_t)((uintptr_t)m - (uintptr_t)m0));
	{ const uintptr_t end0 = (uintptr_t)m0 + bytes + HP, end = (uintptr_t)m + ((bytes + 4095) & ~(size_t)4095); if (end0 > end) (void)munmap((void*)end, (size_t)(end0 - end)); }
	(void)madvise(m, bytes, MADV_HUGEPAGE);
	unsigned long mask[16]; memset(mask, 0, sizeof(mask)); mask[node / 64] |= 1UL << (node % 64);
	if (syscall(SYS_mbind, m, bytes, 2 /* MPOL_BIND */, mask, 1024UL, 0U) != 0) { (void)munmap(m, bytes); return nullptr; }
	const long ps = sysconf(_SC_PAGESIZE);
	for (size_t o = 0; o < bytes; o += (size_t)(ps > 0 ? ps : 4096)) ((volatile unsigned char*)m)[o] = 0;
	if (hipHostRegister(m, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); (void)munmap(m, bytes); return nullptr; }
	return (unsigned char*)m;
}
// the NUMA node a probe found best for a device's delivery arenas, kept for the process (every engine and batch context of the device starts from it)
static int hao_arena_node_of[64] = { -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
	-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };
// GB/s of one device-to-host copy of nb bytes into `host` on the batch's copy stream (HIP events around it); -1 when it cannot be measured
static double hao_arena_rate(hipStream_t st, unsigned char *host, const void *dsrc, size_t nb)
{
	hipEvent_t e0 = nullptr, e1 = nullptr; double r = -1;
	if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
		hipEventRecord(e0, st) == hipSuccess && hipMemcpyAsync(host, dsrc, nb, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess) {
		float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) r = (double)nb / ((double)ms * 1e6);
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	(void)hipGetLastError();
	return r;
}
static inline double hao_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int hao_scan_u32(hao_ctx *c, const uint32_t *in, uint64_t *out, uint64_t n_plus1)
{
	auto it = rocprim::make_transform_iterator(in, U32ToU64());
	return hao_excl_scan_u64(c, it, out, n_plus1);
}

// exact-overlap flags of the batch's final ol->list (device resident, current output set)
static int hao_exact_run(hao_ctx *c)
{
	hao_ctx::Batch &B = *c->batch;
	if (B.exact_valid) return HAO_OK;
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_exact_check needs the bases of the target reads: single-device mode only"); return HAO_EUNSUPP; }
	HIP_TRY(B.O().exact.reserve(B.n_ol + 1));
	if (B.n_ol) {
		hao_exact_args a;
		a.ol = B.O().ol_out.p; a.n_ol = B.n_ol; a.rid_base = c->rid_base; a.packed = c->d_packed.p; a.pk_off = c->d_pk_off.p; a.len = c->d_len.p;
		a.nsite_off = c->has_n ? c->d_nsite_off.p : nullptr; a.nsite = c->has_n ? c->d_nsite.p : nullptr; a.flags = B.O().exact.p;
		hipLaunchKernelGGL(hao_exact_check_kernel, dim3((unsigned)((B.n_ol + 3) / 4)), dim3(256), 0, c->stream, a);
		HAO_CHECK_LAUNCH();
	}
	B.exact_valid = true;
	return HAO_OK;
}

// f3 on the device end to end (hao_grid.cuh): window / candidate pairs of the current batch's final ol->list on the grid, in text order, into c->al_task; then the distance-only
// window alignment over them where they lie (hao_al_ed_resident, hao_f3.hip): results in c->al_res.  No host round trip but the two totals.
int hao_al_ed_resident(hao_ctx *c, uint64_t n_tasks, uint32_t nword);      // (hao_f3.hip)
static int hao_ed_grid_run(hao_ctx *c, uint32_t wl, uint32_t thre, uint64_t *n_tasks)
{
	hao_ctx::Batch &B = *c->batch; *n_tasks = 0; c->al_grid_n = 0;
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_window_ed_grid needs the bases of both reads: single-device mode only"); return HAO_EUNSUPP; }
	if (wl == 0 || thre > HAO_ED_MAX_THRE) { hao_set_err(c, "hao_window_ed_grid: window length 0 or threshold beyond the widest band"); return HAO_EINVAL; }
	const uint64_t n = B.n; const uint32_t nword = (2 * thre + 1 + 63) / 64;
	if (n == 0 || B.n_ol == 0) return HAO_OK;
	DevBuf<uint64_t> &nwin = c->al_k1, &wbase = c->al_k2;      // (the upload path's key buffers: free here)
	HIP_TRY(nwin.reserve(n + 2)); HIP_TRY(wbase.reserve(n + 2));
	hipLaunchKernelGGL(ed_grid_nwin_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream, c->d_len.p, B.lo, n, wl, nwin.p); HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, nwin.p, wbase.p, n + 1)) return rc;
	uint64_t W = 0; HIP_TRY(hipMemcpyAsync(&W, wbase.p + n, 8, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream));
	DevBuf<uint64_t> &cnt = c->al_path, off; HIP_TRY(cnt.reserve(W + 2)); HIP_TRY(off.reserve(W + 2));
	HIP_TRY(hipMemsetAsync(cnt.p + W, 0, 8, c->stream));
	const dim3 g_((unsigned)((n + 3) / 4)), b_(256);
	hipLaunchKernelGGL((ed_grid_kernel<false>), g_, b_, 0, c->stream, B.O().ol_out.p, B.O().fin_off.p, c->d_len.p, B.lo, n, wl, thre, nword, wbase.p, cnt.p, (hao_ed_task_t*)nullptr); HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, cnt.p, off.p, W + 1)) return rc;
	uint64_t T = 0; HIP_TRY(hipMemcpyAsync(&T, off.p + W, 8, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream));
	if (T >= (1ULL << 32)) { hao_set_err(c, "hao_window_ed_grid: more than 2^32 pairs in one batch"); return HAO_EUNSUPP; }
	HIP_TRY(c->al_task.reserve(T + 1));
	if (T) { hipLaunchKernelGGL((ed_grid_kernel<true>), g_, b_, 0, c->stream, B.O().ol_out.p, B.O().fin_off.p, c->d_len.p, B.lo, n, wl, thre, nword, wbase.p, off.p, c->al_task.p); HAO_CHECK_LAUNCH(); }
	HIP_TRY(hipStreamSynchronize(c->stream));      // (off is a local buffer)
	off.release();
	*n_tasks = T; c->al_grid_n = T;
	return T ? hao_al_ed_resident(c, T, nword) : HAO_OK;
}

// Queue the copy of the current batch's results into the slot's pinned arena (copy stream, after everything on the compute stream so far).
static int hao_deliver_enqueue(hao_ctx *c)
{
	hao_ctx::Batch &B = *c->batch; const int s = B.cur; hao_ctx::Batch::OutSet &O = B.O(); const uint64_t n = B.n; const uint32_t parts = B.dl_parts;
	auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
	const bool ol = parts & HAO_DELIVER_OL, cl = parts & HAO_DELIVER_CL, ex = parts & HAO_DELIVER_EXACT;
	size_t o_oloff = 0, o_ol = o_oloff + (ol ? al((n + 1) * 8) : 0), o_fcoff = o_ol + (ol ? al(B.n_ol * sizeof(hao_ovlp_wire_t)) : 0), o_fc = o_fcoff + (ol ? al((B.n_ol + 1) * 8) : 0);
	size_t o_choff = o_fc + (ol ? al(B.n_fcw * 4) : 0), o_cloff = o_choff + (cl ? al((n + 1) * 8) : 0), o_qmoff = o_cloff + (cl ? al((n + 1) * 8) : 0), o_hdr = o_qmoff + (cl ? al((n + 1) * 8) : 0);
	const bool q16 = O.qmz16;      // the minimizer tables in 2 + 2 bytes per minimizer (hao_qtab16_kernel) instead of 8
	size_t o_qmz = o_hdr + (cl ? al(B.n_chains * sizeof(hao_chain_hdr_t)) : 0), o_qmc = o_qmz + (cl ? al(B.n_mz * (q16 ? 2 : sizeof(hao_qmz_t))) : 0), o_bits = o_qmc + (cl && q16 ? al(B.n_mz * 2) : 0);
	const uint64_t nw_ = cl ? (B.n_anchor + 63) / 64 : 0;      // 64-position words of the batch's bit stream (positions = seed hits)
	const uint64_t nr4_ = cl ? nw_ / 4 + 1 : 0;      // rank directory entries on the wire: one per 256 positions
	size_t o_rank = o_bits + (cl ? al(nw_ * 8) : 0), o_codes = o_rank + (cl ? al(nr4_ * 4) : 0), o_exc = o_codes + (cl ? al(B.n_codes) : 0);
	size_t o_ex = o_exc + (cl ? al(B.n_exc * sizeof(hao_exc_t)) : 0), total = o_ex + (ex ? al(B.n_ol) : 0);
	if (total > B.arena_cap[s] || B.arena_bad[s]) {
		const bool redo_ = B.arena_bad[s]; B.arena_bad[s] = false;      // (hao_deliver_wait saw this slot's last batch copied at less than 40 GB/s: the probe below tries every NUMA node)
		B.arena_free(s);
		const size_t want = (total + total / 4 + (1 << 20) + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);      // (a multiple of 2 MB: hao_arena_alloc_bound)
		const double t0_ = hao_now();
		const int node_ = c->sw.arena_numa ? hao_gpu_numa_node(c->device) : -1;
		// MPOL_BIND first: "preferred" silently falls over to the far socket when the GPU's node is short of FREE pages (a process that has just generated or
		// parsed gigabytes of reads leaves it full of page cache) - the same box then delivers at 36 instead of 52 GB/s; bound, the kernel reclaims instead.
		// If the bound allocation fails, once more with the preference only.
		hipError_t he_ = hipErrorOutOfMemory; const char *how_ = "default policy";
		if (B.arena_node >= 0) if (unsigned char *m_ = hao_arena_alloc_bound(want, B.arena_node)) { B.arena[s] = m_; B.arena_reg[s] = true; he_ = hipSuccess; how_ = "by hand on the node an earlier probe chose"; }
		if (he_ != hipSuccess && node_ >= 0 && c->sw.arena_numa != 1) {
			hao_mempolicy_guard g_(node_, 2 /* MPOL_BIND */);
			if (g_.applied) {
				he_ = hipHostMalloc((void**)&B.arena[s], want, (c->sw.arena_numa == 2) ? hipHostMallocNumaUser : hipHostMallocDefault);
				if (he_ != hipSuccess) { B.arena[s] = nullptr; (void)hipGetLastError(); } else how_ = "bound";
			}
		}
		if (he_ != hipSuccess) {
			hao_mempolicy_guard g_(node_, 1 /* MPOL_PREFERRED */);
			he_ = hipHostMalloc((void**)&B.arena[s], want, (c->sw.arena_numa == 2 && g_.applied) ? hipHostMallocNumaUser : hipHostMallocDefault);
			if (he_ == hipSuccess && g_.applied) how_ = "preferred";
		}
		// where did the pages land?  hipHostMalloc does not always honour the calling thread's policy (one run of round 6 delivered configs[2] at 28.6 GB/s and, with
		// arenas allocated later in the same process, at 53.5): if fewer than 28 of 32 sampled pages are on the GPU's node, the arena is allocated again by hand
		int on_ = -1;
		if (he_ == hipSuccess && node_ >= 0) {
			on_ = hao_pages_on_node(B.arena[s], want, node_);
			if (!B.arena_reg[s] && ((on_ >= 0 && on_ < 28) || c->sw.arena_numa == 4)) {      // (HAO_ARENA_NUMA=4: always by hand - tests)
				if (unsigned char *m_ = hao_arena_alloc_bound(want, node_)) { (void)hipHostFree(B.arena[s]); B.arena[s] = m_; B.arena_reg[s] = true; how_ = "mmap + mbind + hipHostRegister"; on_ = hao_pages_on_node(m_, want, node_); }
			}
		}
		// What the placement is worth is MEASURED: one run in eight of round 6 still delivered at 28.7 instead of 50 GB/s (same box, next process: 49.8) with every page
		// reported on the GPU's node.  A 128 MB copy into the new arena is timed; below 50 GB/s a 128 MB buffer bound to each NUMA node in turn gets the same copy and the
		// arena moves to the best node when that is 10 % faster.  (Arenas of 64 MB and more; HAO_DBG_TEST=arena_probe=1: always and whatever the size - tests.)
		if (he_ == hipSuccess && c->sw.arena_numa && (want >= ((size_t)64 << 20) || c->sw.arena_probe) && B.hits.p) {
			const size_t nb = std::min<size_t>(std::min<size_t>(want, (size_t)128 << 20), B.hits.cap * sizeof(hao_hit_t)) & ~(size_t)4095;
			if (nb >= 4096) {
				(void)hao_arena_rate(B.copy_stream, B.arena[s], B.hits.p, nb);      // (first touch of the mapping)
				const double r0 = hao_arena_rate(B.copy_stream, B.arena[s], B.hits.p, nb);
				if ((r0 >= 0 && r0 < 50.0) || c->sw.arena_probe || redo_) {      // (a good arena: 55 - 57 GB/s with the device otherwise idle, as it is here)
					int best_k = -1; double best = r0;
					for (int k = 0; k < 16; ++k) {
						unsigned char *m_ = hao_arena_alloc_bound(nb, k); if (!m_) continue;
						(void)hao_arena_rate(B.copy_stream, m_, B.hits.p, nb);
						const double rk = hao_arena_rate(B.copy_stream, m_, B.hits.p, nb);
						(void)hipHostUnregister(m_); (void)munmap(m_, nb);
						if (c->sw.dltime || c->sw.arena_probe) fprintf(stderr, "[deliver] arena %d probe: NUMA node %d %.1f GB/s\n", s, k, rk);
						if (rk > best * 1.1) { best = rk; best_k = k; }
					}
					if (best_k >= 0) if (unsigned char *m_ = hao_arena_alloc_bound(want, best_k)) { B.arena_cap[s] = want; B.arena_free(s); B.arena[s] = m_; B.arena_reg[s] = true; B.arena_node = best_k; if (c->device >= 0 && c->device < 64) hao_arena_node_of[c->device] = best_k; how_ = "moved after the probe"; }      // (arena_free unmaps arena_cap bytes of a registered arena)
					fprintf(stderr, "[hao] delivery arena %d: %.1f GB/s from the device as allocated (GPU NUMA node %d, %s)%s\n", s, r0, node_, how_, best_k >= 0 ? "" : "; no NUMA node does better");
					if (best_k >= 0) fprintf(stderr, "[hao] delivery arena %d: moved to NUMA node %d (%.1f GB/s)\n", s, best_k, best);
				}
			}
		}
		if (c->sw.dltime) fprintf(stderr, "[deliver] arena %d: %zu MB, GPU NUMA node %d (requested mode %d, allocated %s, %d of 32 sampled pages on the node)\n", s, want >> 20, node_, c->sw.arena_numa, he_ == hipSuccess ? how_ : "FAILED", on_);
		HIP_TRY(he_);
		B.arena_cap[s] = want; B.t_alloc += hao_now() - t0_;
	}
	unsigned char *a = B.arena[s];
	HIP_TRY(hipEventRecord(B.ev_ready[s], c->stream));
	HIP_TRY(hipStreamWaitEvent(B.copy_stream, B.ev_ready[s], 0));
	HIP_TRY(hipEventRecord(B.ev_cstart[s], B.copy_stream));      // (the copy itself, without the wait behind the previous batch's: hao_deliver_wait checks its rate)
	auto cp = [&](size_t off, const void *src, size_t bytes) -> hipError_t { return bytes ? hipMemcpyAsync(a + off, src, bytes, hipMemcpyDeviceToHost, B.copy_stream) : hipSuccess; };
	hao_delivery_t &d = B.dl[s];
	d.rid_lo = B.lo; d.n_reads = n; d.n_ol = d.n_fc = d.n_chains = d.n_cl = d.n_exc = d.n_codes = d.n_pos = 0; d.bytes = 0;
	if (ol && n) {
		HIP_TRY(cp(o_oloff, O.fin_off.p, (n + 1) * 8)); HIP_TRY(cp(o_ol, O.ol_wire.p, B.n_ol * sizeof(hao_ovlp_wire_t)));
		HIP_TRY(cp(o_fcoff, O.fcw_off.p, B.n_ol * 8)); HIP_TRY(cp(o_fc, O.fcw.p, B.n_fcw * 4));
		((uint64_t*)(a + o_fcoff))[B.n_ol] = B.n_fcw;      // end of the last cigar (a host-side word next to, not inside, the region the copy writes)
		d.n_ol = B.n_ol; d.n_fc = B.n_fcw; d.ol_off = (const uint64_t*)(a + o_oloff); d.ol = (const hao_ovlp_wire_t*)(a + o_ol); d.fc_off = (const uint64_t*)(a + o_fcoff); d.fc = (const uint32_t*)(a + o_fc);
		d.bytes += (n + 1) * 8 + B.n_ol * (sizeof(hao_ovlp_wire_t) + 8) + B.n_fcw * 4;
	}
	if (cl && n) {
		HIP_TRY(cp(o_choff, O.ch_off.p, (n + 1) * 8)); HIP_TRY(cp(o_cloff, O.cl_off.p, (n + 1) * 8)); HIP_TRY(cp(o_qmoff, O.qm_off.p, (n + 1) * 8));
		HIP_TRY(cp(o_hdr, O.hdr.p, B.n_chains * sizeof(hao_chain_hdr_t))); if (q16) { HIP_TRY(cp(o_qmz, O.qmz_pos.p, B.n_mz * 2)); HIP_TRY(cp(o_qmc, O.qmz_cnt.p, B.n_mz * 2)); } else HIP_TRY(cp(o_qmz, O.qmz.p, B.n_mz * sizeof(hao_qmz_t))); HIP_TRY(cp(o_exc, O.exc.p, B.n_exc * sizeof(hao_exc_t)));
		HIP_TRY(cp(o_bits, O.bits.p, nw_ * 8)); HIP_TRY(cp(o_rank, O.rank4.p, nr4_ * 4));
		HIP_TRY(cp(o_codes, O.codes.p, B.n_codes));
		d.n_chains = B.n_chains; d.n_cl = B.n_cl; d.n_exc = B.n_exc; d.n_codes = B.n_codes; d.n_pos = B.n_anchor; d.ch_off = (const uint64_t*)(a + o_choff); d.cl_off = (const uint64_t*)(a + o_cloff); d.qm_off = (const uint64_t*)(a + o_qmoff);
		d.chains = (const hao_chain_hdr_t*)(a + o_hdr); d.qmz = q16 ? nullptr : (const hao_qmz_t*)(a + o_qmz); d.qmz_pos = q16 ? (const uint16_t*)(a + o_qmz) : nullptr; d.qmz_cnt = q16 ? (const uint16_t*)(a + o_qmc) : nullptr; d.cl_bits = (const uint64_t*)(a + o_bits); d.cl_rank = (const uint32_t*)(a + o_rank); d.cl_codes = a + o_codes; d.cl_exc = (const hao_exc_t*)(a + o_exc);
		d.bytes += 3 * (n + 1) * 8 + B.n_chains * sizeof(hao_chain_hdr_t) + B.n_mz * (q16 ? 4 : sizeof(hao_qmz_t)) + nw_ * 8 + nr4_ * 4 + B.n_codes + B.n_exc * sizeof(hao_exc_t);
	}
	if (ex && n) { HIP_TRY(cp(o_ex, O.exact.p, B.n_ol)); d.exact = a + o_ex; d.n_ol = B.n_ol; d.bytes += B.n_ol; }
	HIP_TRY(hipEventRecord(B.ev_done[s], B.copy_stream));
	B.dl_pending[s] = true;
	return HAO_OK;
}

static int hao_deliver_init(hao_ctx *c, hao_ctx::Batch &B)
{
	if (B.dl_ready) return HAO_OK;
	HIP_TRY(hipStreamCreateWithFlags(&B.copy_stream, hipStreamNonBlocking));
	for (int x = 0; x < 2; ++x) { HIP_TRY(hipEventCreate(&B.ev_ready[x])); HIP_TRY(hipEventCreate(&B.ev_done[x])); HIP_TRY(hipEventCreate(&B.ev_cstart[x])); }
	if (c->device >= 0 && c->device < 64) B.arena_node = hao_arena_node_of[c->device];
	B.dl_ready = true;
	return HAO_OK;
}

// parts = 0: results stay in HBM (blocking API).  parts != 0 (hao_overlap_batch_async): the batch computes into output set `dl_seq & 1`, packs cl->list
// into the wire format and queues the copy of everything asked for into that slot's pinned arena on the copy stream.
static int hao_overlap_run(hao_ctx *c, uint64_t lo, uint64_t hi, const hao_pass_t &ps, uint32_t parts = 0, int *slot_out = nullptr)
{
	if (ps.apend_be != 1 || ps.is_accurate != 1 || ps.gen_off != 1 || ps.mcopy_num > HAO_MCOPY_MAX || ps.ocv_w == 0) { hao_set_err(c, "unsupported h_ec_lchain arguments"); return HAO_EUNSUPP; }
	if (int rc = hao_view_refresh(c)) return rc;
	if (!c->has_pt) { hao_set_err(c, "hao_pt_gen must run before hao_overlap_batch"); return HAO_EINVAL; }
	if (!c->batch) c->batch = new hao_ctx::Batch();
	hao_ctx::Batch &B = *c->batch; const double t_run0 = hao_now();
	c->al_grid_n = 0;      // (the window-alignment grid of the previous batch's overlaps is stale)
	B.valid = false; B.host_valid = false; B.cl_valid = false; B.exact_valid = false; B.h_exact.clear(); B.lo = lo; B.n = hi - lo; B.dl_parts = parts; B.n_exc = 0;
	const uint64_t n = B.n;
	if (parts) {
		if (int rc = hao_deliver_init(c, B)) return rc;
		B.cur = (int)(B.dl_seq++ & 1);
		if (slot_out) *slot_out = B.cur;
	}
	// the output set about to be written may still be feeding a copy (its previous async batch): wait for that copy, never for the other slot's
	if (B.dl_ready && B.dl_pending[B.cur]) { const double t0_ = hao_now(); HIP_TRY(hipEventSynchronize(B.ev_done[B.cur])); B.dl_pending[B.cur] = false; B.t_evsync += hao_now() - t0_; }
	if (parts) { memset(&B.dl[B.cur], 0, sizeof(hao_delivery_t)); B.dl[B.cur].rid_lo = lo; B.dl[B.cur].n_reads = n; }
	if (n == 0) { B.n_anchor = B.n_groups = B.n_chains = B.n_cl = B.n_ol = B.n_fc = B.n_fcw = B.n_mz = 0; B.valid = true; return HAO_OK; }      // (an empty delivery: nothing to copy, the view stays zeroed)
	// minimizer range of the batch (host knows the per-read offsets? keep a host copy once)
	if (c->h_ix_mz_off.size() != c->n_reads + 1) {
		c->h_ix_mz_off.resize(c->n_reads + 1);
		HIP_TRY(hipMemcpy(c->h_ix_mz_off.data(), c->d_ix_mz_off.p, (c->n_reads + 1) * 8, hipMemcpyDeviceToHost));
	}
	const uint64_t glo = c->rid_base + lo;      // global read id of the first query (== lo when unsharded); minimizer arrays are indexed locally
	B.mz0 = c->h_ix_mz_off[lo]; B.n_mz = c->h_ix_mz_off[hi] - B.mz0;
	const uint64_t nm = B.n_mz;
	// (no host-to-device copies inside a batch: with the delivery path's bulk copy of the previous batch in flight they would queue behind it on the DMA engines)
	if (B.wgt_hi != ps.high_occ || B.wgt_lo != ps.low_occ || !B.wgt.p) {      // seed weights depend on the pass's occurrence thresholds only: uploaded when those change
		std::vector<uint32_t> wt; hao_seed_weight_table(ps.high_occ, ps.low_occ, wt);
		HIP_TRY(B.wgt.reserve(4096)); HIP_TRY(hipMemcpy(B.wgt.p, wt.data(), 4096 * 4, hipMemcpyHostToDevice));
		B.wgt_hi = ps.high_occ; B.wgt_lo = ps.low_occ; B.wgt_max = *std::max_element(wt.begin(), wt.end());
	}
	HIP_TRY(B.q_pos.reserve(nm + 1)); HIP_TRY(B.q_cnt.reserve(nm + 1));
	HIP_TRY(B.s_start.reserve(nm + 1)); HIP_TRY(B.s_pk.reserve(nm + 1)); HIP_TRY(B.s_n.reserve(nm + 1)); HIP_TRY(B.a_off.reserve(nm + 2)); HIP_TRY(B.seg.reserve(n + 2));
	HIP_TRY(c->d_err.reserve(2)); HIP_TRY(hipMemsetAsync(c->d_err.p, 0, 4, c->stream));
	// Q1: every minimizer's lookup result was computed when the index was built (hao_index_finish_kernel; in sharded mode by the owner of its hash, hao_tables.hpp)
	if (!c->lk_valid) { hao_set_err(c, "index without per-minimizer lookup results"); return HAO_EINVAL; }
	hipLaunchKernelGGL(seed_unpack_kernel, dim3((unsigned)((nm + 256) / 256)), dim3(256), 0, c->stream, c->d_ix_lk.p, c->d_ix_mz_info.p, B.mz0, nm, B.wgt.p, B.s_start.p, B.s_n.p, B.q_pos.p, B.q_cnt.p, B.s_pk.p);
	HAO_CHECK_LAUNCH();
	if (int rc = hao_scan_u32(c, B.s_n.p, B.a_off.p, nm + 1)) return rc;
	hipLaunchKernelGGL(seed_segments_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream, c->d_ix_mz_off.p, lo, n, B.mz0, B.a_off.p, B.seg.p);
	HAO_CHECK_LAUNCH();
	hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)(B.a_off.p + nm), 1, c->peek_d);
	HAO_CHECK_LAUNCH();
	const double ts0_ = hao_now();
	HIP_TRY(hipStreamSynchronize(c->stream));
	B.t_s1 += hao_now() - ts0_; B.t_pre += ts0_ - t_run0;
	B.n_anchor = c->peek_h[0];
	c->timer.mark("q_lookup");
	const uint64_t A = B.n_anchor;
	if (A >= (1ULL << 32)) { hao_set_err(c, "batch produces >= 2^32 anchors: use a smaller read range"); return HAO_EUNSUPP; }
	HIP_TRY(B.hits.reserve(A + 1));
	uint64_t max_q = 1;
	for (uint64_t r = lo; r < hi; ++r) max_q = std::max<uint64_t>(max_q, c->h_ix_mz_off[r + 1] - c->h_ix_mz_off[r]);
	const uint64_t sum_q = c->h_ix_mz_off[hi] - c->h_ix_mz_off[lo];      // minimizers of the batch's reads
	int tb = 1; while ((1ULL << tb) < c->n_total) ++tb;
	HIP_TRY(B.g_cnt.reserve(n + 2)); HIP_TRY(B.g_off.reserve(n + 2)); HIP_TRY(B.g_tmp.reserve(A + 1));
	HIP_TRY(B.stats.reserve(3 * HAO_NCLS + 7)); HIP_TRY(hipMemsetAsync(B.stats.p, 0, (3 * HAO_NCLS + 7) * 8, c->stream));
	unsigned long long *d_slow_cnt = B.stats.p, *d_cls_cnt = B.stats.p + HAO_NCLS + 4;   // [0..NCLS] slow groups per class + their hits
	{
		// Q2-Q5 in one kernel: index records -> bins -> sorted k_mer_hits + group lists (no anchor keys in memory)
		hao_seed_args sa_; B.seed_path = 0;
		sa_.mz_off = c->d_ix_mz_off.p; sa_.mz_info = c->d_ix_mz_info.p; sa_.rid_lo = lo; sa_.mz0 = B.mz0; sa_.s_start = B.s_start.p; sa_.s_n = B.s_n.p; sa_.a_off = B.a_off.p; sa_.seg = B.seg.p;
		sa_.sinfo = c->d_ix_sinfo.p; sa_.len = c->d_len_all.p; sa_.q_pos = B.q_pos.p; sa_.q_cnt = B.q_cnt.p; sa_.hits = B.hits.p; sa_.g_tmp = B.g_tmp.p; sa_.g_cnt = B.g_cnt.p; sa_.n_sel = n; sa_.tb = tb;
		sa_.qcap = (uint32_t)std::min<uint64_t>((max_q + 63) & ~63ULL, HAO_QTAB_CAP);
		sa_.dbg = nullptr; sa_.hq = nullptr;
		if (parts & HAO_DELIVER_CL) {      // the wire format's code array (one byte per seed hit, 0x08 = nothing to say) and, for the quick check's codes, every hit's minimizer index
			HIP_TRY(B.hcode.reserve(A + 64));
			// (the quick-check kernels write every position of every group they see; only the debug paths that bypass them need the array pre-filled)
			if (c->sw.seq_chain || c->sw.dp_seqtail || c->sw.dp_nospec) { const uint64_t n16 = (A + 31) / 16; hipLaunchKernelGGL(hao_fill16_kernel, dim3((unsigned)std::min<uint64_t>((n16 + 255) / 256, 1u << 14)), dim3(256), 0, c->stream, (hao_fill_v4*)B.hcode.p, n16, 0x08080808u); HAO_CHECK_LAUNCH(); }
			HIP_TRY(B.hq.reserve(A + 64)); sa_.hq = B.hq.p;
		}
		if (c->sw.seedphase) { HIP_TRY(B.dbgbuf.reserve(64)); HIP_TRY(hipMemsetAsync(B.dbgbuf.p, 0, 512, c->stream)); sa_.dbg = B.dbgbuf.p; }
		HIP_TRY(B.ovf_list.reserve(3 * (n + 1)));
		unsigned long long *d_ovf = B.stats.p + 3 * HAO_NCLS + 3, *d_ovf2 = B.stats.p + 3 * HAO_NCLS + 4, *d_ovf0 = B.stats.p + 3 * HAO_NCLS + 5;
		uint32_t *ovf1 = B.ovf_list.p, *ovf2 = B.ovf_list.p + (n + 1), *ovf0 = B.ovf_list.p + 2 * (n + 1);
		const size_t lds_tile = std::max<size_t>((size_t)512 * (sizeof(hao_stage_t) + 4), 12 * 512);      // staged tile (>= the 12 bytes per slot of the bin sort it shares memory with)
		const size_t lds_q = 12 * (size_t)sa_.qcap + 16;
		size_t lds1 = (size_t)22 * 512 + lds_tile + lds_q, lds2 = (size_t)22 * 1024 + std::max<size_t>(lds_tile, 12 * 1024) + lds_q,
			   lds3 = (size_t)22 * 2048 + std::max<size_t>(lds_tile, 12 * 2048) + lds_q;      // (third launch: 2048 slots = up to 1760 bins per id-range round)
		auto launch = [&](auto k1, auto k2, auto k3) -> int {
			{     // beyond the default dynamic LDS limit: opt in (the CU has 160 KB)
				HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
				HIP_TRY(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
				HIP_TRY(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
			}
			hipLaunchKernelGGL(k1, dim3((unsigned)n), dim3(256), lds1, c->stream, sa_, (const uint32_t*)nullptr, (const unsigned long long*)nullptr, ovf1, d_ovf);
			HAO_CHECK_LAUNCH();
			hipLaunchKernelGGL(k2, dim3((unsigned)n), dim3(256), lds2, c->stream, sa_, (const uint32_t*)ovf1, (const unsigned long long*)d_ovf, ovf2, d_ovf2);
			HAO_CHECK_LAUNCH();
			hipLaunchKernelGGL(k3, dim3((unsigned)n), dim3(256), lds3, c->stream, sa_, (const uint32_t*)ovf2, (const unsigned long long*)d_ovf2, (uint32_t*)nullptr, (unsigned long long*)nullptr);
			HAO_CHECK_LAUNCH();
			return HAO_OK;
		};
		if (max_q <= HAO_QTAB_CAP && !c->sw.seed_noql && c->sw.seed_lds && (double)A * 100.0 <= (double)c->sw.seed_lds_ratio * (double)sum_q * (double)std::max(1, c->hom_cov) && c->n_total < HAO_L5_MASK && c->ix_n_pos + c->sw.ix_pad < (1ULL << 40)) {
			B.seed_path = 2;
			// the list-major kernel (hao_query5.cuh): one persistent workgroup per CU, a read's position lists read once with adjacent lanes on adjacent records into LDS,
			// merged by target there.  The reads it leaves (more than 1536 minimizers, more records than the LDS holds, more than seed_merge_maxn hits) go through the
			// table kernels below it (512-slot launch over the overflow list, then the launches without staged tiles).  Batches of reads across repeat families - hundreds of
			// targets per read, a merge step each - keep the table kernels (341 against 226 ms per pass of the repeat-rich 250 Mb set, profiles/r06/seed_ab.txt).  What tells
			// them apart whatever the coverage: seed hits per (query minimizer x coverage peak) - 1.0 on repeat-free reads at 30 x and at 40 x (every minimizer meets the reads
			// that cover it), 0.5 on ONT reads (1 % error), 1.4 on the repeat-rich set; the list-major kernel takes batches up to seed_lds_ratio = 1.2.
			lds2 = hao_seed3_lds<10>::FIXED + lds_q; lds3 = hao_seed3_lds<11>::FIXED + lds_q;
			auto k1 = seed_bin_kernel<9, 1, 512, true>; auto k2 = seed_bin3_kernel<10, 1, 4>; auto k3 = seed_bin3_kernel<11, 2, 4>;
			HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
			HIP_TRY(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
			HIP_TRY(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
			{
				const uint64_t *sinfo_ = c->d_ix_sinfo.p; const uint32_t *len_ = c->d_len_all.p; const uint64_t *spk_ = B.s_pk.p;
				const unsigned g_ = (unsigned)std::min<uint64_t>(n, (uint64_t)c->n_cu);
				const bool b16_ = c->max_len_all < 65536, wide_ = max_q > 2 * HAO_L5_THREADS;      // offsets of the staged records in 16 bits; reads with more than 1024 minimizers: three per thread
				auto go_ = [&](auto k0, size_t lds_) -> int {
					HIP_TRY(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));
					hipLaunchKernelGGL(k0, dim3(g_), dim3(HAO_L5_THREADS), lds_, c->stream, sa_, sinfo_, len_, spk_, (uint32_t)c->sw.seed_merge_maxn, 176u /* wave 0's share in 1/1024: profiles/r06/seed_ab.txt */, ovf0, d_ovf0, B.stats.p + 3 * HAO_NCLS + 6 /* the kernel's read cursor: zero */);
					return HAO_OK;
				};
				int rc_;
				if (c->sw.seedphase && b16_ && !wide_) rc_ = go_(seed_lds_kernel<true, 2, 14, true>, hao_l5_lds<true, 2>::TOTAL);
				else if (b16_) rc_ = wide_ ? go_(seed_lds_kernel<true, 3, 8, false>, hao_l5_lds<true, 2>::TOTAL) : go_(seed_lds_kernel<true, 2, 16, false>, hao_l5_lds<true, 2>::TOTAL);
				else rc_ = wide_ ? go_(seed_lds_kernel<false, 3, 8, false>, hao_l5_lds<false, 2>::TOTAL) : go_(seed_lds_kernel<false, 2, 16, false>, hao_l5_lds<false, 2>::TOTAL);
				if (rc_) return rc_;
			}
			HAO_CHECK_LAUNCH();
			hipLaunchKernelGGL(k1, dim3((unsigned)n), dim3(256), lds1, c->stream, sa_, (const uint32_t*)ovf0, (const unsigned long long*)d_ovf0, ovf1, d_ovf);
			HAO_CHECK_LAUNCH();
			hipLaunchKernelGGL(k2, dim3((unsigned)n), dim3(256), lds2, c->stream, sa_, (const uint32_t*)ovf1, (const unsigned long long*)d_ovf, ovf2, d_ovf2);
			HAO_CHECK_LAUNCH();
			hipLaunchKernelGGL(k3, dim3((unsigned)n), dim3(256), lds3, c->stream, sa_, (const uint32_t*)ovf2, (const unsigned long long*)d_ovf2, (uint32_t*)nullptr, (unsigned long long*)nullptr);
			HAO_CHECK_LAUNCH();
		}
		else if (max_q <= HAO_QTAB_CAP && !c->sw.seed_noql) {      // every read's minimizer table fits the LDS; reads whose bins overflow the 512-slot table: the launches without staged tiles (hao_query3.cuh)
			lds2 = hao_seed3_lds<10>::FIXED + lds_q; lds3 = hao_seed3_lds<11>::FIXED + lds_q;
			if (int rc = launch(seed_bin_kernel<9, 0, 512, true>, seed_bin3_kernel<10, 1, 4>, seed_bin3_kernel<11, 2, 4>)) return rc;
		}
		else if (int rc = launch(seed_bin_kernel<9, 0, 512, false>, seed_bin_kernel<10, 1, 512, false>, seed_bin_kernel<11, 2, 512, false>)) return rc;
	}
	if (c->sw.seedphase) { unsigned long long d_[64]; HIP_TRY(hipMemcpy(d_, B.dbgbuf.p, 512, hipMemcpyDeviceToHost));
		if (d_[9]) { fprintf(stderr, "[seed lds] merge us per read, waves 0 - 7:"); for (int w_ = 0; w_ < 8; ++w_) fprintf(stderr, " %.2f", d_[16 + w_] / 100.0 / d_[9]); fprintf(stderr, "\n"); }
		if (d_[9]) fprintf(stderr, "[seed lds] reads %llu  avg us per read (wave 0): stage %.2f  barrier %.2f  prepare %.2f  barrier %.2f  issue loads %.2f  splitters %.2f  merge %.2f  barrier + groups %.2f  barrier %.2f\n", d_[9],
			d_[0] / 100.0 / d_[9], d_[1] / 100.0 / d_[9], d_[2] / 100.0 / d_[9], d_[3] / 100.0 / d_[9], d_[4] / 100.0 / d_[9], d_[5] / 100.0 / d_[9], d_[6] / 100.0 / d_[9], d_[7] / 100.0 / d_[9], d_[8] / 100.0 / d_[9]);
		else if (d_[3]) fprintf(stderr, "[seed] blocks %llu  avg us: count pass %.1f  sort+scan %.1f  scatter pass %.1f\n", d_[3], d_[0] / 100.0 / d_[3], d_[1] / 100.0 / d_[3], d_[2] / 100.0 / d_[3]); }
	c->timer.mark("q_sort_bins");
	HIP_TRY(B.cls_cc.reserve(HAO_NCLS * (n + 1) + 1)); HIP_TRY(B.cls_co.reserve(HAO_NCLS * (n + 1) + 1));
	hipLaunchKernelGGL(groups_classify_kernel, dim3((unsigned)((n + 4) / 4)), dim3(256), 0, c->stream, B.g_tmp.p, B.seg.p, B.g_cnt.p, n, B.cls_cc.p);
	HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, B.cls_cc.p, B.cls_co.p, HAO_NCLS * (n + 1))) return rc;
	hipLaunchKernelGGL(groups_layout_kernel, dim3(1), dim3(64), 0, c->stream, B.cls_co.p, n, d_cls_cnt);
	HAO_CHECK_LAUNCH();
	unsigned long long lay[HAO_NCLS + 1];
	hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)d_cls_cnt, HAO_NCLS + 1, c->peek_d);
	HAO_CHECK_LAUNCH();
	const double ts1_ = hao_now();
	HIP_TRY(hipStreamSynchronize(c->stream));
	B.t_s2 += hao_now() - ts1_;
	for (int x = 0; x <= HAO_NCLS; ++x) lay[x] = c->peek_h[x];
	const uint64_t G = B.n_groups = lay[HAO_NCLS];
	hao_cls_layout L; unsigned long long cls_cnt[HAO_NCLS];
	for (int x = 0; x <= HAO_NCLS; ++x) L.base[x] = lay[x];
	for (int x = 0; x < HAO_NCLS; ++x) cls_cnt[x] = L.base[x + 1] - L.base[x];
	HIP_TRY(B.g_start.reserve(G + 1)); HIP_TRY(B.g_read.reserve(G + 1)); HIP_TRY(B.g_cls.reserve(G + 1)); HIP_TRY(B.glist.reserve(G + 1)); HIP_TRY(B.slow.reserve(G + 1));
	hipLaunchKernelGGL(groups_compact_kernel, dim3((unsigned)((n + 4) / 4)), dim3(256), 0, c->stream, B.g_tmp.p, B.seg.p, B.cls_co.p, n, glo, c->d_len_all.p, B.g_off.p, B.g_start.p, B.g_read.p, B.g_cls.p, B.glist.p);
	HAO_CHECK_LAUNCH();
	c->timer.mark("q_groups");
	// Q6 chain: quick check per size class, biggest first; the groups it does not settle go to the DP kernel of their class on a side
	// stream, so the long sequential DPs of big groups run under the quick checks of the smaller classes
	HIP_TRY(B.ohits.reserve(A + 1));
	HIP_TRY(B.fcs.reserve(A + 6 * G + 1)); HIP_TRY(B.rec.reserve(G * HAO_MCOPY_MAX + 1)); HIP_TRY(B.nch.reserve(G + 2)); HIP_TRY(B.nout.reserve(G + 2));
	hao_chain_par par = hao_chain_params(c->opt.k, ps);
	if (G) {
		hao_chain_args ca;
		ca.hits = B.hits.p; ca.g_start = B.g_start.p; ca.g_read = B.g_read.p; ca.g_off = B.g_off.p; ca.seg = B.seg.p; ca.n_groups = G; ca.rid_lo = glo; ca.len = c->d_len_all.p; ca.par = par;
		ca.dbg_qc = nullptr; ca.hq = nullptr; ca.hcode = nullptr; ca.ohq = nullptr; ca.exc_every = (uint32_t)c->sw.exc_every;
		if (parts & HAO_DELIVER_CL) { HIP_TRY(B.ohq.reserve(A + 64)); ca.hq = B.hq.p; ca.hcode = B.hcode.p; ca.ohq = B.ohq.p; }
		if (c->sw.qcphase) { HIP_TRY(B.dbgbuf.reserve(8)); HIP_TRY(hipMemsetAsync(B.dbgbuf.p, 0, 64, c->stream)); ca.dbg_qc = B.dbgbuf.p; }
		ca.stats = d_slow_cnt; ca.dbg_stats = c->sw.dp_stats ? 1 : 0;
		ca.dbg_seq = c->sw.seq_chain ? 1 : (c->sw.dp_seqtail ? 3 : (c->sw.dp_nospec ? 4 : 0));
		// per-hit DP scratch in global memory is only touched by groups beyond the LDS variants (and the sequential debug path)
		const bool need_scratch = cls_cnt[HAO_NCLS - 1] > 0 || ca.dbg_seq == 1;
		if (need_scratch) { HIP_TRY(B.f.reserve(A + 1)); HIP_TRY(B.ii.reserve(A + 1)); HIP_TRY(B.p.reserve(A + 1)); HIP_TRY(B.t.reserve(A + 1)); HIP_TRY(B.tm.reserve(A + 1)); }
		ca.tm = B.tm.p; ca.f = B.f.p; ca.ii = B.ii.p; ca.p = B.p.p; ca.t = B.t.p; ca.ohits = B.ohits.p; ca.fcs = B.fcs.p; ca.rec = B.rec.p; ca.nch = B.nch.p; ca.nout = B.nout.p;
		if (!B.side_ready) {
			int plo_ = 0, phi_ = 0; (void)hipDeviceGetStreamPriorityRange(&plo_, &phi_);      // same priority class as the engine's stream (see hao_create)
			for (int x = 0; x < HAO_NCLS; ++x) { HIP_TRY(hipStreamCreateWithPriority(&B.side[x], hipStreamNonBlocking, plo_)); HIP_TRY(hipEventCreateWithFlags(&B.ev_qc[x], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&B.ev_dp[x], hipEventDisableTiming)); }
			B.side_ready = true;
		}
		const bool serial = c->sw.dp_serial;      // DP kernels on the main stream (no overlap), for A/B timing
		const int spec_min = 2;      // tiles of <= 64 hits gain nothing from speculation
		const int dbg_seq0 = ca.dbg_seq;
		for (int x = HAO_NCLS - 1; x >= 1; --x) {
			const uint64_t nl = cls_cnt[x]; if (!nl) continue;
			ca.dbg_seq = (dbg_seq0 == 0 && x < spec_min) ? 4 : dbg_seq0;
			const hao_gent *lst = B.glist.p + L.base[x]; uint32_t *slow = B.slow.p + L.base[x];
			hipLaunchKernelGGL(chain_group_kernel, dim3((unsigned)nl), dim3(64), 0, c->stream, ca, lst, nl, slow, x);
			HAO_CHECK_LAUNCH();
			hipStream_t ds = serial ? c->stream : B.side[x];
			if (!serial) { HIP_TRY(hipEventRecord(B.ev_qc[x], c->stream)); HIP_TRY(hipStreamWaitEvent(ds, B.ev_qc[x], 0)); }
			if (x <= 2) hipLaunchKernelGGL(chain_dp128_kernel, dim3((unsigned)std::min<uint64_t>(nl, 256 * 16)), dim3(64), 0, ds, ca, lst, slow, d_slow_cnt + x);
			else if (x <= 4) hipLaunchKernelGGL((chain_dp_kernel<512, true>), dim3((unsigned)std::min<uint64_t>(nl, 256 * 8)), dim3(64), 0, ds, ca, lst, slow, d_slow_cnt + x);
			else hipLaunchKernelGGL((chain_dp_kernel<HAO_DP_CAP, false>), dim3((unsigned)std::min<uint64_t>(nl, 256 * 3)), dim3(64), 0, ds, ca, lst, slow, d_slow_cnt + x);
			HAO_CHECK_LAUNCH();
			if (!serial) HIP_TRY(hipEventRecord(B.ev_dp[x], ds));
		}
		ca.dbg_seq = dbg_seq0;
		if (cls_cnt[0]) {      // groups of <= HAO_TINY_MAX hits: eight per wave through the data-parallel quick check, the rejected ones by one lane each (HAO_DBG_FORCE=seq_chain: all of them)
			const hao_gent *lst0 = B.glist.p + L.base[0]; uint32_t *slow0 = B.slow.p + L.base[0];
			const unsigned nb_lane = (unsigned)std::min<uint64_t>((cls_cnt[0] + 63) / 64, 256 * 64);
			if (ca.dbg_seq == 1) hipLaunchKernelGGL(chain_tiny_kernel, dim3(nb_lane), dim3(64), 0, c->stream, ca, lst0, (uint64_t)cls_cnt[0], (const uint32_t*)nullptr, (const unsigned long long*)nullptr);
			else {
				hipLaunchKernelGGL(chain_pack8_kernel, dim3((unsigned)((cls_cnt[0] + 31) / 32)), dim3(256), 0, c->stream, ca, lst0, (uint64_t)cls_cnt[0], slow0);
				HAO_CHECK_LAUNCH();
				hipLaunchKernelGGL(chain_tiny_kernel, dim3(nb_lane), dim3(64), 0, c->stream, ca, lst0, (uint64_t)cls_cnt[0], (const uint32_t*)slow0, (const unsigned long long*)d_slow_cnt);
			}
			HAO_CHECK_LAUNCH();
		}
		if (ca.dbg_qc) { unsigned long long d_[5]; HIP_TRY(hipMemcpy(d_, B.dbgbuf.p, 40, hipMemcpyDeviceToHost)); if (d_[3]) fprintf(stderr, "[qc] fast groups %llu (avg %.0f hits)  avg us: entry + tile 0 %.2f  scan loop %.2f  cigar + record %.2f\n", d_[3], (double)d_[4] / d_[3], d_[0] / 100.0 / d_[3], d_[1] / 100.0 / d_[3], d_[2] / 100.0 / d_[3]); }
		c->timer.mark("q_chain");
		if (!serial) for (int x = 1; x < HAO_NCLS; ++x) if (cls_cnt[x]) HIP_TRY(hipStreamWaitEvent(c->stream, B.ev_dp[x], 0));
	}
	HIP_TRY(hipMemsetAsync(B.nch.p + G, 0, 4, c->stream)); HIP_TRY(hipMemsetAsync(B.nout.p + G, 0, 4, c->stream));
	c->timer.mark("q_chain_dp");
	// Q7 assembly
	HIP_TRY(B.ch_base.reserve(G + 2)); HIP_TRY(B.cl_base.reserve(G + 2)); HIP_TRY(B.fc_base.reserve(G * HAO_MCOPY_MAX + 2)); HIP_TRY(B.nch64.reserve(G * HAO_MCOPY_MAX + 2));
	if (int rc = hao_scan_u32(c, B.nch.p, B.ch_base.p, G + 1)) return rc;
	if (int rc = hao_scan_u32(c, B.nout.p, B.cl_base.p, G + 1)) return rc;
	hipLaunchKernelGGL(hao_fclen_kernel, dim3((unsigned)((G * HAO_MCOPY_MAX + 256) / 256)), dim3(256), 0, c->stream, B.rec.p, B.nch.p, G, B.nch64.p);
	HAO_CHECK_LAUNCH();
	if (int rc = hao_excl_scan_u64(c, B.nch64.p, B.fc_base.p, G * HAO_MCOPY_MAX + 1)) return rc;
	// no host round trip here: the buffers downstream are sized by bounds known from G and A (<= 3 chains per group, chained hits <= seed
	// hits, fake-cigar entries <= hits + 6 per group); the exact totals are read back once, after the last kernel
	const uint64_t NCmax = G * HAO_MCOPY_MAX, FCmax = A + 6 * G;
	HIP_TRY(B.ol.reserve(NCmax + 1)); HIP_TRY(B.ol_fc_off.reserve(NCmax + 1)); HIP_TRY(B.cd.reserve(NCmax + 1)); HIP_TRY(B.fc_raw.reserve(FCmax + 1)); HIP_TRY(B.perm.reserve(NCmax + 1));
	if (G) {
		hao_asm_args aa;
		aa.g_start = B.g_start.p; aa.g_read = B.g_read.p; aa.g_cls = B.g_cls.p; aa.g_off = B.g_off.p; aa.n_groups = G; aa.rid_lo = glo; aa.ohits = B.ohits.p; aa.hits = B.hits.p; aa.fcs = B.fcs.p; aa.rec = B.rec.p; aa.nch = B.nch.p;
		aa.ch_base = B.ch_base.p; aa.cl_base = B.cl_base.p; aa.fc_base = B.fc_base.p; aa.ol = B.ol.p; aa.ol_fc_off = B.ol_fc_off.p; aa.cd = B.cd.p; aa.fc = B.fc_raw.p;
		hipLaunchKernelGGL(chain_assemble_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream, aa); HAO_CHECK_LAUNCH();
	}
	unsigned long long *d_exc_cnt = B.stats.p + 3 * HAO_NCLS + 2;      // (slot [3 NCLS + 2] of the stats block is free; [3 NCLS + 3] = seed overflow list cursor)
	hao_pack_args pa; memset(&pa, 0, sizeof(pa));
	const uint64_t NW = (A + 63) / 64;      // 64-position words of the bit stream (positions = seed hits)
	unsigned long long *d_n_codes = B.stats.p + 3 * HAO_NCLS + 1;
	// cl->list -> wire format (hao_deliver.cuh): chain headers; codes of the chains the DP compacted; then ONE pass over the code array the quick check
	// filled - bits, rank directory, code bytes of the flagged positions, verbatim list.  (The number of chains is only known on the device here: launches
	// cover the bound, the kernels stop at ch_base[G].)
	// The pack kernels need the chain descriptors (chain_assemble_kernel) and nothing the selection and chain_final_kernel write: they run on a side stream UNDER those
	// (one-wave-per-read kernels that leave most of the device idle) and join the engine's stream in front of the batch's totals.  On the engine's stream they were
	// 9.4 ms of a configs[2] pass that nothing else ran under.  (Their scans have a scratch buffer of their own: c->d_tmp belongs to the scans of the selection.)
	auto pack = [&](hipStream_t ps) -> int {
		if (!G) return HAO_OK;
		hao_ctx::Batch::OutSet &O = B.O();
		hipLaunchKernelGGL(hao_pack_hdr_kernel, dim3((unsigned)((NCmax + 255) / 256)), dim3(256), 0, ps, pa, B.ch_base.p + G); HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL(hao_pack_bits_kernel, dim3((unsigned)((NW * 8 + 256 * HAO_PACK_U - 1) / (256 * HAO_PACK_U))), dim3(256), 0, ps, pa, A, NW, O.bits.p, B.pk_cnt.p, B.pk_ecnt.p); HAO_CHECK_LAUNCH();      // (the verbatim list: by count + scan, in position order)
		size_t tb = 0;
		HIP_TRY(rocprim::exclusive_scan(nullptr, tb, B.pk_cnt.p, O.rank.p, 0u, NW + 1, rocprim::plus<uint32_t>(), ps)); HIP_TRY(B.pk_tmp.reserve(tb + 256));
		HIP_TRY(rocprim::exclusive_scan(B.pk_tmp.p, tb, B.pk_cnt.p, O.rank.p, 0u, NW + 1, rocprim::plus<uint32_t>(), ps));
		HIP_TRY(rocprim::exclusive_scan(B.pk_tmp.p, tb, B.pk_ecnt.p, B.pk_erank.p, 0u, NW + 1, rocprim::plus<uint32_t>(), ps));
		hipLaunchKernelGGL(hao_pack_codes_kernel, dim3((unsigned)((NW * 8 + 255) / 256)), dim3(256), 0, ps, pa, B.hcode.p, A, O.bits.p, O.rank.p, NW, O.codes.p, d_n_codes, B.pk_erank.p); HAO_CHECK_LAUNCH();
		{ const uint64_t n4 = NW / 4 + 1; hipLaunchKernelGGL(hao_rank4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ps, O.rank.p, n4, O.rank4.p); HAO_CHECK_LAUNCH(); }
		return HAO_OK;
	};
	bool pack_on_side = false;
	if (parts & HAO_DELIVER_CL) {
		hao_ctx::Batch::OutSet &O = B.O();
		HIP_TRY(O.hdr.reserve(NCmax + 1)); HIP_TRY(O.exc.reserve(c->sw.exc_cap >= 0 ? (uint64_t)c->sw.exc_cap + 1 : std::max<uint64_t>(1 << 14, A / 256)));
		HIP_TRY(O.bits.reserve(NW + 2)); HIP_TRY(O.rank.reserve(NW + 6)); HIP_TRY(O.rank4.reserve(NW / 4 + 2)); HIP_TRY(B.pk_cnt.reserve(NW + 2)); HIP_TRY(O.codes.reserve(A + 16));
		HIP_TRY(B.pk_ecnt.reserve(NW + 2)); HIP_TRY(B.pk_erank.reserve(NW + 2));
		HIP_TRY(O.ch_off.reserve(n + 2)); HIP_TRY(O.cl_off.reserve(n + 2)); HIP_TRY(O.qm_off.reserve(n + 2));
		O.qmz16 = c->max_len_all < 65536 && B.wgt_max < 256 && !c->sw.qmz_raw;
		if (O.qmz16) { HIP_TRY(O.qmz_pos.reserve(nm + 1)); HIP_TRY(O.qmz_cnt.reserve(nm + 1)); } else HIP_TRY(O.qmz.reserve(nm + 1));
		pa.cd = B.cd.p; pa.hits = B.hits.p; pa.ohits = B.ohits.p; pa.mz_off = c->d_ix_mz_off.p; pa.seg = B.seg.p; pa.n_sel = n; pa.rid_lo = lo; pa.mz0 = B.mz0; pa.q_pos = B.q_pos.p;
		pa.hq = B.hq.p; pa.ohq = B.ohq.p;
		pa.hdr = O.hdr.p; pa.bytes = B.hcode.p; pa.exc = O.exc.p; pa.exc_cnt = d_exc_cnt; pa.exc_every = (uint32_t)c->sw.exc_every;
		pa.exc_cap = c->sw.exc_cap >= 0 ? std::min<uint64_t>(O.exc.cap, (uint64_t)c->sw.exc_cap) : O.exc.cap;
		hipStream_t ps = c->stream;
		if (B.side_ready && !c->sw.dp_serial) {      // (the side streams exist once a batch had groups; HAO_DBG_FORCE=dp_serial keeps everything on the engine's stream)
			if (!B.ev_pk0) { HIP_TRY(hipEventCreateWithFlags(&B.ev_pk0, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&B.ev_pk1, hipEventDisableTiming)); }
			ps = B.side[0]; pack_on_side = true;
			HIP_TRY(hipEventRecord(B.ev_pk0, c->stream)); HIP_TRY(hipStreamWaitEvent(ps, B.ev_pk0, 0));
		}
		HIP_TRY(hipMemsetAsync(B.pk_cnt.p + NW, 0, 4, ps));      // (the scans run over NW + 1 counts: their last output is the total)
		HIP_TRY(hipMemsetAsync(B.pk_ecnt.p + NW, 0, 4, ps));
		if (int rc = pack(ps)) return rc;
		hipLaunchKernelGGL(hao_read_ranges_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, ps, B.g_off.p, B.ch_base.p, B.cl_base.p, c->d_ix_mz_off.p, lo, B.mz0, n, O.ch_off.p, O.cl_off.p, O.qm_off.p);
		HAO_CHECK_LAUNCH();
		if (nm && O.qmz16) { hipLaunchKernelGGL(hao_qtab16_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, ps, B.q_pos.p, B.q_cnt.p, nm, O.qmz_pos.p, O.qmz_cnt.p, c->d_err.p); HAO_CHECK_LAUNCH(); }
		else if (nm) { hipLaunchKernelGGL(hao_qtab_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, ps, B.q_pos.p, B.q_cnt.p, nm, O.qmz.p); HAO_CHECK_LAUNCH(); }
		if (pack_on_side) HIP_TRY(hipEventRecord(B.ev_pk1, ps));
	}
	c->timer.mark("q_assemble");
	// Q8 selection
	{
		uint64_t o = 0;      // coverage windows of the pruning scan: len / ocv_w + 2 per read; offsets by a device scan, the total (a size) from the host's copy of the lengths
		for (uint64_t r = 0; r < n; ++r) o += c->h_len_all[glo + r] / par.ocv_w + 2;
		HIP_TRY(B.cc_off.reserve(n + 2)); HIP_TRY(B.cc.reserve(o + 1)); HIP_TRY(B.n_final.reserve(n + 2)); HIP_TRY(B.fc_final.reserve(n + 2));
		HIP_TRY(B.O().fin_off.reserve(n + 2)); HIP_TRY(B.fcf_off.reserve(n + 2)); HIP_TRY(B.nch64.reserve(n + 2));
		hipLaunchKernelGGL(hao_cc_count_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream, c->d_len_all.p, glo, n, (uint64_t)par.ocv_w, B.nch64.p);
		HAO_CHECK_LAUNCH();
		if (int rc = hao_excl_scan_u64(c, B.nch64.p, B.cc_off.p, n + 1)) return rc;
	}
	const uint64_t NC = NCmax;
	HIP_TRY(B.key_xs.reserve(NC + 1)); HIP_TRY(B.key_sc.reserve(NC + 1)); HIP_TRY(B.key_al.reserve(NC + 1));
	hao_sel_args sa;
	HIP_TRY(B.key_tmp.reserve(5 * NC + 8));
	sa.key_xs = B.key_xs.p; sa.key_sc = B.key_sc.p; sa.key_al = B.key_al.p; sa.key_tmp = B.key_tmp.p;
	sa.ol = B.ol.p; sa.g_off = B.g_off.p; sa.ch_base = B.ch_base.p; sa.cl_base = B.cl_base.p; sa.cd = B.cd.p; sa.hits = B.hits.p; sa.ohits = B.ohits.p; sa.n_sel = n; sa.rid_lo = glo; sa.len = c->d_len_all.p; sa.cc_off = B.cc_off.p; sa.cc = B.cc.p;
	sa.perm = B.perm.p; sa.n_final = B.n_final.p; sa.fc_final = B.fc_final.p; sa.max_n_chain = par.max_n_chain; sa.ocv_w = par.ocv_w; sa.chain_cutoff = par.chain_cutoff;
	sa.dbg = nullptr; sa.dbg_seq_prune = c->sw.seq_prune ? 1 : 0;
	if (c->sw.selphase) { HIP_TRY(B.dbgbuf.reserve(8)); HIP_TRY(hipMemsetAsync(B.dbgbuf.p, 0, 64, c->stream)); sa.dbg = B.dbgbuf.p; }
	// three launches split by chain count: the common reads (<= 128 chains) need 5 KB of LDS per wave and fill the CUs; 512- and 1024-chain slices for
	// repeat-rich reads (beyond 1024 chains the keys stay in global scratch)
	hipLaunchKernelGGL((chain_select_kernel<1, 128>), dim3((unsigned)(n + 1)), dim3(64), 0, c->stream, sa, (int64_t)0, (int64_t)129);
	HAO_CHECK_LAUNCH();
	{	// 129 .. 4096 chains: four waves per read share the sorts; beyond: keys in global scratch, one wave
		hipLaunchKernelGGL((chain_select4_kernel<512>), dim3((unsigned)n), dim3(256), 0, c->stream, sa, (int64_t)129, (int64_t)513);
		HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL((chain_select4_kernel<1024>), dim3((unsigned)n), dim3(256), 0, c->stream, sa, (int64_t)513, (int64_t)1025);
		HAO_CHECK_LAUNCH();
		// reads that cross repeat families: thousands of chains (250 Mb repeat-rich set: a quarter of the reads have more than 1024).  With the keys in global scratch
		// and one wave per read those took 17 ms per batch; 64 / 128 KB of LDS per read keeps them on the four-wave path
		hipLaunchKernelGGL((chain_select4_kernel<2048>), dim3((unsigned)n), dim3(256), 0, c->stream, sa, (int64_t)1025, (int64_t)2049);
		HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL((chain_select4_kernel<4096>), dim3((unsigned)n), dim3(256), 0, c->stream, sa, (int64_t)2049, (int64_t)4097);
		HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL((chain_select_kernel<1, 1024>), dim3((unsigned)(n + 1)), dim3(64), 0, c->stream, sa, (int64_t)4097, (int64_t)INT64_MAX);
	}
	HAO_CHECK_LAUNCH();
	if (int rc = hao_scan_u32(c, B.n_final.p, B.O().fin_off.p, n + 1)) return rc;
	if (int rc = hao_excl_scan_u64(c, B.fc_final.p, B.fcf_off.p, n + 1)) return rc;
	if (sa.dbg) { unsigned long long d_[5]; HIP_TRY(hipMemcpy(d_, B.dbgbuf.p, 40, hipMemcpyDeviceToHost)); if (d_[4]) fprintf(stderr, "[select] reads %llu  avg us: score sort %.1f  prune %.1f  position sort %.1f  weak filter %.1f\n", d_[4], d_[0] / 100.0 / d_[4], d_[1] / 100.0 / d_[4], d_[2] / 100.0 / d_[4], d_[3] / 100.0 / d_[4]); }
	c->timer.mark("q_select");
	HIP_TRY(B.O().ol_out.reserve(NCmax + 1)); HIP_TRY(B.O().fc_out.reserve(FCmax + 1)); HIP_TRY(B.O().fc_out_off.reserve(NCmax + 2));
	unsigned long long *d_n_fcw = B.stats.p + 3 * HAO_NCLS;      // (slot [3 NCLS] of the stats block is free) words of the fake cigars that travel raw
	const bool fcw_ = (parts & HAO_DELIVER_OL) != 0;
	if (fcw_) { HIP_TRY(B.O().fcw_off.reserve(NCmax + 2)); HIP_TRY(B.O().fcw.reserve(3 * (FCmax + 1))); HIP_TRY(B.O().ol_wire.reserve(NCmax + 1)); }      // main region: entries - overlaps words; raw overlaps behind it: two words per entry
	hipLaunchKernelGGL(chain_final_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, B.ol.p, B.ol_fc_off.p, B.fc_raw.p, B.perm.p, B.g_off.p, B.ch_base.p,
					   B.O().fin_off.p, B.fcf_off.p, n, B.O().ol_out.p, B.O().fc_out.p, B.O().fc_out_off.p, fcw_ ? B.O().fcw.p : (uint32_t*)nullptr, fcw_ ? B.O().fcw_off.p : (uint64_t*)nullptr, d_n_fcw, (uint32_t)c->sw.fc_raw_every);
	HAO_CHECK_LAUNCH();
	if (fcw_) { hipLaunchKernelGGL(hao_ol_wire_kernel, dim3((unsigned)std::min<uint64_t>((NCmax + 255) / 256, 4096)), dim3(256), 0, c->stream, B.O().ol_out.p, B.O().fin_off.p + n, B.O().ol_wire.p); HAO_CHECK_LAUNCH(); }
	if (pack_on_side) HIP_TRY(hipStreamWaitEvent(c->stream, B.ev_pk1, 0));      // the pack kernels have run under the selection
	c->timer.mark("q_final");
	unsigned long long slow_st[HAO_NCLS + 4], n_exc = 0;
	{	// the totals of the batch: one wave gathers them into mapped host memory
		auto peek = [&](const void *src, int nw, int at) { hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)src, nw, c->peek_d + at); };
		peek(B.ch_base.p + G, 1, 0); peek(B.cl_base.p + G, 1, 1); peek(B.fc_base.p + G * HAO_MCOPY_MAX, 1, 2); peek(B.O().fin_off.p + n, 1, 3); peek(B.fcf_off.p + n, 1, 4);
		peek(d_slow_cnt, HAO_NCLS + 4, 8); peek(d_exc_cnt, 1, 5); peek(d_n_codes, 1, 6); peek(d_n_fcw, 1, 7); peek(B.stats.p + 3 * HAO_NCLS + 3, 3, 24); peek(c->d_err.p, 1, 27);
		HAO_CHECK_LAUNCH();
		const double ts2_ = hao_now();
		HIP_TRY(hipStreamSynchronize(c->stream));
		B.t_s3 += hao_now() - ts2_; B.t_run += hao_now() - t_run0; ++B.t_nrun;
		if (c->sw.dltime && (B.t_nrun & 15) == 0) fprintf(stderr, "[batch] %llu runs (parts %u): total %.1f ms  before sync1 %.1f  sync1 %.1f  sync2 %.1f  sync3 %.1f\n", (unsigned long long)B.t_nrun, parts, B.t_run * 1e3, B.t_pre * 1e3, B.t_s1 * 1e3, B.t_s2 * 1e3, B.t_s3 * 1e3);
		B.n_chains = c->peek_h[0]; B.n_cl = c->peek_h[1]; B.n_fc_raw = c->peek_h[2]; B.n_ol = c->peek_h[3]; B.n_fc = c->peek_h[4];
		for (int x = 0; x < HAO_NCLS + 4; ++x) slow_st[x] = c->peek_h[8 + x];
		B.seed_left[0] = c->peek_h[26]; B.seed_left[1] = c->peek_h[24]; B.seed_left[2] = c->peek_h[25];      // reads left by the first seed launch / by the 512-slot / by the 1024-slot table
		if (parts & HAO_DELIVER_CL) { n_exc = c->peek_h[5]; B.n_codes = G ? c->peek_h[6] : 0; }
		if ((parts & HAO_DELIVER_CL) && (uint32_t)c->peek_h[27]) { hao_set_err(c, "a minimizer position or seed weight that does not fit the packed minimizer table"); return HAO_EUNSUPP; }      // (hao_qtab16_kernel; the host's bounds rule it out)
		if ((parts & HAO_DELIVER_OL) && (c->peek_h[7] >> 63)) { hao_set_err(c, "an overlap without fake-cigar entries: the packed cigar layout holds at least one per overlap"); return HAO_EUNSUPP; }
		B.n_fcw = (parts & HAO_DELIVER_OL) ? (B.n_fc - B.n_ol) + c->peek_h[7] : 0;      // main region + the raw overlaps' words
	}
	if ((parts & HAO_DELIVER_CL) && n_exc > pa.exc_cap) {      // more verbatim hits than the list holds: grow it and pack again (the sources are untouched)
		HIP_TRY(B.O().exc.reserve(n_exc + 1024)); pa.exc = B.O().exc.p; pa.exc_cap = B.O().exc.cap;
		HIP_TRY(hipMemsetAsync(d_exc_cnt, 0, 8, c->stream));
		HIP_TRY(hipMemsetAsync(B.pk_cnt.p + NW, 0, 4, c->stream)); HIP_TRY(hipMemsetAsync(B.pk_ecnt.p + NW, 0, 4, c->stream));
		if (int rc = pack(c->stream)) return rc;
		hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)d_exc_cnt, 1, c->peek_d + 5); HAO_CHECK_LAUNCH();
		hipLaunchKernelGGL(hao_peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)d_n_codes, 1, c->peek_d + 6); HAO_CHECK_LAUNCH();
		HIP_TRY(hipStreamSynchronize(c->stream));
		B.n_codes = c->peek_h[6];
		n_exc = c->peek_h[5];
	}
	B.n_exc = n_exc;
	B.n_generic = 0; for (int x = 0; x < HAO_NCLS; ++x) B.n_generic += slow_st[x];
	B.n_generic_hits = slow_st[HAO_NCLS];
	if (c->sw.dp_stats) { fprintf(stderr, "[dp] slow groups by class:"); for (int x = 0; x < HAO_NCLS; ++x) fprintf(stderr, " %llu/%llu", slow_st[x], cls_cnt[x]);
		fprintf(stderr, "  hits %llu  dp range %llu  spec-committed %llu  spec-failures %llu\n", slow_st[HAO_NCLS], slow_st[HAO_NCLS + 3], slow_st[HAO_NCLS + 1], slow_st[HAO_NCLS + 2]); }
	B.valid = true;
	if (parts & HAO_DELIVER_EXACT) { if (int rc = hao_exact_run(c)) return rc; }
	if (parts) { const double t0_ = hao_now(); const int rc_ = hao_deliver_enqueue(c); B.t_enq += hao_now() - t0_; ++B.t_n; if (c->sw.dltime && (B.t_n & 15) == 0) fprintf(stderr, "[deliver] %llu batches: slot wait %.1f ms, enqueue %.1f ms (arena alloc %.1f ms)\n", (unsigned long long)B.t_n, B.t_evsync * 1e3, B.t_enq * 1e3, B.t_alloc * 1e3); return rc_; }
	return HAO_OK;
}

// cl->list of the batch as tagged k_mer_hits in HBM: built on demand from the chain descriptors (the blocking fetch API and the digests read it;
// the streaming delivery path packs straight from the descriptors and never needs it)
static int hao_batch_materialize_cl(hao_ctx *c)
{
	hao_ctx::Batch &B = *c->batch;
	if (B.cl_valid) return HAO_OK;
	HIP_TRY(B.cl.reserve(B.n_cl + 1));
	if (B.n_chains) {
		hipLaunchKernelGGL(chain_materialize_kernel, dim3((unsigned)((B.n_chains + 3) / 4)), dim3(256), 0, c->stream, B.cd.p, B.n_chains, B.hits.p, B.ohits.p, B.cl.p);
		HAO_CHECK_LAUNCH();
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	B.cl_valid = true;
	return HAO_OK;
}

// host copies for the fetch API (one bulk download per batch)
static int hao_batch_download(hao_ctx *c)
{
	hao_ctx::Batch &B = *c->batch;
	if (B.host_valid) return HAO_OK;
	if (int rc = hao_batch_materialize_cl(c)) return rc;
	const uint64_t n = B.n;
	B.h_seg.assign(n + 1, 0); B.h_fin_off.assign(n + 1, 0); B.h_cl_off.assign(n + 1, 0);
	B.h_hits.resize(B.n_anchor); B.h_cl.resize(B.n_cl); B.h_ol.resize(B.n_ol); B.h_fc.resize(B.n_fc); B.h_fc_out_off.assign(B.n_ol + 1, 0);
	if (n) {
		HIP_TRY(hipMemcpy(B.h_seg.data(), B.seg.p, (n + 1) * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(B.h_fin_off.data(), B.O().fin_off.p, (n + 1) * 8, hipMemcpyDeviceToHost));
		std::vector<uint64_t> goff(n + 1), clb(B.n_groups + 1);
		HIP_TRY(hipMemcpy(goff.data(), B.g_off.p, (n + 1) * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(clb.data(), B.cl_base.p, (B.n_groups + 1) * 8, hipMemcpyDeviceToHost));
		for (uint64_t r = 0; r <= n; ++r) B.h_cl_off[r] = clb[goff[r]];
	}
	if (B.n_anchor) HIP_TRY(hipMemcpy(B.h_hits.data(), B.hits.p, B.n_anchor * sizeof(hao_hit_t), hipMemcpyDeviceToHost));
	if (B.n_cl) HIP_TRY(hipMemcpy(B.h_cl.data(), B.cl.p, B.n_cl * sizeof(hao_hit_t), hipMemcpyDeviceToHost));
	if (B.n_ol) {
		HIP_TRY(hipMemcpy(B.h_ol.data(), B.O().ol_out.p, B.n_ol * sizeof(hao_ovlp_t), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(B.h_fc_out_off.data(), B.O().fc_out_off.p, B.n_ol * 8, hipMemcpyDeviceToHost));
	}
	B.h_fc_out_off[B.n_ol] = B.n_fc;
	if (B.n_fc) HIP_TRY(hipMemcpy(B.h_fc.data(), B.O().fc_out.p, B.n_fc * 8, hipMemcpyDeviceToHost));
	B.host_valid = true;
	return HAO_OK;
}
