// Exchange layer for the sharded (one process per GPU) mode.
//
// Two backends behind one interface:
//   * RCCL (ncclComm over xGMI): production.  Variable-size exchanges are grouped
//     ncclSend/ncclRecv (all-to-all-v) or per-root ncclBroadcast (all-gather-v) - direct
//     point-to-point over the 7 xGMI links per GPU, no ring all-reduce on bulk data; the only
//     all-reduce is the 32 KB k-mer histogram.
//   * loopback: several engines inside ONE process on ONE GPU, one host thread each, exchanging
//     through device-to-device copies between two barriers.  It exists so the partitioning and
//     offset logic of the sharded path can be parity-tested on a single-GPU box.
#pragma once
#include <pthread.h>
#include <rccl/rccl.h>
#include <vector>
#include "hao_ctx.hpp"

struct hao_loop_group {
	int world; pthread_barrier_t bar;
	std::vector<const void*> ptr, ptr2; std::vector<std::vector<uint64_t> > cnt;   // per rank: published buffer(s) + per-destination counts
	std::vector<std::vector<uint64_t> > hostv;
};

struct hao_comm {
	int rank = 0, world = 1;
	ncclComm_t nccl = nullptr; hao_loop_group *loop = nullptr;
	DevBuf<char> ag_tmp;      // padded slots of the balanced all-gather-v
	DevBuf<uint64_t> sc_a, sc_b;      // scratch of the small host-value collectives (allocated once: no hipMalloc / hipFree on the timed path)
	bool active() const { return world > 1 || nccl || loop; }
	void release() { ag_tmp.release(); sc_a.release(); sc_b.release(); }
};
static bool hao_comm_is_active(const hao_comm *cm) { return cm->active(); }

// Every small collective below also carries the status of the LOCAL phase that preceded it on each rank (local_rc): a rank whose allocation /
// kernel / sort failed still takes part (with empty data) and every rank returns the same error afterwards, so nobody is left waiting in the next
// collective for a peer that has already returned.
static int hao_comm_verdict(hao_ctx *c, const hao_comm &cm, const std::vector<uint64_t> &status, int local_rc)
{
	if (local_rc) return local_rc;
	for (int r = 0; r < cm.world; ++r) if (status[r]) { hao_set_err(c, "rank " + std::to_string(r) + " failed in the preceding local phase (code -" + std::to_string(status[r]) + ")"); return -(int)status[r]; }
	return HAO_OK;
}

#define NCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { hao_set_err(c, std::string(#expr) + ": " + ncclGetErrorString(_r)); return HAO_ENODEV; } } while (0)

// all ranks learn nv values of every rank: out[r * nv + i] = value i of rank r (one collective); returns the agreed status (see above)
static int hao_comm_allgather_u64n(hao_ctx *c, hao_comm &cm, const uint64_t *v, int nv, std::vector<uint64_t> &out, int local_rc = 0)
{
	const int W = cm.world, nw = nv + 1;
	out.assign((size_t)W * nv, 0);
	if (W == 1 && !cm.nccl && !cm.loop) { for (int i = 0; i < nv; ++i) out[i] = v[i]; return local_rc; }
	std::vector<uint64_t> mine(v, v + nv), all((size_t)W * nw, 0), st(W, 0); mine.push_back((uint64_t)(-local_rc));
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->hostv[cm.rank] = mine;
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < W; ++r) for (int i = 0; i < nw; ++i) all[(size_t)r * nw + i] = g->hostv[r][i];
		pthread_barrier_wait(&g->bar);
	} else {
		HIP_TRY(cm.sc_a.reserve((size_t)W * nw + 1));
		HIP_TRY(hipMemcpyAsync(cm.sc_a.p + (size_t)cm.rank * nw, mine.data(), 8 * (size_t)nw, hipMemcpyHostToDevice, c->stream));
		NCCL_TRY(ncclAllGather(cm.sc_a.p + (size_t)cm.rank * nw, cm.sc_a.p, (size_t)nw, ncclUint64, cm.nccl, c->stream));
		HIP_TRY(hipMemcpyAsync(all.data(), cm.sc_a.p, 8 * (size_t)W * nw, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	for (int r = 0; r < W; ++r) { for (int i = 0; i < nv; ++i) out[(size_t)r * nv + i] = all[(size_t)r * nw + i]; st[r] = all[(size_t)r * nw + nv]; }
	return hao_comm_verdict(c, cm, st, local_rc);
}
static int hao_comm_allgather_u64(hao_ctx *c, hao_comm &cm, uint64_t v, std::vector<uint64_t> &out, int local_rc = 0)
{ return hao_comm_allgather_u64n(c, cm, &v, 1, out, local_rc); }

// element-wise sum of a host int64 vector over all ranks (the 4096-bin histogram); one more element carries the status
static int hao_comm_allreduce_i64(hao_ctx *c, hao_comm &cm, int64_t *v, size_t n, int local_rc = 0)
{
	if (cm.world == 1 && !cm.nccl && !cm.loop) return local_rc;
	std::vector<int64_t> w(v, v + n); w.push_back(local_rc ? 1 : 0);
	if (local_rc) std::fill(w.begin(), w.begin() + n, 0);
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->hostv[cm.rank].assign((uint64_t*)w.data(), (uint64_t*)w.data() + n + 1);
		pthread_barrier_wait(&g->bar);
		std::vector<int64_t> s(n + 1, 0);
		for (int r = 0; r < cm.world; ++r) for (size_t i = 0; i <= n; ++i) s[i] += (int64_t)g->hostv[r][i];
		pthread_barrier_wait(&g->bar);
		w = s;
	} else {
		HIP_TRY(cm.sc_b.reserve(n + 2));
		HIP_TRY(hipMemcpyAsync(cm.sc_b.p, w.data(), 8 * (n + 1), hipMemcpyHostToDevice, c->stream));
		NCCL_TRY(ncclAllReduce(cm.sc_b.p, cm.sc_b.p, n + 1, ncclInt64, ncclSum, cm.nccl, c->stream));
		HIP_TRY(hipMemcpyAsync(w.data(), cm.sc_b.p, 8 * (n + 1), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	memcpy(v, w.data(), n * 8);
	if (local_rc) return local_rc;
	if (w[n]) { hao_set_err(c, std::to_string((long long)w[n]) + " rank(s) failed in the preceding local phase"); return HAO_ENODEV; }
	return HAO_OK;
}

// all-gather-v of device arrays of `esz`-byte elements: out = concatenation in rank order. counts[] = elements per rank.
static int hao_comm_allgatherv(hao_ctx *c, hao_comm &cm, const void *src, uint64_t n_mine, size_t esz, void *out, const std::vector<uint64_t> &counts)
{
	std::vector<uint64_t> disp(cm.world + 1, 0);
	for (int r = 0; r < cm.world; ++r) disp[r + 1] = disp[r] + counts[r];
	if (cm.world == 1 && !cm.nccl && !cm.loop) { if (n_mine) HIP_TRY(hipMemcpyAsync(out, src, n_mine * esz, hipMemcpyDeviceToDevice, c->stream)); return HAO_OK; }
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		HIP_TRY(hipStreamSynchronize(c->stream));
		g->ptr[cm.rank] = src;
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) if (counts[r]) HIP_TRY(hipMemcpyAsync((char*)out + disp[r] * esz, g->ptr[r], counts[r] * esz, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	{	// nearly equal parts (hash-range partitions are): ONE ring all-gather over padded slots - every link busy the whole time - then the slots are
		// packed; only lopsided exchanges fall back to one broadcast per root
		uint64_t maxc = 0, total = disp[cm.world]; for (int r = 0; r < cm.world; ++r) maxc = std::max(maxc, counts[r]);
		if (maxc && maxc * (uint64_t)cm.world <= total + total / 4 + 4096) {
			const size_t slot = (size_t)maxc * esz;
			HIP_TRY(cm.ag_tmp.reserve(slot * cm.world + 16));
			if (n_mine) HIP_TRY(hipMemcpyAsync(cm.ag_tmp.p + slot * cm.rank, src, n_mine * esz, hipMemcpyDeviceToDevice, c->stream));
			NCCL_TRY(ncclAllGather(cm.ag_tmp.p + slot * cm.rank, cm.ag_tmp.p, slot, ncclChar, cm.nccl, c->stream));      // in place: send = recv + rank * count
			for (int r = 0; r < cm.world; ++r) if (counts[r]) HIP_TRY(hipMemcpyAsync((char*)out + disp[r] * esz, cm.ag_tmp.p + slot * r, counts[r] * esz, hipMemcpyDeviceToDevice, c->stream));
			return HAO_OK;
		}
	}
	// own part must sit in the output before it is broadcast from there
	if (n_mine) HIP_TRY(hipMemcpyAsync((char*)out + disp[cm.rank] * esz, src, n_mine * esz, hipMemcpyDeviceToDevice, c->stream));
	NCCL_TRY(ncclGroupStart());
	for (int r = 0; r < cm.world; ++r) if (counts[r]) { void *p = (char*)out + disp[r] * esz; NCCL_TRY(ncclBroadcast(p, p, counts[r] * esz, ncclChar, r, cm.nccl, c->stream)); }
	NCCL_TRY(ncclGroupEnd());
	return HAO_OK;
}

// all-to-all-v of u64 elements: rank sends src[sdisp[d] .. sdisp[d]+scnt[d]) to d; receives into out in source-rank order.
// rcnt is filled with the received counts. out must hold sum(rcnt) (caller sizes it after the count exchange: see hao_comm_exchange_counts).
static int hao_comm_exchange_counts(hao_ctx *c, hao_comm &cm, const std::vector<uint64_t> &scnt_in, std::vector<uint64_t> &rcnt, int local_rc = 0)
{
	const int W = cm.world;
	std::vector<uint64_t> scnt(scnt_in); scnt.resize(W, 0);
	if (local_rc) std::fill(scnt.begin(), scnt.end(), 0);          // a failed rank sends nothing, but it does send its status
	rcnt.assign(W, 0);
	if (W == 1 && !cm.nccl && !cm.loop) { rcnt[0] = scnt[0]; return local_rc; }
	std::vector<uint64_t> st(W, 0);
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->cnt[cm.rank] = scnt; g->cnt[cm.rank].push_back((uint64_t)(-local_rc));
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < W; ++r) { rcnt[r] = g->cnt[r][cm.rank]; st[r] = g->cnt[r][W]; }
		pthread_barrier_wait(&g->bar);
		return hao_comm_verdict(c, cm, st, local_rc);
	}
	std::vector<uint64_t> snd(2 * W), rcv(2 * W, 0);
	for (int r = 0; r < W; ++r) { snd[2 * r] = scnt[r]; snd[2 * r + 1] = (uint64_t)(-local_rc); }
	HIP_TRY(cm.sc_a.reserve(2 * W + 1)); HIP_TRY(cm.sc_b.reserve(2 * W + 1));
	HIP_TRY(hipMemcpyAsync(cm.sc_a.p, snd.data(), 16 * W, hipMemcpyHostToDevice, c->stream));
	NCCL_TRY(ncclGroupStart());
	for (int r = 0; r < W; ++r) { NCCL_TRY(ncclSend(cm.sc_a.p + 2 * r, 2, ncclUint64, r, cm.nccl, c->stream)); NCCL_TRY(ncclRecv(cm.sc_b.p + 2 * r, 2, ncclUint64, r, cm.nccl, c->stream)); }
	NCCL_TRY(ncclGroupEnd());
	HIP_TRY(hipMemcpyAsync(rcv.data(), cm.sc_b.p, 16 * W, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	for (int r = 0; r < W; ++r) { rcnt[r] = rcv[2 * r]; st[r] = rcv[2 * r + 1]; }
	return hao_comm_verdict(c, cm, st, local_rc);
}

static int hao_comm_alltoallv_u64(hao_ctx *c, hao_comm &cm, const uint64_t *src, const std::vector<uint64_t> &scnt, const std::vector<uint64_t> &sdisp,
								  uint64_t *out, const std::vector<uint64_t> &rcnt)
{
	std::vector<uint64_t> rdisp(cm.world + 1, 0);
	for (int r = 0; r < cm.world; ++r) rdisp[r + 1] = rdisp[r] + rcnt[r];
	if (cm.world == 1 && !cm.nccl && !cm.loop) { if (scnt[0]) HIP_TRY(hipMemcpyAsync(out, src + sdisp[0], scnt[0] * 8, hipMemcpyDeviceToDevice, c->stream)); return HAO_OK; }
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		HIP_TRY(hipStreamSynchronize(c->stream));
		g->ptr[cm.rank] = src; g->cnt[cm.rank] = sdisp;        // publish my send displacements
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) if (rcnt[r]) HIP_TRY(hipMemcpyAsync(out + rdisp[r], (const uint64_t*)g->ptr[r] + g->cnt[r][cm.rank], rcnt[r] * 8, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	NCCL_TRY(ncclGroupStart());
	for (int r = 0; r < cm.world; ++r) {
		if (scnt[r]) NCCL_TRY(ncclSend(src + sdisp[r], scnt[r], ncclUint64, r, cm.nccl, c->stream));
		if (rcnt[r]) NCCL_TRY(ncclRecv(out + rdisp[r], rcnt[r], ncclUint64, r, cm.nccl, c->stream));
	}
	NCCL_TRY(ncclGroupEnd());
	return HAO_OK;
}

// all-to-all-v of TWO parallel u64 arrays with the same counts (minimizer hash + record): one grouped exchange
static int hao_comm_alltoallv2_u64(hao_ctx *c, hao_comm &cm, const uint64_t *src1, const uint64_t *src2, const std::vector<uint64_t> &scnt, const std::vector<uint64_t> &sdisp,
								   uint64_t *out1, uint64_t *out2, const std::vector<uint64_t> &rcnt)
{
	std::vector<uint64_t> rdisp(cm.world + 1, 0);
	for (int r = 0; r < cm.world; ++r) rdisp[r + 1] = rdisp[r] + rcnt[r];
	if (cm.world == 1 && !cm.nccl && !cm.loop) {
		if (scnt[0]) { HIP_TRY(hipMemcpyAsync(out1, src1 + sdisp[0], scnt[0] * 8, hipMemcpyDeviceToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(out2, src2 + sdisp[0], scnt[0] * 8, hipMemcpyDeviceToDevice, c->stream)); }
		return HAO_OK;
	}
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		HIP_TRY(hipStreamSynchronize(c->stream));
		g->ptr[cm.rank] = src1; g->ptr2[cm.rank] = src2; g->cnt[cm.rank] = sdisp;
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) if (rcnt[r]) {
			HIP_TRY(hipMemcpyAsync(out1 + rdisp[r], (const uint64_t*)g->ptr[r] + g->cnt[r][cm.rank], rcnt[r] * 8, hipMemcpyDeviceToDevice, c->stream));
			HIP_TRY(hipMemcpyAsync(out2 + rdisp[r], (const uint64_t*)g->ptr2[r] + g->cnt[r][cm.rank], rcnt[r] * 8, hipMemcpyDeviceToDevice, c->stream));
		}
		HIP_TRY(hipStreamSynchronize(c->stream));
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	NCCL_TRY(ncclGroupStart());
	for (int r = 0; r < cm.world; ++r) {
		if (scnt[r]) { NCCL_TRY(ncclSend(src1 + sdisp[r], scnt[r], ncclUint64, r, cm.nccl, c->stream)); NCCL_TRY(ncclSend(src2 + sdisp[r], scnt[r], ncclUint64, r, cm.nccl, c->stream)); }
		if (rcnt[r]) { NCCL_TRY(ncclRecv(out1 + rdisp[r], rcnt[r], ncclUint64, r, cm.nccl, c->stream)); NCCL_TRY(ncclRecv(out2 + rdisp[r], rcnt[r], ncclUint64, r, cm.nccl, c->stream)); }
	}
	NCCL_TRY(ncclGroupEnd());
	return HAO_OK;
}

// all-gather of one fixed-size slot per rank: buf = world slots of `slot` bytes, the caller has filled slot `rank`
static int hao_comm_allgather_fixed(hao_ctx *c, hao_comm &cm, char *buf, size_t slot)
{
	if ((cm.world == 1 && !cm.nccl && !cm.loop) || slot == 0) return HAO_OK;
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		HIP_TRY(hipStreamSynchronize(c->stream));
		g->ptr[cm.rank] = buf;
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) if (r != cm.rank) HIP_TRY(hipMemcpyAsync(buf + slot * r, (const char*)g->ptr[r] + slot * r, slot, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	NCCL_TRY(ncclAllGather(buf + slot * cm.rank, buf, slot, ncclChar, cm.nccl, c->stream));      // in place: send = recv + rank * count
	return HAO_OK;
}
