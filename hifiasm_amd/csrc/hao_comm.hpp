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
	std::vector<uint64_t> shard_sizes; uint64_t shard_sizes_for = ~0ULL;      // reads per rank, checked once per read set
	bool active() const { return world > 1 || nccl || loop; }
};

#define NCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { hao_set_err(c, std::string(#expr) + ": " + ncclGetErrorString(_r)); return HAO_ENODEV; } } while (0)

// all ranks learn every rank's value (host u64)
static int hao_comm_allgather_u64(hao_ctx *c, hao_comm &cm, uint64_t v, std::vector<uint64_t> &out)
{
	out.assign(cm.world, 0);
	if (cm.world == 1 && !cm.nccl && !cm.loop) { out[0] = v; return HAO_OK; }
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->hostv[cm.rank].assign(1, v);
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) out[r] = g->hostv[r][0];
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	DevBuf<uint64_t> d; HIP_TRY(d.reserve(cm.world + 1));
	HIP_TRY(hipMemcpyAsync(d.p + cm.rank, &v, 8, hipMemcpyHostToDevice, c->stream));
	NCCL_TRY(ncclAllGather(d.p + cm.rank, d.p, 1, ncclUint64, cm.nccl, c->stream));
	HIP_TRY(hipMemcpyAsync(out.data(), d.p, 8 * cm.world, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	d.release();
	return HAO_OK;
}

// the same for nv values per rank: out[r * nv + i] = value i of rank r (one collective instead of nv)
static int hao_comm_allgather_u64n(hao_ctx *c, hao_comm &cm, const uint64_t *v, int nv, std::vector<uint64_t> &out)
{
	out.assign((size_t)cm.world * nv, 0);
	if (cm.world == 1 && !cm.nccl && !cm.loop) { for (int i = 0; i < nv; ++i) out[i] = v[i]; return HAO_OK; }
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->hostv[cm.rank].assign(v, v + nv);
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) for (int i = 0; i < nv; ++i) out[(size_t)r * nv + i] = g->hostv[r][i];
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	DevBuf<uint64_t> d; HIP_TRY(d.reserve((size_t)cm.world * nv + 1));
	HIP_TRY(hipMemcpyAsync(d.p + (size_t)cm.rank * nv, v, 8 * (size_t)nv, hipMemcpyHostToDevice, c->stream));
	NCCL_TRY(ncclAllGather(d.p + (size_t)cm.rank * nv, d.p, (size_t)nv, ncclUint64, cm.nccl, c->stream));
	HIP_TRY(hipMemcpyAsync(out.data(), d.p, 8 * (size_t)cm.world * nv, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	d.release();
	return HAO_OK;
}

// element-wise sum of a host int64 vector over all ranks (the 4096-bin histogram)
static int hao_comm_allreduce_i64(hao_ctx *c, hao_comm &cm, int64_t *v, size_t n)
{
	if (cm.world == 1 && !cm.nccl && !cm.loop) return HAO_OK;
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->hostv[cm.rank].assign((uint64_t*)v, (uint64_t*)v + n);
		pthread_barrier_wait(&g->bar);
		std::vector<int64_t> s(n, 0);
		for (int r = 0; r < cm.world; ++r) for (size_t i = 0; i < n; ++i) s[i] += (int64_t)g->hostv[r][i];
		pthread_barrier_wait(&g->bar);
		memcpy(v, s.data(), n * 8);
		return HAO_OK;
	}
	DevBuf<int64_t> d; HIP_TRY(d.reserve(n + 1));
	HIP_TRY(hipMemcpyAsync(d.p, v, 8 * n, hipMemcpyHostToDevice, c->stream));
	NCCL_TRY(ncclAllReduce(d.p, d.p, n, ncclInt64, ncclSum, cm.nccl, c->stream));
	HIP_TRY(hipMemcpyAsync(v, d.p, 8 * n, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	d.release();
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
static int hao_comm_exchange_counts(hao_ctx *c, hao_comm &cm, const std::vector<uint64_t> &scnt, std::vector<uint64_t> &rcnt)
{
	rcnt.assign(cm.world, 0);
	if (cm.world == 1 && !cm.nccl && !cm.loop) { rcnt[0] = scnt[0]; return HAO_OK; }
	if (cm.loop) {
		hao_loop_group *g = cm.loop;
		g->cnt[cm.rank] = scnt;
		pthread_barrier_wait(&g->bar);
		for (int r = 0; r < cm.world; ++r) rcnt[r] = g->cnt[r][cm.rank];
		pthread_barrier_wait(&g->bar);
		return HAO_OK;
	}
	DevBuf<uint64_t> ds, dr; HIP_TRY(ds.reserve(cm.world + 1)); HIP_TRY(dr.reserve(cm.world + 1));
	HIP_TRY(hipMemcpyAsync(ds.p, scnt.data(), 8 * cm.world, hipMemcpyHostToDevice, c->stream));
	NCCL_TRY(ncclGroupStart());
	for (int r = 0; r < cm.world; ++r) { NCCL_TRY(ncclSend(ds.p + r, 1, ncclUint64, r, cm.nccl, c->stream)); NCCL_TRY(ncclRecv(dr.p + r, 1, ncclUint64, r, cm.nccl, c->stream)); }
	NCCL_TRY(ncclGroupEnd());
	HIP_TRY(hipMemcpyAsync(rcnt.data(), dr.p, 8 * cm.world, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	ds.release(); dr.release();
	return HAO_OK;
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
