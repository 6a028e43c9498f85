// TEST INFRASTRUCTURE: stand-in for <rccl/rccl.h> in the CPU emulation build (tests/simt): the subset hao_comm.hpp uses, between PROCESSES on one host.
// One process = one rank (as with RCCL); "device" buffers are this process's memory.  The transport is a mailbox directory (named in the unique id rank 0 creates):
// a send writes <dir>/<src>_<dst>_<sequence number> and publishes it with a rename, a receive waits for the next file of its (source, destination) pair, reads and
// removes it.  Sends never block, so every pattern of grouped sends and receives completes in any order (ncclGroupStart / End have nothing to do); collectives are
// sends to and receives from every peer.  Semantics kept from RCCL: counts in elements of the data type, in-place all-gather (send buffer = receive buffer +
// rank * count), broadcast per root, sum all-reduce on int64.  A receive that waits longer than HAO_SIMT_RCCL_TIMEOUT seconds (default 300) fails.
// tests/test_dist_cpu.py runs tests/rccl_worker.py --simt under torch.distributed.run (gloo for the launcher side) with 2 and 4 ranks: the RCCL branch of
// hao_comm.hpp - count exchange, all-to-all-v, reverse all-to-all-v, balanced and per-root all-gather-v, histogram all-reduce - between real processes.
#pragma once
#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>

struct hao_simt_comm { int rank, world; std::string dir; std::vector<uint64_t> seq_out, seq_in; };
typedef struct hao_simt_comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
inline size_t hao_simt_nccl_size(ncclDataType_t t) { return t == ncclChar ? 1 : 8; }
inline const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSystemError ? "emulated rccl: mailbox I/O failed or a receive timed out" : "emulated rccl"; }
#define NCCL_MAJOR 2
inline ncclResult_t ncclGetVersion(int *v) { *v = 22605; return ncclSuccess; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	memset(id, 0, sizeof *id);
	const char *base = getenv("TMPDIR"); if (!base || !*base) base = "/tmp";
	snprintf(id->internal, sizeof id->internal, "%s/hao_simt_rccl_%d_%lx", base, (int)getpid(), (unsigned long)time(nullptr));
	return mkdir(id->internal, 0700) == 0 || errno == EEXIST ? ncclSuccess : ncclSystemError;
}
inline ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId id, int rank)
{
	if (n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
	id.internal[sizeof id.internal - 1] = 0;
	hao_simt_comm *m = new hao_simt_comm{rank, n, std::string(id.internal), std::vector<uint64_t>((size_t)n, 0), std::vector<uint64_t>((size_t)n, 0)};
	if (n > 1 && m->dir.empty()) { delete m; return ncclInvalidArgument; }
	*c = m; return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { if (c->rank == 0 && c->world > 1) rmdir(c->dir.c_str()); delete c; } return ncclSuccess; }      // (the directory is empty once every message was received)
inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }

inline ncclResult_t hao_simt_nccl_put(hao_simt_comm *c, int dst, const void *p, size_t bytes)
{
	char tmp[256], fin[256];
	snprintf(fin, sizeof fin, "%s/%d_%d_%llu", c->dir.c_str(), c->rank, dst, (unsigned long long)c->seq_out[dst]);
	snprintf(tmp, sizeof tmp, "%s.part", fin);
	++c->seq_out[dst];
	FILE *f = fopen(tmp, "wb"); if (!f) return ncclSystemError;
	const bool ok = (bytes == 0 || fwrite(p, 1, bytes, f) == bytes);
	if (fclose(f) != 0 || !ok) return ncclSystemError;
	return rename(tmp, fin) == 0 ? ncclSuccess : ncclSystemError;
}
inline ncclResult_t hao_simt_nccl_get(hao_simt_comm *c, int src, void *p, size_t bytes)
{
	char fin[256];
	snprintf(fin, sizeof fin, "%s/%d_%d_%llu", c->dir.c_str(), src, c->rank, (unsigned long long)c->seq_in[src]);
	++c->seq_in[src];
	const char *te = getenv("HAO_SIMT_RCCL_TIMEOUT"); const double limit = te ? atof(te) : 300.0;
	struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
	for (;;) {
		struct stat sb;
		if (stat(fin, &sb) == 0) {
			if ((size_t)sb.st_size != bytes) return ncclInvalidArgument;      // the two sides disagree about the count
			FILE *f = fopen(fin, "rb"); if (!f) return ncclSystemError;
			const bool ok = (bytes == 0 || fread(p, 1, bytes, f) == bytes);
			fclose(f); unlink(fin);
			return ok ? ncclSuccess : ncclSystemError;
		}
		struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
		if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) return ncclSystemError;
		usleep(200);
	}
}
// one rank: a send is matched by the next receive (kept from the single-process stand-in: no files)
struct hao_simt_nccl_pending { const void *p; size_t n; };
inline hao_simt_nccl_pending &hao_simt_nccl_slot() { static hao_simt_nccl_pending s{nullptr, 0}; return s; }
template<class S> inline ncclResult_t ncclSend(const void *p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, S)
{
	if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
	if (c->world == 1) { auto &s = hao_simt_nccl_slot(); if (s.p && s.p != p) return ncclInvalidArgument; s.p = p; s.n = n * hao_simt_nccl_size(t); return ncclSuccess; }
	return hao_simt_nccl_put(c, peer, p, n * hao_simt_nccl_size(t));
}
template<class S> inline ncclResult_t ncclRecv(void *p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, S)
{
	if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
	if (c->world == 1) { auto &s = hao_simt_nccl_slot(); if (!s.p || s.n != n * hao_simt_nccl_size(t)) return ncclInvalidArgument; memmove(p, s.p, s.n); s.p = nullptr; return ncclSuccess; }
	return hao_simt_nccl_get(c, peer, p, n * hao_simt_nccl_size(t));
}
template<class S> inline ncclResult_t ncclAllGather(const void *in, void *out, size_t n, ncclDataType_t t, ncclComm_t c, S)
{
	const size_t b = n * hao_simt_nccl_size(t);
	if ((char*)out + b * c->rank != (const char*)in) memmove((char*)out + b * c->rank, in, b);      // (in place when in = out + rank * count)
	for (int r = 0; r < c->world; ++r) if (r != c->rank) { ncclResult_t e = hao_simt_nccl_put(c, r, in, b); if (e != ncclSuccess) return e; }
	for (int r = 0; r < c->world; ++r) if (r != c->rank) { ncclResult_t e = hao_simt_nccl_get(c, r, (char*)out + b * r, b); if (e != ncclSuccess) return e; }
	return ncclSuccess;
}
template<class S> inline ncclResult_t ncclBroadcast(const void *in, void *out, size_t n, ncclDataType_t t, int root, ncclComm_t c, S)
{
	const size_t b = n * hao_simt_nccl_size(t);
	if (root < 0 || root >= c->world) return ncclInvalidArgument;
	if (c->rank == root) {
		if (out != in) memmove(out, in, b);
		for (int r = 0; r < c->world; ++r) if (r != root) { ncclResult_t e = hao_simt_nccl_put(c, r, in, b); if (e != ncclSuccess) return e; }
		return ncclSuccess;
	}
	return hao_simt_nccl_get(c, root, out, b);
}
template<class S> inline ncclResult_t ncclAllReduce(const void *in, void *out, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, S)
{
	if (t != ncclInt64 && t != ncclUint64) return ncclInvalidArgument;
	std::vector<int64_t> mine((const int64_t*)in, (const int64_t*)in + n), acc(mine), got(n);
	for (int r = 0; r < c->world; ++r) if (r != c->rank) { ncclResult_t e = hao_simt_nccl_put(c, r, mine.data(), n * 8); if (e != ncclSuccess) return e; }
	for (int r = 0; r < c->world; ++r) if (r != c->rank) { ncclResult_t e = hao_simt_nccl_get(c, r, got.data(), n * 8); if (e != ncclSuccess) return e; for (size_t i = 0; i < n; ++i) acc[i] += got[i]; }
	memcpy(out, acc.data(), n * 8);
	return ncclSuccess;
}
