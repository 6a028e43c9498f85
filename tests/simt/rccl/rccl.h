// TEST INFRASTRUCTURE: stand-in for <rccl/rccl.h> in the CPU emulation build (tests/simt).  One process = one rank: collectives over a single rank are copies;
// anything that needs a second rank fails.
#pragma once
#include <string.h>
#include <stddef.h>
typedef struct hao_simt_comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
inline size_t hao_simt_nccl_size(ncclDataType_t t) { return t == ncclChar ? 1 : 8; }
inline const char *ncclGetErrorString(ncclResult_t) { return "emulated rccl"; }
#define NCCL_MAJOR 2
inline ncclResult_t ncclGetVersion(int *v) { *v = 22605; return ncclSuccess; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0, sizeof *id); return ncclSuccess; }
inline ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId, int) { *c = (ncclComm_t)1; return n == 1 ? ncclSuccess : ncclInvalidArgument; }
inline ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }
// a single rank sends to / receives from itself: the pending send is matched by the next receive
struct hao_simt_nccl_pending { const void *p; size_t n; };
inline hao_simt_nccl_pending &hao_simt_nccl_slot() { static hao_simt_nccl_pending s{nullptr, 0}; return s; }
template<class S> inline ncclResult_t ncclSend(const void *p, size_t n, ncclDataType_t t, int, ncclComm_t, S) { auto &s = hao_simt_nccl_slot(); if (s.p && s.p != p) { return ncclInvalidArgument; } s.p = p; s.n = n * hao_simt_nccl_size(t); return ncclSuccess; }
template<class S> inline ncclResult_t ncclRecv(void *p, size_t n, ncclDataType_t t, int, ncclComm_t, S) { auto &s = hao_simt_nccl_slot(); if (!s.p || s.n != n * hao_simt_nccl_size(t)) return ncclInvalidArgument; memmove(p, s.p, s.n); s.p = nullptr; return ncclSuccess; }
template<class S> inline ncclResult_t ncclAllGather(const void *in, void *out, size_t n, ncclDataType_t t, ncclComm_t, S) { memmove(out, in, n * hao_simt_nccl_size(t)); return ncclSuccess; }
template<class S> inline ncclResult_t ncclBroadcast(const void *in, void *out, size_t n, ncclDataType_t t, int, ncclComm_t, S) { memmove(out, in, n * hao_simt_nccl_size(t)); return ncclSuccess; }
template<class S> inline ncclResult_t ncclAllReduce(const void *in, void *out, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t, S) { memmove(out, in, n * hao_simt_nccl_size(t)); return ncclSuccess; }
