// TEST INFRASTRUCTURE - a stand-in for <hip/hip_runtime.h> that lets g++ compile the device sources of hifiasm_amd/csrc (*.cuh) for the CPU and run their
// kernels on an emulated gfx950 workgroup, so that the `-m "not gpu"` suite can execute the KERNEL SOURCES (not a restatement of them) against the oracle.
// Nothing in the product includes this file: tests/simt/*.cpp put this directory first on the include path.
//
// Model: one fiber per work-item, one workgroup at a time, one OS thread.  A wave's fibers run in turns: every runnable lane runs until it reaches a cross-lane
// operation (ballot, shuffle, DPP move, readlane, ds_permute - all built on exchange()), a workgroup barrier, or the end of the kernel; when all live lanes of
// the wave stand at the same cross-lane operation the scheduler publishes their operands and the wave goes on.  This is exact for code that keeps cross-lane
// operations in wave-uniform control flow (what the kernels promise in their comments) and it CHECKS that promise: lanes of one wave standing at different
// operations, or some at a barrier and some not, end the launch with an error instead of a silent wrong answer.  LDS / global atomics are plain memory operations
// (single OS thread); a wave's lanes see each other's LDS stores in lane order, waves interleave only at cross-lane operations and barriers - one legal schedule.
// DPP controls, row masks and ds_permute follow the gfx9 ISA for the encodings hao_common.cuh uses; the harnesses pin them by running the unchanged kernels
// against the oracle (the same kernels are green on the device).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <map>
#include <chrono>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cxxabi.h>
#include <string>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(p, order, scope) (*(volatile decltype(p))(p))
#define __hip_atomic_store(p, v, order, scope) (*(volatile decltype(p))(p) = (v))

struct uint2 { unsigned x, y; }; struct uint4 { unsigned x, y, z, w; }; struct int2 { int x, y; }; struct int4 { int x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; }; struct ushort2 { unsigned short x, y; }; struct uchar4 { unsigned char x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
struct dim3 { unsigned x, y, z; constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((uint64_t)hi << 32 | lo) >> (sh & 31)); }

// ---- host API: one "device" = this process's memory, every stream synchronous ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct hao_simt_stream *hipStream_t; typedef struct hao_simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocNumaUser = 0x20000000 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = getenv("HAO_SIMT_CUS") ? atoi(getenv("HAO_SIMT_CUS")) : 3; return hipSuccess; }      // (three "CUs": a persistent kernel's workgroups each run several reads)
inline hipError_t hipDeviceGetPCIBusId(char *b, int len, int) { snprintf(b, len, "0000:00:00.0"); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
// device memory is not zeroed (0xA5 here), and every allocation sits between guard zones that hipFree checks: a kernel that writes before or past its buffer ends the process
namespace hao_simt_mem { constexpr size_t G = 256; struct Hdr { size_t n; size_t magic; }; }
inline hipError_t hipMalloc(void **p, size_t n)
{
	using namespace hao_simt_mem;
	char *b = (char*)malloc(n + 2 * G); if (!b) { *p = nullptr; return hipErrorOutOfMemory; }
	memset(b, 0x5c, G); memset(b + G, getenv("HAO_SIMT_ZERO") ? 0 : 0xa5, n); memset(b + G + n, 0x5c, G);
	Hdr h{n, 0x68616f73696d74ULL}; memcpy(b, &h, sizeof h);
	*p = b + G; return hipSuccess;
}
template<class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void *p)
{
	using namespace hao_simt_mem;
	if (!p) return hipSuccess;
	char *b = (char*)p - G; Hdr h; memcpy(&h, b, sizeof h);
	bool ok = h.magic == 0x68616f73696d74ULL;
	for (size_t i = sizeof h; ok && i < G; ++i) ok = b[i] == (char)0x5c;
	for (size_t i = 0; ok && i < G; ++i) ok = b[G + h.n + i] == (char)0x5c;
	if (!ok) { fprintf(stderr, "tests/simt: a device buffer of %zu bytes was written outside its bounds\n", h.magic == 0x68616f73696d74ULL ? h.n : (size_t)0); abort(); }
	free(b); return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template<class T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipHostRegisterMapped = 2, hipHostRegisterPortable = 1 };
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
template<class T> inline hipError_t hipHostGetDevicePointer(T **d, void *h, unsigned f) { return hipHostGetDevicePointer((void**)d, h, f); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = (size_t)64 << 30; *tot = (size_t)64 << 30; return hipSuccess; }

namespace hao_simt {
enum { ST_NEW = 0, ST_RUN, ST_COLL, ST_BAR, ST_DONE };
constexpr int MAXT = 1024, STACK = 96 * 1024;
struct Fiber { void *sp; char *stack; int state; uint64_t val; const void *site; };
struct Site { const char *file; int line; };
struct Ctx {
	Fiber f[MAXT]; int nthreads = 0, cur = 0; void *sched_sp = nullptr;
	uint64_t snap[MAXT / 64][64]; uint64_t present[MAXT / 64];
	std::function<void()> body; std::vector<char> dyn_lds; std::string error, kernel; bool ascending = getenv("HAO_SIMT_ASCENDING") != nullptr, wave_reverse = getenv("HAO_SIMT_WAVES") && !strcmp(getenv("HAO_SIMT_WAVES"), "reverse"), fair = getenv("HAO_SIMT_FAIR") != nullptr;
	uint64_t n_exchange = 0, n_barrier = 0, n_switch = 0, n_launch = 0, block_serial = 0, or_serial = ~0ULL;
};
inline Ctx g;

// ---- context switch (x86-64 System V: callee-saved registers + stack pointer) ----
extern "C" void hao_simt_switch(void **save_sp, void *load_sp);
#if defined(__x86_64__)
__asm__(R"(
.text
.weak hao_simt_switch
.type hao_simt_switch,@function
hao_simt_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
.size hao_simt_switch,.-hao_simt_switch
)");
#else
#error "tests/simt: context switch written for x86-64 only"
#endif

// a work-item that cannot go on (it stands at a cross-lane operation or a barrier, or has ended) hands the processor to the next lane of its wave that can run
// (lanes run from the highest down, see run_block), or to the scheduler when there is none
inline void to_scheduler()
{
	++g.n_switch;
	const int me = g.cur, t0 = me & ~63, t1 = std::min(g.nthreads, t0 + 64);
	for (int t = g.ascending ? me + 1 : me - 1; g.ascending ? t < t1 : t >= t0; t += g.ascending ? 1 : -1) if (g.f[t].state == ST_NEW || g.f[t].state == ST_RUN) {
		g.cur = t; threadIdx.x = (unsigned)t; if (g.f[t].state == ST_NEW) g.f[t].state = ST_RUN;
		hao_simt_switch(&g.f[me].sp, g.f[t].sp); return;
	}
	hao_simt_switch(&g.f[me].sp, g.sched_sp);
}
[[noreturn]] inline void fiber_main() { g.body(); g.f[g.cur].state = ST_DONE; to_scheduler(); abort(); }
inline void fiber_init(Fiber &F)
{
	if (!F.stack) F.stack = (char*)aligned_alloc(64, STACK);
	// stack as hao_simt_switch's `ret` expects it: six saved registers, then the entry address; after the ret the stack pointer is 8 mod 16 like after a call
	uintptr_t top = ((uintptr_t)F.stack + STACK) & ~(uintptr_t)15;
	void **sp = (void**)(top - 8);
	*--sp = (void*)&fiber_main;
	for (int i = 0; i < 6; ++i) *--sp = nullptr;
	F.sp = sp; F.state = ST_NEW; F.val = 0; F.site = nullptr;
}

// every cross-lane operation: deposit an operand, get back the operands of all live lanes of my wave (and which lanes those are)
inline const void *site_of(const char *file, int line)      // one record per source position, so that a mismatch can be reported by file and line
{
	static std::vector<Site*> tab; for (Site *t : tab) if (t->file == file && t->line == line) return t;
	tab.push_back(new Site{file, line}); return tab.back();
}
// site: the source position of the operation (file, line) - the return address would differ between the copies an optimising compiler makes of one call
#define HAO_SIMT_SITE_ARGS int line_ = __builtin_LINE(), const char *file_ = __builtin_FILE()
#define HAO_SIMT_SITE (hao_simt::site_of(file_, line_))
__attribute__((noinline)) inline const uint64_t *exchange(uint64_t v, uint64_t *present, const void *site)
{
	Fiber &me = g.f[g.cur]; me.val = v; me.site = site; me.state = ST_COLL;
	to_scheduler();
	const int w = g.cur >> 6; *present = g.present[w]; return g.snap[w];
}
inline void barrier() { g.f[g.cur].state = ST_BAR; to_scheduler(); }

// one workgroup; returns false (and sets g.error) on divergence / deadlock
inline bool run_block()
{
	const int nt = g.nthreads, nw = (nt + 63) / 64;
	++g.block_serial;
	for (int t = 0; t < nt; ++t) fiber_init(g.f[t]);
	auto resume = [&](int t) { g.cur = t; threadIdx.x = (unsigned)t; threadIdx.y = threadIdx.z = 0; if (g.f[t].state == ST_NEW) g.f[t].state = ST_RUN; hao_simt_switch(&g.sched_sp, g.f[t].sp); };
	for (;;) {
		bool progress = false; int live = 0, at_bar = 0;
		for (int wi = 0; wi < nw; ++wi) {
			const int w = g.wave_reverse ? nw - 1 - wi : wi;      // HAO_SIMT_WAVES=reverse / HAO_SIMT_FAIR=1: other legal interleavings of a workgroup's waves (a missing barrier shows)
			const int t0 = w * 64, t1 = std::min(nt, t0 + 64);
			for (bool again = true; again; ) {
				again = false;
				// (highest lane first: in the common single-writer idiom - every lane reads, then `if (lane == 0)` or the first lane of a group writes - the writer runs
				// last, as if in lockstep; the places where that is not enough carry HAO_LOCKSTEP() in the sources)
				if (!g.ascending) { for (int t = t1 - 1; t >= t0; --t) if (g.f[t].state == ST_NEW || g.f[t].state == ST_RUN) { resume(t); progress = true; } }
				else for (int t = t0; t < t1; ++t) if (g.f[t].state == ST_NEW || g.f[t].state == ST_RUN) { resume(t); progress = true; }      // HAO_SIMT_ASCENDING=1: which kernels lean on the lane order?
				int nl = 0, nc = 0, nb = 0; const void *site = nullptr; bool same = true;
				for (int t = t0; t < t1; ++t) {
					const int s = g.f[t].state; if (s == ST_DONE) continue; ++nl;
					if (s == ST_COLL) { if (nc++ == 0) site = g.f[t].site; else if (g.f[t].site != site) same = false; }
					else if (s == ST_BAR) ++nb;
				}
				if (nc && (nb || !same)) {
					char b[300]; snprintf(b, sizeof b, "%s block %u wave %d: lanes stand at different cross-lane operations / barriers (%d at a barrier;", g.kernel.c_str(), blockIdx.x, w, nb); g.error = b;
					std::vector<std::pair<const Site*, int>> seen;
					for (int t = t0; t < t1; ++t) if (g.f[t].state == ST_COLL) { const Site *st = (const Site*)g.f[t].site; bool f = false; for (auto &x : seen) if (x.first == st) { ++x.second; f = true; } if (!f) seen.push_back({st, 1}); }
					for (auto &x : seen) { const char *fn = strrchr(x.first->file, '/'); snprintf(b, sizeof b, " %d at %s:%d", x.second, fn ? fn + 1 : x.first->file, x.first->line); g.error += b; }
					g.error += ")"; return false;
				}
				if (nc && nc == nl) {      // publish the operands; the wave goes on
					uint64_t pr = 0; for (int t = t0; t < t1; ++t) if (g.f[t].state == ST_COLL) { pr |= 1ULL << (t - t0); g.snap[w][t - t0] = g.f[t].val; g.f[t].state = ST_RUN; } else g.snap[w][t - t0] = 0;
					g.present[w] = pr; ++g.n_exchange; again = !g.fair; progress = true;
				}
			}
			for (int t = t0; t < t1; ++t) { const int s = g.f[t].state; if (s != ST_DONE) { ++live; if (s == ST_BAR) ++at_bar; } }
		}
		if (!live) return true;
		if (at_bar == live) { for (int t = 0; t < nt; ++t) if (g.f[t].state == ST_BAR) g.f[t].state = ST_RUN; ++g.n_barrier; progress = true; }
		if (!progress) { char b[200]; snprintf(b, sizeof b, "%s block %u: no work-item can run (%d live, %d at the barrier): a cross-lane operation or barrier inside divergent control flow", g.kernel.c_str(), blockIdx.x, live, at_bar); g.error = b; return false; }
	}
}

// launch<<<grid, block, dyn_lds>>>: `call` invokes the kernel function with its arguments (it runs once per work-item)
inline int launch(dim3 grid, dim3 block, size_t dyn_lds, std::function<void()> call)
{
	static std::recursive_mutex mu; std::lock_guard<std::recursive_mutex> lk(mu);      // host threads of the library launch one at a time
	if (block.y != 1 || block.z != 1 || block.x > (unsigned)MAXT) { g.error = "block shape not modelled"; return 1; }
	g.body = std::move(call); g.nthreads = (int)block.x; g.error.clear();
	g.dyn_lds.assign(dyn_lds + 64, (char)0xa5);      // LDS is not zeroed on the device either
	memset(g.dyn_lds.data() + dyn_lds, 0x5c, 64);      // ... and a kernel that writes past the dynamic LDS it asked for is caught by the guard words behind it
	blockDim = block; gridDim = grid;
	for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned b = 0; b < grid.x; ++b) {
		blockIdx = dim3(b, by, bz); if (!run_block()) return 1;
		for (int i = 0; i < 64; ++i) if (g.dyn_lds[dyn_lds + i] != (char)0x5c) { g.error = g.kernel + ": a work-item wrote past the end of the dynamic LDS"; return 1; }
	}
	++g.n_launch;
	return 0;
}
inline int launch(unsigned grid, unsigned block, size_t dyn_lds, std::function<void()> call) { return launch(dim3(grid), dim3(block), dyn_lds, std::move(call)); }
inline void *dyn_lds() { return g.dyn_lds.data(); }
// HAO_SIMT_TRACE=1: every launch is printed, and SIGTERM / SIGSEGV print where the emulation stands (kernel, block, work-item, backtrace of its fiber)
inline void on_signal(int sig)
{
	void *bt[48]; const int n = backtrace(bt, 48);
	fprintf(stderr, "[simt] signal %d in %s, block %u, work-item %d\n", sig, g.kernel.c_str(), blockIdx.x, g.cur);
	backtrace_symbols_fd(bt, n, 2); _exit(3);
}
// HAO_SIMT_PROF=1: seconds and launches per kernel at exit
struct Prof { std::map<std::string, std::pair<double, uint64_t>> t; ~Prof() { if (!getenv("HAO_SIMT_PROF")) return; for (auto &x : t) fprintf(stderr, "[simt prof] %9.3f s %8llu launches  %s\n", x.second.first, (unsigned long long)x.second.second, x.first.c_str());
	fprintf(stderr, "[simt prof] total: %llu cross-lane operations, %llu barriers, %llu fiber switches, %llu launches\n", (unsigned long long)g.n_exchange, (unsigned long long)g.n_barrier, (unsigned long long)g.n_switch, (unsigned long long)g.n_launch); } };
inline Prof prof;
}      // namespace hao_simt
// hipLaunchKernelGGL(kernel, grid, block, dynamic LDS, stream, args...): the kernel runs to completion here; an emulation error (divergent cross-lane operation,
// barrier deadlock) is fatal - a wrong answer must not look like a kernel's
template<class K, class... A> inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t, A... args)
{
	{ Dl_info di; int st_ = 0; const char *nm = dladdr((void*)kernel, &di) && di.dli_sname ? di.dli_sname : "?"; char *dm = abi::__cxa_demangle(nm, nullptr, nullptr, &st_); std::string k = dm ? dm : nm; free(dm); hao_simt::g.kernel = k.substr(0, k.find('(')); }
	if (getenv("HAO_SIMT_TRACE")) { static bool inst = false; if (!inst) { inst = true; signal(SIGTERM, hao_simt::on_signal); signal(SIGSEGV, hao_simt::on_signal); } fprintf(stderr, "[simt] %s <<<%u, %u, %zu>>>\n", hao_simt::g.kernel.c_str(), grid.x, block.x, dyn_lds); }
	const auto t0_ = std::chrono::steady_clock::now();
	struct Acc { std::chrono::steady_clock::time_point t0; ~Acc() { auto &e = hao_simt::prof.t[hao_simt::g.kernel]; e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++e.second; } } acc_{t0_};
	if (hao_simt::launch(grid, block, dyn_lds, [&] { kernel(args...); })) { fprintf(stderr, "tests/simt: %s\n", hao_simt::g.error.c_str()); abort(); }
}

// ---- the device vocabulary the sources use ----
template<class T> __forceinline__ T min(T a, T b) { return b < a ? b : a; }
template<class T> __forceinline__ T max(T a, T b) { return a < b ? b : a; }
__forceinline__ uint32_t min(uint32_t a, int b) { return min<uint32_t>(a, (uint32_t)b); }
__forceinline__ uint32_t min(int a, uint32_t b) { return min<uint32_t>((uint32_t)a, b); }
__forceinline__ uint64_t min(uint64_t a, uint32_t b) { return min<uint64_t>(a, b); }
__forceinline__ uint64_t min(uint32_t a, uint64_t b) { return min<uint64_t>(a, b); }
__forceinline__ int64_t min(int64_t a, int b) { return min<int64_t>(a, b); }
__forceinline__ int64_t max(int64_t a, int b) { return max<int64_t>(a, b); }
__forceinline__ int64_t min(int a, int64_t b) { return min<int64_t>(a, b); }
__forceinline__ int64_t max(int a, int64_t b) { return max<int64_t>(a, b); }
__forceinline__ uint32_t max(uint32_t a, int b) { return max<uint32_t>(a, (uint32_t)b); }

__forceinline__ void __syncthreads() { hao_simt::barrier(); }
// barrier + OR of a predicate over the workgroup: two accumulators used in turn (a work-item clears the one of the previous call after the barrier of this one:
// everybody has read it by then)
inline int __syncthreads_or(int pred)
{
	static unsigned acc[2]; static int gen[hao_simt::MAXT];
	if (hao_simt::g.block_serial != hao_simt::g.or_serial) { hao_simt::g.or_serial = hao_simt::g.block_serial; acc[0] = acc[1] = 0; memset(gen, 0, sizeof gen); }
	const int k = gen[threadIdx.x]++ & 1;
	if (pred) acc[k] = 1;
	hao_simt::barrier();
	const int r = (int)acc[k]; acc[k ^ 1] = 0;
	return r;
}
// (__threadfence / __threadfence_block: with the other fences below - a rendezvous of the wave's lanes, because that is what the sources use them for)
__forceinline__ unsigned long long wall_clock64() { return 0; }
__forceinline__ int __popc(unsigned x) { return __builtin_popcount(x); }
__forceinline__ int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
__forceinline__ int __ffsll(long long x) { return __builtin_ffsll(x); }
__forceinline__ int __ffs(int x) { return __builtin_ffs(x); }
__forceinline__ int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
__forceinline__ int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
__forceinline__ unsigned long long __brevll(unsigned long long x) { unsigned long long r = 0; for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ULL) << (63 - i); return r; }
__forceinline__ unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i); return r; }

template<class T, class U> __forceinline__ T atomicAdd(T *p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template<class T, class U> __forceinline__ T atomicSub(T *p, U v) { const T o = *p; *p = (T)(o - (T)v); return o; }
template<class T, class U> __forceinline__ T atomicMax(T *p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template<class T, class U> __forceinline__ T atomicMin(T *p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template<class T, class U> __forceinline__ T atomicOr(T *p, U v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template<class T, class U> __forceinline__ T atomicAnd(T *p, U v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template<class T, class U> __forceinline__ T atomicExch(T *p, U v) { const T o = *p; *p = (T)v; return o; }
template<class T, class U, class V> __forceinline__ T atomicCAS(T *p, U cmp, V v) { const T o = *p; if (o == (T)cmp) *p = (T)v; return o; }

namespace hao_simt {
template<class T> __forceinline__ uint64_t bits_of(T v) { static_assert(sizeof(T) <= 8, "cross-lane operand"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template<class T> __forceinline__ T of_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
__forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
}
__forceinline__ unsigned long long __ballot(int pred, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange(pred ? 1 : 0, &pr, HAO_SIMT_SITE); unsigned long long m = 0;
	for (int i = 0; i < 64; ++i) if ((pr >> i & 1) && s[i]) m |= 1ULL << i;
	return m;
}
#define HAO_SLOAD_U32(dst, ptr) ((dst) = *(ptr))      /* hao_common.cuh: a scalar load (inline assembly on the device) */
#define HAO_OPAQUE_U32(x) ((void)0)      /* hao_common.cuh: a compiler barrier on the device */
#define HAO_LOCKSTEP() do { uint64_t pr_; (void)hao_simt::exchange(0, &pr_, hao_simt::site_of(__FILE__, __LINE__)); } while (0)      /* hao_common.cuh: lanes of a wave run in lockstep */
// the sources use release + acquire fence pairs where a wave's lanes hand data to each other through memory: the release is the rendezvous of the wave's lanes
__forceinline__ void __builtin_amdgcn_fence(int order, const char *, HAO_SIMT_SITE_ARGS) { if (order != __ATOMIC_ACQUIRE) { uint64_t pr; (void)hao_simt::exchange(0, &pr, HAO_SIMT_SITE); } }
__forceinline__ void __threadfence(HAO_SIMT_SITE_ARGS) { uint64_t pr; (void)hao_simt::exchange(0, &pr, HAO_SIMT_SITE); }
__forceinline__ void __threadfence_block(HAO_SIMT_SITE_ARGS) { uint64_t pr; (void)hao_simt::exchange(0, &pr, HAO_SIMT_SITE); }
__forceinline__ int __any(int pred, HAO_SIMT_SITE_ARGS) { return __ballot(pred, line_, file_) != 0; }
__forceinline__ int __all(int pred, HAO_SIMT_SITE_ARGS) { uint64_t pr; const uint64_t *s = hao_simt::exchange(pred ? 1 : 0, &pr, HAO_SIMT_SITE); for (int i = 0; i < 64; ++i) if ((pr >> i & 1) && !s[i]) return 0; return 1; }
template<class T> __forceinline__ T __shfl(T v, int src, int width = 64, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange(hao_simt::bits_of(v), &pr, HAO_SIMT_SITE); const int me = hao_simt::lane_id(), base = me & ~(width - 1), k = base + (src & (width - 1));
	return (pr >> k & 1) ? hao_simt::of_bits<T>(s[k]) : v;
}
template<class T> __forceinline__ T __shfl_up(T v, unsigned d, int width = 64, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange(hao_simt::bits_of(v), &pr, HAO_SIMT_SITE); const int me = hao_simt::lane_id(), k = me - (int)d;
	return (k >= (me & ~(width - 1)) && (pr >> k & 1)) ? hao_simt::of_bits<T>(s[k]) : v;
}
template<class T> __forceinline__ T __shfl_down(T v, unsigned d, int width = 64, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange(hao_simt::bits_of(v), &pr, HAO_SIMT_SITE); const int me = hao_simt::lane_id(), k = me + (int)d;
	return (k < (me & ~(width - 1)) + width && k < 64 && (pr >> k & 1)) ? hao_simt::of_bits<T>(s[k]) : v;
}
template<class T> __forceinline__ T __shfl_xor(T v, int x, int width = 64, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange(hao_simt::bits_of(v), &pr, HAO_SIMT_SITE); const int k = hao_simt::lane_id() ^ x;
	return (k < 64 && (pr >> k & 1)) ? hao_simt::of_bits<T>(s[k]) : v;
}
__forceinline__ int __builtin_amdgcn_readlane(int v, int lane, HAO_SIMT_SITE_ARGS) { uint64_t pr; const uint64_t *s = hao_simt::exchange((uint32_t)v, &pr, HAO_SIMT_SITE); return (int)(uint32_t)s[lane & 63]; }
__forceinline__ int __builtin_amdgcn_readfirstlane(int v, HAO_SIMT_SITE_ARGS) { uint64_t pr; const uint64_t *s = hao_simt::exchange((uint32_t)v, &pr, HAO_SIMT_SITE); return (int)(uint32_t)s[__builtin_ctzll(pr)]; }
// v_mov_b32_dpp (gfx9): row_shr:n 0x110+n, row_shl:n 0x100+n, wave_shl:1 0x130, wave_shr:1 0x138, row_bcast:15 0x142, row_bcast:31 0x143, quad_perm 0x00-0xff;
// a lane whose row row_mask disables, whose bank bank_mask disables or whose source does not exist keeps `old` (bound_ctrl: a missing source reads 0 instead)
__forceinline__ int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange((uint32_t)v, &pr, HAO_SIMT_SITE); const int me = hao_simt::lane_id(), row = me >> 4, inrow = me & 15;
	int src = -1;
	if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; if (inrow + n < 16) src = me + n; }
	else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (inrow >= n) src = me - n; }
	else if (ctrl == 0x130) { if (me < 63) src = me + 1; }
	else if (ctrl == 0x138) { if (me > 0) src = me - 1; }
	else if (ctrl == 0x142) { if (row >= 1) src = row * 16 - 1; }
	else if (ctrl == 0x143) { if (row >= 2) src = 31; }
	else if (ctrl >= 0 && ctrl <= 0xff) src = (me & ~3) | (ctrl >> (2 * (me & 3)) & 3);
	else { fprintf(stderr, "tests/simt: DPP control 0x%x not modelled\n", ctrl); abort(); }
	if (!(row_mask >> row & 1) || !(bank_mask >> (inrow >> 2) & 1)) return old;
	if (src < 0 || !(pr >> src & 1)) return bound_ctrl ? 0 : old;
	return (int)(uint32_t)s[src];
}
// ds_permute_b32: lane i pushes v to lane (addr / 4) % 64; a lane nobody pushes to reads 0 (several pushes to one lane: the highest lane's stays)
__forceinline__ int __builtin_amdgcn_ds_permute(int addr, int v, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange((uint64_t)(uint32_t)addr << 32 | (uint32_t)v, &pr, HAO_SIMT_SITE); const int me = hao_simt::lane_id(); int r = 0;
	for (int i = 0; i < 64; ++i) if ((pr >> i & 1) && (int)((s[i] >> 34) & 63) == me) r = (int)(uint32_t)s[i];
	return r;
}
// ds_bpermute_b32: lane i pulls from lane (addr / 4) % 64
__forceinline__ int __builtin_amdgcn_ds_bpermute(int addr, int v, HAO_SIMT_SITE_ARGS)
{
	uint64_t pr; const uint64_t *s = hao_simt::exchange((uint32_t)v, &pr, HAO_SIMT_SITE); const int k = (addr >> 2) & 63;
	return (pr >> k & 1) ? (int)(uint32_t)s[k] : 0;
}
// v_mbcnt_lo / v_mbcnt_hi: base + the mask's bits below this lane (lo: lanes 0-31 of the mask, hi: lanes 32-63)
__forceinline__ unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) { const int me = hao_simt::lane_id(); return base + (unsigned)__builtin_popcount(me >= 32 ? mask : mask & ((1u << me) - 1)); }
__forceinline__ unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) { const int me = hao_simt::lane_id(); return base + (me <= 32 ? 0u : (unsigned)__builtin_popcount(mask & ((1u << (me - 32)) - 1))); }
__forceinline__ int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width) { const unsigned sh = 32 - width; return (int)((unsigned)v >> off << sh) >> sh; }
__forceinline__ unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned width) { return width >= 32 ? v >> off : (v >> off) & ((1u << width) - 1); }
// v_bitop3_b32: bit i of the result = bit (a_i << 2 | b_i << 1 | c_i) of the truth table
__forceinline__ uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, unsigned tt)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; ++i) { const unsigned ix = (a >> i & 1) << 2 | (b >> i & 1) << 1 | (c >> i & 1); r |= (uint32_t)(tt >> ix & 1) << i; }
	return r;
}
