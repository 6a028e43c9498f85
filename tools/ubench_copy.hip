// How does this runtime carry a big device-to-host hipMemcpyAsync into pinned memory, and what does it cost a streaming kernel that runs beside it?  (measurement aid,
// not part of the product: the delivery path's question, DESIGN.md 5 "delivery")
//   A  copy alone on a fresh non-blocking stream
//   B  a streaming kernel alone (16 GB read + written, the seed stage's traffic shape)
//   C  the kernel on stream 0, the copy on stream 1 behind hipStreamWaitEvent on an event recorded on stream 0 in front of the kernel (what hao_deliver_queue does)
//   D  the same without the stream dependency: the host waits for the event, then queues the copy
//   E  the copy split into 8 pieces of 128 MB (C's ordering)
// Each case prints the copy's and the kernel's HIP-event times; under `rocprofv3 --kernel-trace --memory-copy-trace` the copy shows either as MEMORY_COPY rows (an SDMA
// engine) or as __amd_rocclr_copyBuffer kernels (a blit kernel on the compute units).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_copy tools/ubench_copy.hip ; run: tools/ubench_copy [copy MB = 1024]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_kernel(const v4 *__restrict__ in, v4 *__restrict__ out, uint64_t n)
{ for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { v4 v = in[i]; v.x += 1; out[i] = v; } }
__global__ void tiny_kernel(uint32_t *p) { if (threadIdx.x == 0) p[0] += 1; }

int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 1024, nb = mb << 20, kn = (size_t)8 << 30;      // kernel: 8 GB in, 8 GB out
	uint8_t *d, *h; v4 *ki, *ko; uint32_t *t;
	CK(hipMalloc(&d, nb)); CK(hipHostMalloc(&h, nb, hipHostMallocDefault)); CK(hipMalloc(&ki, kn)); CK(hipMalloc(&ko, kn)); CK(hipMalloc(&t, 64));
	CK(hipMemset(d, 1, nb)); CK(hipMemset(ki, 2, kn)); CK(hipMemset(t, 0, 64)); memset(h, 0, nb);
	hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	hipEvent_t k0, k1, c0, c1, rdy; CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1)); CK(hipEventCreate(&rdy));
	auto kern = [&]() { CK(hipEventRecord(k0, s0)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, s0, ki, ko, kn / 16); CK(hipEventRecord(k1, s0)); };
	auto copy = [&](int pieces) { CK(hipEventRecord(c0, s1)); for (int p = 0; p < pieces; ++p) CK(hipMemcpyAsync(h + nb / pieces * p, d + nb / pieces * p, nb / pieces, hipMemcpyDeviceToHost, s1)); CK(hipEventRecord(c1, s1)); };
	auto report = [&](const char *name, bool has_k, bool has_c) {
		CK(hipDeviceSynchronize()); float km = 0, cm = 0; if (has_k) CK(hipEventElapsedTime(&km, k0, k1)); if (has_c) CK(hipEventElapsedTime(&cm, c0, c1));
		printf("%-40s kernel %8.2f ms (%6.0f GB/s)   copy %8.2f ms (%5.1f GB/s)\n", name, km, has_k ? 4.0 * 2 * kn / km / 1e6 : 0.0, cm, has_c ? nb / cm / 1e6 : 0.0); fflush(stdout);
	};
	for (int rep = 0; rep < 2; ++rep) {
		copy(1); report("A copy alone", false, true);
		kern(); report("B kernel alone", true, false);
		hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s0, t); CK(hipEventRecord(rdy, s0)); CK(hipStreamWaitEvent(s1, rdy, 0)); kern(); copy(1); report("C kernel | copy behind a stream wait", true, true);
		hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s0, t); CK(hipEventRecord(rdy, s0)); kern(); CK(hipEventSynchronize(rdy)); copy(1); report("D kernel | copy queued after a host wait", true, true);
		hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s0, t); CK(hipEventRecord(rdy, s0)); CK(hipStreamWaitEvent(s1, rdy, 0)); kern(); copy(8); report("E kernel | copy in 8 pieces", true, true);
	}
	return 0;
}
