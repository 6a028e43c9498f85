// What does the memory system give the SEED STAGE's access pattern when no computation is attached to it?  (measurement aid, not part of the product)
// The stage reads ~36 M position lists of ~28 eight-byte records each, scattered over a 2.3 GB index, and writes one 16-byte k_mer_hit per record to a sequential
// 16 GB array (configs[2]: 993 M records per launch).  This program does exactly that and nothing else, in the two read shapes the engine's kernels use:
//   rows  : a lane walks ITS OWN list with 32-byte reads (four records), 8 lists per lane, lists of a wave interleaved step by step (round 5's seed_merge_kernel<8, 4>: every read
//           moves a line nobody else in the wave wants);
//   lists : adjacent lanes read adjacent records of one list after the other, each record read once (one pass of the table kernels' walk; they walk twice);
// and, as the yardstick, `copy`: the same bytes (8 in, 16 out per record) streamed from and to sequential addresses.  Output positions are sequential per wave in all
// three (both engines write their hits coalesced), so the difference between the lines is the gather alone.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip ; run: tools/ubench_gather [records_in_millions=993] [index_in_millions=290]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct hit { uint32_t a, b, c, d; };
struct rec4 { uint64_t a, b, c, d; };
#define LIST_LEN 28u      // records per list (30x coverage: a k-mer's position list holds one record per read that carries it)

__global__ void fill_kernel(uint64_t *p, uint64_t n) { for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ULL; }

// rows: a wave = 512 lists (8 per lane); step s: every lane reads four records of each of its lists and writes them
__global__ __launch_bounds__(256, 3) void rows_kernel(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ start, uint64_t n_lists, hit *out)
{
	const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
	const uint64_t l0 = wave * 512;
	if (l0 >= n_lists) return;
	uint64_t st[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { const uint64_t l = l0 + i * 64 + lane; st[i] = l < n_lists ? start[l] : start[0]; }
	hit *o = out + l0 * LIST_LEN;
	for (uint32_t s = 0; s < LIST_LEN; s += 4) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const rec4 v = *(const rec4*)(idx + st[i] + s);
			// (the merge kernel's stores: the hits of a step leave in row order, consecutive lanes -> consecutive 16-byte slots)
			hit *q = o + ((uint64_t)(s / 4) * 8 + i) * 256 + lane * 4;
			q[0] = hit{(uint32_t)v.a, (uint32_t)(v.a >> 32), 1, 2}; q[1] = hit{(uint32_t)v.b, (uint32_t)(v.b >> 32), 1, 2};
			q[2] = hit{(uint32_t)v.c, (uint32_t)(v.c >> 32), 1, 2}; q[3] = hit{(uint32_t)v.d, (uint32_t)(v.d >> 32), 1, 2};
		}
	}
}
// lists: a wave takes 512 lists too, one after the other: lane j reads record j of the list (28 of 64 lanes busy per list: two lists per pass of the wave)
__global__ __launch_bounds__(256) void lists_kernel(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ start, uint64_t n_lists, hit *out)
{
	const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
	const uint64_t l0 = wave * 512;
	if (l0 >= n_lists) return;
	hit *o = out + l0 * LIST_LEN;
	for (uint32_t k = 0; k < 512; k += 2) {      // two lists per step: lanes 0-27 and 32-59
		const uint64_t l = l0 + k + (lane >> 5); const uint32_t j = lane & 31;
		if (l < n_lists && j < LIST_LEN) { const uint64_t v = idx[start[l] + j]; o[(uint64_t)(k + (lane >> 5)) * LIST_LEN + j] = hit{(uint32_t)v, (uint32_t)(v >> 32), 1, 2}; }
	}
}
__global__ __launch_bounds__(256) void copy_kernel(const uint64_t *__restrict__ idx, uint64_t n, uint64_t n_idx, hit *out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint64_t v = idx[i % n_idx]; out[i] = hit{(uint32_t)v, (uint32_t)(v >> 32), 1, 2}; }
}

// copy16: the yardstick without the modulo and with 16-byte accesses (round 5's copy_kernel did a 64-bit % per element and 8-byte loads and read 4.36 TB/s; the guide
// measures 6.29 TB/s for a float4 copy): 16 bytes in (two records from a sequential array as large as the reads need), 32 bytes out per work-item
__global__ __launch_bounds__(256) void copy16_kernel(const ulonglong2 *__restrict__ in, uint64_t n2, hit *out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) {
		const ulonglong2 v = in[i];
		out[2 * i] = hit{(uint32_t)v.x, (uint32_t)(v.x >> 32), 1, 2}; out[2 * i + 1] = hit{(uint32_t)v.y, (uint32_t)(v.y >> 32), 1, 2};
	}
}

int main(int argc, char **argv)
{
	const uint64_t n_rec = (uint64_t)(argc > 1 ? atof(argv[1]) : 993.0) * 1000000ULL, n_idx = (uint64_t)(argc > 2 ? atof(argv[2]) : 290.0) * 1000000ULL;
	const uint64_t n_lists = n_rec / LIST_LEN, n_out = n_lists * LIST_LEN + 512 * LIST_LEN;
	uint64_t *idx, *start; hit *out;
	CK(hipMalloc(&idx, (n_idx + 64) * 8)); CK(hipMalloc(&start, (n_lists + 512) * 8)); CK(hipMalloc(&out, (n_out + 4096) * sizeof(hit)));
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, idx, n_idx + 64);
	uint64_t *big; CK(hipMalloc(&big, (n_rec + 64) * 8)); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, big, n_rec + 64);
	std::vector<uint64_t> h(n_lists + 512); uint64_t x = 88172645463325252ULL;
	for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % (n_idx - LIST_LEN - 8); }      // list starts: anywhere in the index, 8-byte aligned (as the engine's)
	CK(hipMemcpy(start, h.data(), h.size() * 8, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const unsigned g = (unsigned)((n_lists + 2047) / 2048);      // 4 waves of 512 lists per block
	auto run = [&](const char *name, int which) {
		float best = 1e30f;
		for (int rep = 0; rep < 4; ++rep) {
			CK(hipEventRecord(e0, 0));
			if (which == 0) hipLaunchKernelGGL(rows_kernel, dim3(g), dim3(256), 0, 0, idx, start, n_lists, out);
			else if (which == 1) hipLaunchKernelGGL(lists_kernel, dim3(g), dim3(256), 0, 0, idx, start, n_lists, out);
			else if (which == 3) hipLaunchKernelGGL(copy16_kernel, dim3(256 * 8), dim3(256), 0, 0, (const ulonglong2*)big, n_lists * LIST_LEN / 2, out);
			else hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, idx, n_lists * LIST_LEN, n_idx, out);
			CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
			float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
		}
		const double bytes = 24.0 * (double)(n_lists * LIST_LEN);
		printf("%-6s %8.3f ms  %7.1f GB/s of (8 in + 16 out) bytes  = %.3f of 8 TB/s   [%llu M records, %llu M lists of %u, index %.2f GB]\n", name, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0,
			   (unsigned long long)(n_lists * LIST_LEN / 1000000), (unsigned long long)(n_lists / 1000000), LIST_LEN, n_idx * 8 / 1e9);
	};
	run("copy16", 3); run("copy", 2); run("lists", 1); run("rows", 0);
	return 0;
}
