// Micro-benchmark (not part of the product): (hash u64, index u32) sort of m uniformly distributed keys,
//   a) one 64-bit radix_sort_pairs            (what ha_pt_gen does today)
//   b) top-bits partition with a 16-bit side key + segmented sort of the low bits
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_sort.hip -o gpurun_out/bench_sort && gpurun_out/bench_sort 215000000
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { uint32_t lo, hi, idx; };
static __device__ __host__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void fill(uint64_t *x, uint64_t m, uint64_t distinct) { uint64_t i = blockIdx.x * 256ULL + threadIdx.x; if (i < m) x[i] = mix(mix(i) % distinct + 1); }
struct TopBits { const uint64_t *x; int sh; __device__ uint32_t operator()(uint64_t i) const { return (uint32_t)(x[i] >> sh); } };
struct MkRec { const uint64_t *x; __device__ Rec operator()(uint64_t i) const { uint64_t v = x[i]; return Rec{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)i}; } };
struct RecKey { __device__ uint64_t operator()(const Rec &r) const { return (uint64_t)r.hi << 32 | r.lo; } };
struct RecIdx { __device__ uint32_t operator()(const Rec &r) const { return r.idx; } };
__global__ void seg_bounds(const uint32_t *k, uint64_t m, uint32_t nseg, uint32_t *off)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x; if (s > nseg) return;
	uint64_t lo = 0, hi = m;
	while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (k[mid] < s) lo = mid + 1; else hi = mid; }
	off[s] = (uint32_t)lo;
}
__global__ void cmp(const uint64_t *a, const uint64_t *b, const uint32_t *va, const uint32_t *vb, uint64_t m, unsigned long long *bad)
{ uint64_t i = blockIdx.x * 256ULL + threadIdx.x; if (i < m && (a[i] != b[i] || va[i] != vb[i])) atomicAdd(bad, 1ULL); }
int main(int argc, char **argv)
{
	uint64_t m = argc > 1 ? strtoull(argv[1], 0, 10) : 215000000ULL;
	int b1 = argc > 2 ? atoi(argv[2]) : 16;
	uint64_t distinct = argc > 3 ? strtoull(argv[3], 0, 10) : m / 30;
	const int sh = 64 - b1; const uint32_t nseg = 1u << b1;
	uint64_t *x, *sa, *sb; uint32_t *ia, *oa, *ob, *k16, *k16o, *off; Rec *rec; void *tmp; unsigned long long *bad;
	CK(hipMalloc(&x, m * 8)); CK(hipMalloc(&sa, m * 8)); CK(hipMalloc(&sb, m * 8)); CK(hipMalloc(&ia, m * 4)); CK(hipMalloc(&oa, m * 4)); CK(hipMalloc(&ob, m * 4));
	CK(hipMalloc(&k16o, m * 4)); CK(hipMalloc(&rec, m * sizeof(Rec))); CK(hipMalloc(&off, (nseg + 2) * 4)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
	size_t tcap = 8ULL << 30; CK(hipMalloc(&tmp, tcap));
	fill<<<(unsigned)((m + 255) / 256), 256>>>(x, m, distinct);
	hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
	for (int rep = 0; rep < 3; ++rep) {
		size_t tb = 0;
		auto iota = rocprim::make_counting_iterator<uint32_t>(0);
		CK(hipEventRecord(e0));
		CK(rocprim::radix_sort_pairs(nullptr, tb, x, sa, iota, oa, m, 0, 64, 0)); if (tb > tcap) return 2;
		CK(rocprim::radix_sort_pairs(tmp, tb, x, sa, iota, oa, m, 0, 64, 0));
		CK(hipEventRecord(e1));
		auto kin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), TopBits{x, sh});
		auto vin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), MkRec{x});
		CK(rocprim::radix_sort_pairs(nullptr, tb, kin, k16o, vin, rec, m, 0, b1, 0)); if (tb > tcap) return 2;
		CK(rocprim::radix_sort_pairs(tmp, tb, kin, k16o, vin, rec, m, 0, b1, 0));
		seg_bounds<<<(nseg + 256) / 256, 256>>>(k16o, m, nseg, off);
		CK(hipEventRecord(e2));
		auto k2 = rocprim::make_transform_iterator(rec, RecKey{});
		auto v2 = rocprim::make_transform_iterator(rec, RecIdx{});
		CK(rocprim::segmented_radix_sort_pairs(nullptr, tb, k2, sb, v2, ob, (unsigned)m, nseg, off, off + 1, 0, sh, 0)); if (tb > tcap) return 2;
		CK(rocprim::segmented_radix_sort_pairs(tmp, tb, k2, sb, v2, ob, (unsigned)m, nseg, off, off + 1, 0, sh, 0));
		CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
		float a, b, c2; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); CK(hipEventElapsedTime(&c2, e2, e3));
		printf("m=%llu b1=%d  one-shot %.2f ms | partition %.2f + segmented %.2f = %.2f ms\n", (unsigned long long)m, b1, a, b, c2, b + c2);
	}
	cmp<<<(unsigned)((m + 255) / 256), 256>>>(sa, sb, oa, ob, m, bad);
	unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
	printf("mismatches %llu\n", hb);
	return hb != 0;
}
