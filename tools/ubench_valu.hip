// Instruction issue-rate micro-benchmark for gfx950 (measurement aid, not part of the product): how many cycles a SIMD spends per wave64
// instruction of each kind, relative to v_add_u32.  Eight independent dependency chains per lane, 8 waves per SIMD, so latency is hidden
// and the number is the issue cost.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run: ./ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITERS 2048

#define KERNEL32(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) { \
	uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1; \
	for (int i = 0; i < ITERS; ++i) { \
		asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
			: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20"); \
	} \
	out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }

// operand numbering: %0..%7 = a0..a7, %8 = b
#define A_ADD(x) "v_add_u32 %" #x ", %" #x ", %8\n"
#define A_XOR(x) "v_xor_b32 %" #x ", %" #x ", %8\n"
#define A_MIN(x) "v_min_u32 %" #x ", %" #x ", %8\n"
#define A_ALIGN(x) "v_alignbit_b32 %" #x ", %" #x ", %8, 11\n"
#define A_BFREV(x) "v_bfrev_b32 %" #x ", %" #x "\n"
#define A_MULLO(x) "v_mul_lo_u32 %" #x ", %" #x ", %8\n"
#define A_BFE(x) "v_bfe_u32 %" #x ", %" #x ", 3, 17\n"
#define A_BFI(x) "v_bfi_b32 %" #x ", %8, %" #x ", %" #x "\n"
#define A_LSHLOR(x) "v_lshl_or_b32 %" #x ", %" #x ", 2, %8\n"
#define A_PERM(x) "v_perm_b32 %" #x ", %" #x ", %8, %8\n"
#define A_DPP_WSHR(x) "v_mov_b32_dpp %" #x ", %" #x " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_DPP_RSHR(x) "v_mov_b32_dpp %" #x ", %" #x " row_shr:3 row_mask:0xf bank_mask:0xf\n"
#define A_MIN_DPP(x) "v_min_u32_dpp %" #x ", %" #x ", %" #x " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_CMPSEL(x) "v_cmp_lt_u32 vcc, %" #x ", %8\nv_cndmask_b32 %" #x ", %" #x ", %8, vcc\n"
#define A_BPERM(x) "ds_bpermute_b32 %" #x ", %8, %" #x "\ns_waitcnt lgkmcnt(4)\n"
#define A_SWAP32(x) "v_permlane32_swap_b32 %" #x ", %8\n"
#define A_SWAP16(x) "v_permlane16_swap_b32 %" #x ", %8\n"
#define A_BCNT(x) "v_bcnt_u32_b32 %" #x ", %" #x ", %8\n"
#define A_ADD3(x) "v_add3_u32 %" #x ", %" #x ", %8, %8\n"
#define A_XOR3(x) "v_xor3_b32 %" #x ", %" #x ", %8, %8\n"
#define A_LSHLADD(x) "v_lshl_add_u32 %" #x ", %" #x ", 3, %8\n"
#define A_MAD24(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %8\n"
#define A_AND(x) "v_and_b32 %" #x ", %" #x ", %8\n"
#define A_OR(x) "v_or_b32 %" #x ", %" #x ", %8\n"
#define A_SUB(x) "v_sub_u32 %" #x ", %" #x ", %8\n"
#define A_NOT(x) "v_not_b32 %" #x ", %" #x "\n"
#define A_MOV(x) "v_mov_b32 %" #x ", %8\n"
#define A_LSHL(x) "v_lshlrev_b32 %" #x ", 3, %" #x "\n"
#define A_LSHR(x) "v_lshrrev_b32 %" #x ", 3, %" #x "\n"
#define A_CNDMASK(x) "v_cndmask_b32 %" #x ", %" #x ", %8, vcc\n"
#define A_CMP(x) "v_cmp_lt_u32 vcc, %" #x ", %8\n"
#define A_ADDSAT(x) "v_add_i32 %" #x ", %" #x ", %8 clamp\n"
#define A_MUL24(x) "v_mul_u32_u24 %" #x ", %" #x ", %8\n"
#define A_MBCNT(x) "v_mbcnt_lo_u32_b32 %" #x ", %8, %" #x "\n"
#define A_READLANE(x) "v_readlane_b32 s20, %" #x ", 5\n"
#define A_ADDCO(x) "v_add_co_u32 %" #x ", vcc, %" #x ", %8\n"

KERNEL32(k_add, A_ADD) KERNEL32(k_xor, A_XOR) KERNEL32(k_min, A_MIN) KERNEL32(k_align, A_ALIGN) KERNEL32(k_bfrev, A_BFREV) KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_bfe, A_BFE) KERNEL32(k_bfi, A_BFI) KERNEL32(k_lshlor, A_LSHLOR) KERNEL32(k_perm, A_PERM) KERNEL32(k_dpp_wshr, A_DPP_WSHR) KERNEL32(k_dpp_rshr, A_DPP_RSHR)
KERNEL32(k_min_dpp, A_MIN_DPP) KERNEL32(k_cmpsel, A_CMPSEL) KERNEL32(k_bperm, A_BPERM) KERNEL32(k_swap32, A_SWAP32) KERNEL32(k_swap16, A_SWAP16) KERNEL32(k_bcnt, A_BCNT)
KERNEL32(k_add3, A_ADD3) KERNEL32(k_lshladd, A_LSHLADD) KERNEL32(k_mad24, A_MAD24)
KERNEL32(k_and, A_AND) KERNEL32(k_or, A_OR) KERNEL32(k_sub, A_SUB) KERNEL32(k_not, A_NOT) KERNEL32(k_mov, A_MOV) KERNEL32(k_lshl, A_LSHL) KERNEL32(k_lshr, A_LSHR) KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_cmp, A_CMP) KERNEL32(k_addsat, A_ADDSAT) KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mbcnt, A_MBCNT) KERNEL32(k_readlane, A_READLANE) KERNEL32(k_addco, A_ADDCO)

#define KERNEL64(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) { \
	uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = (uint64_t)seed << 20 | 1; uint32_t c = seed | 1; \
	for (int i = 0; i < ITERS; ++i) { \
		asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
			: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
	} \
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); }
// %0..%7 64-bit chains, %8 = b (64-bit), %9 = c (32-bit)
#define B_LSHLADD64(x) "v_lshl_add_u64 %" #x ", %" #x ", 3, %8\n"
#define B_LSHL64(x) "v_lshlrev_b64 %" #x ", 5, %" #x "\n"
#define B_LSHR64(x) "v_lshrrev_b64 %" #x ", 5, %" #x "\n"
#define B_MAD64(x) "v_mad_u64_u32 %" #x ", vcc, %9, %9, %" #x "\n"
#define B_CMP64(x) "v_cmp_lt_u64 vcc, %" #x ", %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\n"
#define B_MOV64(x) "v_mov_b64 %" #x ", %8\n"
#define B_PKADD(x) "v_pk_add_u16 %9, %9, %9\n"
KERNEL64(k_lshladd64, B_LSHLADD64) KERNEL64(k_lshl64, B_LSHL64) KERNEL64(k_lshr64, B_LSHR64) KERNEL64(k_mad64, B_MAD64)

// 64-bit compare + 2 cndmask (a 64-bit min), written in C so the compiler picks its forms
__global__ __launch_bounds__(256) void k_min64(uint32_t *out, uint32_t seed) {
	uint64_t a[8]; for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x + seed) * (uint64_t)(2 * j + 3) * 0x9E3779B97F4A7C15ull; uint64_t b = (uint64_t)seed * 0xD6E8FEB86659FD93ull;
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int r = 0; r < 2; ++r)
#pragma unroll
			for (int j = 0; j < 8; ++j) { a[j] = a[j] < b ? a[j] : b; asm volatile("" : "+v"(a[j])); b += 0x1234567; }
	}
	uint64_t s = 0; for (int j = 0; j < 8; ++j) s ^= a[j]; out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ s >> 32); }

// the two hash formulations: compiler's (mul forms) and explicit shift/add forms
__device__ __forceinline__ uint64_t h_c(uint64_t key) { key = ~key + (key << 21); key ^= key >> 24; key = key + (key << 3) + (key << 8); key ^= key >> 14; key = key + (key << 2) + (key << 4); key ^= key >> 28; key += key << 31; return key; }
__device__ __forceinline__ uint64_t la(uint64_t a, int s, uint64_t b) { uint64_t d; switch (s) { case 0: asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b)); break; case 2: asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(d) : "v"(a), "v"(b)); break;
	case 3: asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(d) : "v"(a), "v"(b)); break; default: asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(d) : "v"(a), "v"(b)); } return d; }
__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return (uint64_t)hi << 32 | lo; }
__device__ __forceinline__ uint64_t h_s(uint64_t key) {
	uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
	uint64_t t = mk64(lo << 21, __builtin_amdgcn_alignbit(hi, lo, 11)); key = la(~key, 0, t);                 // ~key + (key << 21)
	lo = (uint32_t)key; hi = (uint32_t)(key >> 32); key = mk64(lo ^ __builtin_amdgcn_alignbit(hi, lo, 24), hi ^ (hi >> 24));
	{ uint64_t k9 = la(key, 3, key), k16 = la(key, 4, 0); key = la(k16, 4, k9); }                                   // * 265
	lo = (uint32_t)key; hi = (uint32_t)(key >> 32); key = mk64(lo ^ __builtin_amdgcn_alignbit(hi, lo, 14), hi ^ (hi >> 14));
	key = la(la(key, 2, key), 2, key);                                                                              // * 21
	lo = (uint32_t)key; hi = (uint32_t)(key >> 32); key = mk64(lo ^ __builtin_amdgcn_alignbit(hi, lo, 28), hi ^ (hi >> 28));
	lo = (uint32_t)key; hi = (uint32_t)(key >> 32); key = la(mk64(lo << 31, __builtin_amdgcn_alignbit(hi, lo, 1)), 0, key);
	return key; }
template<int V> __global__ __launch_bounds__(256) void k_hash(uint32_t *out, uint32_t seed) {
	uint64_t a[8]; for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x + seed) * (uint64_t)(2 * j + 3);
	for (int i = 0; i < ITERS / 8; ++i) {
#pragma unroll
		for (int j = 0; j < 8; ++j) a[j] = V ? h_s(a[j]) : h_c(a[j]);
	}
	uint64_t s = 0; for (int j = 0; j < 8; ++j) s ^= a[j]; out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ s >> 32); }
__global__ void k_hash_check(uint64_t *out) { uint64_t x = 0x0123456789abcdefull * (threadIdx.x + 1); out[threadIdx.x] = h_c(x) ^ h_s(x); }

// LDS: ds_read_b32 / ds_or_b32 rates
__global__ __launch_bounds__(256) void k_ldsor(uint32_t *out, uint32_t seed) {
	__shared__ uint32_t s[1024]; for (int i = threadIdx.x; i < 1024; i += 256) s[i] = 0; __syncthreads();
	uint32_t x = threadIdx.x * 2654435761u + seed;
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int r = 0; r < 16; ++r) { atomicOr(&s[(x >> 7) & 1023], 1u << (x & 31)); x = x * 1664525u + 1013904223u; }
	}
	__syncthreads(); out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] ^ x; }

typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int ncu = p.multiProcessorCount, blocks = ncu * 8; const double clk = p.clockRate * 1e3;      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
	uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4 + 4096);
	struct T { const char *name; kfn f; double insts_per_iter; };
	std::vector<T> ts = { {"v_add_u32", k_add, 16}, {"v_xor_b32", k_xor, 16}, {"v_min_u32", k_min, 16}, {"v_alignbit_b32", k_align, 16}, {"v_bfrev_b32", k_bfrev, 16}, {"v_mul_lo_u32", k_mullo, 16},
		{"v_bfe_u32", k_bfe, 16}, {"v_bfi_b32", k_bfi, 16}, {"v_lshl_or_b32", k_lshlor, 16}, {"v_perm_b32", k_perm, 16}, {"v_add3_u32", k_add3, 16}, {"v_lshl_add_u32", k_lshladd, 16},
		{"v_mad_u32_u24", k_mad24, 16}, {"v_bcnt_u32_b32", k_bcnt, 16},
		{"v_and_b32", k_and, 16}, {"v_or_b32", k_or, 16}, {"v_sub_u32", k_sub, 16}, {"v_not_b32", k_not, 16}, {"v_mov_b32", k_mov, 16}, {"v_lshlrev_b32", k_lshl, 16}, {"v_lshrrev_b32", k_lshr, 16},
		{"v_cndmask_b32 (vcc)", k_cndmask, 16}, {"v_cmp_lt_u32 (vcc)", k_cmp, 16}, {"v_add_i32 clamp", k_addsat, 16}, {"v_mul_u32_u24", k_mul24, 16}, {"v_mbcnt_lo_u32_b32", k_mbcnt, 16},
		{"v_readlane_b32", k_readlane, 16}, {"v_add_co_u32", k_addco, 16},
		{"v_mov_dpp wave_shr:1", k_dpp_wshr, 16}, {"v_mov_dpp row_shr:3", k_dpp_rshr, 16}, {"v_min_u32_dpp wave_shr:1", k_min_dpp, 16}, {"v_cmp_lt_u32+v_cndmask", k_cmpsel, 16},
		{"ds_bpermute_b32", k_bperm, 16}, {"v_permlane32_swap", k_swap32, 16}, {"v_permlane16_swap", k_swap16, 16},
		{"v_lshl_add_u64", k_lshladd64, 16}, {"v_lshlrev_b64", k_lshl64, 16}, {"v_lshrrev_b64", k_lshr64, 16}, {"v_mad_u64_u32", k_mad64, 16},
		{"min64 (C: cmp_u64 + 2 cndmask)", k_min64, 16}, {"hash64 compiler form (per hash)", k_hash<0>, 1.0}, {"hash64 shift/add form (per hash)", k_hash<1>, 1.0}, {"ds_or_b32 (random word)", k_ldsor, 16} };
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	printf("device %s, %d CUs, %.0f MHz; cycles = SIMD cycles per wave64 instruction (4 SIMDs per CU, 8 waves per SIMD resident)\n", p.gcnArchName, ncu, clk / 1e6);
	double base = 0;
	for (auto &t : ts) {
		for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, 12345u);
		hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		// waves per SIMD = blocks * 4 / (ncu * 4) = 8; instructions per wave = ITERS * insts_per_iter
		const double n_iter = (t.insts_per_iter == 1.0) ? ITERS : ITERS;      // hash kernels: ITERS hashes per lane in total (ITERS/8 iterations x 8 chains)
		const double inst_per_simd = 8.0 * n_iter * t.insts_per_iter;
		const double cyc = ms * 1e-3 * clk / inst_per_simd;
		if (!base) base = cyc;
		printf("%-36s %8.3f ms  %7.2f cycles/inst  (%.2fx v_add_u32)\n", t.name, ms, cyc, cyc / base);
	}
	uint64_t *chk; hipMalloc(&chk, 64 * 8); hipLaunchKernelGGL(k_hash_check, dim3(1), dim3(64), 0, 0, chk); uint64_t h[64]; hipMemcpy(h, chk, 512, hipMemcpyDeviceToHost);
	uint64_t bad = 0; for (int i = 0; i < 64; ++i) bad |= h[i]; printf("hash forms agree: %s\n", bad ? "NO" : "yes");
	return 0;
}
