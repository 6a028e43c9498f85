// Common device/host helpers for the hao engine (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "hao.h"

#define HAO_WAVE 64
#define HAO_N_COUNTS 4096      // histogram bins; counts saturate at 4095 (htab.cpp:13-15)
#define HAO_MAX_COUNT 4095
#define HAO_CNT_DUMMY ((1u << 28) - 1)   // count field of a non-candidate window slot (sketch.cpp:470)

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

// yak_hash64_64 (htab.h:149-159): invertible 64-bit mix of one bit-plane.
__host__ __device__ __forceinline__ uint64_t hao_hash64(uint64_t key)
{
	key = ~key + (key << 21);
	key ^= key >> 24;
	key = key + (key << 3) + (key << 8);
	key ^= key >> 14;
	key = key + (key << 2) + (key << 4);
	key ^= key >> 28;
	key += key << 31;
	return key;
}

// The same hash as the sketch kernel spends it - two per k-mer, 16 k-mers per lane - in the instruction forms gfx950 issues cheapest (tools/ubench_valu.hip: the form
// above costs 86 SIMD cycles per hash as the compiler legalises its 64-bit multiplications - v_mad_u64_u32 for each half with moves between the register pairs -,
// this one ~65): a multiplication by a small constant c as lo * c on v_mad_u64_u32 with (hi * c) << 32 as its addend; the first step ~key + (key << 21) as
// key * (2^21 - 1) - 1 with a 24-bit multiplication for the high half (keys below 2^56: a bit plane of a k-mer of at most 56 bases); and the last step,
// key += key << 31 = a multiplication by 2^31 + 1, taken out: hash(a) + hash(b) = fin(part(a) + part(b)).  Bit-identical to hao_hash64 (checked for 10^8 random keys on
// the host, and by every sketch test).
#define HAO_MUL64_C(key, c) { const uint32_t lo_ = (uint32_t)(key), hi_ = (uint32_t)((key) >> 32); const uint64_t p_ = (uint64_t)lo_ * (c); const uint32_t h2_ = (uint32_t)(p_ >> 32) + hi_ * (c); (key) = (uint64_t)h2_ << 32 | (uint32_t)p_; }
__host__ __device__ __forceinline__ uint64_t hao_hash64_part56(uint64_t key)      // key < 2^56
{
	{	const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
		const uint64_t p = (uint64_t)lo * 0x1fffffu + 0xffffffffu; const uint32_t h2 = (uint32_t)(p >> 32) + ((hi & 0xffffffu) * 0x1fffffu + 0xffffffffu); key = (uint64_t)h2 << 32 | (uint32_t)p; }      // (the mask is free: v_mul_u32_u24 reads 24 bits, and it keeps the compiler from putting the 64-bit product back together)
	key ^= key >> 24;
	HAO_MUL64_C(key, 265u)
	key ^= key >> 14;
	HAO_MUL64_C(key, 21u)
	key ^= key >> 28;
	return key;
}
__host__ __device__ __forceinline__ uint64_t hao_hash64_fin(uint64_t s) { HAO_MUL64_C(s, 0x80000001u) return s; }
// hash of a k-mer from its two bit planes (yak_hash_long, htab.h:161-166): K <= 56 takes the cheap form
template<int K> __host__ __device__ __forceinline__ uint64_t hao_hash_planes(uint64_t p0, uint64_t p1)
{
	if (K <= 56) return hao_hash64_fin(hao_hash64_part56(p0) + hao_hash64_part56(p1));
	return hao_hash64(p0) + hao_hash64(p1);
}

// minimizer / index-position record bit layout (htab.h:13-22): rid:28 | pos:27 | rev:1 | span:8
__host__ __device__ __forceinline__ uint64_t hao_info_pack(uint32_t rid, uint32_t pos, uint32_t rev, uint32_t span)
{ return (uint64_t)(rid & 0xfffffffu) | (uint64_t)(pos & 0x7ffffffu) << 28 | (uint64_t)(rev & 1) << 55 | (uint64_t)(span & 0xff) << 56; }
__host__ __device__ __forceinline__ uint32_t hao_info_rid(uint64_t v)  { return (uint32_t)(v & 0xfffffffu); }
__host__ __device__ __forceinline__ uint32_t hao_info_pos(uint64_t v)  { return (uint32_t)(v >> 28 & 0x7ffffffu); }
__host__ __device__ __forceinline__ uint32_t hao_info_rev(uint64_t v)  { return (uint32_t)(v >> 55 & 1); }
__host__ __device__ __forceinline__ uint32_t hao_info_span(uint64_t v) { return (uint32_t)(v >> 56); }

// 2-bit base of a read in the reference read-store layout (4 bases/byte, first base in bits 7..6)
__device__ __forceinline__ uint32_t hao_base_at(const uint8_t *rd, uint32_t i)
{ return (rd[i >> 2] >> (6 - 2 * (i & 3))) & 3; }

// binary search in a sorted u64 array: index of key or -1
__host__ __device__ __forceinline__ int64_t hao_bsearch(const uint64_t *a, uint64_t n, uint64_t key)
{
	uint64_t lo = 0, hi = n;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a[m] < key) lo = m + 1; else hi = m; }
	return (lo < n && a[lo] == key) ? (int64_t)lo : -1;
}

__device__ __forceinline__ int hao_lane() { return threadIdx.x & 63; }
// The lanes of a wave run in lockstep: an LDS word that several lanes read in one statement may be rewritten by one of them in the next.  HAO_LOCKSTEP() marks
// those places; it is nothing on the device and a rendezvous of the wave's lanes in the CPU emulation of the kernels (tests/simt, where lanes are fibers).
#ifndef HAO_LOCKSTEP
#define HAO_LOCKSTEP()
#endif

// A 4-byte load through the SCALAR cache from a wave-uniform address, result in an SGPR, waited for at once.  For the rare uniform fallback of a value that normally
// comes from LDS: written as a plain C++ load next to the LDS load the compiler may fold the two into one flat load (select of the addresses) whose s_waitcnt
// vmcnt(0) then waits for every vector load and store the wave has in flight (hao_query5.cuh).  The CPU emulation of the kernels reads the word directly.
#ifndef HAO_SLOAD_U32
#define HAO_SLOAD_U32(dst, ptr) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst) : "s"(ptr) : "memory")
#endif

// A value the compiler must not reason about (tests' switches inside hot loops: keeps a division's set-up inside the branch that uses it).  Nothing in the CPU emulation.
#ifndef HAO_OPAQUE_U32
#define HAO_OPAQUE_U32(x) asm volatile("" : "+v"(x))
#endif

// Cross-lane moves on the DPP path (one VALU op, no LDS crossbar trip like ds_bpermute).  gfx9 controls: 0x110+n row_shr:n (inside rows of
// 16 lanes), 0x142 row_bcast:15 (lane 15 of a row -> the next row, use row_mask 0xa), 0x143 row_bcast:31 (lane 31 -> rows 2,3, row_mask 0xc),
// 0x138 wave_shr:1.  Lanes without a valid source keep `old`.  Call only in wave-uniform control flow.
template<int CTRL, int ROWMASK> __device__ __forceinline__ int hao_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xf, false); }
// value of the previous lane; lane 0 gets `fill`
__device__ __forceinline__ int hao_wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t hao_wave_shr1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }
// inclusive segmented sum over the wave: fl = 1 where a segment starts; on return fl = "a segment starts at or before this lane"
__device__ __forceinline__ void hao_seg_scan_add(int32_t &x, int &fl)
{
#define HAO_SEG_STEP(CTRL, RM) { const int x2 = hao_dpp<CTRL, RM>(0, x), f2 = hao_dpp<CTRL, RM>(0, fl); if (!fl) x += x2; fl |= f2; }
	HAO_SEG_STEP(0x111, 0xf) HAO_SEG_STEP(0x112, 0xf) HAO_SEG_STEP(0x114, 0xf) HAO_SEG_STEP(0x118, 0xf) HAO_SEG_STEP(0x142, 0xa) HAO_SEG_STEP(0x143, 0xc)
#undef HAO_SEG_STEP
}

// wave-wide reductions on the DPP path (the result is uniform): inclusive scan steps, total in lane 63
__device__ __forceinline__ int32_t hao_wave_max_i32(int32_t v)
{
#define HAO_RED_STEP(CTRL, RM) v = max(v, hao_dpp<CTRL, RM>(INT32_MIN, v));
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int64_t hao_wave_sum_i64(int64_t v)
{
#define HAO_RED_STEP(CTRL, RM) { const uint32_t lo2 = (uint32_t)hao_dpp<CTRL, RM>(0, (int)(uint32_t)v), hi2 = (uint32_t)hao_dpp<CTRL, RM>(0, (int)(uint32_t)((uint64_t)v >> 32)); v += (int64_t)((uint64_t)hi2 << 32 | lo2); }
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), 63) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63));
}
__device__ __forceinline__ uint32_t hao_wave_min_u32(uint32_t v)
{
#define HAO_RED_STEP(CTRL, RM) v = min(v, (uint32_t)hao_dpp<CTRL, RM>(-1, (int)v));
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum over the wave (DPP)
__device__ __forceinline__ uint32_t hao_wave_incl_scan_u32(uint32_t x)
{
#define HAO_RED_STEP(CTRL, RM) x += (uint32_t)hao_dpp<CTRL, RM>(0, (int)x);
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return x;
}
// a + b clamped to [INT32_MIN, INT32_MAX] (the compiler turns this form into v_add_i32 ... clamp)
__host__ __device__ __forceinline__ int32_t hao_add_sat_i32(int32_t a, int32_t b) { int32_t t; return __builtin_add_overflow(a, b, &t) ? (a < 0 ? INT32_MIN : INT32_MAX) : t; }
// a + b clamped to 2^32 - 1 (v_add_u32 ... clamp), and the wave's sum of that kind (uniform result)
__host__ __device__ __forceinline__ uint32_t hao_add_sat_u32(uint32_t a, uint32_t b) { uint32_t t; return __builtin_add_overflow(a, b, &t) ? 0xffffffffu : t; }
__device__ __forceinline__ uint32_t hao_wave_sum_sat_u32(uint32_t v)
{
#define HAO_RED_STEP(CTRL, RM) v = hao_add_sat_u32(v, (uint32_t)hao_dpp<CTRL, RM>(0, (int)v));
	HAO_RED_STEP(0x111, 0xf) HAO_RED_STEP(0x112, 0xf) HAO_RED_STEP(0x114, 0xf) HAO_RED_STEP(0x118, 0xf) HAO_RED_STEP(0x142, 0xa) HAO_RED_STEP(0x143, 0xc)
#undef HAO_RED_STEP
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// bits of the wave-uniform mask m below this lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ uint32_t hao_mbcnt(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// the value lane `src` holds (src wave-uniform: v_readlane_b32)
__device__ __forceinline__ uint32_t hao_bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ int32_t hao_bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t hao_readlane_i64(int64_t v, int l)
{ return (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l) << 32); }
// value of the next lane; lane 63 gets `fill` (DPP wave_shl:1)
__device__ __forceinline__ uint32_t hao_wave_shl1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false); }

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { hao_set_err(c, std::string(#expr) + ": " + hipGetErrorString(_e)); return HAO_ENODEV; } } while (0)
