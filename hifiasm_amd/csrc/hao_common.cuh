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

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { hao_set_err(c, std::string(#expr) + ": " + hipGetErrorString(_e)); return HAO_ENODEV; } } while (0)
