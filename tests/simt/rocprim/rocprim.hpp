// TEST INFRASTRUCTURE: stand-in for <rocprim/rocprim.hpp> in the CPU emulation build (tests/simt): the device-wide primitives the library calls, with rocPRIM's
// signatures (temporary-storage query when the first argument is null) and sequential bodies.  Sorts are stable on the requested key bits like the radix sorts.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
#include <type_traits>
#include <vector>

namespace rocprim {
template<class T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };
template<class T> struct equal_to { bool operator()(const T &a, const T &b) const { return a == b; } };
template<class T> struct double_buffer {
	T *buf[2]; int sel = 0;
	double_buffer(T *cur, T *alt) { buf[0] = cur; buf[1] = alt; }
	T *current() const { return buf[sel]; } T *alternate() const { return buf[sel ^ 1]; } void swap() { sel ^= 1; }
};
template<class It, class F> struct transform_iterator {
	It it; F f;
	auto operator[](size_t i) const { return f(it[i]); }
	auto operator*() const { return f(*it); }
	transform_iterator operator+(size_t n) const { return transform_iterator{it + n, f}; }
};
template<class It, class F> transform_iterator<It, F> make_transform_iterator(It it, F f) { return transform_iterator<It, F>{it, f}; }
template<class T> struct counting_iterator {
	T v;
	T operator[](size_t i) const { return (T)(v + (T)i); } T operator*() const { return v; }
	counting_iterator operator+(size_t n) const { return counting_iterator{(T)(v + (T)n)}; }
};
template<class T> counting_iterator<T> make_counting_iterator(T v) { return counting_iterator<T>{v}; }
template<class T> struct constant_iterator {
	T v;
	T operator[](size_t) const { return v; } T operator*() const { return v; }
	constant_iterator operator+(size_t) const { return *this; }
};
template<class T> constant_iterator<T> make_constant_iterator(T v) { return constant_iterator<T>{v}; }

#define HAO_SIMT_TMP_QUERY() do { if (!tmp) { tb = 256; return hipSuccess; } } while (0)

template<class In, class Out, class Init, class Op>
hipError_t exclusive_scan(void *tmp, size_t &tb, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	Init acc = init;
	for (size_t i = 0; i < n; ++i) { const Init v = (Init)in[i]; out[i] = acc; acc = op(acc, v); }
	return hipSuccess;
}
template<class In, class Out, class Op>
hipError_t inclusive_scan(void *tmp, size_t &tb, In in, Out out, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	if (!n) return hipSuccess;
	using T = decltype(op(in[0], in[0])); T acc = (T)in[0]; out[0] = acc;
	for (size_t i = 1; i < n; ++i) { acc = op(acc, in[i]); out[i] = acc; }
	return hipSuccess;
}
template<class In, class Out, class Init, class Op>
hipError_t reduce(void *tmp, size_t &tb, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	Init acc = init;
	for (size_t i = 0; i < n; ++i) acc = op(acc, (Init)in[i]);
	*out = acc;
	return hipSuccess;
}

template<class K> inline uint64_t key_bits(const K &k, unsigned b0, unsigned b1)
{
	static_assert(std::is_unsigned<K>::value, "radix keys: unsigned integers in this library");
	const uint64_t v = (uint64_t)k >> b0; const unsigned w = b1 - b0;
	return w >= 64 ? v : v & ((1ULL << w) - 1);
}
template<class K> hipError_t radix_sort_keys(void *tmp, size_t &tb, const K *in, K *out, size_t n, unsigned b0 = 0, unsigned b1 = 8 * sizeof(K), hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	std::vector<K> v(in, in + n);
	std::stable_sort(v.begin(), v.end(), [&](const K &a, const K &b) { return key_bits(a, b0, b1) < key_bits(b, b0, b1); });
	std::copy(v.begin(), v.end(), out);
	return hipSuccess;
}
template<class K> hipError_t radix_sort_keys(void *tmp, size_t &tb, double_buffer<K> &db, size_t n, unsigned b0 = 0, unsigned b1 = 8 * sizeof(K), hipStream_t s = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	const hipError_t e = radix_sort_keys(tmp, tb, (const K*)db.current(), db.alternate(), n, b0, b1, s); db.swap(); return e;
}
template<class K, class V> hipError_t radix_sort_pairs(void *tmp, size_t &tb, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned b0 = 0, unsigned b1 = 8 * sizeof(K), hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	std::vector<size_t> ix(n); std::iota(ix.begin(), ix.end(), (size_t)0);
	std::stable_sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return key_bits(kin[a], b0, b1) < key_bits(kin[b], b0, b1); });
	std::vector<K> k(n); std::vector<V> v(n);
	for (size_t i = 0; i < n; ++i) { k[i] = kin[ix[i]]; v[i] = vin[ix[i]]; }
	std::copy(k.begin(), k.end(), kout); std::copy(v.begin(), v.end(), vout);
	return hipSuccess;
}
template<class K, class V> hipError_t radix_sort_pairs(void *tmp, size_t &tb, double_buffer<K> &dk, double_buffer<V> &dv, size_t n, unsigned b0 = 0, unsigned b1 = 8 * sizeof(K), hipStream_t s = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	const hipError_t e = radix_sort_pairs(tmp, tb, (const K*)dk.current(), dk.alternate(), (const V*)dv.current(), dv.alternate(), n, b0, b1, s); dk.swap(); dv.swap(); return e;
}
template<class T, class Cmp> hipError_t merge_sort(void *tmp, size_t &tb, const T *in, T *out, size_t n, Cmp cmp, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	std::vector<T> v(in, in + n); std::stable_sort(v.begin(), v.end(), cmp); std::copy(v.begin(), v.end(), out);
	return hipSuccess;
}
// select by flags: (in, flags, out, count, n) - and by predicate: (in, out, count, n, pred); told apart by where the size stands
template<class In, class Flags, class Out, class Cnt, class N, typename std::enable_if<std::is_integral<N>::value && !std::is_integral<Cnt>::value, int>::type = 0>
hipError_t select(void *tmp, size_t &tb, In in, Flags flags, Out out, Cnt cnt, N n, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	size_t k = 0; for (size_t i = 0; i < (size_t)n; ++i) if (flags[i]) out[k++] = in[i];
	*cnt = k; return hipSuccess;
}
template<class In, class Out, class Cnt, class N, class Pred, typename std::enable_if<std::is_integral<N>::value && !std::is_integral<Pred>::value, int>::type = 0>
hipError_t select(void *tmp, size_t &tb, In in, Out out, Cnt cnt, N n, Pred pred, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	size_t k = 0; for (size_t i = 0; i < (size_t)n; ++i) { const auto v = in[i]; if (pred(v)) out[k++] = v; }
	*cnt = k; return hipSuccess;
}
template<class In, class UK, class UC, class NR>
hipError_t run_length_encode(void *tmp, size_t &tb, In in, unsigned int n, UK ukeys, UC ucnt, NR n_runs, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	size_t r = 0;
	for (size_t i = 0; i < n; ) { size_t j = i + 1; while (j < n && in[j] == in[i]) ++j; ukeys[r] = in[i]; ucnt[r] = (decltype(+ucnt[0]))(j - i); ++r; i = j; }
	*n_runs = r; return hipSuccess;
}
template<class Keys, class Vals, class UK, class Agg, class NU, class Op, class Eq>
hipError_t reduce_by_key(void *tmp, size_t &tb, Keys keys, Vals vals, size_t n, UK ukeys, Agg agg, NU n_unique, Op op, Eq eq, hipStream_t = nullptr, bool = false)
{
	HAO_SIMT_TMP_QUERY();
	size_t r = 0;
	for (size_t i = 0; i < n; ) { auto a = vals[i]; size_t j = i + 1; while (j < n && eq(keys[j], keys[i])) { a = op(a, vals[j]); ++j; } ukeys[r] = keys[i]; agg[r] = a; ++r; i = j; }
	*n_unique = r; return hipSuccess;
}
}      // namespace rocprim
