// On-disk formats as first-class I/O: the engine's tables and read store written in the reference's own resume format, so that a
// GPU-built index can be handed to a stock hifiasm (load_pt_index, htab.cpp:1432-1550; Assembly.cpp:2078):
//   <prefix>.pt_flt          'f' + high-count filter table, 'h' + position index (4096 sub-tables), round / coverage tail  (write_pt_index, htab.cpp:1367-1430)
//   <prefix>.pt_flt.bin      the read store (write_All_reads, Process_Read.cpp:69-125: the same layout as *.ec.bin)
//   <prefix>.pt_flt.paf.bin  per-read overlap lists (empty: none have been computed yet)
// The tables are klib khashl open-addressing tables dumped raw (khashl.h:137-149): bucket count, bits, element count, the `used` bitmap and the
// bucket array; any placement the reference's own probe sequence can find is valid, so the writer builds them with that probe (khashl.h:99:
// bucket = hash * 2654435769 >> (32 - bits), then linear).  Host code only: the tables come from the engine's host views.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

static inline uint32_t hao_kh_bits(uint64_t count) { uint32_t b = 2; while ((1ULL << b) < count * 2 + 1) ++b; return b; }      // load <= 1/2
static inline uint32_t hao_kh_h2b(uint32_t hash, uint32_t bits) { return (uint32_t)(hash * 2654435769U) >> (32 - bits); }

// khashl `_save` of a table whose buckets are `bsz`-byte packed records starting with a u64 key; hash32(key) as the table's hash function defines it
template<typename Hash>
static bool hao_kh_write(FILE *fp, const std::vector<uint64_t> &keys, const void *vals, size_t vsz, Hash hash32)
{
	const uint64_t n = keys.size(); const size_t bsz = 8 + vsz;
	uint32_t n_buckets = 0, bits = 0, count = (uint32_t)n; uint8_t ff = 0;
	if (n == 0) { return fwrite(&n_buckets, 4, 1, fp) == 1 && fwrite(&bits, 4, 1, fp) == 1 && fwrite(&count, 4, 1, fp) == 1 && fwrite(&ff, 1, 1, fp) == 1 && fwrite(&ff, 1, 1, fp) == 1; }
	bits = hao_kh_bits(n); n_buckets = 1u << bits;
	std::vector<uint32_t> used(n_buckets < 32 ? 1 : n_buckets >> 5, 0); std::vector<uint8_t> bk((size_t)n_buckets * bsz, 0);
	const uint32_t mask = n_buckets - 1;
	for (uint64_t i = 0; i < n; ++i) {
		uint32_t b = hao_kh_h2b(hash32(keys[i]), bits);
		while (used[b >> 5] >> (b & 31) & 1) b = (b + 1) & mask;
		used[b >> 5] |= 1u << (b & 31);
		memcpy(&bk[(size_t)b * bsz], &keys[i], 8);
		if (vsz) memcpy(&bk[(size_t)b * bsz + 8], (const uint8_t*)vals + i * vsz, vsz);
	}
	ff = 1;
	return fwrite(&n_buckets, 4, 1, fp) == 1 && fwrite(&bits, 4, 1, fp) == 1 && fwrite(&count, 4, 1, fp) == 1 && fwrite(&ff, 1, 1, fp) == 1 &&
		   fwrite(used.data(), 4, used.size(), fp) == used.size() && fwrite(&ff, 1, 1, fp) == 1 && fwrite(bk.data(), bsz, n_buckets, fp) == n_buckets;
}

static int hao_index_save_impl(hao_ctx *c, const char *prefix, int32_t number_of_round, const char *const *names)
{
	if (!c->has_ft || !c->has_pt) { hao_set_err(c, "hao_index_save: hao_ft_gen and hao_pt_gen must have run"); return HAO_EINVAL; }
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_index_save: single-device mode only (a sharded engine holds a slice of the read store)"); return HAO_EUNSUPP; }
	if (int rc = hao_pt_download(c)) return rc;
	const std::string base = std::string(prefix) + ".pt_flt";
	FILE *fp = fopen(base.c_str(), "wb");
	if (!fp) { hao_set_err(c, "cannot write " + base); return HAO_EINVAL; }
	bool ok = true;
	{	// 'f': yak_ft_t = map u64 -> int16, hash = low 32 bits of the key (kh_hash_dummy); value = count, INT16_MAX above max_kmer_cnt (gen_hh, htab.cpp:1038-1062)
		std::vector<int16_t> v(c->h_ft_vals.size());
		for (size_t i = 0; i < v.size(); ++i) v[i] = c->h_ft_vals[i] == INT32_MAX ? INT16_MAX : (int16_t)c->h_ft_vals[i];
		ok = ok && fwrite("f", 1, 1, fp) == 1 && hao_kh_write(fp, c->h_ft_keys, v.data(), 2, [](uint64_t k) { return (uint32_t)k; });
	}
	{	// 'h': ha_pt_t: sub-table = low `pre` bits of the hash; key = hash >> pre << 12 | count, value = offset of the key's list in the sub-table's
		// position array; hash function = key >> 12 (yak_ct_hash, htab.cpp:123), equality ignores the count bits
		const int32_t k = c->opt.k, pre = 12; const uint64_t tot = c->h_ix_keys.size(), tot_pos = c->h_ix_pos.size();
		ok = ok && fwrite("h", 1, 1, fp) == 1 && fwrite(&k, 4, 1, fp) == 1 && fwrite(&pre, 4, 1, fp) == 1 && fwrite(&tot, 8, 1, fp) == 1 && fwrite(&tot_pos, 8, 1, fp) == 1;
		std::vector<std::vector<uint64_t> > idx(1u << pre);
		for (uint64_t i = 0; i < tot; ++i) idx[c->h_ix_keys[i] & ((1u << pre) - 1)].push_back(i);
		std::vector<uint64_t> keys, vals, a;
		for (uint32_t s = 0; s < (1u << pre) && ok; ++s) {
			keys.clear(); vals.clear(); a.clear();
			for (uint64_t i : idx[s]) {
				const uint64_t o = c->h_ix_off[i], n = c->h_ix_off[i + 1] - o;
				keys.push_back((c->h_ix_keys[i] >> pre) << 12 | n); vals.push_back(a.size());
				a.insert(a.end(), c->h_ix_pos.begin() + o, c->h_ix_pos.begin() + o + n);
			}
			const uint64_t na = a.size();
			ok = hao_kh_write(fp, keys, vals.data(), 8, [](uint64_t key) { return (uint32_t)(key >> 12); }) && fwrite(&na, 8, 1, fp) == 1 && (na == 0 || fwrite(a.data(), 8, na, fp) == na);
		}
	}
	{	const int32_t hom = c->hom_cov, het = c->het_cov, mnc = c->max_n_chain;
		ok = ok && fwrite(&number_of_round, 4, 1, fp) == 1 && fwrite(&hom, 4, 1, fp) == 1 && fwrite(&het, 4, 1, fp) == 1 && fwrite(&mnc, 4, 1, fp) == 1; }
	fclose(fp);
	if (!ok) { hao_set_err(c, "short write on " + base); return HAO_EINVAL; }
	// ---- the read store (write_All_reads, Process_Read.cpp:69-125) ----
	const uint64_t n = c->n_reads;
	std::vector<uint8_t> packed(c->n_pk_bytes + 1); std::vector<uint64_t> pk_off(n + 1); std::vector<uint32_t> ns;
	HIP_TRY(hipMemcpy(packed.data(), c->d_packed.p, c->n_pk_bytes, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(pk_off.data(), c->d_pk_off.p, (n + 1) * 8, hipMemcpyDeviceToHost));
	if (c->has_n) { ns.resize(c->h_nsite_off[n] + 1); HIP_TRY(hipMemcpy(ns.data(), c->d_nsite.p, c->h_nsite_off[n] * 4, hipMemcpyDeviceToHost)); }
	fp = fopen((base + ".bin").c_str(), "wb");
	if (!fp) { hao_set_err(c, "cannot write " + base + ".bin"); return HAO_EINVAL; }
	{
		const int32_t adapter = 0; const uint64_t index_size = n, name_index_size = n + 1, total_bases = c->n_bases; uint64_t total_name = 0;
		std::vector<uint64_t> name_index(n + 1, 0), len64(n); std::string all_names;
		for (uint64_t i = 0; i < n; ++i) { const std::string nm = names && names[i] ? std::string(names[i]) : "r" + std::to_string(i); name_index[i] = all_names.size(); all_names += nm; len64[i] = c->h_len[i]; }
		name_index[n] = total_name = all_names.size();
		ok = fwrite(&adapter, 4, 1, fp) == 1 && fwrite(&index_size, 8, 1, fp) == 1 && fwrite(&name_index_size, 8, 1, fp) == 1 && fwrite(&n, 8, 1, fp) == 1 &&
			 fwrite(&total_bases, 8, 1, fp) == 1 && fwrite(&total_name, 8, 1, fp) == 1;
		for (uint64_t i = 0; i < n && ok; ++i) {      // N sites: count, then positions (u64)
			uint64_t cnt = c->has_n ? c->h_nsite_off[i + 1] - c->h_nsite_off[i] : 0;
			ok = fwrite(&cnt, 8, 1, fp) == 1;
			for (uint64_t j = 0; j < cnt && ok; ++j) { const uint64_t p = ns[c->h_nsite_off[i] + j]; ok = fwrite(&p, 8, 1, fp) == 1; }
		}
		ok = ok && fwrite(len64.data(), 8, n, fp) == n;
		for (uint64_t i = 0; i < n && ok; ++i) ok = fwrite(&packed[pk_off[i]], 1, c->h_len[i] / 4 + 1, fp) == c->h_len[i] / 4 + 1;
		std::vector<uint8_t> trio(n, 0);      // AMBIGU is re-set by the loader (htab.cpp:1519)
		const int32_t hom = c->hom_cov, het = c->het_cov;
		ok = ok && (total_name == 0 || fwrite(all_names.data(), 1, total_name, fp) == total_name) && fwrite(name_index.data(), 8, n + 1, fp) == n + 1 &&
			 fwrite(trio.data(), 1, n, fp) == n && fwrite(&hom, 4, 1, fp) == 1 && fwrite(&het, 4, 1, fp) == 1;
	}
	fclose(fp);
	if (!ok) { hao_set_err(c, "short write on " + base + ".bin"); return HAO_EINVAL; }
	// ---- overlap lists: none yet (is_fully_corrected, is_abnormal, length = 0 per read; write_pt_index, htab.cpp:1411-1421) ----
	fp = fopen((base + ".paf.bin").c_str(), "wb");
	if (!fp) { hao_set_err(c, "cannot write " + base + ".paf.bin"); return HAO_EINVAL; }
	ok = fwrite(&n, 8, 1, fp) == 1;
	{ const uint8_t z8 = 0; const uint32_t z32 = 0; for (uint64_t i = 0; i < n && ok; ++i) ok = fwrite(&z8, 1, 1, fp) == 1 && fwrite(&z8, 1, 1, fp) == 1 && fwrite(&z32, 4, 1, fp) == 1; }
	fclose(fp);
	if (!ok) { hao_set_err(c, "short write on " + base + ".paf.bin"); return HAO_EINVAL; }
	return HAO_OK;
}
