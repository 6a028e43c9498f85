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
#include <sys/stat.h>
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

// ---------------------------------------------------------------------------------------
// The reader: an index dump of a stock hifiasm (write_pt_index, htab.cpp:1367-1430: <prefix>.pt_flt + .pt_flt.bin) - or one written by
// hao_index_save - becomes the engine's state, as load_pt_index (htab.cpp:1432-1550) makes it the reference's: read store, high-count filter
// table, position index, coverage peaks and max_n_chain.  What the file does not hold is derived here: the query side's read-ordered minimizers
// (the reference re-sketches every query read; the engine sketches all reads once with the loaded filter table) and every minimizer's lookup result.
// ---------------------------------------------------------------------------------------
// bytes between the file position and the end of the file (sizes read from a file are checked against it before anything is allocated for them)
// (the size comes from fstat: no seeking, so the stdio buffer survives - the loader asks once per read and once per sub-table)
static uint64_t hao_file_left(FILE *fp)
{
	struct stat sb; const long at = ftell(fp);
	if (at < 0 || fstat(fileno(fp), &sb) != 0 || sb.st_size <= at) return 0;
	return (uint64_t)sb.st_size - (uint64_t)at;
}
static bool hao_kh_read(FILE *fp, size_t vsz, std::vector<uint64_t> &keys, std::vector<uint8_t> &vals)
{
	uint32_t n_buckets = 0, bits = 0, count = 0; uint8_t ff = 0; keys.clear(); vals.clear();
	if (fread(&n_buckets, 4, 1, fp) != 1 || fread(&bits, 4, 1, fp) != 1 || fread(&count, 4, 1, fp) != 1 || fread(&ff, 1, 1, fp) != 1) return false;
	// khashl _save (khashl.h:137-149): n_buckets = 1 << bits (0 for a table that never grew), count <= n_buckets, and the arrays that follow must be in the file
	if (bits > 31 || (n_buckets != 0 && n_buckets != (1u << bits)) || count > n_buckets) return false;
	if ((uint64_t)n_buckets * (8 + vsz) + (n_buckets >> 3) > hao_file_left(fp) + 16) return false;
	std::vector<uint32_t> used;
	if (ff) { used.resize(n_buckets < 32 ? 1 : n_buckets >> 5); if (fread(used.data(), 4, used.size(), fp) != used.size()) return false; }
	if (fread(&ff, 1, 1, fp) != 1) return false;
	if (!ff) return count == 0;
	const size_t bsz = 8 + vsz; std::vector<uint8_t> bk((size_t)n_buckets * bsz);
	if (n_buckets && fread(bk.data(), bsz, n_buckets, fp) != n_buckets) return false;
	if (used.empty()) return count == 0;
	keys.reserve(count); vals.reserve((size_t)count * vsz);
	for (uint32_t b = 0; b < n_buckets; ++b)
		if (used[b >> 5] >> (b & 31) & 1) { uint64_t k; memcpy(&k, &bk[(size_t)b * bsz], 8); keys.push_back(k); vals.insert(vals.end(), &bk[(size_t)b * bsz + 8], &bk[(size_t)b * bsz + 8] + vsz); }
	return keys.size() == count;
}

// lookup result of every read-ordered minimizer in the loaded index (what hao_index_finish_kernel leaves at build time)
__global__ void hao_lk_fill_kernel(const uint64_t *mz_x, uint64_t n_mz, hao_pt_dev pt, uint64_t *lk)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_mz) return;
	uint64_t st = 0; const uint32_t n = hao_pt_lookup(pt, mz_x[i], &st);
	lk[i] = n ? (st | (uint64_t)n << 48) : 0;
}


static int hao_index_load_impl(hao_ctx *c, const char *prefix, int32_t *number_of_round)
{
	if (hao_is_sharded(c)) { hao_set_err(c, "hao_index_load: single-device mode only"); return HAO_EUNSUPP; }
	c->has_ft = false; c->has_pt = false; c->lk_valid = false; c->h_ix_valid = false;      // a load that fails half-way leaves "no index", not the previous one over new reads
	const std::string base = std::string(prefix) + ".pt_flt";
	FILE *fp = fopen(base.c_str(), "rb");
	if (!fp) { hao_set_err(c, "cannot read " + base); return HAO_EINVAL; }
	auto bad = [&](const std::string &what) { if (fp) fclose(fp); hao_set_err(c, "hao_index_load: " + what); return HAO_EINVAL; };
	char mode = 0; bool have_ft = false, have_pt = false;
	std::vector<uint64_t> ftk; std::vector<int32_t> ftv;
	if (fread(&mode, 1, 1, fp) != 1) return bad("empty " + base);
	if (mode == 'f') {
		std::vector<uint64_t> k; std::vector<uint8_t> v;
		if (!hao_kh_read(fp, 2, k, v)) return bad("filter table of " + base);
		std::vector<std::pair<uint64_t, int32_t> > kv(k.size());
		for (size_t i = 0; i < k.size(); ++i) {      // ha_ft_cnt's view of the value (htab.cpp:1064-1070); the device view's 13-bit code field holds 1 .. 4095 and "above the maximum"
			int16_t x; memcpy(&x, &v[2 * i], 2);
			if (x != INT16_MAX && (x < 1 || x > 4095)) return bad("filter table of " + base + ": a count outside 1 .. 4095");
			kv[i] = std::make_pair(k[i], x == INT16_MAX ? INT32_MAX : (int32_t)x);
		}
		std::sort(kv.begin(), kv.end());
		ftk.resize(kv.size()); ftv.resize(kv.size());
		for (size_t i = 0; i < kv.size(); ++i) { ftk[i] = kv[i].first; ftv[i] = kv[i].second; }
		have_ft = true;
		if (fread(&mode, 1, 1, fp) != 1) mode = 0;
	}
	struct Ent { uint64_t hash; uint32_t sub, cnt; uint64_t off; };
	std::vector<Ent> ents; std::vector<std::vector<uint64_t> > pos;
	if (mode == 'h') {
		int32_t k = 0, pre = 0; uint64_t tot = 0, tot_pos = 0;
		if (fread(&k, 4, 1, fp) != 1 || fread(&pre, 4, 1, fp) != 1 || fread(&tot, 8, 1, fp) != 1 || fread(&tot_pos, 8, 1, fp) != 1 || pre < 0 || pre > 20) return bad("index header of " + base);
		if (k != c->opt.k) return bad("the index was built with k = " + std::to_string(k) + ", the engine runs with k = " + std::to_string(c->opt.k));
		if (tot > hao_file_left(fp) / 16 || tot_pos > hao_file_left(fp) / 8) return bad("index header of " + base + ": more keys / positions than the file holds");
		pos.resize((size_t)1 << pre); ents.reserve(tot);
		std::vector<uint64_t> kk; std::vector<uint8_t> vv;
		for (uint32_t s = 0; s < (1u << pre); ++s) {
			uint64_t na = 0;
			if (!hao_kh_read(fp, 8, kk, vv) || fread(&na, 8, 1, fp) != 1) return bad("sub-table " + std::to_string(s) + " of " + base);
			if (na > hao_file_left(fp) / 8) return bad("positions of sub-table " + std::to_string(s) + ": more than the file holds");
			pos[s].resize(na);
			if (na && fread(pos[s].data(), 8, na, fp) != na) return bad("positions of sub-table " + std::to_string(s));
			for (size_t i = 0; i < kk.size(); ++i) {      // key = hash >> pre << 12 | count (htab.cpp:122-124, 303-314), value = offset of its list
				Ent e; e.hash = ((kk[i] >> 12) << pre) | s; e.sub = s; e.cnt = (uint32_t)(kk[i] & 4095); memcpy(&e.off, &vv[8 * i], 8);
				if (e.off > na || e.cnt > na - e.off) return bad("a list of sub-table " + std::to_string(s) + " leaves its position array");
				ents.push_back(e);
			}
		}
		if (ents.size() != tot) return bad("key count of " + base);
		have_pt = true;
	}
	if (!have_ft || !have_pt) return bad(base + " holds no filter table / position index");
	int32_t rounds = 0, hom = -1, het = -1, mnc = 100;
	if (fread(&rounds, 4, 1, fp) != 1 || fread(&hom, 4, 1, fp) != 1 || fread(&het, 4, 1, fp) != 1 || fread(&mnc, 4, 1, fp) != 1) return bad("tail of " + base);
	fclose(fp); fp = nullptr;
	if (number_of_round) *number_of_round = rounds;
	// ---- the read store (load_All_reads, Process_Read.cpp:127-232) ----
	fp = fopen((base + ".bin").c_str(), "rb");
	if (!fp) { hao_set_err(c, "cannot read " + base + ".bin"); return HAO_EINVAL; }
	int32_t adapter = 0; uint64_t index_size = 0, name_index_size = 0, n = 0, total_bases = 0, total_name = 0;
	if (fread(&adapter, 4, 1, fp) != 1 || fread(&index_size, 8, 1, fp) != 1 || fread(&name_index_size, 8, 1, fp) != 1 || fread(&n, 8, 1, fp) != 1 || fread(&total_bases, 8, 1, fp) != 1 ||
		fread(&total_name, 8, 1, fp) != 1 || n >= (1ULL << 28) || n > hao_file_left(fp) / 16) return bad("header of " + base + ".bin");      // (a read costs at least its N-site count and its length: 16 bytes)
	std::vector<uint64_t> ns_off(n + 1, 0), len64(n), pk_off(n + 1, 0); std::vector<uint32_t> ns, len(n); std::vector<uint8_t> packed;
	for (uint64_t i = 0; i < n; ++i) {
		uint64_t cnt = 0; if (fread(&cnt, 8, 1, fp) != 1 || cnt > hao_file_left(fp) / 8) return bad("N sites of read " + std::to_string(i));
		for (uint64_t j = 0; j < cnt; ++j) { uint64_t p; if (fread(&p, 8, 1, fp) != 1 || p >= (1ULL << 27)) return bad("N sites of read " + std::to_string(i)); ns.push_back((uint32_t)p); }
		ns_off[i + 1] = ns.size();
	}
	if (n && fread(len64.data(), 8, n, fp) != n) return bad("read lengths");
	for (uint64_t i = 0; i < n; ++i) { if (len64[i] >= (1ULL << 27)) return bad("read longer than 2^27"); len[i] = (uint32_t)len64[i]; pk_off[i + 1] = pk_off[i] + len64[i] / 4 + 1; }
	for (uint64_t i = 0; i < n; ++i) for (uint64_t j = ns_off[i]; j < ns_off[i + 1]; ++j) if (ns[j] >= len[i]) return bad("an N site of read " + std::to_string(i) + " lies beyond the read");
	if (pk_off[n] > hao_file_left(fp)) return bad("packed reads: shorter than the lengths say");
	packed.resize(pk_off[n] + 1);
	if (pk_off[n] && fread(packed.data(), 1, pk_off[n], fp) != pk_off[n]) return bad("packed reads");
	fclose(fp); fp = nullptr;      // (names, trio flags and the second copy of the peaks are not the engine's business)
	if (int rc = hao_set_reads(c, packed.data(), pk_off.data(), len.data(), n, ns_off.data(), ns.empty() ? nullptr : ns.data())) return rc;
	// ---- filter table ----
	c->h_ft_keys = ftk; c->h_ft_vals = ftv; const uint64_t nf = ftk.size();
	HIP_TRY(c->d_ft_keys.reserve(nf + 1)); HIP_TRY(c->d_ft_vals.reserve(nf + 1));
	if (nf) { HIP_TRY(hipMemcpyAsync(c->d_ft_keys.p, ftk.data(), nf * 8, hipMemcpyHostToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(c->d_ft_vals.p, ftv.data(), nf * 4, hipMemcpyHostToDevice, c->stream)); }
	if (int rc = hao_build_bucket(c, c->d_ft_keys.p, nf, 16, c->d_ft_bucket)) return rc;
	if (int rc = hao_ft_build_hash(c, nf)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->has_ft = true; c->ft_peak_hom = -1; c->ft_peak_het = -1; c->ft_cutoff = 0; memset(c->ft_hist, 0, sizeof(c->ft_hist));
	// ---- position index: keys ascending, every key's list in file order (= (rid, pos) order) ----
	std::sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.hash < b.hash; });
	const uint64_t nk = ents.size(); uint64_t np = 0; for (const Ent &e : ents) np += e.cnt;
	std::vector<uint64_t> keys(nk), start(nk), sinfo(np); std::vector<uint32_t> cnt(nk);
	{ uint64_t o = 0; for (uint64_t i = 0; i < nk; ++i) { const Ent &e = ents[i]; keys[i] = e.hash; start[i] = o; cnt[i] = e.cnt; memcpy(sinfo.data() + o, pos[e.sub].data() + e.off, (size_t)e.cnt * 8); o += e.cnt; } }
	HIP_TRY(c->d_ix_keys.reserve(nk + 1)); HIP_TRY(c->d_ix_start.reserve(nk + 1)); HIP_TRY(c->d_ix_cnt.reserve(nk + 1)); HIP_TRY(c->d_ix_sinfo.reserve(np + 8));
	if (nk) { HIP_TRY(hipMemcpyAsync(c->d_ix_keys.p, keys.data(), nk * 8, hipMemcpyHostToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(c->d_ix_start.p, start.data(), nk * 8, hipMemcpyHostToDevice, c->stream));
			  HIP_TRY(hipMemcpyAsync(c->d_ix_cnt.p, cnt.data(), nk * 4, hipMemcpyHostToDevice, c->stream)); }
	if (np) HIP_TRY(hipMemcpyAsync(c->d_ix_sinfo.p, sinfo.data(), np * 8, hipMemcpyHostToDevice, c->stream));
	c->ix_n_keys = nk; c->ix_n_pos = np; c->ix_n_sorted = np; c->ix_pad = 0;
	int bits = 16; while ((1ULL << bits) < nk / 2 && bits < 26) ++bits;
	if (int rc = hao_build_bucket(c, c->d_ix_keys.p, nk, bits, c->d_ix_bucket)) return rc;
	c->ix_bucket_bits = bits;
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->hom_cov = hom; c->het_cov = het; c->max_n_chain = mnc; memset(c->pt_hist, 0, sizeof(c->pt_hist));
	// ---- query side: read-ordered minimizers (with the loaded filter table) and their lookup results ----
	if (n == 0) { hao_set_err(c, "hao_index_load: no reads"); return HAO_EINVAL; }
	if (int rc = hao_sketch_run(c, 0, n, 1, c->opt.sample_dist, 1)) return rc;
	std::swap(c->d_ix_mz_x, c->d_mz_x); std::swap(c->d_ix_mz_info, c->d_mz_info); std::swap(c->d_ix_mz_off, c->d_mz_off);
	c->ix_n_mz = c->sk_total; c->sk_n = 0;
	const uint64_t m = c->ix_n_mz;
	if (m >= (1ULL << 32)) { hao_set_err(c, "more than 2^32 minimizers on one device"); return HAO_EUNSUPP; }
	HIP_TRY(c->d_ix_lk.reserve(m + 1));
	if (m) { hipLaunchKernelGGL(hao_lk_fill_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, c->d_ix_mz_x.p, m, hao_pt_view(c), c->d_ix_lk.p); HAO_CHECK_LAUNCH(); }
	c->lk_valid = true;
	c->h_ix_mz_off.resize(n + 1);
	HIP_TRY(hipMemcpyAsync(c->h_ix_mz_off.data(), c->d_ix_mz_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->has_pt = true; c->h_ix_valid = false;
	return HAO_OK;
}


// ---------------------------------------------------------------------------------------
// *.ovlp.source.bin / *.ovlp.reverse.bin (write_ma_hit_ts / load_ma_hit_ts / write_ma / read_ma, Overlaps.cpp:23280-23469): reader and writer of the format (include/hao.h)
// ---------------------------------------------------------------------------------------
#define HAO_MA_DISK_BYTES 42
static int hao_ovlp_bin_read_impl(const char *path, uint64_t *n_reads, uint8_t **flags, uint64_t **off, hao_ma_hit_t **hits)
{
	*n_reads = 0; *flags = nullptr; *off = nullptr; *hits = nullptr;
	FILE *fp = fopen(path, "rb");
	if (!fp) return HAO_EINVAL;
	int64_t n = 0;
	std::vector<uint8_t> fl; std::vector<uint64_t> of; std::vector<hao_ma_hit_t> hv;
	auto bad = [&]() { fclose(fp); return HAO_EINVAL; };
	if (fread(&n, 8, 1, fp) != 1 || n < 0 || (uint64_t)n > hao_file_left(fp) / 6) return bad();      // (a read costs at least its two flags and its length)
	fl.resize(2 * (size_t)n); of.assign((size_t)n + 1, 0);
	for (int64_t i = 0; i < n; ++i) {
		uint32_t len = 0;
		if (fread(&fl[2 * i], 1, 2, fp) != 2 || fread(&len, 4, 1, fp) != 1 || (uint64_t)len > hao_file_left(fp) / HAO_MA_DISK_BYTES) return bad();
		of[i + 1] = of[i] + len;
		unsigned char rec[HAO_MA_DISK_BYTES];
		for (uint32_t k = 0; k < len; ++k) {
			if (fread(rec, 1, HAO_MA_DISK_BYTES, fp) != HAO_MA_DISK_BYTES) return bad();
			hao_ma_hit_t h; memset(&h, 0, sizeof h);
			memcpy(&h.qns, rec, 8); memcpy(&h.qe, rec + 8, 4); memcpy(&h.tn, rec + 12, 4); memcpy(&h.ts, rec + 16, 4); memcpy(&h.te, rec + 20, 4);
			h.el = rec[24]; h.no_l_indel = rec[25];
			memcpy(&h.ml, rec + 26, 4); memcpy(&h.rev, rec + 30, 4); memcpy(&h.bl, rec + 34, 4); memcpy(&h.del, rec + 38, 4);
			hv.push_back(h);
		}
	}
	if (hao_file_left(fp) != 0) return bad();      // trailing bytes: not this format
	fclose(fp);
	*flags = (uint8_t*)malloc(fl.size() + 1); *off = (uint64_t*)malloc(of.size() * 8); *hits = (hao_ma_hit_t*)malloc((hv.size() + 1) * sizeof(hao_ma_hit_t));
	if (!*flags || !*off || !*hits) { free(*flags); free(*off); free(*hits); *flags = nullptr; *off = nullptr; *hits = nullptr; return HAO_ENOMEM; }
	if (!fl.empty()) memcpy(*flags, fl.data(), fl.size());
	memcpy(*off, of.data(), of.size() * 8);
	if (!hv.empty()) memcpy(*hits, hv.data(), hv.size() * sizeof(hao_ma_hit_t));
	*n_reads = (uint64_t)n;
	return HAO_OK;
}
static int hao_ovlp_bin_write_impl(const char *path, uint64_t n_reads, const uint8_t *flags, const uint64_t *off, const hao_ma_hit_t *hits)
{
	if (!path || !off || (n_reads && !flags)) return HAO_EINVAL;
	for (uint64_t i = 0; i < n_reads; ++i) if (off[i + 1] < off[i] || off[i + 1] - off[i] > 0xffffffffULL) return HAO_EINVAL;
	FILE *fp = fopen(path, "wb");
	if (!fp) return HAO_EINVAL;
	const int64_t n = (int64_t)n_reads; bool ok = fwrite(&n, 8, 1, fp) == 1;
	for (uint64_t i = 0; ok && i < n_reads; ++i) {
		const uint32_t len = (uint32_t)(off[i + 1] - off[i]);
		ok = fwrite(flags + 2 * i, 1, 2, fp) == 2 && fwrite(&len, 4, 1, fp) == 1;
		for (uint64_t k = off[i]; ok && k < off[i + 1]; ++k) {
			const hao_ma_hit_t &h = hits[k]; unsigned char rec[HAO_MA_DISK_BYTES];
			memcpy(rec, &h.qns, 8); memcpy(rec + 8, &h.qe, 4); memcpy(rec + 12, &h.tn, 4); memcpy(rec + 16, &h.ts, 4); memcpy(rec + 20, &h.te, 4);
			rec[24] = h.el; rec[25] = h.no_l_indel;
			memcpy(rec + 26, &h.ml, 4); memcpy(rec + 30, &h.rev, 4); memcpy(rec + 34, &h.bl, 4); memcpy(rec + 38, &h.del, 4);
			ok = fwrite(rec, 1, HAO_MA_DISK_BYTES, fp) == HAO_MA_DISK_BYTES;
		}
	}
	ok = (fclose(fp) == 0) && ok;
	return ok ? HAO_OK : HAO_EINVAL;
}
