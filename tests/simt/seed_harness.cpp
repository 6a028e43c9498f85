// TEST INFRASTRUCTURE: the seed stage's kernels (hifiasm_amd/csrc/hao_query.cuh, hao_query3.cuh - the SOURCES the device library is built from, compiled by g++
// against tests/simt/hip/hip_runtime.h) run on the emulated workgroup, launched the way hao_batch.hpp launches them (seed_unpack_kernel, the scan,
// seed_segments_kernel, then the three tiers: 512 slots for every read, 1024 and 2048 slots for the reads that overflowed).  Called by tests/test_simt_seed_cpu.py
// only, which compares the hits and the group lists with the oracle's restatement of minimizers_qgen0.
#include "hao_query3.cuh"
#include "hao_query5.cuh"
#include <execinfo.h>
#include <signal.h>
static void simt_segv(int) { void *bt[40]; int n = backtrace(bt, 40); fprintf(stderr, "SIGSEGV in work-item %d of block %u\n", hao_simt::g.cur, blockIdx.x); backtrace_symbols_fd(bt, n, 2); _exit(3); }

namespace {
struct Sim {
	std::vector<uint64_t> s_start, a_off, seg, g_tmp, g_cnt, s_pk; std::vector<uint32_t> s_n, q_pos, q_cnt, ovf; std::vector<hao_hit_t> hits; std::vector<uint16_t> hq;
};
int fail(char *err, int cap, const std::string &m) { snprintf(err, cap, "%s", m.c_str()); return 1; }
}

// mode 0: the table kernels as the library launches them (QL instances; overflowing reads through seed_bin3_kernel);
// 2: HAO_DBG_FORCE=noql (per-minimizer tables possibly in global memory: qcap_force > 0 caps the LDS table to force that path)
// 12 / 13: the list-major kernel (hao_query5.cuh) takes every read first; the reads it leaves go through the table kernels as in mode 0 (stats[6] = reads it left)
// blocks: the reads to run in the first launch (others keep empty output); returns 0 or 1 with a message
extern "C" int simt_seed_run(uint64_t n, const uint64_t *mz_off, const uint64_t *mz_info, const uint64_t *lk, const uint32_t *wgt, const uint64_t *sinfo, const uint32_t *len, uint64_t n_total,
		int mode, uint32_t qcap_force, const uint32_t *blocks, uint32_t n_blocks, int want_hq,
		uint64_t *seg_out, uint32_t *hits_out, uint64_t hits_cap, uint64_t *g_tmp_out, uint64_t *g_cnt_out, uint16_t *hq_out, uint64_t *stats, char *err, int errcap)
{
	if (getenv("SIMT_DBG")) signal(SIGSEGV, simt_segv);
	hao_simt::g.n_exchange = hao_simt::g.n_barrier = hao_simt::g.n_switch = 0;
	Sim S; const uint64_t nm = mz_off[n];
	S.s_start.assign(nm + 1, 0); S.s_n.assign(nm + 1, 0); S.q_pos.assign(nm + 1, 0); S.q_cnt.assign(nm + 1, 0); S.a_off.assign(nm + 2, 0); S.seg.assign(n + 2, 0); S.s_pk.assign(nm + 1, 0);
	using hao_simt::launch;
	if (launch((unsigned)((nm + 256) / 256), 256, 0, [&] { seed_unpack_kernel(lk, mz_info, 0, nm, wgt, S.s_start.data(), S.s_n.data(), S.q_pos.data(), S.q_cnt.data(), S.s_pk.data()); })) return fail(err, errcap, hao_simt::g.error);
	for (uint64_t i = 0; i <= nm; ++i) S.a_off[i + 1] = S.a_off[i] + S.s_n[i];      // hao_scan_u32 (exclusive, nm + 1 entries + the total)
	if (launch((unsigned)((n + 256) / 256), 256, 0, [&] { seed_segments_kernel(mz_off, 0, n, 0, S.a_off.data(), S.seg.data()); })) return fail(err, errcap, hao_simt::g.error);
	const uint64_t A = S.a_off[nm];
	if (A > hits_cap) return fail(err, errcap, "hits_cap too small");
	S.hits.assign(A + 1, hao_hit_t{0, 0, 0, 0}); S.g_tmp.assign(A + 1, 0); S.g_cnt.assign(n + 2, ~0ULL); S.hq.assign(A + 64, 0xffff); S.ovf.assign(2 * (n + 1), 0);
	uint64_t max_q = 1; for (uint64_t r = 0; r < n; ++r) max_q = std::max<uint64_t>(max_q, mz_off[r + 1] - mz_off[r]);
	int tb = 1; while ((1ULL << tb) < n_total) ++tb;
	hao_seed_args sa; memset(&sa, 0, sizeof sa);
	sa.mz_off = mz_off; sa.mz_info = mz_info; sa.rid_lo = 0; sa.mz0 = 0; sa.s_start = S.s_start.data(); sa.s_n = S.s_n.data(); sa.a_off = S.a_off.data(); sa.seg = S.seg.data();
	sa.sinfo = sinfo; sa.len = len; sa.q_pos = S.q_pos.data(); sa.q_cnt = S.q_cnt.data(); sa.hits = S.hits.data(); sa.g_tmp = S.g_tmp.data(); sa.g_cnt = S.g_cnt.data(); sa.n_sel = n; sa.tb = tb;
	sa.qcap = (uint32_t)std::min<uint64_t>((max_q + 63) & ~63ULL, HAO_QTAB_CAP);
	if (mode == 2 && qcap_force) sa.qcap = qcap_force;
	sa.dbg = nullptr; sa.hq = want_hq ? S.hq.data() : nullptr;
	if (mode != 2 && max_q > HAO_QTAB_CAP) return fail(err, errcap, "QL instances need every read's minimizer table in LDS");
	unsigned long long ovf_cnt[2] = {0, 0}; uint32_t *ovf1 = S.ovf.data(), *ovf2 = S.ovf.data() + (n + 1);
	const size_t lds_tile = std::max<size_t>((size_t)512 * (sizeof(hao_stage_t) + 4), 12 * 512), lds_q = 12 * (size_t)sa.qcap + 16;
	size_t lds1 = (size_t)22 * 512 + lds_tile + lds_q, lds2 = (size_t)22 * 1024 + std::max<size_t>(lds_tile, 12 * 1024) + lds_q, lds3 = (size_t)22 * 2048 + std::max<size_t>(lds_tile, 12 * 2048) + lds_q;
	if (mode == 0 || mode >= 12) { lds2 = hao_seed3_lds<10>::FIXED + lds_q; lds3 = hao_seed3_lds<11>::FIXED + lds_q; }
	const uint32_t *nil32 = nullptr; const unsigned long long *nil64 = nullptr;
	std::vector<uint32_t> ovf0(n + 4, 0); unsigned long long ovf0_cnt = 0, next_read = 0;      // (next_read: the kernel's read cursor, zero at launch)
	const uint32_t max_n = getenv("SIMT_SEED_MAXN") ? (uint32_t)atoi(getenv("SIMT_SEED_MAXN")) : 0xffffffffu;      // reads with more seed hits go to the table kernels
	if (mode == 12 || mode == 13) {      // the list-major kernel (hao_query5.cuh): persistent workgroups of 512 work-items, every read of the set; SIMT_SEED_GRID workgroups (default 3: every workgroup
		// runs several reads through its pipeline); mode 12: 16-bit offsets when every read is shorter than 64 kb (what the library does), mode 13: 32-bit offsets
		bool b16 = mode == 12; for (uint64_t r = 0; r < n_total; ++r) if (len[r] >= 65536) b16 = false;
		const unsigned grid = (unsigned)std::min<uint64_t>(n, getenv("SIMT_SEED_GRID") ? (uint64_t)atoi(getenv("SIMT_SEED_GRID")) : 3);
		const bool wide = max_q > 2 * HAO_L5_THREADS || getenv("SIMT_SEED_WIDE");      // (three minimizers per thread and fewer record registers: what the library launches for batches with long reads)
		std::function<void()> call;
		if (b16 && wide) call = [&] { seed_lds_kernel<true, 3, 8, false>(sa, sinfo, len, S.s_pk.data(), max_n, 176u, ovf0.data(), &ovf0_cnt, &next_read); };
		else if (b16) call = [&] { seed_lds_kernel<true, 2, 16, false>(sa, sinfo, len, S.s_pk.data(), max_n, 176u, ovf0.data(), &ovf0_cnt, &next_read); };
		else if (wide) call = [&] { seed_lds_kernel<false, 3, 8, false>(sa, sinfo, len, S.s_pk.data(), max_n, 176u, ovf0.data(), &ovf0_cnt, &next_read); };
		else call = [&] { seed_lds_kernel<false, 2, 16, false>(sa, sinfo, len, S.s_pk.data(), max_n, 176u, ovf0.data(), &ovf0_cnt, &next_read); };
		if (launch(grid, HAO_L5_THREADS, b16 ? hao_l5_lds<true, 2>::TOTAL : hao_l5_lds<false, 2>::TOTAL, call)) return fail(err, errcap, hao_simt::g.error);
		stats[6] = ovf0_cnt;
		if (ovf0_cnt) {
			std::function<void()> call = [&] { seed_bin_kernel<9, 1, 512, true>(sa, ovf0.data(), &ovf0_cnt, ovf1, &ovf_cnt[0]); };
			if (launch((unsigned)ovf0_cnt, 256, lds1, call)) return fail(err, errcap, hao_simt::g.error);
		}
	}
	// first launch: the chosen reads only (a block per read; blockIdx.x = the read)
	if (mode < 12) {
		hao_simt::g.body = nullptr;
		for (uint32_t b = 0; b < n_blocks; ++b) {
			std::function<void()> call;
			if (mode == 2) call = [&] { seed_bin_kernel<9, 0, 512, false>(sa, nil32, nil64, ovf1, &ovf_cnt[0]); };
			else call = [&] { seed_bin_kernel<9, 0, 512, true>(sa, nil32, nil64, ovf1, &ovf_cnt[0]); };
			hao_simt::g.body = call; hao_simt::g.nthreads = 256; hao_simt::g.error.clear(); hao_simt::g.dyn_lds.assign(lds1 + 64, (char)0xa5);
			blockDim = {256, 1, 1}; gridDim = {(unsigned)n, 1, 1}; blockIdx = {blocks[b], 0, 0};
			if (!hao_simt::run_block()) return fail(err, errcap, hao_simt::g.error);
		}
	}
	stats[2] = ovf_cnt[0];
	if (ovf_cnt[0]) {
		std::function<void()> call;
		if (mode == 0 || mode >= 12) call = [&] { seed_bin3_kernel<10, 1, 4>(sa, ovf1, &ovf_cnt[0], ovf2, &ovf_cnt[1]); };
		else call = [&] { seed_bin_kernel<10, 1, 512, false>(sa, ovf1, &ovf_cnt[0], ovf2, &ovf_cnt[1]); };
		if (launch((unsigned)ovf_cnt[0], 256, lds2, call)) return fail(err, errcap, hao_simt::g.error);
	}
	stats[3] = ovf_cnt[1];
	if (ovf_cnt[1]) {
		std::function<void()> call;
		if (mode == 0 || mode >= 12) call = [&] { seed_bin3_kernel<11, 2, 4>(sa, ovf2, &ovf_cnt[1], (uint32_t*)nullptr, (unsigned long long*)nullptr); };
		else call = [&] { seed_bin_kernel<11, 2, 512, false>(sa, ovf2, &ovf_cnt[1], (uint32_t*)nullptr, (unsigned long long*)nullptr); };
		if (launch((unsigned)ovf_cnt[1], 256, lds3, call)) return fail(err, errcap, hao_simt::g.error);
	}
	memcpy(seg_out, S.seg.data(), (n + 1) * 8);
	for (uint64_t i = 0; i < A; ++i) { hits_out[4 * i] = S.hits[i].w0; hits_out[4 * i + 1] = S.hits[i].offset; hits_out[4 * i + 2] = S.hits[i].self_offset; hits_out[4 * i + 3] = S.hits[i].cnt; }
	memcpy(g_tmp_out, S.g_tmp.data(), A * 8); memcpy(g_cnt_out, S.g_cnt.data(), (n + 1) * 8);
	if (want_hq) memcpy(hq_out, S.hq.data(), A * 2);
	stats[0] = hao_simt::g.n_exchange; stats[1] = hao_simt::g.n_barrier; stats[4] = A; stats[5] = hao_simt::g.n_switch;
	return 0;
}

// the cross-lane vocabulary on its own (tests pin the emulator's DPP / permute semantics against closed forms): op 0 inclusive scan (hao_wave_incl_scan_u32),
// 1 wave max (hao_wave_max_i32), 2 previous lane (hao_wave_shr1), 3 next lane (hao_wave_shl1), 4 hao_match_key<9>, 5 hao_match_bits (9 bits), 6 hao_seed_locate over ao[]
static void vocab_kernel(int op, const uint32_t *in, uint64_t *out, const uint32_t *ao, uint32_t nk)
{
	const int lane = hao_lane(); const uint32_t v = in[threadIdx.x];
	uint64_t r = 0;
	if (op == 0) r = hao_wave_incl_scan_u32(v);
	else if (op == 1) r = (uint64_t)(int64_t)hao_wave_max_i32((int32_t)v);
	else if (op == 2) r = hao_wave_shr1(v, 0xabcdu);
	else if (op == 3) r = hao_wave_shl1(v, 0xabcdu);
	else if (op == 4) r = hao_match_key<9>(v & 511, (v >> 31) != 0);
	else if (op == 5) r = hao_match_bits(v & 511, (v >> 31) != 0, 9);
	else if (op == 6) { uint32_t kc = 0, k = 0; for (uint32_t x0 = 0; x0 < ao[nk]; x0 += 64) { k = hao_seed_locate(ao, nk, kc, x0, lane); if (x0 + lane < ao[nk]) out[x0 + lane] = k; } return; }
	out[threadIdx.x] = r;
}
extern "C" int simt_vocab(int op, const uint32_t *in, uint64_t *out, const uint32_t *ao, uint32_t nk, char *err, int errcap)
{
	if (hao_simt::launch(1, 64, 0, [&] { vocab_kernel(op, in, out, ao, nk); })) return fail(err, errcap, hao_simt::g.error);
	return 0;
}

// the emulator's own alarms, on kernels that deserve them: 0 a ballot only half of the wave reaches (the others go straight to the barrier), 1 two ballots at
// different source positions, 2 a cross-lane operation one lane reaches while the others of its wave stand at the barrier,
// 3 a store past the dynamic LDS the launch asked for, 4 (control) the same shapes written correctly.  Returns 1 with the message when the launch was refused.
static void bad_kernel(int which, uint32_t *out)
{
	uint32_t *lds = (uint32_t*)hao_simt::dyn_lds(); const int lane = hao_lane();
	if (which == 0) { if (lane < 32) out[threadIdx.x] = (uint32_t)__popcll(__ballot(1)); __syncthreads(); }
	else if (which == 1) {      // (operations are told apart by source LINE)
		unsigned long long b;
		if (lane & 1) b = __ballot(lane > 3);
		else b = __ballot(lane > 5);
		out[threadIdx.x] = (uint32_t)b;
	}
	else if (which == 2) { if (lane == 0) out[0] = (uint32_t)__popcll(__ballot(1)); __syncthreads(); }
	else if (which == 3) { lds[threadIdx.x] = 1; if (threadIdx.x == 7) lds[64 + 3] = 2; __syncthreads(); out[threadIdx.x] = lds[(threadIdx.x + 1) & 63]; }
	else { const unsigned long long b = __ballot(lane < 32); lds[threadIdx.x & 63] = (uint32_t)__popcll(b); __syncthreads(); out[threadIdx.x] = lds[(threadIdx.x + 1) & 63]; }
}
extern "C" int simt_selfcheck(int which, uint32_t *out, char *err, int errcap)
{
	if (hao_simt::launch(1, 64, 256, [&] { bad_kernel(which, out); })) return fail(err, errcap, hao_simt::g.error);
	return 0;
}
