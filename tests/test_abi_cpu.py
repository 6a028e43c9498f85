"""CPU-side checks of the drop-in boundary (no compute calls): the C-ABI library loads, exports every
symbol include/hao.h declares, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hao.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hao_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from hifiasm_amd import api
    L = C.CDLL(api.lib_path())
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hao.h but not exported by libhao.so"
    assert set(api.ABI_SYMBOLS) <= set(names)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from hifiasm_amd.api import Engine, HaoError
    with pytest.raises(HaoError):
        Engine(0)


def test_product_never_touches_oracle():
    """nothing under hifiasm_amd/ or include/ may import, link or name the oracle"""
    bad = []
    for base in ("hifiasm_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hpp", ".cuh", ".hip", ".h", ".c")) and fn != "build.py":   # build.py only COMPILES the oracle
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"oracle_py|liboracle|hao_oracle|hao_or_", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_product_never_touches_the_emulated_library():
    """tests/simt (the device sources compiled for an emulated workgroup) is test infrastructure: nothing under hifiasm_amd/, include/ or integration/, nor
    bench.py, names it; __graft_entry__.py only BUILDS it (build()), smoke() runs on libhao.so"""
    bad = []
    for base in ("hifiasm_amd", "include", "integration"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hpp", ".cuh", ".hip", ".h", ".c", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"libhao_simt|simt_build|simt_suite|hao_simt::|tests/simt/_build", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    assert not re.search(r"simt", open(os.path.join(ROOT, "bench.py")).read())
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "simt" not in entry[entry.index("def smoke"):], "smoke() must run the device library"


def test_opt_defaults_match_reference():
    """init_opt defaults (CommandLines.cpp:243-380) mirrored by hao_opt_default"""
    from hifiasm_amd import api
    o = api.Opt()
    api.lib().hao_opt_default(C.byref(o))
    assert (o.k, o.w, o.hpc, o.sample_dist, o.rewin, o.min_hist_cnt, o.max_kmer_cnt, o.max_n_chain) == (51, 51, 1, 500, 1000, 5, 2000, 100)
    assert o.high_factor == 5.0 and o.is_ont == 0


def test_index_partition_headroom():
    """Only a rank's HASH PARTITION of the index is limited to 2^32 records (its sort carries a 32-bit arrival index); the replicated index itself is not
    (48-bit list starts; tests/test_gpu_sharded.py::test_replicated_index_beyond_2_32, tests/test_gpu_altpaths.py::test_index_positions_beyond_2_32).  What
    BASELINE.json's multi-GPU workloads need against that, from the minimizer densities the REFERENCE measured on the full-size fixtures (sum of count x
    histogram of ha_pt_gen = minimizers of the pass): a partition of configs[3] / [4] on 8 GPUs, and of a 50x human set whose replicated index exceeds 2^32."""
    import numpy as np
    from hifiasm_amd.workloads import WORKLOADS
    limit = 1 << 32
    src = open(os.path.join(ROOT, "hifiasm_amd", "csrc", "hao_tables.hpp")).read()
    assert "minimizers in the replicated index" not in src          # the round-3 refusal is gone
    out = {}
    for fixture, target in (("chr1_250M_hifi30x", "human3G_hifi40x"), ("ont50M_30x", "ont_human_30x")):
        g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
        h = g["pt_hist"].astype(np.int64)
        n_mz = int((h * np.arange(h.size)).sum())
        gs, cov = WORKLOADS[fixture][0], WORKLOADS[fixture][1]
        density = n_mz / float(gs * cov)                       # minimizers per sequenced base
        tg, tcov = WORKLOADS[target][0], WORKLOADS[target][1]
        need = density * tg * tcov
        out[target] = (density, need, need / 8 / limit)
        assert need / 8 < limit / 4, (target, need)            # a partition on 8 GPUs: far below (and ~60 B per record while it is built: 2^32 records would not fit a device)
    d_hifi = out["human3G_hifi40x"][0]
    assert d_hifi * 3.1e9 * 50 > limit                         # human 50x: the replicated index is beyond 2^32 records - allowed now - ...
    assert d_hifi * 3.1e9 * 50 / 8 < limit / 4                 # ... while its partitions are not
    print("[index headroom]", {k: (round(v[0], 5), f"{v[1] / 1e9:.2f} G minimizers", f"partition on 8 GPUs: {100 * v[2]:.0f} % of 2^32") for k, v in out.items()})


def test_rank_memory_plan_of_the_multi_gpu_configs():
    """Every resident buffer of one rank, phase by phase (hifiasm_amd/memplan.py: ha_ft_gen with its hash-range passes, ha_pt_gen incl. the all-gather slots that
    become the replicated index, the all-reads pass with both delivery sets), for BASELINE.json's configs[3] (3 Gb human, 40x HiFi, 8 GPUs) and configs[4]
    (30x ONT, 8 GPUs) against the 288 GB of an MI355X - with the minimizer densities the REFERENCE measured on the full-size fixtures.  configs[3]'s ha_ft_gen
    is the phase that does not fit in one pass (15 Gbases per rank: four 8-byte-per-base buffers during the exchange); the plan must say so and fit with passes.
    (tests/test_gpu_zz_rankshare.py runs that rank's share through ha_ft_gen on a device and compares the measured peak with this plan.)"""
    import numpy as np
    from hifiasm_amd import memplan
    from hifiasm_amd.workloads import WORKLOADS, n_reads_of
    src = open(os.path.join(ROOT, "hifiasm_amd", "csrc", "hao_tables.hpp")).read()
    assert f"#define HAO_FT_BYTES_PER_SLOT {memplan.FT_PER_SLOT}" in src and f"#define HAO_FT_BYTES_PER_SLOT_SHARDED {memplan.FT_PER_SLOT_SHARDED}" in src      # the plan and the engine use the same figures
    assert "#define HAO_FT_CHUNK_SLOTS (1ULL << 28)" in src and memplan.FT_CHUNK_SLOTS == 1 << 28 and f"#define HAO_FT_RUN_BYTES_PER_SLOT {memplan.FT_RUN_PER_SLOT}" in src
    plans = {}
    for fixture, target in (("chr1_250M_hifi30x", "human3G_hifi40x"), ("ont50M_30x", "ont_human_30x")):
        g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
        h = g["pt_hist"].astype(np.int64)
        gs, cov, rl, err = WORKLOADS[fixture][:4]
        density = float((h * np.arange(h.size)).sum()) / float(gs * cov)
        tg, tcov, trl, terr = WORKLOADS[target][:4]
        hits_per_read = 0.92 * density * trl * tcov          # configs[2]: 11.9 k seed hits per read for 0.0287 x 15 000 x 30 = 12.9 k (list length ~ coverage)
        # (configs[4]: noisy reads are counted through the Bloom filter, the reference's default - an exact count would keep ~0.4 distinct k-mers per base)
        pl = memplan.rank_plan(float(tg) * tcov, n_reads_of(target), 8, density, hits_per_read, float(tg), err=terr, bloom=terr > 0.005)
        plans[target] = pl
        for phase in ("ft_gen", "pt_gen", "all_reads_pass"):
            assert pl[phase] < 0.9 * memplan.HBM_BYTES, (target, phase, pl[phase] / 1e9)
    p3 = plans["human3G_hifi40x"]
    assert p3["passes_ft"] >= 2                                 # one pass would need 46 B x 15 Gbases = 690 GB
    assert memplan.FT_PER_SLOT_SHARDED * 3e9 * 40 / 8 > memplan.HBM_BYTES
    one = memplan.rank_plan(250e6 * 30, 500_000, 1, 0.02873, 11_900, 250e6)      # configs[2] on one GPU: everything would fit at once (20 B x 7.5 Gbases = 150 GB); two passes because a pass's
    assert one["passes_ft"] == 2 and one["peak"] < 0.9 * memplan.HBM_BYTES       # buffers stop at 2^32 slots (allocating bigger ones costs more than hashing the reads again)
    assert memplan.ft_passes(4.0e9, 280e9, False) == 1 and memplan.ft_passes(4.4e9, 280e9, False) == 2
    print("[rank memory plan, 8 GPUs]", {k: {q: (round(v[q] / 1e9, 1) if isinstance(v[q], float) else v[q]) for q in ("passes_ft", "ft_gen", "pt_gen", "all_reads_pass", "index", "batches_per_pass")} for k, v in plans.items()})


def test_ctypes_mirrors_match_the_header(tmp_path):
    """hifiasm_amd/api.py mirrors four structs of include/hao.h by hand: a C program prints the header's sizes and field offsets, ctypes must agree
    (the delivery view and the chain header changed shape in round 3)."""
    import subprocess
    from hifiasm_amd import api
    src = tmp_path / "sz.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "hao.h"
#define S(t) printf(#t " %zu\\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\\n", offsetof(t, f))
int main(void) {
    S(hao_opt_t); O(hao_opt_t, high_factor); O(hao_opt_t, hg_size);
    S(hao_pass_t); O(hao_pass_t, mcopy_rate); O(hao_pass_t, ocv_w);
    S(hao_chain_hdr_t); O(hao_chain_hdr_t, pos);
    S(hao_delivery_t); O(hao_delivery_t, n_pos); O(hao_delivery_t, bytes); O(hao_delivery_t, ol_off); O(hao_delivery_t, chains); O(hao_delivery_t, cl_exc); O(hao_delivery_t, copy_ms);
    S(hao_hit_t); S(hao_ovlp_t); S(hao_exc_t); S(hao_qmz_t); S(hao_ed_task_t); S(hao_ed_result_t); S(hao_trace_result_t);
    return 0;
}
''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    got = {k: int(v) for k, v in got.items()}
    for name, cls in (("hao_opt_t", api.Opt), ("hao_pass_t", api.Pass), ("hao_chain_hdr_t", api.ChainHdr), ("hao_delivery_t", api.Delivery)):
        assert C.sizeof(cls) == got[name], name
    for key, v in got.items():
        if "." in key:
            t, f = key.split(".")
            cls = {"hao_opt_t": api.Opt, "hao_pass_t": api.Pass, "hao_chain_hdr_t": api.ChainHdr, "hao_delivery_t": api.Delivery}[t]
            assert getattr(cls, f).offset == v, key
    # record sizes the tests and the decoder assume (numpy views of the delivered arrays)
    assert (got["hao_hit_t"], got["hao_ovlp_t"], got["hao_exc_t"], got["hao_qmz_t"], got["hao_ed_task_t"], got["hao_ed_result_t"], got["hao_trace_result_t"]) == (16, 48, 32, 8, 40, 8, 24)
