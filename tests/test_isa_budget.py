"""CPU (hipcc cross-compiles gfx950 without a GPU): the resource budgets the hot kernels' occupancy rests on, read from the compiler's own
assembly (`hipcc -save-temps`, tools/isa_stats.py).  DESIGN.md section 4: two 512-thread scatter workgroups per CU need <= 128 VGPRs per lane
and <= 80 KiB of LDS each; the 1024-thread kernels (k = 13 scatter, k <= 8 LDS histogram, histogram pass) one workgroup per CU within 160 KiB;
the kernels of the headline path must not spill.  Round 4 found a kernel the compiler had serialised this way (31 VALU instructions per
element behind every LDS atomic): what the back end made of the source is part of what is tested."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    import isa_stats
    d = tmp_path_factory.mktemp("isa")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-save-temps",
           "-o", str(d / "lib.so"), os.path.join(ROOT, "kmerdb_amd", "csrc", "kdb_engine.hip"), "-lz", "-lpthread"]
    subprocess.check_call(cmd, cwd=str(d), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    s = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
    assert len(s) == 1
    return isa_stats.kernel_stats(str(d / s[0]))


def _one(isa, *needles):
    hits = [(n, v) for n, v in isa.items() if all(x in n for x in needles)]
    assert len(hits) == 1, (needles, [n[:120] for n, _ in hits])
    return hits[0][1]


def test_headline_scatter_kernel_keeps_two_workgroups_per_cu_and_does_not_spill(isa):
    # k = 12, drop mode, canonical, equal-length reads: BASELINE config 2's kernel
    v = _one(isa, "scatter_bases_kernel<unsigned int, unsigned short, 512, 64, 16, false, true, 12, 512, false>")
    assert v["scratch"] == 0 and v["vgprs"] <= 128 and 2 * v["lds"] <= LDS_PER_CU, v
    assert v["ds_rtn_atomics"] >= 16                      # (the sixteen slot requests of a tile are there, unrolled)
    # N-expansion mode and ragged batches: the same two workgroups per CU (the tile images carry the N lists)
    for needle in ("512, 64, 16, true, true, 12, 512, false>", "512, 64, 16, true, true, 12, 512, true>", "512, 64, 16, false, true, 12, 512, true>"):
        w = _one(isa, "scatter_bases_kernel<unsigned int, unsigned short, " + needle)
        assert w["vgprs"] <= 128 and 2 * w["lds"] <= LDS_PER_CU, (needle, w)


def test_two_level_kernels_keep_two_workgroups_per_cu(isa):
    # the 64-byte-line forms (options l1_wide_lines = 0 / l2_wide_lines = 0): two workgroups of 512 threads per CU
    for needle in ("scatter_bases_kernel<unsigned int, kdb::u24, 256, 64, 8, false, true, 0, 512, false>",
                   "scatter_bases_kernel<unsigned long, unsigned int, 512, 32, 8, false, true, 0, 512, false>",
                   "scatter_bases_kernel<unsigned int, kdb::u24, 256, 64, 8, false, true, 15, 512, false>",      # config 3's level 1, shifts compiled in (round 5)
                   "scatter_ids_kernel<kdb::u24, unsigned short, 512, 64, false, 512>", "scatter_ids_kernel<kdb::u24, unsigned short, 512, 64, true, 512>",
                   "scatter_ids_kernel<unsigned int, unsigned short, 512, 64, false, 512>"):
        v = _one(isa, needle)
        assert v["scratch"] == 0 and v["vgprs"] <= 128 and 2 * v["lds"] <= LDS_PER_CU, (needle, v)


def test_kernels_that_write_128_byte_pieces_fit_one_workgroup_of_1024_threads_per_cu(isa):
    """Round 5, the default: rings of twice the elements, 128 KiB of them per workgroup, sixteen waves -- 128 VGPRs per lane, no spill on the
    paths of BASELINE's configurations (k = 12 headline, config 3's two levels, config 4's two levels)."""
    for needle in ("scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, false, true, 12, 1024, false>",      # the headline
                   "scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, false, true, 12, 1024, true>",
                   "scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, false, true, 11, 1024, false>",      # k = 9 ... 11, compiled in
                   "scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, false, true, 9, 1024, false>",
                   "scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, false, true, 0, 1024, false>",       # (another bucket field: generic)
                   "scatter_bases_kernel<unsigned int, kdb::u24w, 128, 256, 16, false, true, 15, 1024, false>",      # config 3, level 1: one placement round per tile
                   "scatter_bases_kernel<unsigned int, kdb::u24w, 128, 256, 16, false, true, 0, 1024, false>",
                   "scatter_bases_kernel<unsigned int, kdb::u24w, 256, 128, 8, false, true, 15, 1024, false>",       # (option l1_one_round = 0)
                   "scatter_bases_kernel<unsigned int, kdb::u24w, 256, 128, 8, false, true, 0, 1024, false>",
                   "scatter_bases_kernel<unsigned long, kdb::u32w, 512, 64, 8, false, true, 17, 1024, false>",       # config 4, level 1 (compiled for k = 17)
                   "scatter_bases_kernel<unsigned long, kdb::u32w, 512, 64, 8, false, true, 0, 1024, false>",        # k = 17, level 1, generic
                   "scatter_ids_kernel<kdb::u24, kdb::u16w, 512, 128, true, 1024>", "scatter_ids_kernel<kdb::u24, kdb::u16w, 512, 128, false, 1024>",
                   "scatter_ids_kernel<unsigned int, kdb::u16w, 512, 128, false, 1024>"):
        v = _one(isa, needle)
        # (config 3's compiled one-round kernel keeps three dwords in scratch at 126 VGPRs and is still 4 % faster than the generic one beside it,
        #  which does not spill: profiles/r05/l1_one_round_ab.txt)
        # (and the forms for ragged batches keep three dwords there since they hold two tiles' record-start offsets: still 13 % faster than before)
        spill_ok = 16 if ("128, 256, 16, false, true, 15, 1024, false" in needle or needle.endswith("1024, true>")) else 0
        assert v["scratch"] <= spill_ok and v["vgprs"] <= 128 and v["lds"] <= LDS_PER_CU, (needle, v)
    # N-expansion mode: the tile images carry the N lists too; still one workgroup per CU (spills are tolerated there, as at k = 13)
    for needle in ("scatter_bases_kernel<unsigned int, kdb::u16w, 512, 128, 16, true, true, 12, 1024, false>",
                   "scatter_bases_kernel<unsigned int, kdb::u24w, 256, 128, 8, true, true, 0, 1024, true>"):
        v = _one(isa, needle)
        assert v["vgprs"] <= 128 and v["lds"] <= LDS_PER_CU, (needle, v)


def test_one_workgroup_per_cu_kernels_fit_the_lds(isa):
    for needle in ("scatter_bases_kernel<unsigned int, unsigned short, 1024, 64, 16, false, true, 13, 1024, false>",
                   "count_smallk_kernel<false, true, true, false>", "page_hist_kernel<true>", "page_hist_kernel<false>"):
        v = _one(isa, needle)
        assert v["vgprs"] <= 128 and v["lds"] <= LDS_PER_CU, (needle, v)


def test_16_bit_histogram_adds_stay_straight_line(isa):
    """The eight returning atomics of a page chunk are issued together: not one `s_waitcnt lgkmcnt(0)` per atomic (round 4: there were 69 in
    this kernel, one behind each of its atomics, and a dozen register copies with each)."""
    v = _one(isa, "page_hist_kernel<true>")
    assert v["ds_rtn_atomics"] >= 32 and v["waits_lgkmcnt0"] <= v["ds_rtn_atomics"], v
    assert v["v_mov"] <= 10 * v["ds_rtn_atomics"], v
    w = _one(isa, "count_smallk_kernel<false, true, true, false>")
    assert w["scratch"] == 0 and w["ds_rtn_atomics"] >= 16, w
