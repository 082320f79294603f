"""The generated code of the kernels that retire asm register loads with hand-counted waits is audited at build time (CPU test):
no instruction may touch the destination of an asm load between the load and the wait that names it (ADVICE r04, medium;
cdna_hip_programming.md section 5.7 item 1).  The checker itself is tested on hand-written listings first."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
import isa_audit as A  # noqa: E402

CSRC = os.path.join(ROOT, "mastermetastyletransfer_amd", "csrc")


def _audit(body: str):
    text = "\t.text\nkern:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"
    funcs = A.parse_functions(text)
    assert list(funcs) == ["kern"]
    return A.audit_function(funcs["kern"])


CLEAN = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v[2:3], off
\t;;#ASMEND
\tv_add_u32_e32 v4, v5, v6
\t;;#ASMSTART
\ts_waitcnt vmcnt(3) ; landed v[10:13]
\t;;#ASMEND
\tv_mov_b32_e32 v20, v10
"""


def test_checker_accepts_a_clean_stream():
    v, loads, named = _audit(CLEAN)
    assert v == [] and loads == 1 and named == 1


@pytest.mark.parametrize("touch", ["v_mov_b32_e32 v20, v11", "v_accvgpr_write_b32 a5, v13", "scratch_store_dword off, v10, s0", "v_mov_b32_e32 v12, 0"])
def test_checker_flags_a_touch_between_load_and_wait(touch):
    body = CLEAN.replace("\tv_add_u32_e32 v4, v5, v6\n", "\t" + touch + "\n")
    v, _, _ = _audit(body)
    assert len(v) == 1 and "in flight" in v[0]


def test_checker_follows_the_loop_back_edge():
    # the load at the bottom of the loop is retired at the top of the next iteration; the copy at the top (before the wait) is a violation
    # only because of the back edge
    body = """
.LBB0_1:
\tv_mov_b32_e32 v30, v10
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; landed v[10:13]
\t;;#ASMEND
\tv_mov_b32_e32 v31, v10
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v[2:3], off
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt vmcnt(0)
"""
    v, _, _ = _audit(body)
    assert len(v) == 1 and "v_mov_b32_e32 v30, v10" in v[0]


def test_checker_counts_a_counted_wait_without_names_as_nothing_and_a_zero_wait_as_everything():
    body = CLEAN.replace("s_waitcnt vmcnt(3) ; landed v[10:13]", "s_waitcnt vmcnt(3)")
    v, _, _ = _audit(body)
    assert len(v) == 2                     # the consumer, and the load that is never retired
    body = CLEAN.replace("s_waitcnt vmcnt(3) ; landed v[10:13]", "s_waitcnt vmcnt(0)")
    assert _audit(body)[0] == []
    body = CLEAN.replace("global_load_dwordx4 v[10:13], v[2:3], off", "ds_read_b128 v[10:13], v2").replace("s_waitcnt vmcnt(3) ; landed v[10:13]", "s_waitcnt vmcnt(0)")
    assert len(_audit(body)[0]) == 2       # an LDS read is not retired by a vector-memory wait


@pytest.mark.parametrize("src", ["gemm_x3.hip", "mlp_x3.hip"])
def test_generated_code_never_touches_a_register_in_flight(src):
    """every instantiation in the file, built with build.py's flags for it (bf16 and IEEE-half halves, every tile form incl. the
    MASTER_X3_TILE variants and the 8-wave C = 128 MLP, the data-gradient epilogue)"""
    res, loads, named = A.audit_source(os.path.join(CSRC, src))
    assert loads > 0 and named > 0, "the audit saw no asm loads / naming waits: the markers have changed"
    assert res == {}, "\n".join(f"{k}: {v[:5]}" for k, v in res.items())


# ---- the queue-counting analysis: kernels whose counted waits do not name registers (the bf16-mode kernels) ------------------------------
def _count(body: str):
    text = "\t.text\nkern:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"
    return A.audit_function_counting(A.parse_functions(text)["kern"])


READS8 = "".join(f"\t;;#ASMSTART\n\tds_read_b128 v[{10 + 4 * i}:{13 + 4 * i}], v2 offset:{1024 * i}\n\t;;#ASMEND\n" for i in range(8))


def test_counting_checker_retires_all_but_the_youngest():
    # eight reads, lgkmcnt(4): the first four have landed (in-order queue), the last four have not
    ok = READS8 + "\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32_e32 v60, v10\n\tv_mov_b32_e32 v61, v25\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v62, v41\n"
    v, loads = _count(ok)
    assert v == [] and loads == 8
    bad = READS8 + "\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32_e32 v60, v26\n\ts_waitcnt lgkmcnt(0)\n"
    v, _ = _count(bad)
    assert len(v) == 1 and "v26" in v[0]
    # a compiler-issued LDS store behind the reads takes one of the "youngest" slots: lgkmcnt(4) now leaves only three asm reads in flight
    mixed = READS8 + "\tds_write_b32 v3, v4\n\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32_e32 v60, v26\n\ts_waitcnt lgkmcnt(0)\n"
    assert _count(mixed)[0] == []
    # the two queues are separate: a vector-memory wait retires no LDS read
    assert len(_count(READS8 + "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v60, v10\n\ts_waitcnt lgkmcnt(0)\n")[0]) == 1


def test_counting_checker_reports_what_it_cannot_analyse():
    # a branch around a memory operation while asm reads are in flight: the two paths reach the join with different queues
    body = READS8 + "\ts_cbranch_scc1 .LBB0_2\n\tds_write_b32 v3, v4\n.LBB0_2:\n\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32_e32 v60, v10\n\ts_waitcnt lgkmcnt(0)\n"
    v, _ = _count(body)
    assert any("not analysable" in x for x in v)
    # a scalar load (may complete out of order) next to asm reads in flight
    v, _ = _count(READS8 + "\ts_load_dword s4, s[0:1], 0x0\n\ts_waitcnt lgkmcnt(0)\n")
    assert any("scalar memory" in x for x in v)


@pytest.mark.parametrize("src", ["mlp_bf16.hip", "mlp_c128_bf16.hip", "linear_ts_bf16.hip", "conv_ring_bf16.hip", "attn_block_bf16.hip",
                                 "window_attn_proj_bf16.hip", "mlp_2w_bf16.hip", "mlp_8w_bf16.hip", "mlp_x3.hip"])
def test_counted_waits_of_the_bf16_kernels_retire_every_asm_read_before_its_use(src):
    """the bf16-mode kernels pipeline their LDS fragment reads through inline asm with counted lgkmcnt waits that do not name registers:
    the queues are simulated over each kernel's flow graph (every instantiation, build.py's flags); mlp_x3.hip passes both analyses"""
    res, loads = A.audit_source_counting(os.path.join(CSRC, src))
    assert res == {}, "\n".join(f"{k}: {v[:5]}" for k, v in res.items())
    assert loads > 0 or src in ("mlp_2w_bf16.hip", "mlp_8w_bf16.hip")          # (the two-workgroup MLP form leaves its reads to the compiler: nothing to retire by hand)
