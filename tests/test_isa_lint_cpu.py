"""The counted-wait contract (tools/isa_lint.py), run on the CPU: no GPU needed, hipcc cross-compiles.

* the shipped build: `make` (a no-op when current), then the lint over build/*.s against csrc/isa_contract.json and the embedded code
  objects of lib/libtsd.so - what __graft_entry__.build() asserts as well;
* the lint must FAIL on the bug it exists for: kernels_chain.hip compiled with -DTSD_CHAIN_STRICT_WAIT=0 re-introduces round 4's
  vmcnt(15 + 25) - the source counts 25 plain loads, hipcc issues 17 - and rule R1 has to name those waits;
* hand-written listings for each rule, so the analysis itself is pinned (loops, alternative chains, merged loads, partial tiles).
"""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-diffusion.mojo_amd")
CSRC = os.path.join(PKG, "csrc")
sys.path.insert(0, os.path.join(PKG, "tools"))
import isa_lint  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
needs_hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc on this host")


@needs_hipcc
def test_shipped_build_keeps_the_counted_wait_contract():
    subprocess.check_call(["make", "-C", CSRC, "-j8", "ARCH=gfx950"], stdout=subprocess.DEVNULL)
    viol = isa_lint.run(lib=isa_lint.DEFAULT_LIB, quiet=True)
    assert not viol, "\n".join(viol[:20])


@needs_hipcc
def test_contract_table_covers_every_counted_wait_of_the_tile_pipelines():
    import json
    table = json.load(open(isa_lint.DEFAULT_CONTRACT))
    meta = table.pop("_meta")
    assert len(meta["code_sha256"]) == 64 and "clang version" in meta["hipcc"] and os.path.exists(os.path.join(ROOT, meta["record"])), meta
    counted = {k: sum(int(r["waits"].split("counted=")[1].split()[0]) for r in t.values()) for k, t in table.items()}
    assert counted["kernels_gemm"] >= 200 and counted["kernels_chain"] >= 100, counted
    assert counted["kernels_attn8"] >= 4, counted   # the 8-wave flash attention: two alternatives per pass (waves with one / two DMA pieces per tile)
    assert counted["kernels_attn"] == 0 and counted["kernels_norm"] == 0 and counted["kernels_elementwise"] == 0, counted
    # no MFMA kernel may touch scratch (3848 scratch accesses were lost once without anybody noticing)
    spills = {n: r["regs"] for t in table.values() for n, r in t.items() if "regs" in r and "scratch=0" not in r["regs"]}
    # known: the 32-query d = 40 attention variant (TSD_ATTN_QB=1, not the default) spills one register.  (Until round 5 the N = 128
    # fused-skip conv tiles kept a 48-byte address table in scratch: two loops whose tails hipcc merged, now one loop.)
    known = {"_Z17flash_attn_kernelILi40ELi1EEv5AttnK"}
    assert set(spills) <= known, spills


@needs_hipcc
def test_lint_fails_on_the_round4_wait(tmp_path):
    """-DTSD_CHAIN_STRICT_WAIT=0: the first waits of out_proj / conv_out count 25 bias + residual loads of which hipcc issues 17."""
    out = tmp_path / "build"
    out.mkdir()
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DTSD_CHAIN_STRICT_WAIT=0", "-save-temps=obj",
                           "-c", os.path.join(CSRC, "kernels_chain.hip"), "-o", str(out / "kernels_chain.o")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    viol = isa_lint.run(build_dir=str(out), contract=None, quiet=True)
    r1 = [v for v in viol if " R1 " in v and "attn_chain_kernelILi0E" in v]
    assert r1, viol[:5]
    assert any("other loads" in v and "vmcnt(40)" in v for v in r1), r1[:3]
    # and the shipped flavour of the same file is clean under the same rules
    out2 = tmp_path / "build2"
    out2.mkdir()
    subprocess.check_call(["make", "-C", CSRC, "-j8", "ARCH=gfx950"], stdout=subprocess.DEVNULL)  # (a no-op when current; run in isolation it builds)
    shutil.copy(os.path.join(CSRC, "build", "kernels_chain-hip-amdgcn-amd-amdhsa-gfx950.s"), str(out2))
    assert isa_lint.run(build_dir=str(out2), contract=None, quiet=True) == []


# ---- re-blessing the table needs the GPU evidence for exactly this library ------------------------------------------------------
def _record(tmp_path, **over):
    import json
    rec = {"code_sha256": isa_lint.code_sha256(isa_lint.DEFAULT_LIB), "race_loops": 1000, "race_distinct": 1, "jitter_loops": 300,
           "jitter_matches_shipped": True}
    rec.update(over)
    p = tmp_path / "record.json"
    p.write_text(json.dumps(rec))
    return str(p)


@needs_hipcc
def test_rebless_is_refused_without_the_evidence_record(tmp_path):
    """`isa_lint.py --update` used to rewrite 1089 lines of isa_contract.json on request; the rule that a re-bless goes with the jitter run and
    >= 1000 determinism loops was prose.  Now: no record, a record of another build, too few loops or more than one result -> refused, the
    table is not touched; the command line exits with status 2."""
    subprocess.check_call(["make", "-C", CSRC, "-j8", "ARCH=gfx950"], stdout=subprocess.DEVNULL)
    table = tmp_path / "contract.json"
    for rec in (str(tmp_path / "missing.json"), _record(tmp_path, code_sha256="0" * 64), _record(tmp_path, race_loops=200),
                _record(tmp_path, race_distinct=2), _record(tmp_path, jitter_loops=100), _record(tmp_path, jitter_matches_shipped=False)):
        v = isa_lint.run(lib=isa_lint.DEFAULT_LIB, contract=str(table), update=True, record=rec, quiet=True, out=open(os.devnull, "w"))
        assert v and all(x.startswith("update refused") for x in v), v
        assert not table.exists()
    r = subprocess.run([sys.executable, os.path.join(PKG, "tools", "isa_lint.py"), "--lib", "--contract", str(table), "--update", "--record",
                        str(tmp_path / "missing.json")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 2 and "REFUSED" in r.stdout, (r.returncode, r.stdout[-500:])
    assert not table.exists()


@needs_hipcc
def test_rebless_with_the_record_names_compiler_and_code_objects_and_other_compilers_only_warn(tmp_path):
    """With the record of this very library the table is written and carries `_meta` (hipcc version, sha256 of the code objects, the record).
    A drifted figure then fails R4 under the same compiler - and is a WARNING under another one (a ROCm point release must not turn a
    correct library into a failed build; R0-R3 stay fatal)."""
    import json
    subprocess.check_call(["make", "-C", CSRC, "-j8", "ARCH=gfx950"], stdout=subprocess.DEVNULL)
    table = tmp_path / "contract.json"
    assert isa_lint.run(lib=isa_lint.DEFAULT_LIB, contract=str(table), update=True, record=_record(tmp_path), quiet=True) == []
    t = json.load(open(table))
    assert t["_meta"]["code_sha256"] == isa_lint.code_sha256(isa_lint.DEFAULT_LIB) and "clang version" in t["_meta"]["hipcc"], t["_meta"]
    assert isa_lint.run(lib=isa_lint.DEFAULT_LIB, contract=str(table), quiet=True) == []
    name = next(n for n, r in t["kernels_gemm"].items() if "regs" in r)
    t["kernels_gemm"][name]["regs"] = "vgpr=1 agpr=0 sgpr_spill=0 vgpr_spill=0 scratch=0"
    json.dump(t, open(table, "w"))
    v = isa_lint.run(lib=isa_lint.DEFAULT_LIB, contract=str(table), quiet=True)
    assert len(v) == 1 and "R4" in v[0] and "regs drifted" in v[0], v
    t["_meta"]["hipcc"] = "HIP version: 9.9.9 | some other clang version 99"
    json.dump(t, open(table, "w"))
    w = []
    assert isa_lint.run(lib=isa_lint.DEFAULT_LIB, contract=str(table), quiet=True, warnings=w) == []
    assert any("compiler differs" in x for x in w) and any("regs drifted" in x for x in w), w


# ---- hand-written listings -----------------------------------------------------------------------------------------------------
def _lint_text(tmp_path, body):
    d = tmp_path / "b"
    d.mkdir(exist_ok=True)
    txt = "\t.type\tk,@function\nk:\n" + textwrap.dedent(body) + "\ts_endpgm\n.Lfunc_end0:\n"
    (d / "x-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(txt)
    return isa_lint.run(build_dir=str(d), contract=None, quiet=True)


DMA = "\tbuffer_load_dwordx4 v1, s[8:11], s2 offen lds\n"
LOAD = "\tglobal_load_dwordx4 v[2:5], v[8:9], off\n"
STORE = "\tglobal_store_dwordx4 v[8:9], v[2:5], off\n"


def W(n, dma, other=0, ppt=0, ppt2=0):
    return ";;#ASMSTART\n\ts_waitcnt vmcnt(%d) ; tsd-wait dma=%d other=%d ppt=%d,%d\n;;#ASMEND\n" % (n, dma, other, ppt, ppt2)


def test_ring_loop_with_whole_tiles_is_clean(tmp_path):
    body = DMA * 6 + ".LBB0_1:\n" + W(3, 3, 0, 3) + "\ts_barrier\n" + DMA * 3 + "\ts_cbranch_scc1 .LBB0_1\n" + W(0, 0)
    assert _lint_text(tmp_path, body) == []


def test_dropped_dma_piece_breaks_the_tile_rule(tmp_path):
    body = DMA * 6 + ".LBB0_1:\n" + W(3, 3, 0, 3) + "\ts_barrier\n" + DMA * 2 + "\ts_cbranch_scc1 .LBB0_1\n" + W(0, 0)
    v = _lint_text(tmp_path, body)
    assert v and all(" R2 " in x for x in v), v


def test_merged_loads_behind_the_dma_pieces_are_caught(tmp_path):
    ok = DMA * 4 + LOAD * 5 + W(7, 2, 5, 2)
    assert _lint_text(tmp_path, ok) == []
    merged = DMA * 4 + LOAD * 3 + W(7, 2, 5, 2)   # the compiler merged 5 loads into 3: four DMA pieces stay in flight, two were meant
    v = _lint_text(tmp_path, merged)
    assert len(v) == 1 and " R1 " in v[0] and "4 LDS-DMA instructions stay in flight where 2 were intended" in v[0], v


def test_extra_loads_or_stores_only_make_a_wait_stricter(tmp_path):
    body = DMA * 4 + LOAD * 2 + STORE + W(2, 2, 0, 2)
    assert _lint_text(tmp_path, body) == []


def test_every_path_into_a_wait_is_checked(tmp_path):
    # one arm of a branch issues the counted loads, the other does not
    body = DMA * 2 + "\ts_cbranch_scc1 .LBB0_2\n" + LOAD * 2 + ".LBB0_2:\n" + W(2, 0, 2)
    v = _lint_text(tmp_path, body)
    assert len(v) == 1 and " R1 " in v[0], v


def test_untagged_and_inconsistent_waits(tmp_path):
    v = _lint_text(tmp_path, DMA * 2 + ";;#ASMSTART\n\ts_waitcnt vmcnt(1)\n;;#ASMEND\n")
    assert len(v) == 1 and " R0 " in v[0] and "without a tsd-wait tag" in v[0], v
    v = _lint_text(tmp_path, DMA * 4 + W(3, 2, 0, 2))
    assert any("tag says" in x for x in v), v
    v = _lint_text(tmp_path, DMA * 4 + W(3, 3, 0, 2))
    assert any("whole number of tiles" in x for x in v), v
    # compiler-inserted waits (outside ASMSTART / ASMEND) are the compiler's business
    assert _lint_text(tmp_path, LOAD * 3 + "\ts_waitcnt vmcnt(1)\n") == []


def test_chain_of_alternatives_counts_as_one_wait(tmp_path):
    begin = ";;#ASMSTART\n\t; tsd-wait-alt begin\n;;#ASMEND\n"
    end = ";;#ASMSTART\n\t; tsd-wait-alt end\n;;#ASMEND\n"
    # hipcc's structurised layout: two independent skips around two waits - a graph walk finds a way around both
    chain = "\ts_cbranch_vccz .LBB0_3\n" + W(0, 0) + ".LBB0_3:\n\ts_cbranch_vccnz .LBB0_4\n" + W(2, 2, 0, 2) + ".LBB0_4:\n"
    loop = DMA * 4 + ".LBB0_1:\n%s" + chain + "%s\ts_barrier\n" + DMA * 2 + "\ts_cbranch_scc1 .LBB0_1\n" + W(0, 0)
    assert _lint_text(tmp_path, loop % (begin, end)) == []
    v = _lint_text(tmp_path, loop % ("", ""))
    assert v and all(" R2 " in x for x in v), v   # without the markers: "no source wait within ... on some path"
    # a vector-memory instruction between the markers is not a chain of waits
    v = _lint_text(tmp_path, DMA * 2 + begin + LOAD + W(2, 2, 0, 2) + end)
    assert any(" R3 " in x for x in v), v
