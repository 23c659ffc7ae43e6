#!/usr/bin/env python3
"""isa_lint - the counted-wait contract of libtsd's kernels, checked on the compiler's own gfx950 assembly.

Why.  The tile pipelines of kernels_gemm.hip and kernels_chain.hip keep several LDS-DMA tiles in flight and wait with
`s_waitcnt vmcnt(N)`, N > 0: "everything but my N youngest vector-memory instructions has completed".  That is only the
statement the source means if the N youngest instructions ARE what the source counted - and hipcc may merge, split or move
plain loads.  Round 4 shipped `vmcnt(15 + 25)` where the compiler had issued 17 loads, not 25: 14 of 1500 denoise loops
differed and every test was green.  This tool turns the accounting into a build-time check.

Input.  `make` compiles every .hip file with -save-temps=obj; build/<file>-hip-amdgcn-amd-amdhsa-gfx950.s is the assembly the
shipped object was assembled from.  Source-written waits sit between `;;#ASMSTART` / `;;#ASMEND` and carry their accounting
(lds_dma.h `wait_vm_counted`):   s_waitcnt vmcnt(N) ; tsd-wait dma=<D> other=<E> ppt=<P>,<P2>

Checks, per kernel, on the control-flow graph rebuilt from the labels and branches of the listing:
  R0  the immediate equals D + E, and D is a whole number of tiles (x*P + y*P2);
  R1  on EVERY path into a counted wait, at most D of the N youngest vector-memory instructions (those it leaves in flight) are
      LDS-DMA instructions - otherwise pieces of the tile the wait is for stay in flight (the round-4 bug: 17 plain loads
      where E said 25, so 23 DMA pieces instead of 15).  Extra loads or stores in the window only make a wait stricter
      (gfx9 counts loads and stores in one in-order counter);
  R2  between a counted wait and the source wait before it (any path) the wave issues whole tiles: the number of LDS-DMA
      instructions is x*P + y*P2 - a merged, dropped or duplicated DMA instruction breaks this;
  R3  no indirect branch in a kernel with counted waits (the graph would be incomplete); between the `tsd-wait-alt begin / end`
      markers of an if / else chain of waits (exactly one executes - hipcc lowers the chain to independent skips a graph walk
      could bypass, so the end marker counts as "a source wait has executed") there are source waits only, on every path;
  R4  the kernel's figures - vector-memory instruction counts, source / compiler-inserted vmcnt waits, registers, spills,
      scratch bytes, and per counted wait the set of window and interval compositions found - equal the committed table
      csrc/isa_contract.json.  Any drift fails until a person re-blesses the table (`--update`), which by the rules of
      DESIGN.md goes with scripts/jitter_check.sh and >= 1000 determinism loops on the GPU;
  R5  (with --lib) the vector-memory / wait instruction stream of every kernel in the shipped libtsd.so, disassembled from its
      embedded code objects, equals the listing's: the listing IS what ships.

The table carries the compiler it was blessed with (`_meta.hipcc`) and the sha256 of the blessed library's gfx950 code objects
(`_meta.code_sha256`).  R4 / R5 compare figures that belong to ONE compiler: under another hipcc (a ROCm point release, an EXTRA= flag) or
without the llvm tools they are reported as warnings, R0-R3 - the safety rules - stay fatal everywhere.

Re-blessing (`--update`) is not a formality: it REFUSES to write unless `--record FILE` (default profiles/bless_record.json) is the evidence
of a GPU run on exactly the library being blessed - `code_sha256` equal to this library's, >= 1000 identical denoise loops on the shipped
build and >= 300 on the -DTSD_JITTER build that reproduce the shipped bits (scripts/bless_evidence.sh writes it on the GPU box).

Usage:  isa_lint.py [--build-dir DIR] [--contract FILE | --no-contract] [--lib [libtsd.so]] [--update [--record FILE]] [--only FILE ...] [-q]
Exit status 0 = clean, 1 = violations (printed one per line), 2 = --update refused.
"""
import argparse
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
from collections import OrderedDict

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
DEFAULT_BUILD = os.path.join(CSRC, "build")
DEFAULT_CONTRACT = os.path.join(CSRC, "isa_contract.json")
DEFAULT_LIB = os.path.join(os.path.dirname(HERE), "lib", "libtsd.so")
DEFAULT_RECORD = os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles", "bless_record.json")
MIN_RACE_LOOPS, MIN_JITTER_LOOPS = 1000, 300
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LLVM_BIN = os.environ.get("TSD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

VMEM_RE = re.compile(r"^(buffer|tbuffer|global|flat|scratch|image)_(load|store|atomic|gather|sample)")
VMCNT_RE = re.compile(r"vmcnt\((\d+)\)")
TAG_RE = re.compile(r"tsd-wait dma=(\d+) other=(\d+) ppt=(\d+),(\d+)")
LABEL_RE = re.compile(r"^([A-Za-z_.$][\w.$]*):")
WINDOW_CAP = 63      # vmcnt is a 6-bit counter
INTERVAL_CAP = 256   # more vector-memory instructions than this between two source waits = "unbounded"


def classify(mn, text):
    """'D' LDS-DMA, 'L' load, 'S' store / atomic, None = not a vector-memory instruction."""
    if not VMEM_RE.match(mn):
        return None
    if "_store" in mn or "_atomic" in mn:
        return "S"
    toks = text.replace(",", " ").split()
    if "lds" in toks[1:] or "_lds_" in mn:
        return "D"
    return "L"


class Wait:
    __slots__ = ("n", "src", "tag", "line")

    def __init__(self, n, src, tag, line):
        self.n, self.src, self.tag, self.line = n, src, tag, line


class Mark:
    """`; tsd-wait-alt begin` / `end` (lds_dma.h): brackets an if / else chain of source waits of which exactly one executes."""
    __slots__ = ("begin", "line")

    def __init__(self, begin, line):
        self.begin, self.line = begin, line


class Block:
    __slots__ = ("label", "events", "succ", "pred", "indirect", "closed")

    def __init__(self, label):
        self.label, self.events, self.succ, self.pred, self.indirect = label, [], [], [], False
        self.closed = False   # ended in s_branch / s_endpgm: no fall-through


def parse_listing(path):
    """-> (OrderedDict function -> [Block...] with .succ / .pred as indices, meta: kernel -> dict of figures)."""
    kernels = OrderedDict()
    meta = {}
    cur = None          # blocks of the function being read
    by_label = None
    in_asm = False
    func = None
    types = set()
    with open(path) as f:
        lines = f.readlines()
    # kernel metadata (YAML at the end of the listing): one record per kernel
    rec = None
    for ln in lines:
        s = ln.strip()
        if s.startswith("- .agpr_count:") or s.startswith("- .args:"):
            rec = {}
        if rec is not None:
            m = re.match(r"-?\s*\.(agpr_count|vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|name):\s*(\S+)", s)
            if m:
                rec[m.group(1)] = m.group(2)
            if s.startswith(".wavefront_size:") or s.startswith("amdhsa.target:"):
                if "name" in rec:
                    meta[rec["name"]] = {k: int(v) for k, v in rec.items() if k != "name"}
                rec = None
    for no, ln in enumerate(lines, 1):
        s = ln.strip()
        if s.startswith(".type") and "@function" in s:
            types.add(s.split()[1].split(",")[0])
            continue
        m = LABEL_RE.match(ln)
        if m:
            lab = m.group(1)
            if lab in types and func is None:
                func = lab
                cur = [Block(lab)]
                by_label = {lab: 0}
                in_asm = False
                continue
            if func is not None:
                if lab.startswith(".Lfunc_end"):
                    _link(cur, by_label)
                    kernels[func] = cur
                    func = None
                    cur = None
                    continue
                # a new basic block: the previous one falls through unless it ended in s_branch / s_endpgm
                prev = len(cur) - 1
                cur.append(Block(lab))
                by_label[lab] = len(cur) - 1
                if not cur[prev].closed:
                    cur[prev].succ.append(len(cur) - 1)
                continue
        if func is None or not s:
            continue
        if s.startswith("; tsd-wait-alt"):
            if cur[-1].closed:
                cur.append(Block(None))
            cur[-1].events.append(Mark(s.endswith("begin"), no))
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s[0] in ";." or s.startswith("//"):
            continue
        text = s.split(";")[0].strip()
        if not text:
            continue
        mn = text.split()[0]
        b = cur[-1]
        if b.closed:
            # instructions behind an unconditional branch without a label of their own: unreachable
            cur.append(Block(None))
            b = cur[-1]
        if mn == "s_waitcnt":
            mv = VMCNT_RE.search(text)
            if mv:
                t = TAG_RE.search(s)
                b.events.append(Wait(int(mv.group(1)), in_asm, tuple(int(x) for x in t.groups()) if t else None, no))
            continue
        k = classify(mn, text)
        if k:
            b.events.append(k)
            continue
        if mn == "s_branch":
            b.succ.append(text.split()[1])
            b.closed = True
        elif mn.startswith("s_cbranch"):
            b.succ.append(text.split()[-1])
            cur.append(Block(None))   # the fall-through is a block of its own
            b.succ.append(len(cur) - 1)
        elif mn in ("s_endpgm", "s_endpgm_saved"):
            b.closed = True
        elif mn in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            b.indirect = True
            b.closed = True
    return kernels, meta


def _link(blocks, by_label):
    for b in blocks:
        succ = []
        for s in b.succ:
            if isinstance(s, str):
                if s in by_label:
                    succ.append(by_label[s])
                else:
                    b.indirect = True  # a branch out of the function
            else:
                succ.append(s)
        b.succ = sorted(set(succ))
        b.pred = []
    for i, b in enumerate(blocks):
        for s in b.succ:
            blocks[s].pred.append(i)


def _back_states(blocks, b0, i0, interval, cap):
    """Walk backwards from just before event i0 of block b0 over every path of the graph.
    interval = False: the WINDOW of a wait - stop after `cap` vector-memory instructions, at a vmcnt(0) wait (everything older has
    completed) or at the function entry; returns the set of (dma, load, store) compositions.
    interval = True: back to the previous SOURCE wait - a single one, or the end marker of a chain of alternatives (one of them has
    executed) - or the entry; only LDS-DMA instructions are counted; returns the set of (dma,) counts, and overflow = True if some
    path holds more than `cap` of them."""
    start = (b0, i0, 0, 0, 0)
    seen = {start}
    work = [start]
    out = set()
    overflow = False
    while work:
        b, i, d, l, s = work.pop()
        ev = blocks[b].events
        done = False
        while i > 0:
            e = ev[i - 1]
            i -= 1
            if isinstance(e, Mark):
                if interval and not e.begin:
                    out.add((d,)); done = True; break
                continue
            if isinstance(e, Wait):
                if interval:
                    if e.src:
                        out.add((d,)); done = True; break
                elif e.n == 0:
                    out.add((d, l, s)); done = True; break
                continue
            if e == "D": d += 1
            elif interval: continue
            elif e == "L": l += 1
            else: s += 1
            if d + l + s >= cap:
                if interval:
                    overflow = True
                    out.add((d,))
                else:
                    out.add((d, l, s))
                done = True; break
        if done:
            continue
        if not blocks[b].pred:
            out.add((d,) if interval else (d, l, s))  # function entry
            continue
        for p in blocks[b].pred:
            st = (p, len(blocks[p].events), d, l, s)
            if st not in seen:
                seen.add(st)
                work.append(st)
    return out, overflow


def _check_chain(blocks, b0, i0):
    """Forward from a `begin` marker: every path reaches an `end` marker through source waits only.  -> list of complaints."""
    bad = []
    seen = set()
    work = [(b0, i0 + 1)]
    nwaits = 0
    while work:
        b, i = work.pop()
        ev = blocks[b].events
        stop = False
        while i < len(ev):
            e = ev[i]
            i += 1
            if isinstance(e, Mark):
                if e.begin:
                    bad.append("a second begin marker (line %d) before the end of the chain" % e.line)
                stop = True
                break
            if isinstance(e, Wait):
                if e.src:
                    nwaits += 1
                else:
                    bad.append("compiler-inserted vmcnt wait inside the chain (line %d)" % e.line)
                continue
            bad.append("vector-memory instruction inside the chain")
            stop = True
            break
        if stop:
            continue
        if not blocks[b].succ:
            bad.append("a path from the begin marker leaves the kernel without passing the end marker")
            continue
        for n in blocks[b].succ:
            if n not in seen:
                seen.add(n)
                work.append((n, 0))
    if not nwaits:
        bad.append("no source wait between the markers")
    return bad


def _tiles_ok(d, p, p2):
    ps = [x for x in (p, p2) if x > 0]
    if d == 0:
        return True
    if not ps:
        return False
    if len(ps) == 1:
        return d % ps[0] == 0
    a, b = ps
    return any((d - a * x) >= 0 and (d - a * x) % b == 0 for x in range(d // a + 1))


def fmt_set(st):
    return " ".join("/".join(str(x) for x in t) for t in sorted(st)) if st else "-"


def analyse_kernel(name, blocks, meta):
    """-> (record for the contract table, [violations])."""
    viol = []
    nD = nL = nS = nsrc = ncomp = ncounted = nalt = 0
    sites = []
    indirect = any(b.indirect for b in blocks)
    for bi, b in enumerate(blocks):
        for ei, e in enumerate(b.events):
            if isinstance(e, Mark):
                if e.begin:
                    nalt += 1
                    for msg in sorted(set(_check_chain(blocks, bi, ei))):
                        viol.append("R3 %s: chain of alternative waits at line %d: %s" % (name, e.line, msg))
                continue
            if not isinstance(e, Wait):
                if e == "D": nD += 1
                elif e == "L": nL += 1
                else: nS += 1
                continue
            if not e.src:
                ncomp += 1
                continue
            nsrc += 1
            if e.n == 0:
                continue
            ncounted += 1
            where = "%s: line %d: s_waitcnt vmcnt(%d)" % (name, e.line, e.n)
            if e.tag is None:
                viol.append("R0 %s: a source-written counted wait without a tsd-wait tag (use wait_vm_counted)" % where)
                continue
            D, E, P, P2 = e.tag
            if D + E != e.n:
                viol.append("R0 %s: tag says dma=%d other=%d" % (where, D, E))
            if not _tiles_ok(D, P, P2):
                viol.append("R0 %s: dma=%d is not a whole number of tiles of %d / %d pieces" % (where, D, P, P2))
            win, _ = _back_states(blocks, bi, ei, False, e.n)
            for (d, l, s) in sorted(win):
                if d > D:
                    viol.append("R1 %s: the source counts %d LDS-DMA pieces + %d other loads; on some path the %d youngest vector-memory "
                                "instructions are dma/load/store = %d/%d/%d: %d LDS-DMA instructions stay in flight where %d were intended"
                                % (where, D, E, e.n, d, l, s, d, D))
            itv = set()
            if D > 0:
                itv, over = _back_states(blocks, bi, ei, True, INTERVAL_CAP)
                if over:
                    viol.append("R2 %s: no source wait within %d LDS-DMA instructions on some path into it" % (where, INTERVAL_CAP))
                for (d,) in sorted(itv):
                    if not _tiles_ok(d, P, P2):
                        viol.append("R2 %s: %d LDS-DMA instructions since the previous source wait on some path - not whole tiles of "
                                    "%d / %d pieces" % (where, d, P, P2))
            sites.append("vmcnt(%d) dma=%d other=%d ppt=%d,%d | window dma/load/store %s | dma since the previous source wait %s" % (e.n, D, E, P, P2, fmt_set(win), fmt_set(itv)))
    if indirect and ncounted:
        viol.append("R3 %s: indirect branch / call in a kernel with counted waits" % name)
    rec = OrderedDict()
    rec["vmem"] = "dma=%d load=%d store=%d" % (nD, nL, nS)
    rec["waits"] = "source=%d counted=%d chains=%d compiler_vmcnt=%d" % (nsrc, ncounted, nalt, ncomp)
    m = meta.get(name)
    if m:
        rec["regs"] = "vgpr=%d agpr=%d sgpr_spill=%d vgpr_spill=%d scratch=%d" % (
            m.get("vgpr_count", -1), m.get("agpr_count", -1), m.get("sgpr_spill_count", -1), m.get("vgpr_spill_count", -1),
            m.get("private_segment_fixed_size", -1))
    if sites:
        # consecutive identical sites are folded: "<site> x<count>"
        folded = []
        for st in sites:
            if folded and folded[-1][0] == st:
                folded[-1][1] += 1
            else:
                folded.append([st, 1])
        rec["sites"] = [st if n == 1 else "%s  x%d" % (st, n) for st, n in folded]
    return rec, viol


def stream_of_listing(blocks):
    out = []
    for b in blocks:
        for e in b.events:
            if not isinstance(e, Mark):
                out.append("W%d" % e.n if isinstance(e, Wait) else e)
    return out


# ---- the shipped library: embedded code objects -> per-kernel vector-memory / wait stream -------------------------------------
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path, workdir):
    fb = os.path.join(workdir, "fatbin.bin")   # (llvm-objcopy wants a seekable output)
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fb], check=True)
    with open(fb, "rb") as f:
        raw = f.read()
    outs = []
    pos = 0
    while True:
        at = raw.find(MAGIC, pos)
        if at < 0:
            break
        (n,) = struct.unpack_from("<Q", raw, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", raw, p)
            triple = raw[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                path = os.path.join(workdir, "co_%d.elf" % len(outs))
                with open(path, "wb") as f:
                    f.write(raw[at + off:at + off + size])
                outs.append(path)
        pos = at + len(MAGIC)
    return outs


def streams_of_library(lib_path, workdir):
    streams = {}
    for co in code_objects(lib_path, workdir):
        txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE,
                             universal_newlines=True).stdout
        cur = None
        for ln in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
            if m:
                cur = streams.setdefault(m.group(1), [])
                continue
            if cur is None:
                continue
            s = ln.strip()
            if not s:
                continue
            text = s.split("//")[0].strip()
            if not text:
                continue
            mn = text.split()[0]
            if mn == "s_waitcnt":
                mv = VMCNT_RE.search(text)
                if mv:
                    cur.append("W%s" % mv.group(1))
                continue
            k = classify(mn, text)
            if k:
                cur.append(k)
    return streams


def llvm_tools_present():
    return all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("llvm-objcopy", "llvm-objdump"))


def code_sha256(lib_path):
    """sha256 over the gfx950 code objects embedded in the library, in bundle order: what the GPU executes (host code does not count)."""
    import tempfile
    h = hashlib.sha256()
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib_path, td):
            with open(co, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def hipcc_version():
    """One line that names the compiler (HIP version + clang version); '' when hipcc cannot be run."""
    try:
        txt = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return ""
    keep = [ln.strip() for ln in txt.splitlines() if ln.startswith("HIP version") or "clang version" in ln]
    return " | ".join(keep)


def check_record(record, lib):
    """-> list of reasons why `record` does not license blessing `lib` (empty = it does)."""
    if not record or not os.path.exists(record):
        return ["no evidence record %s (scripts/bless_evidence.sh writes it on the GPU box)" % record]
    try:
        rec = json.load(open(record))
    except ValueError as e:
        return ["evidence record %s is not JSON: %s" % (record, e)]
    why = []
    have = code_sha256(lib)
    if rec.get("code_sha256") != have:
        why.append("the record is for another build (code objects %s..., this library %s...)" % (str(rec.get("code_sha256"))[:12], have[:12]))
    if int(rec.get("race_loops", 0)) < MIN_RACE_LOOPS or int(rec.get("race_distinct", 0)) != 1:
        why.append("needs >= %d identical denoise loops on the shipped build with ONE result (record: %s loops, %s distinct)"
                   % (MIN_RACE_LOOPS, rec.get("race_loops"), rec.get("race_distinct")))
    if int(rec.get("jitter_loops", 0)) < MIN_JITTER_LOOPS or not rec.get("jitter_matches_shipped"):
        why.append("needs >= %d loops of the -DTSD_JITTER build reproducing the shipped bits (record: %s loops, matches: %s)"
                   % (MIN_JITTER_LOOPS, rec.get("jitter_loops"), rec.get("jitter_matches_shipped")))
    return why


def run(build_dir=DEFAULT_BUILD, contract=DEFAULT_CONTRACT, lib=None, update=False, only=None, quiet=False, out=sys.stdout, record=None,
        warnings=None):
    """Returns the list of violations (empty = clean).  R4 / R5 findings under a compiler other than the blessed one (or without the llvm
    tools) go to `warnings` (a list, optional) instead: they compare one compiler's figures."""
    warn = warnings if warnings is not None else []
    files = sorted(f for f in os.listdir(build_dir) if f.endswith("-hip-amdgcn-amd-amdhsa-gfx950.s")) if os.path.isdir(build_dir) else []
    if only:
        files = [f for f in files if f in only or f.split("-hip-")[0] in only]
    viol = []
    if not files:
        return ["no compiler listings (*-hip-amdgcn-amd-amdhsa-gfx950.s) in %s: run `make -C csrc` first" % build_dir]
    table = OrderedDict()
    listing_streams = {}
    for fn in files:
        kernels, meta = parse_listing(os.path.join(build_dir, fn))
        ft = OrderedDict()
        for name, blocks in kernels.items():
            rec, v = analyse_kernel(name, blocks, meta)
            viol += ["%s: %s" % (fn.split("-hip-")[0], x) for x in v]
            ft[name] = rec
            listing_streams[name] = stream_of_listing(blocks)
        table[fn.split("-hip-")[0]] = ft
    r45 = []   # R4 / R5 findings: fatal only under the blessed compiler
    if lib and not llvm_tools_present():
        warn.append("R5 skipped: llvm-objcopy / llvm-objdump not in %s (TSD_LLVM_BIN)" % LLVM_BIN)
        lib_checked = None
    else:
        lib_checked = lib
    if lib_checked:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            shipped = streams_of_library(lib, td)
        if not shipped:
            r45.append("R5 no gfx950 code object found in %s" % lib)
        for name, st in listing_streams.items():
            if name not in shipped:
                # device functions that were inlined everywhere have a listing body but no symbol in the object: kernels must be there
                if any(name in t and "regs" in t[name] for t in table.values()):
                    r45.append("R5 %s: kernel of the listing is not in %s" % (name, os.path.basename(lib)))
                continue
            if shipped[name] != st:
                r45.append("R5 %s: the shipped object's vector-memory / wait stream (%d instructions) differs from the listing's (%d): "
                            "the library was not built from these listings" % (name, len(shipped[name]), len(st)))
    blessed_hipcc = None
    if update:
        if viol:
            print("isa_lint: --update refused: the build violates R0-R3", file=out)
            for v in viol:
                print("  " + v, file=out)
            return viol + ["update refused"]
        the_lib = lib or DEFAULT_LIB
        why = check_record(record or DEFAULT_RECORD, the_lib) if llvm_tools_present() else ["llvm tools missing: cannot hash the library's code objects"]
        if why:
            print("isa_lint: --update REFUSED - a re-bless goes with the GPU evidence for exactly this library:", file=out)
            for w in why:
                print("  " + w, file=out)
            return ["update refused: " + w for w in why]
        if only:
            old = json.load(open(contract), object_pairs_hook=OrderedDict) if os.path.exists(contract) else OrderedDict()
            old.update(table)
            table = old
        table.pop("_meta", None)
        full = OrderedDict([("_meta", OrderedDict([("hipcc", hipcc_version()), ("code_sha256", code_sha256(the_lib)),
                                                    ("record", os.path.relpath(record or DEFAULT_RECORD, os.path.dirname(os.path.dirname(HERE))))]))])
        full.update(table)
        with open(contract, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        if not quiet:
            print("isa_lint: wrote %s (%d kernels)" % (contract, sum(len(t) for t in table.values())), file=out)
    elif contract is not None:
        if not os.path.exists(contract):
            viol.append("R4 no committed table %s (isa_lint.py --update writes it)" % contract)
        else:
            want = json.load(open(contract), object_pairs_hook=OrderedDict)
            blessed_hipcc = (want.pop("_meta", None) or {}).get("hipcc")
            for fkey, ft in table.items():
                wt = want.get(fkey)
                if wt is None:
                    r45.append("R4 %s: file not in the committed table" % fkey)
                    continue
                for name, rec in ft.items():
                    if name not in wt:
                        r45.append("R4 %s: %s: kernel not in the committed table" % (fkey, name))
                        continue
                    for key in set(rec) | set(wt[name]):
                        if rec.get(key) != wt[name].get(key):
                            r45.append("R4 %s: %s: %s drifted from the committed table\n      now : %s\n      was : %s"
                                        % (fkey, name, key, json.dumps(rec.get(key)), json.dumps(wt[name].get(key))))
                for name in wt:
                    if name not in ft:
                        r45.append("R4 %s: %s: kernel of the committed table is gone" % (fkey, name))
    # R4 / R5 are statements about ONE compiler's output: under another hipcc they inform, they do not fail the build
    if r45:
        now = hipcc_version()
        if blessed_hipcc and now and now != blessed_hipcc:
            warn.append("compiler differs from the blessed one (%s; table: %s): %d R4 / R5 finding(s) reported as warnings" % (now, blessed_hipcc, len(r45)))
            warn += r45
        else:
            viol += r45
    if not quiet:
        for w in warn:
            print("  warning: " + w, file=out)
    if not quiet:
        nk = sum(len(t) for t in table.values())
        ns = sum(int(re.search(r"counted=(\d+)", r["waits"]).group(1)) for t in table.values() for r in t.values())
        print("isa_lint: %d listings, %d functions, %d counted waits checked on every path%s: %s"
              % (len(files), nk, ns, ", shipped library cross-checked" if lib else "", "%d violation(s)" % len(viol) if viol else "clean"), file=out)
        for v in viol:
            print("  " + v, file=out)
    return viol


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--build-dir", default=DEFAULT_BUILD)
    ap.add_argument("--contract", default=DEFAULT_CONTRACT)
    ap.add_argument("--no-contract", action="store_true", help="rules R0-R3 only (a build that is not the shipped one)")
    ap.add_argument("--lib", nargs="?", const=DEFAULT_LIB, default=None)
    ap.add_argument("--update", action="store_true")
    ap.add_argument("--record", default=None, help="evidence record of the GPU run that licenses --update (default profiles/bless_record.json)")
    ap.add_argument("--only", nargs="*")
    ap.add_argument("-q", "--quiet", action="store_true")
    a = ap.parse_args()
    v = run(a.build_dir, None if a.no_contract else a.contract, a.lib, a.update, a.only, a.quiet, record=a.record)
    sys.exit((2 if any(x.startswith("update refused") for x in v) else 1) if v else 0)


if __name__ == "__main__":
    main()
