"""the counted `s_waitcnt vmcnt(NST)` at the top of a prefetch loop (csrc/wr_quad.h, WR_DMA_PREFETCH) is right only
while at least NST vector-memory instructions are issued between the LDS-DMA pair and that wait -- gfx950 retires loads, stores and
LDS-DMA in order on ONE counter, so "at most NST outstanding" means "the DMA has landed" only if NST younger operations exist.  With
fewer (a store merged or dropped by the compiler, an edit of store_bins_lines / store_hbits) the wave would read the previous
symbol's samples and nothing would fault.  And a vector-memory operation the compiler adds on its own INSIDE such a loop -- a
register spill's reload -- brings a `s_waitcnt vmcnt(0)` with it that waits for every store of the symbol before: the prefetch's
gain is gone without a test failing (round 5's WR_IDX8 experiment did exactly that: +9 %, profiles/r05_ab_idx8_again_not_kept.txt).

This test disassembles the BUILT library (the code object in its .hip_fatbin section; 2 s) and checks, for every prefetch loop of
the kernels with the usual output set (XK = false; the XK instances pick the wait's immediate at run time):
  * the loop holds exactly one vmcnt wait, the counted one, in front of the DMA pair;
  * the vector-memory instructions behind the pair up to the loop's end are at least its immediate;
  * no scratch (spill) access inside the loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd", "wifirx", "libwifirx.so")
LLVM = "/opt/rocm/lib/llvm/bin"

INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$")
VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")


def _disassemble(tmp):
    tools = [shutil.which("objcopy"), os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-objdump")]
    if not os.path.exists(LIB) or any(t is None or not os.path.exists(t) for t in tools):
        pytest.skip("libwifirx.so or the binutils / ROCm LLVM tools are not here")
    # (on a COPY, with an explicit output file: objcopy without one rewrites its INPUT in place -- the library some earlier test of
    #  this process has mapped; the suite then died later with a segmentation fault in whatever native code ran next)
    fat, lib_copy = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "libwifirx_copy.so")
    shutil.copyfile(LIB, lib_copy)
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib_copy, os.path.join(tmp, "objcopy_out.so")],
                          stderr=subprocess.DEVNULL)
    # one bundle per translation unit, back to back
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    text = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        part, co = os.path.join(tmp, "fat%d.bin" % n), os.path.join(tmp, "dev%d.co" % n)
        with open(part, "wb") as f:
            f.write(data[a:b])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                               "--output=" + co])
        text.append(subprocess.run([tools[2], "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def _functions(text):
    """{mangled name: [(address, mnemonic, operands, comment tail)]}"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))      # (address, mnemonic, operands, rest of the comment: a branch's target)
    return out


def _branch_target(ops):
    m = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", ops)
    return int(m.group(1), 16) if m else None


def _prefetch_loops(fns):
    """[(kernel name, EQ, HB, address of the DMA pair, counted wait's immediate, younger vector-memory instructions, other vmcnt
    waits in the loop, scratch accesses in the loop)] of the kernels with the usual output set (XK = false)"""
    out = []
    for name, ins in fns.items():
        m = re.search(r"demod_(batch|stream)_kernelILi(\d)ELb([01])ELb0E", name)       # <EQ, HB, XK = false>
        if not m:
            continue
        base = ins[0][0]
        addr = [i[0] - base for i in ins]
        dma = [k for k, i in enumerate(ins) if i[1] == "global_load_lds_dwordx4"]
        assert len(dma) % 2 == 0 and dma, name
        n_loops = 0
        for d0, d1 in zip(dma[0::2], dma[1::2]):
            assert d1 - d0 <= 4, (name, "a DMA pair is two instructions a few slots apart")
            # the innermost loop around the pair: the first backward branch behind it whose target lies in front of it
            end = head = None
            for k in range(d1 + 1, len(ins)):
                mn = ins[k][1]
                if mn.startswith("s_cbranch") or mn == "s_branch":
                    t = _branch_target(ins[k][3])
                    if t is not None and t <= addr[d0]:
                        end, head = k, addr.index(t) if t in addr else None
                        # (the loop may close with two branches to its head: one that skips a last store under an empty execution
                        #  mask, then the store and the unconditional one)
                        k2 = k + 1
                        while k2 < min(end + 16, len(ins)):
                            if ins[k2][1].startswith("s_cbranch") or ins[k2][1] == "s_branch":
                                t2 = _branch_target(ins[k2][3])
                                if t2 is not None and abs(t2 - t) <= 64 and t2 in addr:       # (a latch block of a few scalar moves in front of the head)
                                    end, head = k2, min(head, addr.index(t2))
                            k2 += 1
                        break
            if end is None or head is None:
                continue                                  # pf_setup's request: straight-line code, followed by a full wait
            body = range(head, end + 1)
            waits = [(k, ins[k][2]) for k in body if ins[k][1] == "s_waitcnt" and "vmcnt" in ins[k][2]]
            counted = [(k, int(re.fullmatch(r"vmcnt\((\d+)\)", w).group(1))) for k, w in waits
                       if k < d0 and re.fullmatch(r"vmcnt\([1-9]\d*\)", w)]
            assert len(counted) == 1, (name, hex(addr[d0]), "the counted wait in front of the DMA pair", [w for _, w in waits])
            kw, nst = counted[0]
            younger = [k for k in list(range(d1 + 1, end + 1)) + list(range(head, kw)) if VMEM.match(ins[k][1])]
            others = [w for k, w in waits if k != kw]
            scratch = sum(1 for k in body if ins[k][1].startswith("scratch_"))
            out.append((name, int(m.group(2)), int(m.group(3)), m.group(1), hex(addr[d0]), nst, len(younger), others, scratch))
            n_loops += 1
        assert n_loops >= 2, (name, "prefetch loops found", n_loops)      # BPSK / QPSK at least (COMB), four elsewhere
    return out


def check(tmp):
    """the assertions; run in a process of its own (see the test below)"""
    loops = _prefetch_loops(_functions(_disassemble(tmp)))
    assert len(loops) >= 8 * 2 * 2 - 8                 # four equalisers x planes on / off x batch / stream, 2 .. 4 loops each
    # correctness, every instance: at least as many younger vector-memory instructions as the wait leaves outstanding
    for name, eq, hb, kind, where, nst, younger, others, scratch in loops:
        assert younger >= nst, (name, where, "vmcnt(%d) with %d younger vector-memory instructions" % (nst, younger))
    # performance: nothing but the counted wait and no spill access in the loops -- asserted for EVERY instance with the usual output
    # set: four equalisers, with and without plane output, batch and stream kernels.  (Until the 64-QAM staging addressed a bin by
    # one lane constant instead of two, wr_quad.h store_bins_lines, the STA instances reloaded a spilled register in their 64-QAM loop.)
    dirty = {}
    for name, eq, hb, kind, where, nst, younger, others, scratch in loops:
        if others or scratch:
            dirty[(kind, eq, hb)] = dirty.get((kind, eq, hb), 0) + 1
    known = {}                  # (instance -> loops allowed to be dirty: none)
    for key, n in dirty.items():
        assert key in known and n <= known[key], ("spill reload / full wait inside a prefetch loop", key, n, dirty)
    return len(loops)


def test_counted_waits_of_the_prefetch_loops(tmp_path):
    """Runs check() in a child interpreter (400 000 lines of disassembly parsed outside the pytest process); a failed assertion
    arrives as the child's traceback."""
    import sys
    if not os.path.exists(LIB):
        pytest.skip("libwifirx.so not built")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(tmp_path)], capture_output=True, text=True, timeout=300)
    if p.returncode == 77:
        pytest.skip(p.stdout.strip() or "tools missing")
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "prefetch loops checked" in p.stdout


if __name__ == "__main__":
    import sys
    try:
        n = check(sys.argv[1])
    except pytest.skip.Exception as e:          # tools not here
        print(e)
        sys.exit(77)
    print("%d prefetch loops checked" % n)
