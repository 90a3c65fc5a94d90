"""Static audit of k_attn_decode2 (csrc/lm_kernels.hip) - no GPU needed.  The kernel's K/V tile loads are asm statements that hipcc
does not track: it must never read, copy or overwrite a destination register between the load and the counted wait that covers it
(a v_mov there copies the STALE value while the data lands in the old register).  For every instantiation this lists
  * scratch (spill) instructions,
  * for every group of 16 tile loads: every instruction between the group and the wait that covers it which touches one of the
    group's destination registers.
Usage: python tools/isa_audit_attn2.py   (exit code 1 when a violation is found)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlx-audio-swift_amd", "csrc")


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-I" + CSRC,
                            "-I" + os.path.join(ROOT, "include"), "-S", os.path.join(CSRC, "lm_kernels.hip"), "-o", asm], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        txt = open(asm).read()
    bad = 0
    for name in re.findall(r"^(_Z\w*k_attn_decode2\w+):", txt, re.M):
        i = txt.index("\n" + name + ":")
        body = txt[i: txt.index(".Lfunc_end", i)]
        lines = [x.split(";")[0].rstrip() for x in body.split("\n")]
        lines = [l for l in lines if l.strip()]
        # simulate the vmcnt FIFO over the listing: every vector memory instruction (asm loads, compiler loads / stores, scratch
        # spills - they all count in vmcnt on gfx9) enters the queue in program order and retires in order; s_waitcnt vmcnt(N) retires
        # all but the youngest N.  A register written by a load still in the queue must not be read or written by anything else.
        n_scratch = sum("scratch_" in l for l in lines)
        fifo = []                                   # (line, dest register set) of outstanding vector memory operations
        viol = 0
        for k, l in enumerate(lines):
            st = l.strip()
            op = st.split()[0] if st.split() else ""
            if st.startswith((".LBB",)) or op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
                inflight_asm = [e for e in fifo if e[2]]
                if inflight_asm:
                    print(f"   line {k}: control flow / barrier ({st}) with {len(inflight_asm)} asm tile loads in flight")
                    viol += 1
                if st.startswith(".LBB"):
                    fifo = []                       # every path into a label has drained its asm loads (checked at the branch above)
                continue
            m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", st)
            if m:
                n = int(m.group(1))
                if n == 0:
                    fifo = []
                elif len(fifo) > n:
                    fifo = fifo[len(fifo) - n:]
                continue
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", st)
            used = set()
            for tk in toks:
                used |= regs_of(tk)
            busy = set()
            for e in fifo:
                if e[2]:
                    busy |= e[1]
            is_vmem = op.startswith(("global_", "buffer_", "scratch_", "flat_"))
            if is_vmem:
                dest = regs_of(st.split()[1].rstrip(",")) if "load" in op else set()
                asm_tile = "global_load_dwordx4" in st and ", s[" in st
                asm_pro = op == "global_load_dwordx4" and not asm_tile and " nt" not in st     # prologue slab / table loads (the touch loads are nt)
                srcs = used - dest
                if srcs & busy or (dest & busy):
                    print(f"   line {k}: {st}  <- touches a register an asm load is still writing")
                    viol += 1
                if op.startswith("scratch_") and any(e[2] for e in fifo):
                    print(f"   line {k}: {st}  <- spill while asm loads are in flight (it also shifts the counted waits)")
                    viol += 1
                fifo.append((k, dest, asm_tile or asm_pro))
                continue
            if used & busy:
                print(f"   line {k}: {st}  <- touches a register an asm load is still writing")
                viol += 1
        print(f"{name}: {len(lines)} instructions, {n_scratch} scratch ops, {viol} violations")
        bad += viol
    print("violations:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
