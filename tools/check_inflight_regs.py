"""Lint for hand-placed vmcnt waits: no instruction may touch the destination registers of a vector-memory load that is
still in flight.

    python tools/check_inflight_regs.py OBJECT.o KERNEL_NAME_SUBSTRING [--sgpr-only]

The bf16 fused kernel issues its stash loads as inline assembly and waits for them with hand-placed `s_waitcnt vmcnt(N)`
(csrc/inr_siren_bf16_impl.h: the compiler's own wait insertion answered the first use with vmcnt(0)).  The compiler does
not know those registers are written asynchronously: a copy or spill it decides to put between the load and the wait
would read stale data -- and whether it does depends on register allocation, i.e. on unrelated code.  This walks the
kernel's instructions in program order with the in-order vmcnt model of gfx9 (every vector-memory instruction counts;
`s_waitcnt vmcnt(N)` retires all but the N youngest) and reports any instruction that reads or writes a register of a
load not yet retired.  Linear walk: branches are not followed (loop bodies are checked once, in layout order)."""
import os, re, subprocess, sys

LL = "/opt/rocm/lib/llvm/bin"


def kernel_lines(obj, pat):
    tmp = "/tmp/_chk_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["cp", obj, tmp + "/x.o"], check=True)
    subprocess.run([f"{LL}/llvm-objdump", "--offloading", "x.o"], cwd=tmp, capture_output=True)
    devs = [f for f in os.listdir(tmp) if f.startswith("x.o.") and "gfx950" in f]
    txt = []  # (an object without device code -- the host-only API unit -- has nothing to check)
    if devs:
        txt = subprocess.run([f"{LL}/llvm-objdump", "-d", tmp + "/" + devs[0]], capture_output=True, text=True).stdout.split("\n")
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    funcs, cur = {}, None
    for l in txt:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and l.strip():
            funcs[cur].append(l.split("//")[0].strip())
    return {k: v for k, v in funcs.items() if pat in k}


def regs(tok):
    """VGPR numbers named by an operand token: v12, v[3:6]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")


def check(lines):
    fifo, bad = [], []  # fifo: (destination registers or empty set, text), oldest first
    for i, l in enumerate(lines):
        op = l.split()[0] if l else ""
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if op.startswith("s_waitcnt"):
            if m:
                n = int(m.group(1))
                fifo = fifo[len(fifo) - n:] if n < len(fifo) else fifo
                if n == 0:
                    fifo = []
            continue
        inflight = set().union(*[d for d, _ in fifo]) if fifo else set()
        if inflight:
            touched = regs(l.split(None, 1)[1]) if len(l.split(None, 1)) > 1 else set()
            hit = touched & inflight
            if hit:
                src = [t for d, t in fifo if d & hit]
                bad.append((i, l, sorted(hit), src[0]))
        if VMEM.match(op):
            dest = set()
            # destinations are tracked for buffer loads (what the inline assembly and the raw-buffer builtins emit); the
            # compiler's own global loads sit in if / else diamonds this linear walk would misread
            if op.startswith("buffer_load") and "lds" not in l:
                dest = regs(l.split(None, 1)[1].split(",")[0])
            fifo.append((dest, l))
    return bad


def sregs(tok):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", tok):
        out.add(int(a))
    return out


def check_sgpr_hazard(lines):
    """gfx9: an SGPR written by a VALU instruction (v_readlane, v_readfirstlane, v_cmp ...) must not be read by a
    vector-memory instruction within the next 5 wait states.  The compiler's hazard recognizer pads its own memory
    instructions; it does not look inside inline assembly -- an SGPR-spill reload (v_readlane) right in front of an
    inline-assembly load made that load read a stale offset (timing-dependent)."""
    bad = []
    for i, l in enumerate(lines):
        op = l.split()[0] if l else ""
        if not VMEM.match(op) or len(l.split(None, 1)) < 2:
            continue
        need = sregs(l.split(None, 1)[1])
        if not need:
            continue
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            p = lines[j]
            pop = p.split()[0] if p else ""
            if pop.startswith("s_cbranch") or pop.startswith("s_branch") or pop == "s_barrier":
                break  # (another path: not followed)
            if pop.startswith("v_") and len(p.split(None, 1)) > 1:
                dst = p.split(None, 1)[1].split(",")[0]
                hit = sregs(dst) & need
                if hit:
                    bad.append((i, l, sorted(hit), p))
                    break
            m = re.match(r"s_nop (\d+)", p)
            states += int(m.group(1)) + 1 if m else 1
            j -= 1
    return bad


if __name__ == "__main__":
    sgpr_only = "--sgpr-only" in sys.argv  # (the in-flight walk misreads the if / else diamonds of compiler-scheduled loads)
    argv = [a for a in sys.argv if a != "--sgpr-only"]
    found = kernel_lines(argv[1], argv[2])
    rc = 0
    for name, lines in found.items():
        bad = [] if sgpr_only else check(lines)
        print("%s: %d instructions, %d touch a register in flight" % (name[:80], len(lines), len(bad)))
        for i, l, hit, src in bad[:20]:
            print("   line %d: %s   <- v%s of: %s" % (i, l[:90], hit, src[:70]))
        haz = check_sgpr_hazard(lines)
        print("%s: %d vector-memory instructions read an SGPR a VALU wrote < 5 wait states earlier" % (name[:40], len(haz)))
        for i, l, hit, src in haz[:20]:
            print("   line %d: %s   <- s%s written by: %s" % (i, l[:80], hit, src[:60]))
        rc |= 1 if bad or haz else 0
    sys.exit(rc)
