#!/usr/bin/env python3
"""Diagnostic: per-loop instruction mix of one kernel of a built object (what sits beside the MFMAs of each
GEMM loop).  Usage: python tools/loopstats.py <object.o> <mangled-kernel-substring>"""
import collections, os, re, subprocess, sys, tempfile
obj, pat = sys.argv[1], sys.argv[2]
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", os.path.abspath(obj)], cwd=tmp, check=True, capture_output=True)
co = [f for f in os.listdir(os.path.dirname(os.path.abspath(obj))) if f.startswith(os.path.basename(obj) + ".") and "gfx950" in f]
d = os.path.dirname(os.path.abspath(obj))
src = os.path.join(d, co[0])
asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", src], check=True, capture_output=True, text=True).stdout
for f in os.listdir(d):
    if f.startswith(os.path.basename(obj) + "."):
        os.remove(os.path.join(d, f))
lines = asm.split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
for n, i in enumerate(starts):
    if pat in lines[i]:
        body = lines[i:(starts[n + 1] if n + 1 < len(starts) else len(lines))]
        break
else:
    sys.exit("kernel not found; have:\n" + "\n".join(lines[i] for i in starts))
ins = []
for li, l in enumerate(body):
    m = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), li))
base = ins[0][0]
off2idx = {a - base: i for i, (a, _, _) in enumerate(ins)}
print(body[0])
for i, (a, mn, li) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        m = re.search(r"<.*\+0x([0-9a-f]+)>", body[li])
        if not m:
            continue
        t = off2idx.get(int(m.group(1), 16))
        if t is None or t >= i:
            continue
        c = collections.Counter()
        for (_, m2, _) in ins[t:i + 1]:
            if m2.startswith("v_mfma"): c["mfma"] += 1
            elif m2 in ("v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"): c["trans"] += 1
            elif m2.startswith("v_"): c["valu"] += 1
            elif m2.startswith("s_nop"): c["s_nop"] += 1
            elif m2.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif m2.startswith("s_"): c["salu"] += 1
            elif m2.startswith("ds_"): c["ds"] += 1
            elif m2.split("_")[0] in ("buffer", "global", "scratch", "flat"):
                c[m2.split("_")[0] + ("_st" if "store" in m2 else "_ld")] += 1
            else: c[m2] += 1
        if c["mfma"] >= 8:
            print(f"loop @{ins[t][2]}..{li} n={i + 1 - t}", dict(c))
