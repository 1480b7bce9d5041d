"""Diagnostic: vector loads that the compiler waits for with s_waitcnt vmcnt(0) a few instructions later (a serialized
memory round trip; inside a loop or an unrolled epilogue these add up).

    python tools/serialized_loads.py OBJECT.o KERNEL_NAME_SUBSTRING"""
import os, re, subprocess, sys
obj, pat = sys.argv[1], sys.argv[2]
LL = "/opt/rocm/lib/llvm/bin"
tmp = "/tmp/_sl"; os.makedirs(tmp, exist_ok=True)
subprocess.run(["cp", obj, tmp + "/x.o"], check=True)
subprocess.run([LL + "/llvm-objdump", "--offloading", "x.o"], cwd=tmp, capture_output=True)
dev = [f for f in os.listdir(tmp) if f.startswith("x.o.") and "gfx950" in f][0]
txt = subprocess.run([LL + "/llvm-objdump", "-d", tmp + "/" + dev], capture_output=True, text=True).stdout.split("\n")
cur, funcs = None, {}
for l in txt:
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur:
        funcs[cur].append(l.split("//")[0].strip())
for name, body in funcs.items():
    if pat not in name:
        continue
    hits = []
    for i, l in enumerate(body):
        if re.search(r"s_waitcnt.*vmcnt\(0\)", l):
            for j in range(i - 1, max(i - 5, 0), -1):
                if re.match(r"(global_load|buffer_load|flat_load)", body[j]):
                    hits.append((i, body[j][:48])); break
    # group consecutive similar hits
    print(name[:100], "serialized loads:", len(hits))
    last = None; run = 0
    for i, h in hits:
        key = h.split()[0]
        if last and key == last[1] and i - last[0] < 40:
            run += 1
        else:
            if last: print("   x%-3d near instr %-6d %s" % (run, last[0], last[2]))
            run = 1
        last = (i, key, h)
    if last: print("   x%-3d near instr %-6d %s" % (run, last[0], last[2]))
