"""Diagnostic: where a kernel's scratch (register spill) instructions sit relative to its loops.

    python tools/spill_locations.py OBJECT.o KERNEL_NAME_SUBSTRING

Prints, per matching kernel, a histogram {length of the innermost enclosing loop in instructions (-1: none): count}."""
import re,sys,subprocess,os
obj=sys.argv[1]; pat=sys.argv[2]
LL="/opt/rocm/lib/llvm/bin"
tmp="/tmp/_spl"; os.makedirs(tmp,exist_ok=True)
for f in os.listdir(tmp): os.remove(os.path.join(tmp,f))
subprocess.run(["cp",obj,tmp+"/x.o"],check=True)
subprocess.run([f"{LL}/llvm-objdump","--offloading","x.o"],cwd=tmp,capture_output=True)
dev=[f for f in os.listdir(tmp) if f.startswith("x.o.") and "gfx950" in f][0]
txt=subprocess.run([f"{LL}/llvm-objdump","-d",tmp+"/"+dev],capture_output=True,text=True).stdout.split("\n")
# split into functions
funcs={}; cur=None
for l in txt:
    m=re.match(r'^[0-9a-f]+ <(.*)>:',l)
    if m: cur=m.group(1); funcs[cur]=[]; continue
    if cur: funcs[cur].append(l)
for name,lines in funcs.items():
    if pat not in name: continue
    addr={}
    for i,l in enumerate(lines):
        m=re.search(r'//\s*([0-9A-F]{12}):',l)
        if m: addr[int(m.group(1),16)]=i
    loops=[]
    for i,l in enumerate(lines):
        m=re.match(r'\s*s_c?branch\w*\s+(\d+)',l)
        if m:
            off=int(m.group(1))
            if off>=32768:
                a=int(re.search(r'//\s*([0-9A-F]{12}):',l).group(1),16)
                t=addr.get(a+4+(off-65536)*4)
                if t is not None: loops.append((t,i))
    sc=[i for i,l in enumerate(lines) if 'scratch_' in l]
    inner=0
    for i in sc:
        # innermost loop length containing it
        cont=[(b-a) for a,b in loops if a<=i<=b]
        if cont and min(cont)<3000: inner+=1
    import collections
    hist=collections.Counter()
    for i in sc:
        cont=[(b-a) for a,b in loops if a<=i<=b]
        hist[min(cont) if cont else -1]+=1
    print(sorted(hist.items()))
    print(name[:90], "scratch ops",len(sc),"inside loops shorter than 3000 instrs:",inner, "loops",len(loops))
