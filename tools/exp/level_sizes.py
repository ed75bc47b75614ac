import sys,re,collections
h=collections.Counter(); first=collections.Counter(); n=0
seq=[]
for l in sys.stdin:
    m=re.match(r"\[level\] mode (\d+) batch-level (\d+) nf (\d+) ef (\d+)",l)
    if not m: continue
    mode,i,nf,ef=map(int,m.groups())
    tot=nf+ef
    b=0
    while (1<<b)<=tot: b+=1
    h[b]+=1
    if i==0: first[b]+=1
    n+=1
print("levels",n)
for b in sorted(h): print("nf+ef < 2^%-2d: %6d levels (%5.1f %%), first of a batch %6d"%(b,h[b],100.0*h[b]/n,first[b]))
