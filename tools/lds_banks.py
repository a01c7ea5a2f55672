import itertools,sys
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups+= [[l+32 for l in g] for g in groups]
def cycles(addr16):
    tot=0
    for g in groups:
        banks={}
        for l in g:
            banks.setdefault(addr16[l]%16,set()).add(addr16[l])
        tot+=max(len(v) for v in banks.values())
    return tot
def coords(W,TH,TI,BM):
    out=[]
    for wm in range(BM//64):
      for mt in range(4):
        for kh in range(3):
          for kw in range(3):
            L=[]
            for lane in range(64):
                m=wm*64+mt*16+(lane&15)
                ti=m//W//TH; ty=m//W%TH; tx=m%W
                L.append((ti*(TH+2)+ty+kh, tx+kw, ty+kh))   # global halo row, hx, hy within image
            out.append(L)
    return out
for (W,TH,TI,BM) in ((8,8,2,128),(4,4,8,128)):
    C=coords(W,TH,TI,BM)
    P=TH+2
    for stride in range(W+2, 17):
        best=None
        # s over hy parity-ish: try s tables of period 2,3,4,6 limited
        for per in (1,2,4):
            for T in itertools.product(range(4),repeat=per):
                for xs in (0,1):
                    def f(hyy,hx,hy,lg):
                        hp=hyy*stride+hx
                        return hp*4+(lg ^ T[hyy%per] ^ (2*((hx>>2)&1) if xs else 0))
                    tot=0
                    for L in C:
                        tot+=cycles([f(L[l][0],L[l][1],L[l][2],l>>4) for l in range(64)])
                    r=tot/len(C)
                    if best is None or r<best[0]: best=(r,per,T,xs)
        print(W,'stride',stride,best)
print('unified')
for (W,TH,TI,BM) in ((32,8,1,256),(32,4,1,128),(16,8,1,128),(16,16,1,256),(8,8,2,128),(8,8,4,256),(4,4,8,128),(4,4,16,256),(18,7,1,126)):
    C=coords(W,TH,TI,BM); stride=W+2
    res={}
    for name,g in (('A',lambda hyy,hx:2*(((hyy*stride+hx)>>2)&1)),('B',lambda hyy,hx:2*(hyy&1)),('C',lambda hyy,hx:2*(((hx>>2)^hyy)&1)),
                   ('D',lambda hyy,hx:2*(((hx>>2)&1)) if W>=16 else 2*(hyy&1))):
        tot=0
        for L in C:
            tot+=cycles([ (L[l][0]*stride+L[l][1])*4+((l>>4)^g(L[l][0],L[l][1])) for l in range(64)])
        res[name]=round(tot/len(C),2)
    print(W,TH,TI,res)
