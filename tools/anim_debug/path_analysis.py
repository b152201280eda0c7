import gzip, sys, json
import numpy as np
ROOT='/root/repo'
def read_fasta(path):
    recs={}; name=None; buf=[]
    op=gzip.open if path.endswith('.gz') else open
    for l in op(path,'rt'):
        if l.startswith('>'):
            if name: recs[name]=''.join(buf).upper()
            name=l[1:].split()[0]; buf=[]
        else: buf.append(l.strip())
    if name: recs[name]=''.join(buf).upper()
    return recs
COMP=str.maketrans('ACGT','TGCA')
def read_delta_full(path):
    out=[]; cur=None
    for l in gzip.open(path,'rt'):
        t=l.split()
        if l.startswith('>'): hdr=(t[0][1:],t[1]); continue
        if len(t)==7: cur={'hdr':hdr,'c':tuple(map(int,t[:4])),'err':int(t[4]),'ind':[]}; continue
        if len(t)==1 and cur is not None:
            v=int(t[0])
            if v==0: out.append(cur); cur=None
            else: cur['ind'].append(v)
    return out
def path_stats(r,q,ind):
    i=j=0; m=mm=0; gaps=[]  # gaps: list of run lengths
    cols=[]
    for v in ind:
        n=abs(v)-1
        for t in range(n):
            cols.append('M' if r[i]==q[j] else 'X'); i+=1; j+=1
        if v>0: cols.append('D'); i+=1
        else: cols.append('I'); j+=1
    while i<len(r) and j<len(q):
        cols.append('M' if r[i]==q[j] else 'X'); i+=1; j+=1
    assert i==len(r) and j==len(q),(i,len(r),j,len(q))
    return ''.join(cols)
def score(cols,ma=3,mi=-7,go=-10,ge=-7):
    s=0; prev=''
    for c in cols:
        if c=='M': s+=ma
        elif c=='X': s+=mi
        else: s+= ge if c==prev else go
        prev=c
    return s
def optimal(r,q,ma=3,mi=-7,go=-10,ge=-7):
    # global affine alignment, max score then min errors; O(nm) python/numpy row-wise
    n,m=len(r),len(q); NEG=-10**9
    rq=np.frombuffer(q.encode(),dtype=np.uint8)
    H=np.full(m+1,NEG,dtype=np.int64); He=np.zeros(m+1,dtype=np.int64)
    Y=np.full(m+1,NEG,dtype=np.int64)
    H[0]=0
    # first row: gaps in ref (consume query)
    for j in range(1,m+1):
        H[j]=go+ge*(j-1); He[j]=j
    X=np.full(m+1,NEG,dtype=np.int64); Xe=np.zeros(m+1,dtype=np.int64)
    BIG=1<<20
    for i in range(1,n+1):
        # keys combine score and errors: key = score*BIG - errors
        Hk=H*BIG-He; Xk=np.where(X>NEG//2, X*BIG-Xe, NEG*BIG)
        nXk=np.maximum(Hk+go*BIG-1, Xk+ge*BIG-1)   # consume ref base i (vertical)
        sub=np.where(rq==ord(r[i-1]), ma*BIG, mi*BIG-1)
        diag=np.empty(m+1,dtype=np.int64); diag[0]=NEG*BIG; diag[1:]=Hk[:-1]+sub
        base=np.maximum(diag,nXk)
        # horizontal (consume query) needs sequential scan
        nH=np.empty(m+1,dtype=np.int64); yk=NEG*BIG
        nH[0]=nXk[0]
        b=base.tolist(); out=[0]*(m+1); out[0]=int(nXk[0]); y=NEG*BIG
        for j in range(1,m+1):
            y=max(out[j-1]+go*BIG-1, y+ge*BIG-1)
            out[j]=max(b[j],y)
        nHk=np.array(out,dtype=np.int64)
        H=np.floor_divide(nHk+BIG-1,BIG); He=H*BIG-nHk
        X=np.floor_divide(nXk+BIG-1,BIG); Xe=X*BIG-nXk
        X=np.where(nXk<=NEG*BIG//2,NEG,X)
    return int(H[m]),int(He[m])
if __name__=='__main__':
    pair=sys.argv[1]  # e.g. NC_002696_vs_NC_014100
    a,b=pair.split('_vs_')
    R=read_fasta(f'{ROOT}/tests/golden/genomes/caulobacter/{a}.fna.gz'); Q=read_fasta(f'{ROOT}/tests/golden/genomes/caulobacter/{b}.fna.gz')
    D=read_delta_full(f'{ROOT}/tests/golden/anim/caulobacter/{pair}.delta.gz')
    want=[tuple(map(int,x.split(','))) for x in sys.argv[2:]]
    for d in D:
        if d['c'] not in want: continue
        rs,re,qs,qe=d['c']; r=R[d['hdr'][0]][rs-1:re]
        q=Q[d['hdr'][1]]
        q=q[qs-1:qe] if qs<qe else q[qe-1:qs][::-1].translate(COMP)
        cols=path_stats(r,q,d['ind'])
        errs=sum(c!='M' for c in cols)
        print(d['c'],'mummer errors',d['err'],'recount',errs,'path score',score(cols), 'M',cols.count('M'),'X',cols.count('X'),'D',cols.count('D'),'I',cols.count('I'))
        print('   optimal (score, min errors):', optimal(r,q))
