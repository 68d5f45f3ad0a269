import numpy as np, struct
class R:
    def __init__(s,b): s.b=b; s.p=0
    def u32(s): v=struct.unpack_from('<I',s.b,s.p)[0]; s.p+=4; return v
    def u64(s): v=struct.unpack_from('<Q',s.b,s.p)[0]; s.p+=8; return v
    def take(s,n): v=s.b[s.p:s.p+n]; s.p+=n; return v
def parse_stark(r):
    d={}
    d['pow_bits']=r.u32(); d['blowup']=r.u32(); d['log_last']=r.u32(); d['nq']=r.u64()
    n=r.u64(); d['commitments']=[r.take(32) for _ in range(n)]
    nt=r.u64(); sv=[]
    for t in range(nt):
        nc=r.u64(); cols=[]
        for c in range(nc):
            ns=r.u64(); cols.append([np.frombuffer(r.take(16),dtype='<u4') for _ in range(ns)])
        sv.append(cols)
    d['sampled']=sv
    nd=r.u64(); dec=[]
    for t in range(nd):
        nh=r.u64(); hw=[r.take(32) for _ in range(nh)]
        ncw=r.u64(); cw=np.frombuffer(r.take(4*ncw),dtype='<u4')
        dec.append((hw,cw))
    d['decommit']=dec
    nqv=r.u64(); qv=[]
    for t in range(nqv):
        n=r.u64(); qv.append(np.frombuffer(r.take(4*n),dtype='<u4'))
    d['queried']=qv
    d['nonce']=r.u64()
    def layer():
        n=r.u64(); wit=np.frombuffer(r.take(16*n),dtype='<u4').reshape(-1,4)
        nh=r.u64(); hw=[r.take(32) for _ in range(nh)]
        ncw=r.u64(); cw=r.take(4*ncw)
        com=r.take(32)
        return {'wit':wit,'hw':hw,'com':com}
    d['first']=layer()
    ni=r.u64(); d['inner']=[layer() for _ in range(ni)]
    n=r.u64(); d['last']=np.frombuffer(r.take(16*n),dtype='<u4').reshape(-1,4); d['last_log']=r.u32()
    return d
def parse_with_poseidon(b):
    r=R(b); d={'lp':r.u32(),'lq':r.u32(),'plonk_sum':np.frombuffer(r.take(16),dtype='<u4'),'poseidon_sum':np.frombuffer(r.take(16),dtype='<u4')}
    d.update(parse_stark(r)); assert r.p==len(b),(r.p,len(b)); return d
def parse_without_poseidon(b):
    r=R(b); d={'log_size':r.u32(),'total_sum':np.frombuffer(r.take(16),dtype='<u4')}
    d.update(parse_stark(r)); assert r.p==len(b),(r.p,len(b)); return d
if __name__=='__main__':
    d=parse_without_poseidon(open('/root/reference/examples/last-layer/data/bitcoin_proof.bin','rb').read())
    print({k:(v if not isinstance(v,(list,np.ndarray,bytes,dict)) else type(v)) for k,v in d.items()})
    print('trees',[len(t) for t in d['sampled']],[[len(c) for c in t] for t in d['sampled']])
    print('hw',[len(h) for h,_ in d['decommit']],'qv',[len(q) for q in d['queried']])
    print('first wit',d['first']['wit'].shape,len(d['first']['hw']),'inner',[(l['wit'].shape[0],len(l['hw'])) for l in d['inner']],'last',d['last'].shape,d['last_log'])
    d2=parse_with_poseidon(open('/root/repo/tests/golden/proofs/hybrid_hash.bin','rb').read())
    print('hw',[len(h) for h,_ in d2['decommit']],'qv',[len(q) for q in d2['queried']])
    print('first wit',d2['first']['wit'].shape,len(d2['first']['hw']),'inner',[(l['wit'].shape[0],len(l['hw'])) for l in d2['inner']],'last',d2['last'].shape,d2['last_log'])
