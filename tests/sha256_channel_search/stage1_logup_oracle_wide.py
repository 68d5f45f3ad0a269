import sys, hashlib, struct, itertools
sys.path.insert(0,'/root/repo/tests/sha256_channel_search'); sys.path.insert(0,'/root/repo')
import numpy as np
from parse import *
from tests import oracle_binding as ob
P=2**31-1
d=parse_with_poseidon(open('/root/repo/tests/golden/proofs/hybrid_hash.bin','rb').read())
def qadd(a,b): return [(x+y)%P for x,y in zip(a,b)]
def qsub(a,b): return [(x-y)%P for x,y in zip(a,b)]
def qmulm(a,k): return [(x*k)%P for x in a]
def qinv(a): return [int(x) for x in ob.qm31_inv(np.array(a,dtype=np.uint32))]
S0=qadd([int(x) for x in d['plonk_sum']],[int(x) for x in d['poseidon_sum']])
def logup_ok(z,al):
    s=S0
    for idx,val in ((1,[1,0,0,0]),(2,[0,1,0,0]),(3,[0,0,1,0])):
        den=qsub(qadd(val,qmulm(al,idx)),z)
        if den==[0,0,0,0]: return False
        s=qadd(s,qinv(den))
    return s==[0,0,0,0]
def sha1(b): return hashlib.sha256(b).digest()
def sha2(b): return sha1(sha1(b))
def num_to_bytes(v):
    v=int(v); out=bytearray()
    while v>0: out.append(v&0xff); v>>=8
    if out and out[-1]&0x80: out.append(0)
    return bytes(out)
def u64enc(x):
    lo22=x&((1<<22)-1); m21=(x>>22)&((1<<21)-1); h21=(x>>43)&((1<<21)-1)
    return {'le8':struct.pack('<Q',x),'le8pad32':struct.pack('<Q',x)+b'\0'*24,'le4':struct.pack('<I',x&0xffffffff),'num':num_to_bytes(x),'be8':struct.pack('>Q',x),'be4':struct.pack('>I',x),
            'felt16':struct.pack('<4I',lo22,m21,h21,0),'felt32':struct.pack('<4I',lo22,m21,h21,0)+b'\0'*16,'byte':bytes([x&0xff]),'le4pad32':struct.pack('<I',x)+b'\0'*28,'be8pad32':b'\0'*24+struct.pack('>Q',x)}
lp,lq=d['lp'],d['lq']
def stmt0_variants():
    out={}
    for k in u64enc(1):
        out['sep_'+k]=[u64enc(lp)[k],u64enc(lq)[k]]
        out['cat_'+k]=[u64enc(lp)[k]+u64enc(lq)[k]]
    out['u32pair']=[struct.pack('<2I',lp,lq)]
    out['u32pair_pad32']=[struct.pack('<2I',lp,lq)+b'\0'*24]
    out['felt_lp_lq']=[struct.pack('<4I',lp,lq,0,0)]
    out['felt_lp_lq32']=[struct.pack('<4I',lp,lq,0,0)+b'\0'*16]
    out['twofelts']=[struct.pack('<8I',lp,0,0,0,lq,0,0,0)]
    out['skip']=[]
    out['be_pair']=[struct.pack('>2I',lp,lq)]
    return out
def words(h,end): return list(struct.unpack(('<' if end=='le' else '>')+'8I',h))
def red(x,mode):
    if mode=='mod': return x%P
    if mode=='mask': return x&P
    v=x&P; return 0 if v==P else v
found=[]; tried=0
for Hn,H in (('sha',sha1),('sha2',sha2)):
 for order in ('dr','rd'):
  def mix(dg,data): return H(dg+data) if order=='dr' else H(data+dg)
  for initn,init in (('zero',b'\0'*32),('empty',b''),('root0',None)):
   for sn,sv in stmt0_variants().items():
      dg=mix(init,d['commitments'][0]) if init is not None else d['commitments'][0]
      for x in sv: dg=mix(dg,x)
      dg=mix(dg,d['commitments'][1])
      srcs={}
      nch={'zero':4,'empty':4,'root0':3}[initn] if True else 0
      for c in range(0,6):
        for ch in (None,len(sv)+2, len(sv)+1):
          pre=b'' if ch is None else None
          fmts={'c32':struct.pack('<Q',c)+b'\0'*24,'c8':struct.pack('<Q',c),'c4':struct.pack('<I',c),'c1':bytes([c]),'c32be':b'\0'*24+struct.pack('>Q',c),'c4be':struct.pack('>I',c)}
          if ch is not None:
              fmts={'ch4c4':struct.pack('<2I',ch,c),'ch8c8':struct.pack('<2Q',ch,c),'ch4c4pad':struct.pack('<2I',ch,c)+b'\0'*24,'ch16c16':struct.pack('<Q',ch)+b'\0'*8+struct.pack('<Q',c)+b'\0'*8}
          for k,v in fmts.items():
              tag=k if ch is None else f'{k}@{ch}'
              srcs[('d|'+tag,c)]=H(dg+v); srcs[(tag+'|d',c)]=H(v+dg)
      srcs[('digest',0)]=dg; srcs[('H(d)',0)]=H(dg); srcs[('btc',0)]=H(dg+b'\0'); srcs[('btc',1)]=H(H(dg)+b'\0'); srcs[('H(d)',1)]=H(H(dg))
      names=set(k for k,_ in srcs)
      for nm in names:
        cs=sorted(c for (k,c) in srcs if k==nm)
        for end in ('le','be'):
          for mode in ('mod','mask','mask0'):
            W={c:[red(x,mode) for x in words(srcs[(nm,c)],end)] for c in cs}
            cands=[]
            for c in cs:
                cands.append((f'one@{c}',W[c][:4],W[c][4:]))
                if c+1 in W: cands.append((f'two@{c}',W[c][:4],W[c+1][:4]))
            for cn,z,al in cands:
                for swap in (False,True):
                    tried+=1
                    zz,aa=(al,z) if swap else (z,al)
                    if logup_ok(zz,aa):
                        found.append((Hn,order,initn,sn,nm,end,mode,cn,swap)); print('FOUND',found[-1],flush=True)
print('done',len(found),'tried',tried)
