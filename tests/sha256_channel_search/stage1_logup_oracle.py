import sys, hashlib, struct, itertools
sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0,'/root/repo')
import numpy as np
from parse import *
from tests import oracle_binding as ob
P=2**31-1
d=parse_with_poseidon(open('/root/repo/tests/golden/proofs/hybrid_hash.bin','rb').read())
def qadd(a,b): return [(x+y)%P for x,y in zip(a,b)]
def qsub(a,b): return [(x-y)%P for x,y in zip(a,b)]
def qmulm(a,k): return [(x*k)%P for x in a]
def qinv(a): return [int(x) for x in ob.qm31_inv(np.array(a,dtype=np.uint32))]
def logup_ok(z,al):
    s=[int(x) for x in d['plonk_sum']]; s=qadd(s,[int(x) for x in d['poseidon_sum']])
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
def u64v(x):
    lo22=x&((1<<22)-1); m21=(x>>22)&((1<<21)-1); h21=(x>>43)&((1<<21)-1)
    return {'le8':struct.pack('<Q',x),'le8pad32':struct.pack('<Q',x)+b'\0'*24,'le4':struct.pack('<I',x&0xffffffff),'num':num_to_bytes(x),'be8':struct.pack('>Q',x),'be4':struct.pack('>I',x),
            'felt16':struct.pack('<4I',lo22,m21,h21,0),'felt32':struct.pack('<4I',lo22,m21,h21,0)+b'\0'*16,'skip':None,'byte':bytes([x&0xff])}
def words(h,end):
    return list(struct.unpack(('<' if end=='le' else '>')+'8I',h))
def red(x,mode):
    if mode=='mod': return x%P
    if mode=='mask': return x&P
    if mode=='mask0': 
        v=x&P; return 0 if v==P else v
found=[]
lp,lq=d['lp'],d['lq']
for Hn,H in (('sha',sha1),('sha2',sha2)):
 for order in ('dr','rd'):
  def mix(dg,data): 
      if data is None: return dg
      return H(dg+data) if order=='dr' else H(data+dg)
  for uk in u64v(1):
   for initn,init in (('zero',b'\0'*32),('empty',b''),('sha_empty',sha1(b'')),('sha_zero',sha1(b'\0'*32)),('root0',None)):
    for both in (True,False):   # both log sizes mixed, or combined?
      dg=init
      dg=mix(dg,d['commitments'][0]) if init is not None else d['commitments'][0]; dg=mix(dg,u64v(lp)[uk]); 
      if both: dg=mix(dg,u64v(lq)[uk])
      dg=mix(dg,d['commitments'][1])
      # draw variants
      srcs={}
      for c in (0,1):
          cb={'c32':struct.pack('<Q',c)+b'\0'*24,'c8':struct.pack('<Q',c),'c4':struct.pack('<I',c),'c1':bytes([c]),'c32be':b'\0'*24+struct.pack('>Q',c),'c8be':struct.pack('>Q',c),'c4be':struct.pack('>I',c)}
          for k,v in cb.items():
              srcs[('d|'+k,c)]=H(dg+v); srcs[(k+'|d',c)]=H(v+dg)
      srcs[('digest',0)]=dg; srcs[('H(d)',0)]=H(dg); srcs[('H(H(d))',0)]=H(H(dg))
      # bitcoin style: extract=H(d|0), d'=H(d); second: H(d'|0)
      srcs[('btc',0)]=H(dg+b'\0'); srcs[('btc',1)]=H(H(dg)+b'\0')
      names=set(k for k,_ in srcs)
      for nm in names:
          for end in ('le','be'):
              for mode in ('mod','mask','mask0'):
                  w0=[red(x,mode) for x in words(srcs[(nm,0)],end)]
                  cands=[('one',w0[:4],w0[4:])]
                  if (nm,1) in srcs:
                      w1=[red(x,mode) for x in words(srcs[(nm,1)],end)]
                      cands.append(('two',w0[:4],w1[:4]))
                  for cn,z,al in cands:
                      for swap in (False,True):
                          zz,aa=(al,z) if swap else (z,al)
                          if logup_ok(zz,aa):
                              found.append((Hn,order,uk,initn,both,nm,end,mode,cn,swap)); print('FOUND',found[-1])
print('done',len(found))
