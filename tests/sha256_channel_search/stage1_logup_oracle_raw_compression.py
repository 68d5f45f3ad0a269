import sys, struct, hashlib, itertools
sys.path.insert(0,'/root/repo/tests/sha256_channel_search'); sys.path.insert(0,'/root/repo')
import numpy as np
from parse import *
from tests import oracle_binding as ob
P=2**31-1
K=[0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2]
IV=[0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19]
def rotr(x,n): return ((x>>n)|(x<<(32-n)))&0xffffffff
def compress(state,block):
    w=list(struct.unpack('>16I',block))
    for i in range(16,64):
        s0=rotr(w[i-15],7)^rotr(w[i-15],18)^(w[i-15]>>3); s1=rotr(w[i-2],17)^rotr(w[i-2],19)^(w[i-2]>>10)
        w.append((w[i-16]+s0+w[i-7]+s1)&0xffffffff)
    a,b,c,d,e,f,g,h=state
    for i in range(64):
        S1=rotr(e,6)^rotr(e,11)^rotr(e,25); ch=(e&f)^((~e)&g&0xffffffff); t1=(h+S1+ch+K[i]+w[i])&0xffffffff
        S0=rotr(a,2)^rotr(a,13)^rotr(a,22); maj=(a&b)^(a&c)^(b&c); t2=(S0+maj)&0xffffffff
        h,g,f,e,d,c,b,a=g,f,e,(d+t1)&0xffffffff,c,b,a,(t1+t2)&0xffffffff
    return [(x+y)&0xffffffff for x,y in zip(state,[a,b,c,d,e,f,g,h])]
assert struct.pack('>8I',*compress(IV,b'\x80'+b'\0'*63))==hashlib.sha256(b'').digest()
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
lp,lq=d['lp'],d['lq']
def b2s(b,end): return list(struct.unpack(end+'8I',b))
def s2b(s,end): return struct.pack(end+'8I',*s)
found=0; tried=0
def red(x,m): return x%P if m=='mod' else (x&P if m=='mask' else (0 if (x&P)==P else x&P))
# raw compression channel variants: state is 8 words; mixing data (<=32 bytes) as block = [digest|data] or [data|digest] through compress(IV, block), or compress(state, data-block padded with zeros)
for send in ('>','<'):       # how the 32-byte root maps to words of the block
 for mode in ('iv_dr','iv_rd','chain_data0','chain_0data'):
  def mix(state,data):
      data=data+b'\0'*(32-len(data)) if len(data)<32 else data[:32]
      if mode=='iv_dr': return compress(IV, s2b(state,'>')+data)
      if mode=='iv_rd': return compress(IV, data+s2b(state,'>'))
      if mode=='chain_data0': return compress(state, data+b'\0'*32)
      if mode=='chain_0data': return compress(state, b'\0'*32+data)
  for init in ([0]*8, IV):
   for u in ('le8','le4','be8','be4','felt'):
    def enc(x):
        return {'le8':struct.pack('<Q',x),'le4':struct.pack('<I',x),'be8':struct.pack('>Q',x),'be4':struct.pack('>I',x),'felt':struct.pack('<4I',x,0,0,0)}[u]
    for st0 in ('sep','pair','skip'):
      st=list(init)
      st=mix(st,d['commitments'][0])
      if st0=='sep': st=mix(st,enc(lp)); st=mix(st,enc(lq))
      elif st0=='pair': st=mix(st,enc(lp)+enc(lq))
      st=mix(st,d['commitments'][1])
      for c in range(0,4):
        for cenc in ('le4','le8','be4','be8'):
          cb={'le4':struct.pack('<I',c),'le8':struct.pack('<Q',c),'be4':struct.pack('>I',c),'be8':struct.pack('>Q',c)}[cenc]
          for dmode in ('iv_dc','iv_cd','chain_c'):
            if dmode=='iv_dc': out=compress(IV,s2b(st,'>')+cb+b'\0'*(32-len(cb)))
            elif dmode=='iv_cd': out=compress(IV,cb+b'\0'*(32-len(cb))+s2b(st,'>'))
            else: out=compress(st,cb+b'\0'*(64-len(cb)))
            for wend in ('native','swap'):
              ws=out if wend=='native' else [struct.unpack('<I',struct.pack('>I',x))[0] for x in out]
              for m in ('mod','mask','mask0'):
                w=[red(x,m) for x in ws]
                for z,al in ((w[:4],w[4:]),(w[4:],w[:4])):
                    tried+=1
                    if logup_ok(z,al): found+=1; print('FOUND',send,mode,init==IV,u,st0,c,cenc,dmode,wend,m,flush=True)
print('done',found,'tried',tried)
