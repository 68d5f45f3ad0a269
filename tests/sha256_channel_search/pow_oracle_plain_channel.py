import sys, hashlib, struct, itertools
sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import numpy as np
from parse import *
d=parse_without_poseidon(open('/root/reference/examples/last-layer/data/bitcoin_proof.bin','rb').read())
def sha1(b): return hashlib.sha256(b).digest()
def sha2(b): return hashlib.sha256(hashlib.sha256(b).digest()).digest()
flat=[s for t in d['sampled'] for c in t for s in c]
nonce=d['nonce']; ls=d['log_size']
def num_to_bytes(v):
    v=int(v); out=bytearray()
    while v>0: out.append(v&0xff); v>>=8
    if out and out[-1]&0x80: out.append(0)
    return bytes(out)
def enc_felt(q,enc,H):
    q=[int(x) for x in q]
    if enc=='le16': return struct.pack('<4I',*q)
    if enc=='be16': return struct.pack('>4I',*q)
    if enc=='numcat': return b''.join(num_to_bytes(x) for x in q)
    if enc=='numcat_rev': return b''.join(num_to_bytes(x) for x in q[::-1])
    if enc=='chain_rev':  # bitcoin-circle-stark hash_qm31: start from last coordinate
        r=H(num_to_bytes(q[3])); r=H(num_to_bytes(q[2])+r); r=H(num_to_bytes(q[1])+r); r=H(num_to_bytes(q[0])+r); return r
    if enc=='chain_fwd':
        r=H(num_to_bytes(q[0])); r=H(num_to_bytes(q[1])+r); r=H(num_to_bytes(q[2])+r); r=H(num_to_bytes(q[3])+r); return r
    if enc=='chain_rev_le4':
        r=H(struct.pack('<I',q[3])); r=H(struct.pack('<I',q[2])+r); r=H(struct.pack('<I',q[1])+r); r=H(struct.pack('<I',q[0])+r); return r
    if enc=='sha_le16': return H(struct.pack('<4I',*q))
def u64v(x,H):
    lo22=x&((1<<22)-1); m21=(x>>22)&((1<<21)-1); h21=(x>>43)&((1<<21)-1)
    f=[lo22,m21,h21,0]
    out={'le8':struct.pack('<Q',x),'le8pad32':struct.pack('<Q',x)+b'\0'*24,'le4':struct.pack('<I',x&0xffffffff),'num':num_to_bytes(x),'be8':struct.pack('>Q',x)}
    for e in ('le16','numcat','chain_rev','chain_fwd','chain_rev_le4','sha_le16'): out['felt_'+e]=enc_felt(f,e,H)
    out['felt_m31']=enc_felt([x&0x7fffffff,0,0,0],'le16',H)
    return out
found=[]
for Hn,H in (('sha',sha1),('sha2',sha2)):
 for order in ('dr','rd'):
  def mix(dg,data): return H(dg+data) if order=='dr' else H(data+dg)
  for fenc in ('le16','be16','numcat','numcat_rev','chain_rev','chain_fwd','chain_rev_le4','sha_le16'):
    for fmode in ('all','perqm31'):
      if fmode=='all' and fenc.startswith(('chain','sha_')): continue
      def mixf(dg,felts):
          if fmode=='all': return mix(dg,b''.join(enc_felt(q,fenc,H) for q in felts))
          for q in felts: dg=mix(dg,enc_felt(q,fenc,H))
          return dg
      for uk in u64v(1,H):
        for drawmod in ('none','sha_d','sha_d2'):
            def draw(dg,n=1):
                if drawmod=='none': return dg
                for _ in range(n if drawmod=='sha_d2' else 1): dg=H(dg)
                return dg
            dg=b'\0'*32
            dg=mix(dg,d['commitments'][0]); dg=mix(dg,u64v(ls,H)[uk]); dg=mix(dg,d['commitments'][1]); dg=draw(dg,2)
            dg=mixf(dg,[d['total_sum']]); dg=mix(dg,d['commitments'][2]); dg=draw(dg); dg=mix(dg,d['commitments'][3]); dg=draw(dg)
            dg=mixf(dg,flat); dg=draw(dg)
            dg=mix(dg,d['first']['com']); dg=draw(dg)
            for l in d['inner']: dg=mix(dg,l['com']); dg=draw(dg)
            dg=mixf(dg,list(d['last']))
            for un,uv in u64v(nonce,H).items():
                dn=mix(dg,uv)
                checks={'le_tz':int.from_bytes(dn[:16],'little')&((1<<28)-1)==0,'be_lz':int.from_bytes(dn[:4],'big')>>4==0,'le_last':int.from_bytes(dn[28:],'little')>>4==0,'be_tz':int.from_bytes(dn[28:],'big')&((1<<28)-1)==0}
                for k,v in checks.items():
                    if v: found.append((Hn,order,fenc,fmode,uk,drawmod,un,k)); print('FOUND',found[-1])
print('done',len(found))
