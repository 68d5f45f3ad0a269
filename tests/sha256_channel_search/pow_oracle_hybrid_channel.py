import sys, hashlib, itertools, struct
sys.path.insert(0,'/root/repo')
import numpy as np
from tests import oracle_binding as ob
from tests.conftest import read_proof
proof = read_proof('hybrid_hash.bin')
w = np.frombuffer(proof, dtype=np.uint32)
lay = ob.proof_layout(proof)
print({k:v for k,v in lay.items() if k not in('prefixes','fri_commitments')})
lp,lq = int(w[0]),int(w[1])
sums = w[2:10].copy()
commits=[w[17+8*t:25+8*t].copy() for t in range(4)]
# sampled values: flattened 142 QM31 in order
# use the known sample table: start at word 49+2; trees with prefixes
pos=49+2; samples=[]
cols=[50,60,16,8]
for t in range(4):
    assert w[pos]==cols[t]; pos+=2
    for c in range(cols[t]):
        ns=int(w[pos]); pos+=2
        for s in range(ns):
            samples.append(w[pos:pos+4].copy()); pos+=4
samples=np.concatenate(samples); assert samples.size==142*4 and pos==895
nonce_word=lay['nonce_word']; nonce=int(w[nonce_word])|(int(w[nonce_word+1])<<32)
fri=lay['fri_commitments']
# last layer coeffs
pfx=[p for p in lay['prefixes'] if p[2]=='last_layer_poly'][0]
last=w[pfx[0]+2:pfx[0]+2+4*pfx[1]].copy()
print('nonce',nonce,'n_fri',len(fri),'last',pfx[1])
def sha(b): return hashlib.sha256(b).digest()
def pos_hash(words):  # hash_m31_columns_get_rate
    return ob.hash_node(None, np.asarray(words,dtype=np.uint32).reshape(1,-1))[0]
samples_h = pos_hash(samples).astype('<u4').tobytes()
last_h = pos_hash(last).astype('<u4').tobytes()
sums_b = sums.astype('<u4').tobytes()
sums_h = pos_hash(sums).astype('<u4').tobytes()
def le32(x): return struct.pack('<I',x)
def u64_variants(x):
    lo22=x&((1<<22)-1); m21=(x>>22)&((1<<21)-1); h21=(x>>43)&((1<<21)-1)
    f=struct.pack('<4I',lo22,m21,h21,0)
    return {'le8':struct.pack('<Q',x),'le8pad32':struct.pack('<Q',x)+b'\0'*24,'le4':le32(x&0xffffffff),'felt16':f,'felt32':f+b'\0'*16,
            'be8':struct.pack('>Q',x),'poshash':pos_hash(np.frombuffer(f,dtype='<u4')).astype('<u4').tobytes(),
            'felt1':struct.pack('<4I',x&0x7fffffff,0,0,0) if x<2**31 else f}
found=[]
roots_b=[c.astype('<u4').tobytes() for c in commits]
fri_b=[c.astype('<u4').tobytes() for c in fri]
for order in ('dr','rd'):
  def mix(d,data): return sha(d+data) if order=='dr' else sha(data+d)
  for small in ('raw','poshash'):
    for large in ('poshash','raw'):
      for u64k in ['le8','le8pad32','le4','felt16','felt32','be8','poshash','felt1']:
        for drawmod in ('none','sha_d','sha_d_each'):
          for init in ('zero',):
            d=b'\0'*32
            def draw(d,n=1):
                if drawmod=='none': return d
                for _ in range(n if drawmod=='sha_d_each' else 1): d=sha(d)
                return d
            d=mix(d,roots_b[0])
            d=mix(d,u64_variants(lp)[u64k]); d=mix(d,u64_variants(lq)[u64k])
            d=mix(d,roots_b[1])
            d=draw(d,2)
            d=mix(d,sums_b if small=='raw' else sums_h)
            d=mix(d,roots_b[2]); d=draw(d); d=mix(d,roots_b[3]); d=draw(d)
            d=mix(d,samples_h if large=='poshash' else samples.astype('<u4').tobytes()); d=draw(d)
            for fb in fri_b:
                d=mix(d,fb); d=draw(d)
            d=mix(d,last_h if large=='poshash' else last.astype('<u4').tobytes())
            for u64n in ['le8','le8pad32','felt16','felt32','be8','poshash']:
                dn=mix(d,u64_variants(nonce)[u64n])
                for nm,val in (('le_tz',int.from_bytes(dn[:16],'little')),('be_lz',None),('le_last',int.from_bytes(dn[16:],'little'))):
                    if nm=='be_lz':
                        ok = int.from_bytes(dn[:4],'big')>>4==0
                    else:
                        ok = val & ((1<<28)-1)==0
                    if ok:
                        found.append((order,small,large,u64k,drawmod,u64n,nm)); print('FOUND',found[-1])
print('done',len(found))
