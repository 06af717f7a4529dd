import os, sys
R="/root/repo"; sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, orc
from kvazzup_amd.codec import Decoder, split_nals
kw=dict(slices=3, wpp=1, cip=1, pcm=15, lf_across=1, intra_in_p=40, long_term=1, num_refs=3, tmvp=1)
w,h,period=208,144,8
g=orc.OracleGen(w,h,seed=93,intra_period=period,density=30,sao=1,all_part_modes=1,**kw)
aus=[g.picture() for _ in range(3*period)]; g.close()
rng=np.random.default_rng(777)
gd=Decoder(threads=4, frame_threads=True)
errors=0
for trial in range(120):
    t=int(rng.integers(0,2*period))
    nals=[bytearray(n) for n in split_nals(aus[t])]
    kind=trial%6; i=int(rng.integers(0,len(nals)))
    if kind==0:
        for _ in range(int(rng.integers(1,6))):
            nals[i][min(5+int(rng.integers(0,max(len(nals[i])-5,1))),len(nals[i])-1)]^=1<<int(rng.integers(0,8))
    elif kind==1: nals[i]=nals[i][:max(6,int(rng.integers(6,len(nals[i])+1)))]
    elif kind==2:
        p=int(rng.integers(6,max(7,len(nals[i])))); nals[i][p:p+16]=bytes(rng.integers(0,256,16,dtype=np.uint8))
    elif kind==3 and len(nals)>1: del nals[i]
    elif kind==4 and len(nals)>1:
        j=int(rng.integers(0,len(nals))); nals[i],nals[j]=nals[j],nals[i]
    else: nals.insert(i,bytearray(nals[i]))
    for nal in nals:
        try: gd.decode_nal(bytes(nal),t)
        except RuntimeError as e: errors+=1
print("errors",errors, flush=True)
for t in range(2*period,3*period):
    for n in split_nals(aus[t]):
        try:
            f=gd.decode_nal(bytes(n),t); print(t,(n[4]>>1)&63,"->", None if f is None else f["pts"])
        except RuntimeError as e: print(t,(n[4]>>1)&63,"ERR",e, gd.lib.kvzx_decoder_last_error(gd.h))
eos=bytes([0,0,0,1,72,1])
for k in range(30):
    try:
        f=gd.decode_nal(eos); print("eos",k, None if f is None else f["pts"])
    except RuntimeError as e: print("eos",k,"ERR",e)
