import sys; R="/root/repo"; sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, orc
from kvazzup_amd.pipeline import Pipeline
every=int(sys.argv[1]) if len(sys.argv)>1 else 7
w,h,frames,period=416,240,40,16
pl=Pipeline(w,h,settings={"video/QP":30,"video/Intra":period,"uvgx/wireLossEvery":every},custom=(("me-range",16),))
clip=[np.ascontiguousarray(orc.synth_frame(0,0x5EED0008,w,h,t)) for t in range(frames)]
for f in clip: pl.push_host_paced(f, max_backlog=4)
print("wait", pl.wait(frames, 8000)); print(pl.stats())
pts=[]
while True:
    d=pl.pop_decoded()
    if d is None: break
    pts.append(d["pts"])
print(len(pts), pts)
pl.close()
