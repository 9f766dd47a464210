import sys, numpy as np
sys.path.insert(0,'/root/repo')
import importlib
synth = importlib.import_module('orb-slam2-dualcam_amd.synth')
img,_ = synth.frame_pair(640,480,0,0)
v = img.astype(np.int32)
th=int(sys.argv[1]) if len(sys.argv)>1 else 7
c = v[3:-3,3:-3]
r0=v[6:,3:-3]; r8=v[:-6,3:-3]; r4=v[3:-3,6:]; r12=v[3:-3,:-6]
B=[r>c+th for r in (r0,r4,r8,r12)]; Dk=[r<c-th for r in (r0,r4,r8,r12)]
nb=sum(b.astype(int) for b in B); nd=sum(d.astype(int) for d in Dk)
any2=(nb>=2)|(nd>=2)
adj=np.zeros_like(any2)
for k in range(4):
    adj|= (B[k]&B[(k+1)%4]) | (Dk[k]&Dk[(k+1)%4])
print("any2 frac",any2.mean(),"adjacent frac",adj.mean())
# 8-point test: 4 consecutive of the 8 even ring points
offs=[(3,0),(2,2),(0,3),(-2,2),(-3,0),(-2,-2),(0,-3),(2,-2)]
H,W=v.shape
R=[v[3+dy:H-3+dy,3+dx:W-3+dx] for dy,dx in offs]
B8=[r>c+th for r in R]; D8=[r<c-th for r in R]
p8=np.zeros_like(any2)
for k in range(8):
    p8|= (B8[k]&B8[(k+1)%8]&B8[(k+2)%8]&B8[(k+3)%8]) | (D8[k]&D8[(k+1)%8]&D8[(k+2)%8]&D8[(k+3)%8])
print("8pt frac",p8.mean())
# true corners
ring=[(3,0),(3,1),(2,2),(1,3),(0,3),(-1,3),(-2,2),(-3,1),(-3,0),(-3,-1),(-2,-2),(-1,-3),(0,-3),(1,-3),(2,-2),(3,-1)]
R16=[v[3+dy:H-3+dy,3+dx:W-3+dx] for dy,dx in ring]
Bf=np.stack([r>c+th for r in R16]); Df=np.stack([r<c-th for r in R16])
def arc9(M):
    out=np.zeros(M.shape[1:],bool)
    for k in range(16):
        a=np.ones(M.shape[1:],bool)
        for j in range(9): a&=M[(k+j)%16]
        out|=a
    return out
cor=arc9(Bf)|arc9(Df)
print("true corner frac",cor.mean())
