"""Determinism stress: the same launch many times, every output compared bit for bit with the first (finds races)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccst_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
shapes = [(3,192,160,256,256,False,False),(3,96,80,512,256,False,False),(3,384,320,128,128,True,False),(3,96,80,256,512,False,False),(6,128,128,256,256,False,False),
          (3,384,320,128,128,False,True)]
for (N,H,W,Cin,Cout,pool,ups) in shapes:
    Hs,Ws = (H//2,W//2) if ups else (H,W)
    x = torch.rand(N,Hs,Ws,Cin,generator=g).to(dev)
    w = (torch.randn(Cout,Cin,3,3,generator=g)*(2.0/(9*Cin))**0.5).to(dev)
    b = (torch.randn(Cout,generator=g)*0.05).to(dev)
    pc = ops.pack_conv_weight(w,b,wino=4)
    flags = 1|8|(2 if pool else 0)|(4 if ups else 0)
    xm = ops.absmax(x)
    for name, fn in (("f23", lambda: ops.conv3x3_f23(x,pc,flags,x_absmax=xm)), ("split", lambda: ops.conv3x3_halo_split(x,pc,flags,x_absmax=xm))):
        ref = fn().clone()
        bad, worst = 0, 0.0
        for i in range(reps):
            y = fn()
            if not torch.equal(y, ref):
                bad += 1
                worst = max(worst, float((y-ref).abs().max()))
        print(name, (N,H,W,Cin,Cout,pool,ups), "mismatching launches %d / %d, worst |diff| %.3g (max |y| %.3g)" % (bad, reps, worst, float(ref.abs().max())))
