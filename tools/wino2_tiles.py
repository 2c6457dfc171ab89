import sys, os
sys.path.insert(0, '/root/repo')
import torch
from frtm_vos_amd import ops
dev='cuda:0'
g=torch.Generator().manual_seed(0)
for B,cin,cout,h,w in [(16,64,64,120,214),(16,64,64,60,107),(16,65,64,120,214),(16,64,32,240,427),(8,64,64,120,214),(16,64,64,30,54)]:
    x=torch.relu(torch.randn(B,cin,h,w,generator=g)).to(dev)
    wt=(torch.randn(cout,cin,3,3,generator=g)/(9*cin)**0.5).to(dev)
    wW,_,lay=ops.pack_weights(wt,wino=True)
    fl=2.0*B*h*w*cin*cout*9
    for tile in (0,1,2,3):
        try:
            out=ops.conv2d(x,wW,cout,3,1,1,relu=True,tile=tile,splitk=1,w_layout=2)
        except RuntimeError as e:
            print(B,cin,cout,h,w,'tile',tile,str(e)[:80]); continue
        for _ in range(3): ops.conv2d(x,wW,cout,3,1,1,relu=True,tile=tile,splitk=1,w_layout=2,out=out)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d(x,wW,cout,3,1,1,relu=True,tile=tile,splitk=1,w_layout=2,out=out)
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/20*1e3
        print('%d x %d->%d @%dx%d tile %d: %7.1f us %6.1f TF'%(B,cin,cout,h,w,tile,us,fl/us/1e6),flush=True)
