import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from pixelsplat_amd.decoder import render_cuda
from pixelsplat_amd.synthetic import make_workload
from oracle import raster_ref as R
from tests.cases import oracle_view_inputs
dev=torch.device('cuda')
hw=(256,256); b,v=7,4
ctx,tgt,g,target=make_workload(b,hw,v_ctx=2,v_tgt=v,seed=0)
V=b*v
args=[tgt.extrinsics.reshape(V,4,4).to(dev), tgt.intrinsics.reshape(V,3,3).to(dev), tgt.near.reshape(V).to(dev), tgt.far.reshape(V).to(dev), hw, torch.zeros((V,3),device=dev), g.means.to(dev), g.covariances.to(dev), g.harmonics.to(dev), g.opacities.to(dev)]
img1,aux=render_cuda(*args, views_per_scene=v, return_aux=True)
img2=render_cuda(*args, views_per_scene=v)
print('deterministic:', torch.equal(img1,img2))
vps=aux['view_params'].cpu().numpy(); im=img1.cpu().numpy()
R.lib()
for vi in range(16):
    inp=oracle_view_inputs(g,tgt,vi//v,vi%v,view_params=vps[vi])
    st=R.forward(H=256,W=256,**inp)
    d=np.abs(im[vi]-st.image)
    k=np.unravel_index(d.argmax(), d.shape)
    print(vi, 'linf %.3e'%d.max(), 'at', k, 'gpu %.6f oracle %.6f'%(im[vi][k], st.image[k]), ' n>1e-4:', int((d>1e-4).sum()))
