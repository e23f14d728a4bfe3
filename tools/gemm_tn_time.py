import sys, torch, time
sys.path.insert(0,'/root/repo')
from pixelsplat_amd.epipolar import gemm_tn
dev=torch.device('cuda'); R=57344
for m,n in [(592,128),(128,592),(80,128),(128,80)]:
    a=torch.randn(R,m,device=dev); b=torch.randn(R,n,device=dev)
    for _ in range(3): gemm_tn(a,b)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): gemm_tn(a,b)
    torch.cuda.synchronize(); print(m,n,'%.1f us'%((time.perf_counter()-t0)/20*1e6))
