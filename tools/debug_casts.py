import sys, os, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'fp8-mps-metal_amd'), os.path.join(R,'oracle')]
import fp8_mi355x_native as n, fp8_oracle as o
dev=torch.device('cuda:0')
rng=np.random.default_rng(7)
x=torch.from_numpy((rng.standard_normal(64)*16).astype(np.float32)).to(torch.float16)
got=n.fp8_encode(x.to(dev)).cpu().numpy(); exp=o.encode(x.float().numpy())
print('f16 encode mism idx', np.nonzero(got!=exp)[0][:32])
print(got[:16], exp[:16])
raw=rng.integers(0,256,size=1001,dtype=np.uint8)
t=torch.from_numpy(raw).to(dev)[1:]
got=n.fp8_dequantize(t,None).cpu().view(torch.int16).numpy().view(np.uint16); exp=o.dequantize_f16(raw[1:]).view(np.uint16)
bad=np.nonzero(got!=exp)[0]; print('dequant unaligned mism', bad[:20], [hex(raw[1:][i]) for i in bad[:10]], [hex(got[i]) for i in bad[:10]], [hex(exp[i]) for i in bad[:10]])
for s in (0.5,0.0137,300.0):
    got=n.fp8_dequantize(t,torch.tensor([s])).cpu().view(torch.int16).numpy().view(np.uint16); exp=o.dequantize_f16(raw[1:],s).view(np.uint16)
    bad=np.nonzero(got!=exp)[0]; print('scale',s,'mism',bad[:10],[hex(raw[1:][i]) for i in bad[:6]], [hex(got[i]) for i in bad[:6]], [hex(exp[i]) for i in bad[:6]])
