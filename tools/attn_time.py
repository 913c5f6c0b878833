import sys, os, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/soccdpt_amd') else '.')
from soccdpt_amd.lib import op_vit_attention, PREC_F16
B,N,heads=4,577,12
qkv=(torch.randn(B*N,3*heads*64)*1.5).half().cuda(); out=torch.empty(B*N,heads*64,dtype=torch.float16,device='cuda')
for _ in range(5): op_vit_attention(qkv,out,B,N,heads,PREC_F16)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): op_vit_attention(qkv,out,B,N,heads,PREC_F16)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('SOCCDPT_LIB_PATH','new'), 'vit attention us', e0.elapsed_time(e1)*1e3/50)
