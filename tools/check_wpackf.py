import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from vae_eeg_xai_amd import _lib
from vae_eeg_xai_amd.net import Net
from oracle import vae_oracle as vo
# transposition kernel check
W=torch.randn(9,100,36,device='cuda'); dst=torch.empty_like(W).view(-1)
_lib.check(_lib.lib.veex_transpose_weights_f32(_lib.ptr(W),_lib.ptr(dst),9,100,36,_lib.current_stream()))
assert torch.equal(dst.view(9,36,100), W.transpose(1,2).contiguous())
L,B=28,96
p=vo.init_params(L,'vae',seed=3,jitter=0.1); x=vo.synth_topomaps(B,5); eps=np.random.default_rng(1).standard_normal((B,L)).astype(np.float32)
res=[]
for flag in ('0','1'):
    _lib.setenv('VEEX_WPACKF',flag)
    net=Net(L,'vae'); net.set_weights_dict(p); net.train_fwd_bwd(x,eps); torch.cuda.synchronize(); res.append(net.grads.clone())
    a=net.attribute(x,[1,2]); res.append(a.clone())
print('bit identical:', torch.equal(res[0],res[2]), torch.equal(res[1],res[3]))
