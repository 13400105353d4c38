import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+'/oracle', ROOT+'/tiny-cuda-nn_amd'): sys.path.insert(0,p)
import numpy as np, torch, oracle, tinycudann as tcnn
cfg={"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5}
n=2048
enc=tcnn.Encoding(2,cfg); ref=oracle.create_encoding(2,cfg,alignment=0)
params_h=oracle.half_bits(oracle.Pcg32(7).uniform_strided(ref.n_params,-1.0,1.0))
x=oracle.Pcg32(42).uniform_strided(n*2).reshape(n,2)
want,ctx=ref.forward(x,params_h,want_indices=True)
with torch.no_grad():
    enc.params.copy_(torch.from_numpy(params_h.view(np.float16).astype(np.float32)).cuda())
    got=enc(torch.from_numpy(x).cuda())
g=got.cpu().numpy().view(np.uint16)
bad=np.argwhere(g!=want)
print("mismatches",len(bad),"of",g.size)
print("by column:",np.bincount(bad[:,1],minlength=32))
for (i,j) in bad[:10]:
    print(i,j,g[i,j],want[i,j], np.float16(0).__class__, g[i,j].view(np.float16) if False else None, x[i])
gf=g.view(np.float16).astype(np.float32); wf=want.view(np.float16).astype(np.float32)
print("max abs diff",np.abs(gf-wf).max())
