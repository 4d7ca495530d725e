import os, sys, numpy as np, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests","golden"))
import netgen
import utils.conv2d_func as cf
from cnns_slfp_quantization_amd import layer_specs, _lib
dev=torch.device("cuda:0")
gold=np.load(os.path.join(ROOT,"tests/golden/net224_golden.npz"))
rows=[r for r in layer_specs.nets()["mobilenetv1_imagenet224"]["layers"] if r["kind"]=="conv"]
scales=[(r["Ka"],r["Kw"]) for r in rows]
x=netgen.net_input224(64).to(dev).contiguous(memory_format=torch.channels_last)
G=gold["logits_q8"]
for passes in (3,1):
    cf.options.mfma_passes=passes
    m=netgen.load_bn_stats_(netgen.fill_parameters(netgen.build_mobilenetv1_imagenet(cf.conv2d_Q,8,scales)), gold).to(dev).eval().to(memory_format=torch.channels_last)
    with torch.no_grad():
        L=torch.cat([m(x[i:i+16]) for i in range(0,64,16)]).cpu().numpy()
    d=L-G
    print("passes",passes,"max|d|/max|ref|",np.abs(d).max()/np.abs(G).max(),"l2",np.linalg.norm(d)/np.linalg.norm(G))
    c=G-G.mean(0); dc=d-d.mean(0)
    print("  centered: ||dc||/||c||",np.linalg.norm(dc)/np.linalg.norm(c))
    print("  top1 agree",(L.argmax(1)==G.argmax(1)).mean(),"top5 set agree",np.mean([set(np.argsort(-a)[:5])==set(np.argsort(-b)[:5]) for a,b in zip(L,G)]))
    # rank of per-image deviations: correlation of centered logits per image
    cc=[np.corrcoef((L-L.mean(0))[i],c[i])[0,1] for i in range(64)]
    print("  per-image corr of centered logits: min %.4f mean %.4f"%(min(cc),np.mean(cc)))
