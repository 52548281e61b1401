"""Lists the acx_gemm launches of one XD-shaped (configs[4]) head forward in bf16: shape, epilogue, split-K workspace."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops, _lib as L, init_weights as IW
from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts

seen = collections.Counter()
real = ops.gemm
def spy(a, w, **kw):
    M, K = a.shape[0], w.shape[1]
    seen[(M, w.shape[0], K, kw.get("act", 0), kw.get("residual") is not None, kw.get("amap", 0), kw.get("prec", 0),
          str(a.dtype)[6:], str(kw["out"].dtype)[6:] if kw.get("out") is not None else "-")] += 1
    return real(a, w, **kw)
ops.gemm = spy
hc = IW.XD_HEAD
toks = torch.tensor(lookup_prompts(key="xd")["tokenized_prompts"], dtype=torch.int32)
net = AnomalyCLIP(arch="ViT-B/16", labels_key="xd", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads, dim_heads=None,
                  num_segments=32, seg_length=16, concat_features=False, normal_id=hc.normal_id, stride=1, load_from_features=True,
                  select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=hc.ncrops, num_topk=3, num_bottomk=3,
                  precision="bf16", vit_chunk=160)
net.load_state_dict(IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0), strict=True)
net = net.cuda().eval()
feats = torch.randn(1, hc.ncrops, 512 * 16, 512, device="cuda") * 0.3
with torch.no_grad():
    net(feats, None, torch.zeros(512, device="cuda"), 16, True)
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, "x  M,N,K,act,res,amap,prec,a,c =", k)
