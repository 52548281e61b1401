import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops
F, L, H = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 12
qkv = torch.randn(F * L, 3 * H * 64, device="cuda").bfloat16()
for _ in range(3):
    ops.attention_bf16(qkv, F, L, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention_bf16(qkv, F, L, H)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attention_bf16 F={F} L={L} H={H}: {ms:.4f} ms  {4.0 * F * H * L * L * 64 / ms / 1e9:.1f} TFLOP/s (algorithmic)")
