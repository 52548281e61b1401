import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops
F, L, H = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 12
qkv = torch.randn(F * L, 3 * H * 64, device="cuda")
for _ in range(3):
    ops.attention(qkv, F, L, H, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(qkv, F, L, H, False)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
Lp = (L + 15) // 16 * 16
print(f"attention F={F} L={L} H={H}: {ms:.4f} ms  {4.0 * F * H * L * L * 64 / ms / 1e9:.1f} TFLOP/s (algorithmic)  "
      f"{4.0 * F * H * Lp * Lp * 64 / ms / 1e9:.1f} TFLOP/s (16-padded MFMA work)")
