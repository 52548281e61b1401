"""Development probe: phase timeline of attn_p3_kernel's workgroup 0 (needs a -DAP3_TRACE=1 build through ACX_LIB_PATH):
per wave and item, s_memrealtime (100 MHz) stamps at  0 item start | 1 after B1 | 2 V share issued | 3 Q K^T done | 4 after B2 |
5 softmax done | 6 P V done | 7 stores issued   (stager wave 7: 0 start | 1 K landed | 2 after B1 | 3 V share landed | 4 after B2 | 5 K(next) issued)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

tr = torch.zeros(16 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["ACX_TRACE_PTR"] = str(tr.data_ptr())
from anomalyclip_amd import ops

B, L_, H = 512, 197, 12
qkv = torch.randn(B * L_, 3 * H * 64, device="cuda")
q3 = ops.split_bf16x3(qkv, panel=True)
for _ in range(3):
    tr.zero_()
    ops.attention_p3(q3, B, L_, H)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(16, 8, 8).astype("float64") / 100.0        # us
t0 = t[2, :7, 0].min()
for it in range(2, 8):
    print(f"item {it}:")
    for w in range(8):
        row = t[it, w] - t0
        if w < 7:
            print(f"  wave {w}: start {row[0]:7.2f} | B1 +{row[1] - row[0]:5.2f} | issueV +{row[2] - row[1]:5.2f} | QK^T +{row[3] - row[2]:5.2f} | B2 +{row[4] - row[3]:5.2f} | "
                  f"softmax +{row[5] - row[4]:5.2f} | PV +{row[6] - row[5]:5.2f} | epilogue +{row[7] - row[6]:5.2f} | total {row[7] - row[0]:6.2f}")
        else:
            print(f"  stager: start {row[0]:7.2f} | K landed +{row[1] - row[0]:5.2f} | B1 +{row[2] - row[1]:5.2f} | V share +{row[3] - row[2]:5.2f} | B2 +{row[4] - row[3]:5.2f} | "
                  f"issue K(next) +{row[5] - row[4]:5.2f}")
