"""Timing of the ViT attention kernels at 512 frames x 12 heads x 197 tokens (development probe)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops
from bench import _event_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, L_, H = 512, 197, 12
qkv = torch.randn(B * L_, 3 * H * 64, generator=g, device=dev)
q3 = ops.split_bf16x3(qkv, panel=True)
fl = 4.0 * B * H * L_ * L_ * 64
t32 = _event_time(lambda: ops.attention(qkv, B, L_, H, False), 8)
tp3 = _event_time(lambda: ops.attention_p3(q3, B, L_, H), 8)
print(f"f32 MFMA attention {t32 * 1e3:.3f} ms {fl / t32 / 1e12:.1f} TFLOP/s | planes attention {tp3 * 1e3:.3f} ms "
      f"{fl / tp3 / 1e12:.1f} TF-equiv ({6 * fl / tp3 / 1e12:.0f} bf16 TF)")
