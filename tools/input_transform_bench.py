"""Device time of the input transform (uint8 1080x1920 frames -> float32 CHW network input) for both network sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_input_transform_u8
B = 8
frames = torch.randint(0, 256, (B, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
for net in (256, 384):
    for _ in range(20):
        op_input_transform_u8(frames, net, net)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        op_input_transform_u8(frames, net, net)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    touched = B * net * net * (16 * 3 + 12)            # 4x4x3 source bytes + 3 output floats per output pixel
    print(f"B={B} 1080x1920 -> {net}x{net}: {us:.1f} us per batch ({us / B:.2f} us per frame), {touched / us / 1e3:.1f} GB/s of touched bytes, "
          f"{B * 1080 * 1920 * 3 / us / 1e3:.0f} GB/s of frame bytes")
