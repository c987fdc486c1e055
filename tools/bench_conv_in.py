import sys, torch
sys.path.insert(0, "/root/repo")
import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib
B, H = 64, 64
lat = torch.randn(B, 4, H, H, device="cuda"); w = (torch.randn(320, 36, device="cuda") / 6).bfloat16(); bias = torch.randn(320, device="cuda")
out = torch.empty(B, H, H, 320, dtype=torch.bfloat16, device="cuda")
f = lambda: _lib.check(sda.lib().sdn_conv_in_bf16(lat.data_ptr(), w.data_ptr(), bias.data_ptr(), B, 4, H, H, 320, out.data_ptr(), _lib.stream_ptr()), "c")
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"conv_in B={B}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
