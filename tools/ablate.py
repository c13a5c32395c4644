import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C
CL = torch.channels_last
dev = torch.device("cuda:0")
N, Cin, H, W, Cout, k = 2, 1024, 64, 128, 1024, 3
x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
y = _C.conv_forward(x, w, pad=1)
for _ in range(3): _C.conv_forward(x, w, pad=1, out=y)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): _C.conv_forward(x, w, pad=1, out=y)
e.record(); torch.cuda.synchronize()
print("ablate=%s mode=%d: %.3f ms" % (os.environ.get("DADET_ABLATE", "0"), _C.get_gemm_mode(), s.elapsed_time(e) / 10))
