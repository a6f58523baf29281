import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from centerpose_amd import hip
hip.set_default_precision("f16x3")
x = torch.randn(32, 128, 128, 64, device="cuda"); w = torch.randn(256, 64, 3, 3, device="cuda") / 24
for _ in range(5): y = hip.conv2d_nhwc(x, w, None, None, None, 1, 1, 1)
torch.cuda.synchronize()
L = hip.lib(); buf = (ctypes.c_ulonglong * 2)()
L.cp_debug_read_clk(buf)
print("block lifetime: %d shader ticks, %d x10ns -> %.0f MHz, %.1f us" % (buf[0], buf[1], buf[0] / (buf[1] / 100.0), buf[1] / 100.0))
