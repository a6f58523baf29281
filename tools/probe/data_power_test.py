"""Tuning probe: is the heads conv power-limited?  Same launch on random vs all-zero operands (MFMA toggle activity)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from centerpose_amd import hip
hip.set_default_precision("f16x3")
for name, mk in (("randn", lambda *s: torch.randn(*s, device="cuda")), ("zeros", lambda *s: torch.zeros(*s, device="cuda")),
                 ("randn", lambda *s: torch.randn(*s, device="cuda"))):
    x = mk(32, 128, 128, 64); w = mk(256, 64, 3, 3) / 24
    for _ in range(3): y = hip.conv2d_nhwc(x, w, None, None, None, 1, 1, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): y = hip.conv2d_nhwc(x, w, None, None, None, 1, 1, 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("%s operands: %.3f ms  (%.0f TFLOP/s algorithmic incl. weight pack)" % (name, dt * 1e3, 2 * 524288 * 256 * 576 / dt / 1e12))
