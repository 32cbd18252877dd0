import sys, time
sys.path.insert(0, '/root/repo')
import torch, medfusion_amd as M
from medfusion_amd import blocks as B
from bench import build_pipeline
dev = torch.device("cuda:0")
pipe = build_pipeline(dev, None)
def run(flag):
    B.SIDE_STREAM_RESIDUAL = flag
    pipe.sample(16, (8, 32, 32), steps=10, use_ddim=True, noise=M.PhiloxDeviceNoise(1)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.sample(16, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(1))
    torch.cuda.synchronize()
    return 16 / (time.perf_counter() - t0), out
for rep in range(2):
    for flag in (False, True):
        ips, out = run(flag)
        print(f"side_stream_residual={flag}: {ips:.3f} img/s  checksum {float(out.double().abs().sum()):.6e}")
