"""Time of the reference-stream z draw on the device (`ssn_mt19937_random_sample_f32`): host-visible latency of the call (it
returns when the state after the draw is known) and device time of the whole draw, per shape."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from tc_gan_amd.networks.ssn import device_rand      # noqa: E402

for shape in [(1024, 200, 200), (128, 202, 202), (1024, 204, 204)]:
    rs = np.random.RandomState(0)
    for _ in range(3):
        device_rand(rs, shape, torch.float32)
    torch.cuda.synchronize()
    lat, dev = [], []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        device_rand(rs, shape, torch.float32)
        lat.append(time.perf_counter() - t0)
        b.record()
        torch.cuda.synchronize()
        dev.append(a.elapsed_time(b))
    t0 = time.perf_counter()
    rs.rand(*shape)
    host = time.perf_counter() - t0
    print('%-18s call returns after %.3f ms (min %.3f), device %.3f ms (min %.3f); host rng.rand %.1f ms'
          % (shape, 1e3 * np.median(lat), 1e3 * min(lat), np.median(dev), min(dev), 1e3 * host), flush=True)
