"""Device time of z + W at the C3 shape: the draw with W formed in its own launch (`device_rand_weights`) against the draw
followed by `generate_weight_batch` (the two launches it replaces)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import ssn_numpy as on                      # parameters only
from tc_gan_amd.networks.ssn import device_rand, device_rand_weights    # noqa: E402
from tc_gan_amd.weight_gen import generate_weight_batch                   # noqa: E402

jds = on.new_JDS()
for B, N in ((1024, 100), (128, 101)):
    rs = np.random.RandomState(0)
    def fused(keep):
        return device_rand_weights(rs, B, N, jds['J'], jds['D'], jds['S'], keep_z=keep).W
    def two():
        z = device_rand(rs, (B, 2 * N, 2 * N), torch.float32)
        return generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    for name, fn in (('fused, z dropped', lambda: fused(False)), ('fused, z kept', lambda: fused(True)), ('draw + build_w', two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print('B %4d N %3d  %-18s %.3f ms (min %.3f)' % (B, N, name, np.median(ts), min(ts)), flush=True)
