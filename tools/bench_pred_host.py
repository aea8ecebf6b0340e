"""PCIe-inclusive block-wise predict: numpy source on the host, numpy destinations (the reference's flow).  tools."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import predict as P
from raygun_amd.torch.models import MTLSDModel
dev = torch.device("cuda")
torch.manual_seed(1234)
with torch.device(dev):
    model = MTLSDModel(num_affs=20).eval()
vol = np.random.default_rng(0).integers(0, 256, (404,) * 3, dtype=np.uint8)
for dts, name in ((None, "float32"), ([np.uint8, np.uint8], "uint8")):
    P.predict_volume(model, vol[:196, :196, :196], input_shape=[196] * 3, output_shape=[104] * 3, out_dtypes=dts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d, n = P.predict_volume(model, vol, input_shape=[196] * 3, output_shape=[104] * 3, out_dtypes=dts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"host-resident, {name} destinations: {n} blocks in {dt*1e3:.0f} ms = {n/dt:.1f} blocks/s = {n*104**3/dt/1e6:.1f} Mvox/s", flush=True)
