import sys, os, json, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_cyclegan_gpu as T
fx = np.load(os.path.join(T.G, "cyclegan_step_link_3d64_ngf64.npz")); ref = json.loads(str(fx["loss_dicts"]))
dev = torch.device("cuda")
for opt in (0, -1, 0):
    from raygun_amd import ops
    with ops.option("RG_GEMM5", opt):
        s = T._system_ngf64(dev)
        out = []
        for it in range(2):
            mine, outs, gs = T._step_ngf64(s, dev, it, (1, 1, 64, 64, 64))
            out.append(mine)
    print("GEMM5", opt, {k: (round(out[1][k], 4), round(ref[1][k], 4), round(ref[0][k], 4)) for k in ("l1_loss/cycled_B", "l1_loss/cycled_A", "gan_loss/fake_B", "Total_Loss/G1")}, flush=True)
