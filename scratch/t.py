import torch, ctypes, os, time
t0=time.time()
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libt.so"))
print("cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
p = torch.cuda.get_device_properties(0); print(p)
a = (torch.arange(1024, device="cuda") % 7).to(torch.bfloat16)
out = torch.zeros(64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
rc = lib.rg_test(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))
torch.cuda.synchronize()
print("rc", rc, out[:8].tolist(), time.time()-t0)
with open("/proc/self/maps") as f:
    print([l.split()[-1] for l in f if "amdhip" in l][:3])
import subprocess
print(subprocess.run("nproc; free -g | head -2; lscpu | grep 'Model name'", shell=True, capture_output=True, text=True).stdout)
