import ctypes, os, sys
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
import veloci_amd
L = veloci_amd.lib()
b = L.vq_index_builder_new(10, 0, 10)
h = ctypes.c_void_p()
rc = L.vq_index_build(ctypes.c_void_p(b), 0, ctypes.byref(h))
print(order, "rc", rc, L.vq_last_error())
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
