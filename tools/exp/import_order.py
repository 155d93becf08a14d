import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1]
if order == "crt_first":
    import cudaraytracer_amd as crt
    import torch
else:
    import torch
    import cudaraytracer_amd as crt
torch.cuda.set_device(0)
x = torch.zeros(4, device="cuda")
try:
    c = crt.Context(0); print(order, "context ok"); c.close()
except Exception as e:
    print(order, "FAILED", e)
os.system(f"grep -E 'libamdhip64|libhsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
