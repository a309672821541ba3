import sys; sys.path.insert(0, "/root/repo")
import torch, ctypes as C
from gwen_amd import _lib
L = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.zeros(64, device="cuda")
def ev():
    h = C.c_void_p(); L.gwen_event_create(C.byref(h)); return h
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
pairs = [(ev(), ev()) for _ in range(50)]
torch.cuda.synchronize()
res = []
for a, b in pairs:
    hip.hipEventRecord(a, st); hip.hipEventRecord(b, st)
torch.cuda.synchronize()
ms = C.c_float()
for a, b in pairs:
    L.gwen_event_elapsed_ms(a, b, C.byref(ms)); res.append(ms.value * 1e3)
print("empty bracket us:", sorted(res)[len(res)//2], min(res), max(res))
res = []
y = torch.zeros(64, device="cuda")
for a, b in pairs:
    hip.hipEventRecord(a, st)
    L.gwen_relu_backward_f32(C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 64, st)
    hip.hipEventRecord(b, st)
torch.cuda.synchronize()
for a, b in pairs:
    L.gwen_event_elapsed_ms(a, b, C.byref(ms)); res.append(ms.value * 1e3)
print("tiny kernel bracket us:", sorted(res)[len(res)//2], min(res), max(res))
