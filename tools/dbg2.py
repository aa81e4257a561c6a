import sys
sys.path.insert(0, '.')
from animsnapbases_amd import HipEngine, Comm
e = HipEngine(0, stream=0)
import torch
print("is_available", torch.cuda.is_available(), "count", torch.cuda.device_count())
try:
    t = torch.zeros(4, device="cuda:0")
    print(t.device, hex(t.data_ptr()), "is_available now", torch.cuda.is_available())
except Exception as ex:
    print("cuda tensor failed:", repr(ex)[:300])
