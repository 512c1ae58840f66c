"""Does the priority of the backward-weights side stream matter?  One process, the headline step, three side streams in turn:
torch's default-priority stream, a LOW-priority and a HIGH-priority stream made with hipStreamCreateWithPriority.
    python scratch/r4_prio_probe.py [rounds]"""
import ctypes, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
unet_amd = importlib.import_module("unet-medical-image-contour-segmentation_amd")
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range: least", lo.value, "greatest", hi.value, flush=True)


def make(prio):
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(prio))      # hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda:0"))


dev = torch.device("cuda:0")
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=True).to(dev).to(memory_format=torch.channels_last)
st = unet_amd.TrainStepper(model, lr=1e-5, amp=True, wgrad_stream=True)
g = torch.Generator().manual_seed(1)
im = torch.rand(8, 1, 512, 512, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
mk = torch.randint(0, 2, (8, 512, 512), generator=g).to(dev)
streams = {"default": st.wgrad_stream, "low": make(lo.value), "high": make(hi.value), "none": None}


def run(steps=30, warm=6):
    for _ in range(warm):
        st.step(im, mk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(im, mk)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, s in streams.items():
        st.wgrad_stream = s
        ms = run()
        print(f"round {rep}  side stream {name:8s} {ms:7.3f} ms/step  {8 / ms * 1e3:7.1f} images/s", flush=True)
