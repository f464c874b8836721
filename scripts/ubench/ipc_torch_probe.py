"""Capability probe: can memory of torch's caching allocator be exported with hipIpcGetMemHandle (the ring's receive
buffers are torch tensors handed over as raw device pointers), and does a copy by another process land in it?
parent: tensors of several sizes -> handle of an INTERIOR pointer -> child process opens it and fills the tensor."""
import ctypes as C
import subprocess
import sys

import torch

hip = C.CDLL("libamdhip64.so")


class Handle(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]


def child(hexes):
    torch.cuda.init()
    src = torch.arange(1 << 20, dtype=torch.float64, device="cuda")
    for hx, n in hexes:
        h = Handle.from_buffer_copy(bytes.fromhex(hx))
        p = C.c_void_p()
        e = hip.hipIpcOpenMemHandle(C.byref(p), h, 1)
        print(f"  child: open -> {e} ptr {p.value}", flush=True)
        if e == 0:
            e2 = hip.hipMemcpy(p, C.c_void_p(src.data_ptr()), n * 8, 3)
            print(f"  child: copy of {n} doubles -> {e2}", flush=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        args = sys.argv[1:]
        child([(args[i], int(args[i + 1])) for i in range(0, len(args), 2)])
        sys.exit(0)
    torch.cuda.init()
    pad = torch.zeros(1000, device="cuda")           # something else in the small pool first
    ts = [torch.zeros(n, dtype=torch.float64, device="cuda") for n in (100, 7 * 1031, 1 << 17, 1 << 20)]
    argv = []
    for t in ts:
        h = Handle()
        e = hip.hipIpcGetMemHandle(C.byref(h), C.c_void_p(t.data_ptr()))
        base, size = C.c_void_p(), C.c_size_t()
        e3 = hip.hipMemGetAddressRange(C.byref(base), C.byref(size), C.c_void_p(t.data_ptr()))
        print(f"tensor of {t.numel()} doubles at {t.data_ptr():#x}: hipIpcGetMemHandle -> {e}; allocation base {base.value:#x} size {size.value} ({e3}), "
              f"offset {t.data_ptr() - (base.value or 0)}", flush=True)
        argv += [bytes(h.reserved).hex() if False else bytes(h)[:64].hex(), str(min(t.numel(), 1 << 20))]
    torch.cuda.synchronize()
    subprocess.check_call([sys.executable, __file__] + argv)
    torch.cuda.synchronize()
    for t in ts:
        n = t.numel()
        good = bool((t == torch.arange(n, dtype=torch.float64, device="cuda")).all())
        print(f"tensor of {n} doubles: filled by the other process: {good}", flush=True)
