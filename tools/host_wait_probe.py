"""Which host wait of the HIP runtime leaves the cores alone?  Queues ~1 s of GEMMs, waits for them in one of several ways,
and prints the CPU seconds every thread of the process burnt while waiting (/proc/self/task).

    python tools/host_wait_probe.py                      # all modes, each in its own process
    python tools/host_wait_probe.py MODE                 # "+"-joined options: flags flags_late evblock streamsync devsync side two d2h
"""
import ctypes
import os
import subprocess
import sys
import time


def threads():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            raw = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        f = raw[raw.rindex(")") + 2:].split()
        out[int(tid)] = (int(f[11]) + int(f[12])) / tick
    return out


def run(mode):
    import torch

    torch.cuda.init()
    hip = ctypes.CDLL("libamdhip64.so")
    opts = set(mode.split("+"))
    if "flags" in opts:
        print(f"  hipSetDeviceFlags(blocking) -> {hip.hipSetDeviceFlags(ctypes.c_uint(0x4))}")
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    b = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    if "flags_late" in opts:
        print(f"  hipSetDeviceFlags(blocking), late -> {hip.hipSetDeviceFlags(ctypes.c_uint(0x4))}")
    side = torch.cuda.Stream() if ("side" in opts or "two" in opts) else None
    host = torch.empty(1 << 20, dtype=torch.float16).pin_memory()
    host_dev = None
    if "zerocopy" in opts:
        ptr = ctypes.c_void_p()
        rc = hip.hipHostGetDevicePointer(ctypes.byref(ptr), ctypes.c_void_p(host.data_ptr()), 0)
        print(f"  hipHostGetDevicePointer -> {rc}, same address: {ptr.value == host.data_ptr()}")
        class _Arr:  # __cuda_array_interface__ view of the pinned buffer
            __cuda_array_interface__ = {"shape": (1 << 20,), "typestr": "<f2", "data": (ptr.value, False), "version": 2}
        host_dev = torch.as_tensor(_Arr(), device="cuda")
    t0 = time.perf_counter()
    th0 = threads()
    work = torch.cuda.stream(side) if "side" in opts else torch.cuda.stream(torch.cuda.current_stream())
    with work:
        for _ in range(600):
            c = a @ b
        if "two" in opts:
            with torch.cuda.stream(side):
                for _ in range(100):
                    d = a @ b
            torch.cuda.current_stream().wait_stream(side)
        if "d2h" in opts:
            host.copy_(c.view(-1)[:1 << 20], non_blocking=True)
        if "d2hs" in opts:      # a small read-back, like the per-step frontier / waypoint results
            host[:2048].copy_(c.view(-1)[:2048], non_blocking=True)
        if "zerocopy" in opts:  # the kernel itself stores into the pinned host buffer
            torch.add(c.view(-1)[:2048], 1, out=host_dev[:2048])
        queued = time.perf_counter() - t0
        if "streamsync" in opts:
            torch.cuda.current_stream().synchronize()
        elif "devsync" in opts:
            torch.cuda.synchronize()
        else:
            ev = torch.cuda.Event(blocking="evblock" in opts)
            ev.record()
            ev.synchronize()
    wall = time.perf_counter() - t0
    th1 = threads()
    burnt = sorted(((th1[t] - th0.get(t, 0.0)) for t in th1), reverse=True)[:3]
    print(f"{mode:32s} wall {wall:.2f} s (queued in {queued:.2f} s), busiest threads: "
          + ", ".join(f"{x / wall:.2f}" for x in burnt) + " cores", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for mode in ("event", "evblock", "flags", "flags+evblock", "flags+streamsync", "flags+devsync",
                     "flags+evblock+side", "flags+evblock+d2h", "flags+evblock+d2hs", "flags+evblock+zerocopy",
                     "flags+evblock+two", "flags_late+evblock"):
            subprocess.run([sys.executable, __file__, mode], timeout=60)
