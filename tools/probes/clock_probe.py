"""does the matrix pipe hold its clock?  loops the RPN 3x3 conv GEMM (conv_fwd_split_kernel<2,2,3>) on quiet (zeros) and on
random operands for a few seconds each while a thread samples the GPU's sclk / power from sysfs (hwmon) or rocm-smi"""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def find_sources():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "power1_average", "power1_input"):
            f = os.path.join(h, name)
            if os.path.exists(f):
                out.setdefault(name, []).append(f)
    return out


SRC = find_sources()
print("sysfs sources:", {k: len(v) for k, v in SRC.items()})


def read_all():
    vals = {}
    for k, files in SRC.items():
        best = 0
        for f in files:
            try:
                best = max(best, int(open(f).read().strip()))
            except Exception:
                pass
        vals[k] = best
    return vals


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                              timeout=20).stdout
    except Exception as e:
        return "rocm-smi failed: %r" % (e,)


def run(label, x, w, seconds=4.0):
    y = _C.conv_forward(x, w, pad=1)
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append(read_all())
            time.sleep(0.02)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            _C.conv_forward(x, w, pad=1, out=y)
        n += 50
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    mid = smi() if label.endswith("(smi)") else ""
    stop[0] = True
    th.join()
    ms = s.elapsed_time(e) / n
    gf = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * w.shape[1] * 9 / 1e9
    late = samples[len(samples) // 2:]
    f = [v.get("freq1_input", 0) / 1e6 for v in late]
    p = [max(v.get("power1_average", 0), v.get("power1_input", 0)) / 1e6 for v in late]
    print("%-28s %.3f ms  %.0f TF/s algorithmic (%.0f executed)   sclk MHz min/avg/max %.0f / %.0f / %.0f   power W avg %.0f  (%d samples)" % (
        label, ms, gf / ms, 6 * gf / ms, min(f or [0]), sum(f) / max(len(f), 1), max(f or [0]), sum(p) / max(len(p), 1),
        len(late)))
    if mid:
        print(mid[-1500:])


print(smi()[-800:])
x0 = torch.zeros((2, 1024, 64, 128), device=dev).contiguous(memory_format=CL)
w0 = torch.zeros((1024, 1024, 3, 3), device=dev).contiguous(memory_format=CL)
xr = torch.randn((2, 1024, 64, 128), device=dev).contiguous(memory_format=CL)
wr = (torch.randn((1024, 1024, 3, 3), device=dev) * 0.02).contiguous(memory_format=CL)
run("idle->zeros", x0, w0)
run("randn", xr, wr)
run("zeros again", x0, w0)
run("randn (smi)", xr, wr, seconds=6.0)
