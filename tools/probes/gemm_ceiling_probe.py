#!/usr/bin/env python3
"""What does the vendor GEMM (hipBLASLt through torch.matmul) sustain on THIS box for bf16 x bf16 -> f32-accumulate on
N(0,1) data?  The bf16 sweep kernels are power-limited (board power sits at the cap, the shader clock gives way), so the
practical ceiling of any bf16 MFMA kernel here is what a tuned GEMM reaches under the same cap, not 2.5 PFLOP/s."""
import subprocess
import sys
import threading
import time

import torch

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = [l.split(":")[-1].strip() for l in o.splitlines() if "Package Power" in l]
            s = [l.split("(")[-1].split("Mhz")[0] for l in o.splitlines() if "sclk" in l]
            if p and s:
                samples.append((float(p[0]), float(s[0])))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.1)


for (m, n, k, kind) in [(8192, 8192, 8192, "randn"), (8192, 8192, 8192, "zeros"), (65536, 1024, 768, "randn"), (262144, 1024, 768, "randn"),
                        (16384, 16384, 768, "randn")]:
    a = torch.randn((m, k), generator=g, device=dev).bfloat16() if kind == "randn" else torch.zeros((m, k), device=dev, dtype=torch.bfloat16)
    b = torch.randn((n, k), generator=g, device=dev).bfloat16() if kind == "randn" else torch.zeros((n, k), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.T
    torch.cuda.synchronize()
    reps = max(5, int(3e15 / (2.0 * m * n * k)))
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c = a @ b.T
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
    tail = samples[len(samples) // 2:]
    pw = sum(x[0] for x in tail) / max(len(tail), 1)
    ck = sum(x[1] for x in tail) / max(len(tail), 1)
    print(f"torch.matmul bf16 {m}x{n}x{k} {kind}: {ms:.3f} ms = {tf:.0f} TFLOP/s ({tf / 2500:.3f} of 2.5 PF), power {pw:.0f} W, sclk {ck:.0f} MHz "
          f"({len(samples)} samples, {reps} reps)", flush=True)
    del a, b, c
