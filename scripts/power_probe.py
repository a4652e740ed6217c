#!/usr/bin/env python3
"""Socket power and shader clock beside a running train step ("settle the clock"): 50 steps of one workload with the amdgpu hwmon files of the card sampled
every 10 ms by a host thread (bench.ClockSampler: freq1_input = sclk, power1_input = socket power; sysfs reads only) -> a CSV of the samples and a summary line.
Diagnostic, outside any timed region of bench.py.   python scripts/power_probe.py {2d|3d_bf16|3d_f32} [steps] > profiles/r06_power_<workload>.csv"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "2d"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(1)
    if what == "2d":
        from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
        eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device=dev, seed=0, lr=1e-5)
        x = torch.randn(32, 1, 512, 512, generator=g).to(dev)
        t = torch.randint(0, 2, (32, 512, 512), generator=g).to(dev)
    else:
        from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
        bf = what == "3d_bf16"
        n = 160 if bf else 128
        eng = UNet3DEngine(1, 3, dtype=torch.bfloat16 if bf else torch.float32, device=dev, seed=0, lr=1e-5)
        x = torch.randn(2, 1, n, n, n, generator=g).to(dev)
        t = (torch.rand(2, 3, n, n, n, generator=g) > 0.5).float().to(dev)

    def step():
        eng.forward(x, t, train=True)
        eng.backward()
        eng.optimizer_step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    idle = bench.ClockSampler(dev, period=0.01)
    idle.start()
    time.sleep(0.3)
    idle_sum = idle.stop()
    s = bench.ClockSampler(dev, period=0.01)
    s.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    summ = s.stop()
    print(f"# workload {what}: {steps} steps, {dt / steps * 1e3:.2f} ms/step; hwmon {s.hwmon}")
    print(f"# idle (0.3 s before): {idle_sum}")
    print(f"# under load: {summ}")
    print("t_ms,sclk_mhz,socket_power_w")
    for ts, mhz, w in s.samples:
        print(f"{(ts - t0) * 1e3:.1f},{mhz:.0f},{w:.1f}")


if __name__ == "__main__":
    main()
