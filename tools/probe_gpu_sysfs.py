#!/usr/bin/env python3
"""What the GPU box offers a sampler thread for shader clock and socket power WITHOUT touching HIP (development aid).
Lists the amdgpu sysfs / hwmon files an ordinary user can read, their values and the latency of one read."""
import glob
import os
import time


def rd(path, binary=False):
    t0 = time.perf_counter()
    try:
        data = open(path, "rb").read()
    except OSError as e:
        return "ERR %s" % e, 0.0
    dt = (time.perf_counter() - t0) * 1e3
    return (("%d bytes" % len(data)) if binary else data.decode(errors="replace").strip().replace("\n", " | ")), dt


for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    dev = os.path.join(card, "device")
    if not os.path.exists(os.path.join(dev, "vendor")):
        continue
    print("==", card, rd(os.path.join(dev, "vendor"))[0], rd(os.path.join(dev, "device"))[0])
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent", "current_compute_partition", "power_dpm_force_performance_level"):
        v, dt = rd(os.path.join(dev, name))
        print("  %-40s %-80s %.2f ms" % (name, v[:80], dt))
    v, dt = rd(os.path.join(dev, "gpu_metrics"), binary=True)
    print("  %-40s %-80s %.2f ms" % ("gpu_metrics", v, dt))
    for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
        for f in sorted(os.listdir(hw)):
            p = os.path.join(hw, f)
            if os.path.isfile(p) and (f.startswith(("power", "freq", "temp1", "in0")) or f == "name"):
                v, dt = rd(p)
                print("  %-40s %-80s %.2f ms" % (os.path.relpath(p, dev), v[:80], dt))
try:
    import amdsmi  # noqa: F401

    print("amdsmi importable:", amdsmi.__file__)
except Exception as e:  # noqa: BLE001
    print("amdsmi not importable:", repr(e)[:200])
