"""Median socket power and sclk of a tools/power_trace.sh sample file (samples above 1 kW = the loaded part of the run).
usage: power_summary.py <samples file> <ms per batch>"""
import re
import statistics
import sys

pw, clk = [], []
for line in open(sys.argv[1]):
    m = re.search(r"card0,([0-9.]+)", line)
    nums = re.findall(r"[0-9]+\.?[0-9]*", line)
    w = [float(x) for x in nums if 300.0 <= float(x) <= 1600.0]
    mhz = re.findall(r"\((\d+)Mhz\)", line)   # fclk, mclk, sclk, socclk
    if w and max(w) > 1000:
        pw.append(max(w))
        if len(mhz) >= 3:
            clk.append(int(mhz[2]))
ms = float(sys.argv[2])
if pw:
    p = statistics.median(pw)
    print("%d samples above 1 kW: median %.0f W, sclk %s MHz; %.3f ms per batch -> %.1f J per batch at the socket, %.1f J above the 340 W of resident idle waves"
          % (len(pw), p, statistics.median(clk) if clk else "n/a", ms, p * ms / 1e3, (p - 340.0) * ms / 1e3))
else:
    print("no loaded samples")
