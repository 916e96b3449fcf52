#!/usr/bin/env python3
"""Does the GPU hold the same clock under two runs of the same workload?  Runs a command while sampling the card's shader clock and power
from sysfs (hwmon freq1_input / power1_average, every 20 ms) and prints the command's last JSON line's kernel time next to the clock and
power statistics of the samples taken while the GPU was busy.  (Round 4: the same 4v4 build measures ~20.3 or ~22.4 us per step by process.)
    python tools/micro/clock_watch.py python bench.py --n-agents 4 --steps 2000 --no-cpu-baseline --no-other-workloads --no-live-traffic"""
import glob, json, os, subprocess, sys, threading, time

def find():
    out = []
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        f = os.path.join(h, "freq1_input")
        if os.path.exists(f):
            out.append((f, os.path.join(h, "power1_average") if os.path.exists(os.path.join(h, "power1_average")) else os.path.join(h, "power1_input")))
    return out

def read(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return float("nan")

srcs = find()
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        samples.append((time.time(), [(read(f) / 1e6, read(p) / 1e6) for f, p in srcs]))
        time.sleep(0.02)
th = threading.Thread(target=sampler, daemon=True); th.start()
t0 = time.time()
r = subprocess.run(sys.argv[1:], capture_output=True, text=True)
stop.set(); th.join()
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
res = json.loads(line[-1]) if line else {}
out = {"cards": len(srcs), "us_per_step": res.get("roofline", {}).get("avg_launch_us"), "samples": len(samples), "seconds": round(time.time() - t0, 1)}
for c in range(len(srcs)):
    fr = [s[1][c][0] for s in samples]; pw = [s[1][c][1] for s in samples]
    busy = [(f, p) for f, p in zip(fr, pw) if p == p and p > 0.5 * max(pw)]       # samples taken under load
    if busy:
        fs = sorted(f for f, _ in busy); ps = sorted(p for _, p in busy)
        out[f"card{c}"] = {"busy_samples": len(busy), "sclk_MHz_median": fs[len(fs) // 2], "sclk_MHz_min": fs[0], "sclk_MHz_max": fs[-1],
                           "power_W_median": ps[len(ps) // 2], "power_W_max": ps[-1]}
print(json.dumps(out))
