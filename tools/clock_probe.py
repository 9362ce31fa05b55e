"""What the chip clocks at while it runs the bench step: rocm-smi sampled every 0.5 s beside `tools/ab_step.py` (GPU box only).
Prints min / median / max of sclk, mclk and socket power over the timed replays, and the idle values before the run."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return {"err": str(e)}
    r = {}
    m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    if m:
        r["sclk"] = int(m.group(1))
    m = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", out)
    if m:
        r["mclk"] = int(m.group(1))
    m = re.search(r"(?:Average|Current) (?:Graphics Package|Socket Graphics Package)? ?Power \(W\):\s*([\d.]+)", out)
    if m:
        r["power"] = float(m.group(1))
    if not r:
        r["raw"] = out[-600:]
    return r


print("idle:", sample(), flush=True)
p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "ab_step.py"), "--steps", "60", "--rounds", "3"], stdout=subprocess.PIPE,
                     stderr=subprocess.DEVNULL, text=True)
rows = []
t0 = time.time()
while p.poll() is None:
    s = sample()
    s["t"] = round(time.time() - t0, 1)
    rows.append(s)
    time.sleep(0.5)
print(p.stdout.read().strip().splitlines()[-1])
for r in rows:
    print(r)
busy = [r for r in rows if r.get("power", 0) > 500]
for k in ("sclk", "mclk", "power"):
    v = sorted(r[k] for r in busy if k in r)
    if v:
        print(f"{k} while the step runs (power > 500 W samples: {len(v)}): min {v[0]} median {v[len(v) // 2]} max {v[-1]}")
