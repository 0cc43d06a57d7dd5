#!/usr/bin/env python3
"""Is the step power-managed?  Samples the GPU's socket power, its cap and the shader clock from the amdgpu hwmon files (sysfs; falls back
to `rocm-smi`) every ~20 ms while a child process runs `bench.py --single-mode --no-cpu-baseline` with the given environment, and prints
mean / p90 power and clock over the samples taken while the child was inside its timed region (the busiest 60 % of the samples).

    python3 tools/power_probe.py [ENV=VALUE ...] [-- bench args]
The parent never touches HIP (the child owns the GPU)."""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for key, names in (("power", ("power1_average", "power1_input")), ("cap", ("power1_cap",)), ("sclk", ("freq1_input",)), ("temp", ("temp2_input", "temp1_input"))):
            for n in names:
                p = os.path.join(d, n)
                if key not in out and os.path.isfile(p):
                    try:
                        int(open(p).read().strip()); out[key] = p
                    except Exception:
                        pass
        if "power" in out:
            break
    return out


def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def main():
    args = sys.argv[1:]
    envs, bench = [], []
    if "--" in args:
        i = args.index("--"); envs, bench = args[:i], args[i + 1:]
    else:
        envs = args
    env = dict(os.environ)
    for e in envs:
        k, v = e.split("=", 1); env[k] = v
    files = hwmon_files()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-mode", "--no-cpu-baseline", "--steps", "600", "--warmup", "50"] + bench
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples = []
    smi = None
    while child.poll() is None:
        t = time.time()
        if "power" in files:
            samples.append((t, read(files["power"]), read(files.get("sclk", "")) if "sclk" in files else None, read(files.get("temp", "")) if "temp" in files else None))
            time.sleep(0.02)
        else:      # no readable hwmon: rocm-smi (slow: ~0.3 s per call)
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True)
            smi = (smi or []) + [r.stdout.strip().replace("\n", " | ")]
            time.sleep(0.2)
    line = child.stdout.read().strip().splitlines()[-1] if child.stdout else ""
    import json
    try:
        d = json.loads(line)
        print("bench: %.4f ms/step, %.1f chunks/s" % (d["ms_per_step"], d["value"]))
    except Exception:
        print("bench output unreadable:", line[:200])
    print("env:", " ".join(envs) or "(default)", " files:", {k: os.path.basename(v) for k, v in files.items()})
    if samples:
        pw = sorted(s[1] for s in samples if s[1] is not None)
        if pw:
            busy_thr = pw[int(len(pw) * 0.4)]
            busy = [s for s in samples if s[1] is not None and s[1] >= busy_thr]
            W = [s[1] / 1e6 for s in busy]
            clk = [s[2] / 1e6 for s in busy if s[2]]
            tmp = [s[3] / 1e3 for s in busy if s[3]]
            cap = read(files["cap"]) if "cap" in files else None
            print("samples %d (busy %d): power mean %.0f W, p90 %.0f W, max %.0f W%s" % (len(samples), len(busy), sum(W) / len(W), sorted(W)[int(len(W) * 0.9)], max(W),
                                                                                     ", cap %.0f W" % (cap / 1e6) if cap else ""))
            if clk:
                print("sclk (hwmon, MHz): mean %.0f, p10 %.0f, p90 %.0f" % (sum(clk) / len(clk), sorted(clk)[int(len(clk) * 0.1)], sorted(clk)[int(len(clk) * 0.9)]))
            if tmp:
                print("temperature: mean %.0f C, max %.0f C" % (sum(tmp) / len(tmp), max(tmp)))
    elif smi:
        print("rocm-smi samples:")
        for s in smi[len(smi) // 3:len(smi) // 3 + 6]:
            print("  ", s[:300])
    return 0


if __name__ == "__main__":
    sys.exit(main())
