#!/usr/bin/env python3
"""Runs bench.py once per environment setting and prints the per-kernel milliseconds side by side.
usage: exp_env.py [--workload W] [--steps N] [--lib PATH] NAME=VALUE[,NAME=VALUE..] ...   ("base" = no change;
"wl=W" switches the workload for the settings after it; "args=--flag,value,..." gives bench.py extra arguments)"""
import json, os, subprocess, sys
args = sys.argv[1:]
wl, steps, spp = "san-miguel-class", "3", "32"
while args and args[0].startswith("--"):
    k = args.pop(0)
    v = args.pop(0)
    if k == "--workload": wl = v
    elif k == "--steps": steps = v
    elif k == "--spp": spp = v
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for setting in args:
    if setting.startswith("wl="):
        wl = setting[3:]
        continue
    env = dict(os.environ)
    extra = []
    if setting.startswith("args="):
        extra = setting[5:].split(",")
    elif setting != "base":
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            env[k] = v
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "1", "--workload", wl,
                        "--spp", spp, "--no-extras", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True,
                       timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(setting, "FAILED", p.stderr[-500:])
        continue
    d = json.loads(line[-1])
    k = d["kernels"]
    parts = ["%s %.2f" % (n.replace("extend", "ext").replace("connect", "con").replace("depth0", "d0"), k[n]["avg_launch_ms"])
             for n in sorted(k) if isinstance(k[n], dict)]
    print("%-60s %8.1f Mrays/s  step %.2f ms | %s | other %.2f" % (setting, d["value"], d["ms_per_step"], "  ".join(parts),
                                                               k["other_ms_per_step"]), flush=True)
