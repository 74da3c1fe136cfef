#!/usr/bin/env python3
"""A/B of one kernel-experiment switch on the SAME box: builds (or reuses) the -DKAN_TUNING_KNOBS variant of the library and runs the default bench
workload twice per arm, alternating, with NAME=0 / unset.  The shipped library reads no environment variable; this is a measurement aid.
usage: python tools/ab_env.py KAN_PM_LPT [--workload kan_vgg11] [--steps 30]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from convkan_amd import build
name = sys.argv[1]
extra = sys.argv[2:]
lib = build.build_library(defines=("KAN_TUNING_KNOBS",))
res = {"off": [], "on": []}
for rep in range(2):
    for arm in ("off", "on"):
        env = dict(os.environ, KANCONV_LIB=lib)
        if arm == "off":
            env[name] = "0"
        else:
            env.pop(name, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-aux"] + extra, capture_output=True, text=True, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            sys.stderr.write(r.stderr[-2000:]); raise SystemExit(1)
        d = json.loads(line[0])
        res[arm].append(d["ms_per_step"])
        ks = {k: v["avg_ms"] for k, v in d["roofline"]["kernels"].items()}
        print(f"{name} {arm:3s} rep {rep}: {d['ms_per_step']} ms/step  {ks}", flush=True)
print(json.dumps({"switch": name, "ms_per_step": res}))
