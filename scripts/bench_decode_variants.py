"""Tuning aid: decode ms/token under env-selected kernel variants (each in a child process)."""
import os, subprocess, sys
for env in [{}, {"SPIDER_GEMV_HOIST": "0"}, {"SPIDER_GEMV_HOIST": "1"}]:
    e = dict(os.environ, PYTHONPATH=".", **env)
    o = subprocess.run([sys.executable, "scripts/prof_decode.py", "130"], env=e, capture_output=True, text=True)
    print(env, [l for l in o.stdout.splitlines() if "decode" in l] or o.stderr[-300:])
