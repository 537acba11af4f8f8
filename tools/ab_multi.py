import os, subprocess, sys, json
root = "/root/repo"
libs = sys.argv[2:]
rounds = int(sys.argv[1])
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, RT_MI355X_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-valu", "--cpu-rows", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        if not out: res[l].append(float("nan")); continue
        d = json.loads(out[-1]); res[l].append(d["ms_per_step"] * 1e3)
for l in libs:
    v = res[l]; print("%-22s step us: %s  mean %.1f" % (os.path.basename(l), " ".join("%.1f" % x for x in v), sum(v) / len(v)))
