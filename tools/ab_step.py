"""A/B of library builds on ONE box: tools/step_time.py in child processes, interleaved A, B, A, B ... (each child a fresh
process with GMVAE_HIP_LIB set); medians per build.  argv: rounds config libA libB [...]; 'default' = the in-tree build."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, config, libs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["GMVAE_HIP_LIB"] = os.path.join(root, l)
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "step_time.py"), config, "1.5"], env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", p.stderr[-500:], flush=True)
            continue
        j = json.loads(line[-1])
        res[l].append(j)
        print(f"round {r} {l}: {j['us_per_step_median']:.2f} us/step (min {j['us_per_step_min']:.2f}) levels {j['levels']} timeouts {j['timeouts']}", flush=True)
for l in libs:
    v = sorted(j["us_per_step_median"] for j in res[l])
    if v:
        print(f"== {l}: median of medians {v[len(v) // 2]:.2f} us/step, best {v[0]:.2f}")
