"""A/B of two (or more) builds of libnbx.so (tools/build_variant.sh) on the default launch shapes: one child process per
library and round, round-robin, so that clock drift hits every build alike; prints the median force-kernel time per case.
usage: lib_ab.py [--rounds R] [--cases n:own[:order],...] LIB[=name] LIB[=name] ...   (LIB 'default' = the in-tree build)"""
import json
import os
import subprocess
import sys

CASES = "1048576:131072,262144:262144,1048576:262144,1048576:1048576,524288:524288,131072:131072,16384:16384,65536:65536"


def child(cases):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nbody-demo-2023_amd"))
    import nbx
    out = {}
    for case in cases.split(","):
        f = case.split(":")
        n, own = int(f[0]), int(f[1])
        kw = {}
        if len(f) > 2 and f[2]:
            kw["summation_order"] = {"ref": nbx.ORDER_REFERENCE, "tree": nbx.ORDER_TREE}[f[2]]
        if len(f) > 3:
            kw["bodies_per_lane"] = int(f[3])
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, **kw)
        c.upload(nbx.initial_conditions(n))
        steps = max(3, int(1.2e11 / (float(n) * own)))
        for _ in range(2):
            c.step_local(); c.commit()
        c.sync(); c.profile(True)
        for _ in range(steps):
            c.step_local(); c.commit()
        c.sync()
        st = c.stats(); c.close()
        out[case] = dict(ms=st["force_ms_total"] / st["force_launches_timed"], B=st["bodies_per_lane"], S=st["j_split"], variant=st["kernel_variant"],
                         loop=st["inner_loop"], grid=[st["force_grid_x"], st["force_grid_y"]], pairs=float(n) * own)
    print("RESULT " + json.dumps(out))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child(args[1])
    rounds, cases, libs = 3, CASES, []
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        elif a == "--cases":
            cases = args.pop(0)
        else:
            path, _, name = a.partition("=")
            libs.append((name or path, path))
    res = {name: {} for name, _ in libs}
    for r in range(rounds):
        for name, path in libs:
            env = dict(os.environ)
            env.pop("NBX_LIB", None)
            if path != "default":
                env["NBX_LIB"] = os.path.abspath(path)
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cases], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("child failed for %s:\n%s\n%s" % (name, o.stdout[-2000:], o.stderr[-2000:]), flush=True)
                sys.exit(1)
            for case, v in json.loads(line[0][7:]).items():
                res[name].setdefault(case, []).append(v)
    for case in cases.split(","):
        base = None
        for name, _ in libs:
            v = res[name][case]
            ms = sorted(x["ms"] for x in v)[len(v) // 2]
            base = base or ms
            print("%-24s %-10s variant %d B%d S%d loop %d grid %4dx%-2d median %9.4f ms (min %9.4f)  %5.2f %%  %+5.2f %%" % (
                case, name, v[0]["variant"], v[0]["B"], v[0]["S"], v[0]["loop"], v[0]["grid"][0], v[0]["grid"][1], ms, min(x["ms"] for x in v),
                100 * 20.0 * v[0]["pairs"] / (ms * 1e-3) / 157.3e12, 100 * (base / ms - 1)), flush=True)


if __name__ == "__main__":
    main()
