"""HBM activity of the timed configuration as the driver's sysfs counters see it (round 6): samples
/sys/class/drm/card*/device/{mem_busy_percent,gpu_busy_percent} at ~50 Hz while (1) a plain streaming read of a 8 GiB tensor runs (calibration:
what the counter shows at a known TB/s) and (2) `python bench.py <args>` runs, and reports the samples that fall inside bench.py's timed region
($VISPEC_BENCH_MARK).  rocprofv3's --pmc passes serialise the kernels, so they cannot see the four lanes together; this counter can, coarsely.

    python tools/mem_activity.py out.json [bench args...]"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_counter(name):
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                int(open(p).read())
                return p
            except (OSError, ValueError):
                pass
    return None


class Sampler(threading.Thread):
    def __init__(self, paths, period=0.02):
        super().__init__(daemon=True)
        self.paths, self.period, self.rows, self.stop = paths, period, [], False

    def run(self):
        while not self.stop:
            row = [time.time()]
            for p in self.paths:
                try:
                    row.append(int(open(p).read()))
                except (OSError, ValueError):
                    row.append(-1)
            self.rows.append(row)
            time.sleep(self.period)


def stats(rows, col, t0, t1):
    v = sorted(r[col] for r in rows if t0 <= r[0] <= t1 and r[col] >= 0)
    if not v:
        return None
    return dict(samples=len(v), mean=round(sum(v) / len(v), 1), p10=v[len(v) // 10], p50=v[len(v) // 2], p90=v[(9 * len(v)) // 10], max=v[-1])


def main():
    out, bench_args = sys.argv[1], sys.argv[2:]
    paths = [find_counter("mem_busy_percent"), find_counter("gpu_busy_percent")]
    res = dict(counters=paths)
    if paths[0] is None:
        res["error"] = "no mem_busy_percent under /sys/class/drm/card*/device"
        json.dump(res, open(out, "w"), indent=1)
        print(json.dumps(res))
        return
    smp = Sampler([p for p in paths if p])
    smp.start()
    # (1) calibration in a child process: streaming reads at a known rate
    cal = subprocess.run([sys.executable, "-c", (
        "import torch,time,json\n"
        "a=torch.empty(8<<30,dtype=torch.uint8,device='cuda').view(torch.int32);a.zero_()\n"
        "b=torch.empty_like(a)\n"
        "for fn,name,nb in ((lambda: a.sum(),'read',a.numel()*4),(lambda: b.copy_(a),'copy',2*a.numel()*4)):\n"
        "  fn();torch.cuda.synchronize();t0=time.time();n=0\n"
        "  while time.time()-t0<4: fn();n+=1\n"
        "  torch.cuda.synchronize();t1=time.time()\n"
        "  print(json.dumps(dict(name=name,t0=t0,t1=t1,TBps=n*nb/(t1-t0)/1e12)),flush=True)\n")], capture_output=True, text=True, cwd=ROOT)
    res["calibration"] = []
    for ln in cal.stdout.splitlines():
        if ln.startswith("{"):
            c = json.loads(ln)
            c["mem_busy_percent"] = stats(smp.rows, 1, c["t0"] + 0.5, c["t1"] - 0.2)
            c["gpu_busy_percent"] = stats(smp.rows, 2, c["t0"] + 0.5, c["t1"] - 0.2) if paths[1] else None
            res["calibration"].append(c)
    # (2) the bench
    mark = os.path.join("/tmp", f"vispec_mark_{os.getpid()}.json")
    env = dict(os.environ, VISPEC_BENCH_MARK=mark)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bench_args, capture_output=True, text=True, env=env, cwd=ROOT)
    smp.stop = True
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    m = json.load(open(mark)) if os.path.exists(mark) else None
    res["bench_args"] = bench_args
    if line:
        res["bench"] = dict(value=line["value"], ms_per_step=line["ms_per_step"], frac_region=(line.get("roofline") or {}).get("frac_region"),
                            achieved_region_GBps=(line.get("roofline") or {}).get("achieved_region"))
    else:
        res["bench_error"] = r.stderr[-1500:]
    if m:
        res["timed_region"] = dict(seconds=round(m["t1"] - m["t0"], 2), mem_busy_percent=stats(smp.rows, 1, m["t0"], m["t1"]),
                                   gpu_busy_percent=stats(smp.rows, 2, m["t0"], m["t1"]) if paths[1] else None)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
