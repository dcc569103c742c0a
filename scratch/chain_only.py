"""The configs[2] chain alone (bench.py's config2_chain, without the rest of the bench line):
python scratch/chain_only.py [timed chunks] [warm chunks]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from vsamd import capi
vs = capi.load(os.environ.get("VS_LIB"))
p = bench.make_params(vs, max_corners=400)
import resource, time
def throttled():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat") if l.startswith(("nr_throttled", "throttled_usec", "nr_periods"))}
    except OSError:
        return {}
th0 = throttled()
u0, w0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
r = bench.config2_chain(vs, 0, p, 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
u1, w1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
r["whole_run"] = {"wall_s": round(w1 - w0, 2), "user_s": round(u1.ru_utime - u0.ru_utime, 2), "sys_s": round(u1.ru_stime - u0.ru_stime, 2),
                  "cores_busy": round((u1.ru_utime - u0.ru_utime + u1.ru_stime - u0.ru_stime) / (w1 - w0), 2)}
th1 = throttled()
r["cgroup_cpu"] = {k: th1[k] - th0.get(k, 0) for k in th1}
r.pop("what")
print(json.dumps(r))
