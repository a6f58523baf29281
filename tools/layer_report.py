"""Summarise a CP_PROFILE_DUMP per-launch CSV (kernel,M,N,K,kh,stride,ms,TF) and a bench JSON line."""
import collections
import csv
import json
import sys

bench, layers = sys.argv[1], sys.argv[2]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
d = json.loads(open(bench).read())
# number of timed steps whose launches were profiled (bench.py samples every 4th step); argv[3] overrides
steps = float(d["roofline"].get("timed_steps_sampled") or sys.argv[3])
print("value %.1f img/s  %.2f ms/step  p50 B=1 %s ms  whole-step %.1f TF" % (
    d["value"], d["ms_per_step"], d.get("p50_frame_ms_batch1"), d["whole_step_tflops"]))
for k, v in d["roofline"]["all_conv_kernels"].items():
    print("  %-34s %s" % (k, v))
rows = list(csv.reader(open(layers)))
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault(tuple(r[:6]), []).append(float(r[6]))
out, tot = [], 0.0
for k, v in agg.items():
    ms, n = sum(v) / steps, len(v) / steps
    fl = 2.0 * int(k[1]) * int(k[2]) * int(k[3]) * n
    out.append((ms, k, n, fl / (ms * 1e-3) / 1e12))
    tot += ms
out.sort(reverse=True)
for ms, k, n, tf in out[:top]:
    print("%7.3f ms/step x%-3g %-32s M=%-8s N=%-4s K=%-5s k%s s%s %6.1f TF" % (ms, n, k[0], k[1], k[2], k[3], k[4], k[5], tf))
print("total conv ms/step %.2f" % tot)
