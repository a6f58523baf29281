"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs -> profiles/pmc_traffic.json (HBM bytes per launch per
kernel variant).  Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 128-byte requests
at 64 B, i.e. exactly half of the bytes of a wide (16 B/lane) coalesced read -> x2 (calibrated here on
gn_stats_kernel, which reads a known 512 MiB tensor once: FETCH_SIZE reports 256 MiB); WRITE_SIZE is 1:1
(calibrated on gn_apply_relu_kernel: 512 MiB written, 512 MiB reported).  Counter unit: KiB."""
import collections
import csv
import json
import re
import sys

fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
# what was profiled: bench.py attaches these figures only to a leg of the same workload / batch / precision
meta = {"workload": sys.argv[4] if len(sys.argv) > 4 else "full", "batch": int(sys.argv[5]) if len(sys.argv) > 5 else 64,
        "precision": sys.argv[6] if len(sys.argv) > 6 else "f16x3", "command": sys.argv[7] if len(sys.argv) > 7 else "bench.py"}


def variant(kname):
    m16 = re.search(r"igemm16_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (true|false)>", kname)
    if m16:
        mt, nt, wm, wn = (int(m16.group(i)) for i in range(1, 5))
        dcn, cat = m16.group(5) == "true", m16.group(6) == "true"
        pre = "dcn_igemm16" if dcn else "igemm16_cat" if cat else "igemm16"
        return "%s_f16x3_m%dn%d" % (pre, 32 * mt * wm, 32 * nt * wn)
    mp = re.search(r"igemm16p_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)((?:, (?:true|false))*)>", kname)
    if mp:  # <MT, NT, WM, WN, MULTISRC, FUSE[, GNIN[, GRU]]>
        mt, nt, wm, wn = (int(mp.group(i)) for i in range(1, 5))
        extra = [x.strip() == "true" for x in mp.group(7).split(",")[1:]] if mp.group(7) else []
        gru = len(extra) > 1 and extra[1]
        pre = "igemm16_gru" if gru else "igemm16_head" if mp.group(6) == "true" else "igemm16_cat" if mp.group(5) == "true" else "igemm16"
        return "%s_f16x3_m%dn%d" % (pre, 32 * mt * wm, 32 * nt * wn)
    md = re.search(r"dcn16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", kname)
    if md:  # <MT, NT, WM, WN, OCC>
        mt, nt, wm, wn = (int(md.group(i)) for i in range(1, 5))
        return "dcn_igemm16_f16x3_m%dn%d" % (32 * mt * wm, 32 * nt * wn)
    if "gn_final_kernel" in kname:
        return "gn_final_f32_valu"
    if "dcn16p_kernel<4" in kname:
        return "dcn16p_f16x3_p128n128"
    if "dcn16p_kernel" in kname:
        return "dcn16p_f16x3_p128n64"
    if "dcn16s_kernel" in kname:
        return "dcn16s_f16x3_p128n64"
    if "dcn16t_kernel" in kname:
        return "dcn16t_f16x3_p128n64"
    if "strm16_kernel" in kname:
        return "strm16_f16x3_w32n32"
    if "lowc2_kernel" in kname:
        return "lowc_stem_level0_f16x3"
    if "lowc1s_kernel" in kname:
        return "lowc_3x3s2_c16_rows_f16x3"
    mw = re.search(r"pw16s?_kernel<(\d+)>", kname)
    if mw:  # <NT>: the profile names both forms of the 1x1 stream by their tile
        return "pw16_f16x3_m128n%d" % (32 * int(mw.group(1)))
    mh = re.search(r"halo16_kernel<(\d+), (\d+), (\d+), (\d+)(?:, (?:true|false)(?:, (\d+))?)?", kname)
    if mh:  # <MT, NT, WM, WN[, BDIRECT[, EPI]]>
        mt, nt, wm, wn = (int(mh.group(i)) for i in range(1, 5))
        pre = {"1": "halo16_head", "2": "halo16_gru"}.get(mh.group(5) or "0", "halo16")
        return "%s_f16x3_m%dn%d" % (pre, 32 * mt * wm, 32 * nt * wn)
    ml = re.search(r"lowc_kernel<(\d+), (\d+), (\d+), ", kname)
    if ml:
        return {("4", "1"): "lowc_stem7x7_f16x3", ("16", "1"): "lowc_3x3_c16_f16x3",
                ("16", "2"): "lowc_3x3s2_c16_f16x3"}.get((ml.group(1), ml.group(3)), "lowc")
    m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (true|false)>", kname)
    if not m:
        return re.sub(r"\(.*", "", kname.replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    frag, mt, nt, wm, wn = (int(m.group(i)) for i in range(1, 6))
    dcn, aligned, cat = (m.group(i) == "true" for i in (6, 7, 8))
    bm, bn = frag * mt * wm, frag * nt * wn
    pre = "dcn_igemm" if dcn else "igemm_cat" if cat else "igemm" if aligned else "igemm_unaligned"
    return "%s_f32_%s_m%dn%d" % (pre, "16x16x4" if frag == 16 else "32x32x2", bm, bn)


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        v = variant(r["Kernel_Name"])
        d[v][0] += 1
        d[v][1] += float(r["Counter_Value"])
    return d


f, w = agg(fetch_csv), agg(write_csv)
res = {}
for k in f:
    fb = f[k][1] / f[k][0] * 1024 * 2.0
    wb = (w[k][1] / w[k][0] * 1024) if k in w else 0.0
    res[k] = {"launches_sampled": f[k][0], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
              "hbm_bytes_per_launch": round(fb + wb), "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1; KiB units"}
res["_meta"] = meta
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(res), "kernels")
