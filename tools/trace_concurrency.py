"""How full is the GPU while several lanes run?  From a rocprofv3 kernel trace (start / end timestamp, grid and workgroup size of every
dispatch): over the part of the run where at least four queues are active, the share of time with k kernels in flight, the time-weighted
number of workgroups asked for (a proxy for CUs wanted: the cohort GEMMs and the attention kernels hold a CU per workgroup), and who is in
flight together with whom.
    python tools/trace_concurrency.py <kernel_trace.csv> [out.json]"""
import collections
import csv
import json
import sys

CLASSES = (("gemm_w32_c8", "gemm_cohort"), ("wide8", "gemm_cohort"), ("gemm_w32_wide_kernel", "gemm_cohort"), ("gemm_w32_big", "draft_prefill"), ("gemm_w32_kernel", "gemm_single/draft"),
           ("tree_attn2_partial", "attn_partial"), ("tree_attn_reduce", "attn_merge"), ("splitk_reduce", "reduce"), ("Cijk", "prefill_gemm"),
           ("prefill_attn", "prefill_attn"), ("quant_rows", "quant"))


def klass(name):
    for key, c in CLASSES:
        if key in name:
            return c
    return "other"


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    n_wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], klass(r["Kernel_Name"]), n_wg))
rows.sort()
t0 = rows[0][0]
BIN = 10_000_000  # 10 ms
queues_in_bin = collections.defaultdict(set)
for s, e, q, c, w in rows:
    queues_in_bin[(s - t0) // BIN].add(q)
multi = {b for b, qs in queues_in_bin.items() if len(qs) >= 4}
ev = []
for i, (s, e, q, c, w) in enumerate(rows):
    if (s - t0) // BIN in multi:
        ev.append((s, 1, i))
        ev.append((e, -1, i))
ev.sort()
live = {}
t_prev = None
by_k = collections.Counter()
wg_time = 0.0
wgcap_time = 0.0
class_time = collections.Counter()
pair_time = collections.Counter()
gemm_count_time = collections.Counter()
tot = 0.0
for t, d, i in ev:
    if t_prev is not None and live and t > t_prev:
        dt = t - t_prev
        if dt < BIN:  # (gaps between multi-lane bins are not part of the sample)
            tot += dt
            by_k[min(len(live), 6)] += dt
            w = sum(rows[j][4] for j in live)
            wg_time += dt * w
            wgcap_time += dt * min(w, 256)
            cs = sorted(rows[j][3] for j in live)
            for c in cs:
                class_time[c] += dt
            gemm_count_time[sum(1 for c in cs if c == "gemm_cohort")] += dt
            pair_time["+".join(cs)] += dt
    elif t_prev is not None and not live and t - t_prev < BIN:
        tot += t - t_prev
        by_k[0] += t - t_prev
    if d > 0:
        live[i] = 1
    else:
        live.pop(i, None)
    t_prev = t
out = dict(sample_s=round(tot / 1e9, 3), multi_lane_bins=len(multi),
           share_of_time_with_k_kernels_in_flight={str(k): round(v / tot, 4) for k, v in sorted(by_k.items())},
           mean_kernels_in_flight=round(sum(k * v for k, v in by_k.items()) / tot, 3),
           mean_workgroups_asked=round(wg_time / tot, 1), mean_workgroups_asked_capped_256=round(wgcap_time / tot, 1),
           mean_in_flight_by_class={c: round(v / tot, 3) for c, v in class_time.most_common()},
           share_of_time_with_n_cohort_gemms_in_flight={str(k): round(v / tot, 4) for k, v in sorted(gemm_count_time.items())},
           most_common_sets={k: round(v / tot, 4) for k, v in pair_time.most_common(14)})
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
