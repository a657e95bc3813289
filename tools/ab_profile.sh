#!/bin/bash
# same-box A/B of two trees (this one and a worktree under ./_r02): rocprofv3 kernel stats of tools/graph_stats_probe.py in each
export TMPDIR=/tmp
for t in . _r02; do
  tag=$(basename $(cd $t && pwd)); out=/tmp/ab_$tag; rm -rf $out
  ( cd $GRAFT_REPO_ROOT/$t && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/graph_stats_probe.py > /tmp/ab_$tag.txt 2>&1 )
  grep rounds /tmp/ab_$tag.txt
  cp "$(find $out -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/ab_stats_$tag.csv
  cp "$(find $out -name '*kernel_trace.csv' | head -1)" /tmp/ab_trace_$tag.csv
  python - /tmp/ab_trace_$tag.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 30 % of the run is decode rounds of the third request: busy time and gaps there
n = len(rows); sel = rows[int(n * 0.7):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(sel, sel[1:])]
gaps_pos = [g for g in gaps if g > 0]
print("kernels", len(sel), "span ms", span / 1e6, "busy ms", busy / 1e6, "busy frac", busy / span, "mean gap us", sum(gaps_pos) / len(gaps) / 1e3, "median gap us", sorted(gaps)[len(gaps) // 2] / 1e3)
PY
done
