#!/bin/bash
# PMC passes over one BASELINE configuration (scripts/exp_cfg_one.py): scripts/cfg_pmc.sh <c3|c5|c4>  -> gpurun_out/cfgpmc_<cfg>/summary.txt
# (one counter group per pass; --pmc never combined with tracing)
cfg=$1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/cfgpmc_$cfg
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 $R/scripts/exp_cfg_one.py $cfg > $out/pass$i.log 2>&1
  echo "pass$i [$grp] rc=$?"
done
python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void pygim::", "")
        if not ("k_csr_panel" in k or "k_lds_code" in k or "k_slice_pack" in k):
            continue
        a = agg[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        if row["Counter_Name"] == "FETCH_SIZE" and row.get("Start_Timestamp"):
            d = dur[k]; d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); d[1] += 1
lines = []
for k, d in sorted(agg.items()):
    lines.append(k)
    for c, v in sorted(d.items()):
        lines.append(f"    {c:34s} {v[0] / v[1]:18.1f} per launch ({v[1]} launches)")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and reads HALF the bytes of wide reads on gfx950 -> doubled; WRITE_SIZE (KiB) exact
        hbm = (2.0 * d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] + d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1]) * 1024.0
        ms = dur[k][0] / dur[k][1] * 1e-6 if dur[k][1] else 0
        hit = d.get("TCC_HIT_sum", [0, 1])[0] / max(d.get("TCC_HIT_sum", [0, 1])[0] + d.get("TCC_MISS_sum", [0, 1])[0], 1)
        lines.append(f"    => HBM traffic per launch {hbm / 1e9:.3f} GB (2 x FETCH_SIZE + WRITE_SIZE), kernel {ms:.3f} ms in the FETCH_SIZE pass = {hbm / 1e9 / max(ms, 1e-9) :.1f} GB/ms... "
                     f"{hbm / max(ms * 1e-3, 1e-12) / 1e12:.2f} TB/s; L2 hit rate {hit:.3f}")
# per PRODUCT: the sweep's launches (slice groups, panels) of one product together; the script prints how many products it ran
import re, datetime
products = None
for f in glob.glob(sys.argv[1] + "/pass*.log"):
    m = re.search(r"PRODUCTS (\d+)", open(f).read())
    if m: products = int(m.group(1))
dom = max((k for k in agg if "k_csr_panel" in k or "k_lds_code" in k), key=lambda k: agg[k].get("FETCH_SIZE", [0, 1])[0], default=None)
if dom and products and "FETCH_SIZE" in agg[dom] and "WRITE_SIZE" in agg[dom]:
    d = agg[dom]
    hbm_total = (2.0 * d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024.0
    rec = {"collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "command": f"rocprofv3 --pmc <one counter group per pass> -- python3 scripts/exp_cfg_one.py {sys.argv[2]} (scripts/cfg_pmc.sh)",
           "kernel": dom, "launches_per_product": d["FETCH_SIZE"][1] / products, "hbm_bytes_per_product": hbm_total / products,
           "kernel_ms_per_product": dur[dom][0] * 1e-6 / products if dur[dom][1] else None,
           "l2_hit_rate": d.get("TCC_HIT_sum", [0, 1])[0] / max(d.get("TCC_HIT_sum", [0, 1])[0] + d.get("TCC_MISS_sum", [0, 1])[0], 1),
           "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per MI355X_MICROARCH.md HBM section"}
    json.dump(rec, open(sys.argv[1] + "/traffic.json", "w"), indent=1)
    lines.append("per product: " + json.dumps(rec))
open(sys.argv[1] + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
