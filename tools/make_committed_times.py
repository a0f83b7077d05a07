#!/usr/bin/env python3
"""profiles/committed_kernel_times.json from a collect_profiles.sh run (gpurun_out/<tag>_*): the rocprofv3 average of the
fused message forward (bench.py states it beside its own HIP-event figure) and the kernel groups of one replayed step by
share of the step (bench.py's `kernel_groups`).   python tools/make_committed_times.py <tag>"""
import csv, json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
GROUPS = [
    ("decoder layer (channel-group kernels incl. the 4-column-block Dense forward, which the prior / heads share: 12 of its 30 launches)", r"dec_|skinny_fwd_k<1, 16>"),
    ("optimizer (rank update of the bead-level layers, norm, Adam)", r"grouped_wgrad_t<true>|adam_update|sumsq_partial|optim_finalize|wgrad_gram"),
    ("atom-graph message passing (K2g / K2 / K2b + reductions)", r"equi_msg_|segment_reduce|segment_broadcast"),
    ("atom-level Dense (tile GEMMs) and their weight gradients", r"tile_|gathered_wgrad|grouped_wgrad_t<false>"),
    ("per-batch graph plans + edge records", r"pj_|grp_build|gj_records|csr_|edge_geometry|copyBuffer|fillBuffer"),
    ("prior / heads / loss / decoder tail", r"skinny_|elbo|reconstruct|embedding_rows|pseudo_"),
]
for w in ("chignolin", "dipeptide", "protein2000"):
    entry = {}
    stats = os.path.join(root, "gpurun_out", f"{tag}_{w}_kernel_stats.csv")
    if os.path.exists(stats):
        rows = list(csv.DictReader(open(stats)))
        fwd = [r for r in rows if "equi_msg_fwd_grp_k" in r["Name"]]
        if fwd:
            r = max(fwd, key=lambda r: int(r["TotalDurationNs"]))
            entry["message_forward"] = {"kernel": r["Name"].split("(")[0].replace("void ", ""), "avg_us": float(r["AverageNs"]) / 1e3,
                                        "calls": int(r["Calls"]), "source": f"profiles/{tag}_{w}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --workload {w} --no-cpu-baseline --no-parity`)"}
    seq = os.path.join(root, "gpurun_out", f"{tag}_step_sequence_{w}.txt")
    if os.path.exists(seq):
        tot, per = 0.0, {g: [0.0, 0] for g, _ in GROUPS}
        other = [0.0, 0]
        for line in open(seq):
            m = re.search(r"dur\s+([0-9.]+)\s+grid=.*?x\s*\d+\s+(.*)$", line)
            if not m:
                continue
            dur, name = float(m.group(1)), m.group(2)
            tot += dur
            for g, pat in GROUPS:
                if re.search(pat, name):
                    per[g][0] += dur; per[g][1] += 1
                    break
            else:
                other[0] += dur; other[1] += 1
        groups = [{"group": g, "us": round(v[0], 1), "launches": v[1], "share": round(v[0] / tot, 3)} for g, v in per.items()]
        groups.append({"group": "other (tensor-op launches)", "us": round(other[0], 1), "launches": other[1], "share": round(other[0] / tot, 3)})
        entry["kernel_groups"] = {"source": f"profiles/{tag}_step_sequence_{w}.txt (one replayed step incl. the per-batch graph work, rocprofv3 kernel trace)",
                                  "kernel_time_us": round(tot, 1), "groups": sorted(groups, key=lambda g: -g["us"])}
    if entry:
        out[w] = entry
json.dump(out, open(os.path.join(root, "profiles", "committed_kernel_times.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
